/* accuracy -- recognition accuracy of a labelled codebook on labelled data
 * (accuracy.c:39-218): nearest code per sample from the MI355X engine, per-class tallies
 * and output text as LVQ_PAK prints them. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "accuracy - recognition accuracy (MI355X engine)\n"
    "Required:  -cin file  -din file\nOptional:  -cfout file  -buffer N  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_file = extract_parameter(argc, argv, "-cfout", OPTION);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);

  struct pak_inputs io;
  if (pak_open_inputs(in_data_file, 1, "Can't open data file '%s'\n", in_code_file, 1, "Can't open code file '%s'\n", 0, &io)) exit(1);
  struct entries *data = io.data, *codes = io.codes;
  FILE *ocf = NULL;
  if (out_file && !(ocf = fopen(out_file, "w"))) { fprintf(stderr, "can't open file '%s'\n", out_file); exit(1); }
  set_teach_params(&teach, codes, data, funcname);

  long n = data->num_entries, total = 0, stotal = 0;
  int32_t *idx = malloc(sizeof(int32_t) * (n + 1));
  float *diff = malloc(sizeof(float) * (n + 1));
  if (find_all_winners(&teach, idx, diff, NULL)) exit(1);
  struct hitlist *correct = new_hitlist(), *totals = new_hitlist();
  for (long i = 0; i < n; i++) {                    /* accuracy.c:80-113 */
    int datalabel = get_entry_label(&data->rows[i]);
    if (idx[i] >= 0 && get_entry_label(&codes->rows[idx[i]]) == datalabel) {
      stotal++;
      add_hit(correct, datalabel);
      if (ocf) fprintf(ocf, "1\n");
    } else if (ocf) fprintf(ocf, "0\n");
    add_hit(totals, datalabel);
    total++;
  }
  fprintf(stdout, "\nRecognition accuracy:\n\n");
  for (long k = 0; k < totals->entries; k++) {
    long tot = totals->freq[k], res = hitlist_label_freq(correct, totals->label[k]);
    fprintf(stdout, "%9s: %4ld entries ", find_conv_to_lab((int)totals->label[k]), tot);
    fprintf(stdout, "%6.2f %%\n", 100.0 * (float)res / tot);
  }
  fprintf(stdout, "\nTotal accuracy: %5ld entries %6.2f %%\n\n", total, 100.0 * (float)stotal / total);
  if (ocf) fclose(ocf);
  free_hitlist(correct); free_hitlist(totals); free(idx); free(diff);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
