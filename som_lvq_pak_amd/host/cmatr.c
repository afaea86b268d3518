/* cmatr -- recognition accuracy plus the confusion matrix (LVQ_PAK cmatr.c:40-250): nearest code
 * per sample from the MI355X engine, tallies and text as the reference prints them. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "cmatr - recognition accuracy and confusion matrix (MI355X engine)\n"
    "Required:  -cin file  -din file\nOptional:  -cfout file  -buffer N  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_file = extract_parameter(argc, argv, "-cfout", OPTION);

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 1, 1);
  if (!codes) { fprintf(stderr, "Can't open code file '%s'\n", in_code_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    close_entries(data); close_entries(codes); exit(1);
  }
  FILE *ocf = NULL;
  if (out_file) {
    ifverbose(2) fprintf(stderr, "Classifications are saved to file %s\n", out_file);
    if (!(ocf = fopen(out_file, "w"))) { fprintf(stderr, "\nCannot write to %s\n", out_file); exit(-1); }
  }
  set_teach_params(&teach, codes, data, NULL);

  long n = data->num_entries, total = 0, stotal = 0;
  int32_t *idx = malloc(sizeof(int32_t) * (n + 1)), *ret = malloc(sizeof(int32_t) * (n + 1));
  float *diff = malloc(sizeof(float) * (n + 1));
  if (find_all_winners(&teach, idx, diff, ret)) exit(1);
  struct hitlist *correct = new_hitlist(), *totals = new_hitlist(), *confuzion = new_hitlist();
  for (long i = 0; i < n; i++) {                    /* cmatr.c:77-121 */
    long datalabel = get_entry_label(&data->rows[i]);
    if (ret[i] == 0 || idx[i] < 0) continue;        /* invalid data vector */
    long label = get_entry_label(&codes->rows[idx[i]]);
    if (label == datalabel) {
      stotal++;
      add_hit(correct, datalabel);
      if (ocf) fprintf(ocf, "1\n");
    } else if (ocf) fprintf(ocf, "0\n");
    add_hit(confuzion, datalabel * 65536 + label);
    add_hit(totals, datalabel);
    total++;
  }
  fprintf(stdout, "\nRecognition accuracy:\n\n");
  for (long k = 0; k < totals->entries; k++) {
    fprintf(stdout, "%9s: %4ld entries ", find_conv_to_lab((int)totals->label[k]), totals->freq[k]);
    fprintf(stdout, "%6.2f %%\n", 100.0 * (float)hitlist_label_freq(correct, totals->label[k]) / totals->freq[k]);
  }
  fprintf(stdout, "\nTotal accuracy: %5ld entries %6.2f %%\n\n", total, 100.0 * (float)stotal / total);
  fprintf(stdout, "Confusion matrix:\n\n");
  fprintf(stdout, "          ");
  for (long k = 0; k < totals->entries; k++) fprintf(stdout, " %4s", find_conv_to_lab((int)totals->label[k]));
  fprintf(stdout, "\n\n");
  for (long a = 0; a < totals->entries; a++) {
    fprintf(stdout, "%9s: ", find_conv_to_lab((int)totals->label[a]));
    for (long b = 0; b < totals->entries; b++)
      fprintf(stdout, "%4ld ", hitlist_label_freq(confuzion, totals->label[a] * 65536 + totals->label[b]));
    fprintf(stdout, "\n");
  }
  fprintf(stdout, "\n");
  if (ocf) fclose(ocf);
  free_hitlist(correct); free_hitlist(totals); free_hitlist(confuzion);
  free(idx); free(ret); free(diff);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
