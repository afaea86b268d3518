/* knntest -- recognition accuracy by a k-NN vote against a codebook (LVQ_PAK knntest.c:40-215):
 * the k nearest codes of every sample from the MI355X engine, vote and tallies as the
 * reference keeps them. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "knntest - recognition accuracy by knn test (MI355X engine)\n"
    "Required:  -cin file  -din file\nOptional:  -knn N (default 5, at most 8)  -buffer N  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  int knn = (int)oatoi(extract_parameter(argc, argv, "-knn", OPTION), 5);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  if (knn < 1) knn = 1;

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file %s\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 1, 1);
  if (!codes) { fprintf(stderr, "Can't open codes file %s\n", in_code_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    close_entries(data); close_entries(codes); exit(1);
  }
  set_teach_params(&teach, codes, data, funcname);
  teach.knn = knn;

  long n = data->num_entries, total = 0, stotal = 0;
  int32_t *idx = malloc(sizeof(int32_t) * (n * knn + 1));
  float *diff = malloc(sizeof(float) * (n * knn + 1));
  if (find_all_knn(codes, data, knn, idx, diff)) exit(1);
  struct hitlist *correct = new_hitlist(), *totals = new_hitlist();
  for (long i = 0; i < n; i++) {                    /* knntest.c:103-131 */
    struct hitlist *hits = new_hitlist();
    for (int k = 0; k < knn; k++)
      if (idx[i * knn + k] >= 0) add_hit(hits, get_entry_label(&codes->rows[idx[i * knn + k]]));
    int datalabel = get_entry_label(&data->rows[i]);
    if (hits->entries > 0 && hits->label[0] == datalabel) {
      stotal++;
      add_hit(correct, datalabel);
    }
    add_hit(totals, datalabel);
    total++;
    free_hitlist(hits);
  }
  fprintf(stdout, "\nRecognition accuracy:\n\n");
  for (long k = 0; k < totals->entries; k++) {
    int tot = (int)totals->freq[k], res = (int)hitlist_label_freq(correct, totals->label[k]);
    fprintf(stdout, "%14s: ", find_conv_to_lab((int)totals->label[k]));
    fprintf(stdout, "%6.2f %%\n", 100.0 * (float)res / tot);
  }
  fprintf(stdout, "\nTotal accuracy: %6.2f %%\n\n", 100.0 * (float)stotal / total);
  free_hitlist(correct); free_hitlist(totals); free(idx); free(diff);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
