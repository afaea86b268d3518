/* elimin -- keep only the data entries that a k-NN vote over the whole data set classifies
 * correctly (LVQ_PAK elimin.c:40-169): more same-label than other-label neighbours among the k
 * nearest, the entry itself included.  One all-pairs k-NN pass on the MI355X engine. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

#define KNN 10        /* the reference's own limit (elimin.c:30); this engine stops at 8 */

static const char *usage =
    "elimin - eliminates those entries that are incorrectly classified by knn (MI355X engine)\n"
    "Required:  -din file  -cout file\nOptional:  -knn N (default 5, at most 8)  -v level\n";

int main(int argc, char **argv)
{
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  int knn = (int)oatoi(extract_parameter(argc, argv, "-knn", OPTION), 5);
  if (knn > KNN) { fprintf(stderr, "Can use only %d neighbors", KNN); knn = KNN; }
  if (knn < 1) knn = 1;

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Extra codes are eliminated\n");
  long n = data->num_entries, nkeep = 0;
  int32_t *idx = malloc(sizeof(int32_t) * (n * knn + 1));
  float *diff = malloc(sizeof(float) * (n * knn + 1));
  long *keep = malloc(sizeof(long) * (n + 1));
  if (find_all_knn(data, data, knn, idx, diff)) { fprintf(stderr, "Elimination failed!\n"); exit(1); }
  for (long r = 0; r < n; r++) {                    /* eliminate_codes, elimin.c:76-106 */
    long correct = 0, incorrect = 0;
    int found = 1;
    for (int k = 0; k < knn; k++) if (idx[r * knn + k] < 0) found = 0;
    if (!found) continue;                            /* did not find winners */
    int datalabel = get_entry_label(&data->rows[r]);
    for (int k = 0; k < knn; k++) {
      if (get_entry_label(&data->rows[idx[r * knn + k]]) == datalabel) correct++;
      else incorrect++;
    }
    if (correct > incorrect) keep[nkeep++] = r;
  }
  struct entries *codes = pick_rows(data, keep, nkeep);
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
  invalidate_alphafile(out_code_file);
  free(idx); free(diff); free(keep);
  close_entries(codes); close_entries(data);
  pak_shutdown();
  return 0;
}
