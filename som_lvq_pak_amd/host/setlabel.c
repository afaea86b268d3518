/* setlabel -- label every codebook vector by the majority of its k nearest DATA vectors
 * (LVQ_PAK setlabel.c:43-161): here the data set plays the codebook and the codes are the samples
 * of one k-NN pass on the MI355X engine. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "setlabel - sets the labels of entries by the majority voting (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file\nOptional:  -knn N (default 5, at most 8)  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  int knn = (int)oatoi(extract_parameter(argc, argv, "-knn", OPTION), 5);
  if (knn < 1) knn = 1;

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 0, 1);           /* label_not_needed(1), setlabel.c:129 */
  if (!codes) { fprintf(stderr, "Can't open codes file '%s'\n", in_code_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    close_entries(data); close_entries(codes); exit(1);
  }
  long noc = codes->num_entries;
  int32_t *idx = malloc(sizeof(int32_t) * (noc * knn + 1));
  float *diff = malloc(sizeof(float) * (noc * knn + 1));
  if (find_all_knn(data, codes, knn, idx, diff)) exit(1);             /* "codebook" = data, samples = codes */
  for (long r = 0; r < noc; r++) {                                    /* find_labels, setlabel.c:68-90 */
    struct hitlist *hits = new_hitlist();
    for (int k = 0; k < knn; k++)
      if (idx[r * knn + k] >= 0) add_hit(hits, get_entry_label(&data->rows[idx[r * knn + k]]));
    if (hits->entries > 0) {
      clear_entry_labels(codes, r);
      if (hits->label[0] != LABEL_EMPTY) add_entry_label(codes, r, (int)hits->label[0]);
    }
    free_hitlist(hits);
  }
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
  free(idx); free(diff);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
