/* classify -- label every data vector with the first label of its nearest code
 * (LVQ_PAK classify.c:41-178): winners from the MI355X engine, output files as the reference
 * writes them (-dout: the data with its labels replaced, -cfout: one label per line). */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "classify - finds out the classifications against a given codebook (MI355X engine)\n"
    "Required:  -cin file  -din file  -dout file\nOptional:  -cfout file  -buffer N  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_cf_file = extract_parameter(argc, argv, "-cfout", OPTION);
  char *out_data_file = extract_parameter(argc, argv, "-dout", ALWAYS);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 0, 1);     /* labels optional (label_not_needed, classify.c:125) */
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 1, 1);
  if (!codes) { fprintf(stderr, "Can't open codes file '%s'\n", in_code_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    close_entries(data); close_entries(codes); exit(1);
  }
  FILE *ocf = NULL;
  if (out_cf_file) {
    ifverbose(2) fprintf(stderr, "Classifications are saved to file %s\n", out_cf_file);
    if (!(ocf = fopen(out_cf_file, "w"))) {
      fprintf(stderr, "Cannot write to %s\n", out_cf_file);
      close_entries(data); close_entries(codes); exit(1);
    }
  }
  set_teach_params(&teach, codes, data, funcname);

  long n = data->num_entries;
  int32_t *idx = malloc(sizeof(int32_t) * (n + 1)), *ret = malloc(sizeof(int32_t) * (n + 1));
  float *diff = malloc(sizeof(float) * (n + 1));
  if (find_all_winners(&teach, idx, diff, ret)) exit(1);
  for (long i = 0; i < n; i++) {                    /* classify.c:65-88 */
    int label;
    if (ret[i] == 0 || idx[i] < 0) {
      label = find_conv_to_ind("# empty datavector");          /* sample entirely masked: label kept */
    } else {
      label = get_entry_label(&codes->rows[idx[i]]);
      clear_entry_labels(data, i);
      if (label != LABEL_EMPTY) add_entry_label(data, i, label);
    }
    if (ocf) fprintf(ocf, "%s\n", find_conv_to_lab(label));
  }
  if (ocf) fclose(ocf);
  ifverbose(2) fprintf(stderr, "Output entries are saved to file %s\n", out_data_file);
  save_entries(data, out_data_file);
  free(idx); free(ret); free(diff);
  close_entries(codes); close_entries(data);
  pak_shutdown();
  return 0;
}
