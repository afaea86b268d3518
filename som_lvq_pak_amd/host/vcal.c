/* vcal -- label the units of a trained map by the classes of the samples they win
 * (vcal.c:45-251): winners from the MI355X engine, frequency-ordered label lists as in
 * SOM_PAK. */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "vcal - label map units (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file\nOptional:  -numlabs N (default 1, 0 = all)  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  int numlabs = (int)oatoi(extract_parameter(argc, argv, "-numlabs", OPTION), 1);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  if (numlabs < 0) numlabs = 0;

  struct pak_inputs io;
  if (pak_open_inputs(in_data_file, 0, "Can't open data file '%s'\n", in_code_file, 0, "Can't open code file '%s'\n", 1, &io)) exit(1);
  struct entries *data = io.data, *codes = io.codes;
  set_teach_params(&teach, codes, data, funcname);

  long n = data->num_entries, noc = codes->num_entries;
  int32_t *idx = malloc(sizeof(int32_t) * (n + 1)), *ret = malloc(sizeof(int32_t) * (n + 1));
  float *diff = malloc(sizeof(float) * (n + 1));
  if (find_all_winners(&teach, idx, diff, ret)) exit(1);
  struct hitlist **hits = calloc(noc, sizeof *hits);
  for (long k = 0; k < noc; k++) hits[k] = new_hitlist();
  for (long i = 0; i < n; i++) {                    /* vcal.c:106-129 */
    int datalabel = get_entry_label(&data->rows[i]);
    if (ret[i] == 0 || idx[i] < 0) continue;        /* winner not found: skip the sample */
    if (datalabel != LABEL_EMPTY) add_hit(hits[idx[i]], datalabel);
  }
  for (long k = 0; k < noc; k++) {                  /* vcal.c:144-162 */
    long labs = numlabs == 0 ? hits[k]->entries : (hits[k]->entries < numlabs ? hits[k]->entries : numlabs);
    clear_entry_labels(codes, k);
    for (long j = 0; j < labs; j++) add_entry_label(codes, k, (int)hits[k]->label[j]);
    free_hitlist(hits[k]);
  }
  free(hits); free(idx); free(ret); free(diff);
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
