/* visual -- best-matching unit coordinates and quantization error of every sample
 * (SOM_PAK visual.c:48-231): winners from the MI355X engine, output rows "bx by sqrt(diff)
 * labels-of-the-unit" under a 3-dim header with the map's topology. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "visual - map coordinates and quantization error per sample (MI355X engine)\n"
    "Required:  -cin file  -din file  -dout file\nOptional:  -noskip  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params params;
  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_data_file = extract_parameter(argc, argv, "-dout", ALWAYS);
  int noskip = extract_parameter(argc, argv, "-noskip", OPTION2) != NULL;
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);

  struct entries *data = open_entries(in_data_file, 0, !noskip);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); return 1; }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 0, 1);
  if (!codes) { fprintf(stderr, "can't open code file '%s'\n", in_code_file); return 1; }
  if (codes->topol < TOPOL_HEXA) { fprintf(stderr, "File %s is not a map file\n", in_code_file); return 1; }
  if (data->dimension != codes->dimension) { fprintf(stderr, "Data and codebook vectors have different dimensions"); return 1; }
  set_teach_params(&params, codes, data, funcname);
  set_som_params(&params);

  long n = data->num_entries;
  int32_t *idx = malloc(sizeof(int32_t) * (n + 1)), *ret = malloc(sizeof(int32_t) * (n + 1));
  float *diff = malloc(sizeof(float) * (n + 1));
  int emptylab = find_conv_to_ind("EMPTY_LINE");
  if (find_all_winners(&params, idx, diff, ret)) return 1;

  struct entries *out = calloc(1, sizeof *out);
  out->dimension = 3; out->topol = codes->topol; out->neigh = codes->neigh;
  out->xdim = codes->xdim; out->ydim = codes->ydim; out->num_entries = n;
  out->points = malloc(sizeof(float) * 3 * (n + 1));
  out->rows = calloc(n + 1, sizeof(struct data_entry));
  for (long i = 0; i < n; i++) {
    float *p = out->points + 3 * i;
    out->rows[i].points = p;
    if (ret[i] == 0 || idx[i] < 0) {                       /* empty sample, visual.c:113-121 */
      p[0] = -1; p[1] = -1; p[2] = -1.0;
      add_entry_label(out, i, emptylab);
    } else {
      struct data_entry *w = &codes->rows[idx[i]];
      p[0] = idx[i] % codes->xdim;
      p[1] = idx[i] / codes->xdim;
      p[2] = sqrt(diff[i]);
      for (int k = 0; k < w->num_labs; k++) add_entry_label(out, i, w->labels[k]);
    }
  }
  if (save_entries(out, out_data_file)) return 1;
  close_entries(out); close_entries(data); close_entries(codes);
  free(idx); free(ret); free(diff);
  pak_shutdown();
  return 0;
}
