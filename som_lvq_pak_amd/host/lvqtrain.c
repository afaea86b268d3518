/* lvqtrain -- LVQ1 / OLVQ1 / LVQ2.1 / LVQ3 training on the MI355X engine; the algorithm is
 * chosen by the program name (lvq1, olvq1, lvq2, lvq3) or -type, as in LVQ_PAK
 * (lvqtrain.c:70-112).  -selfuncs is read here (the reference advertises it but never
 * extracts it, lvqtrain.c:61,90,188). */
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "pak.h"

static const char *usage =
    "lvqtrain / lvq1 / olvq1 / lvq2 / lvq3 - train a codebook (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file  -rlen N\n"
    "           -alpha A (optional for olvq1)   lvq2: -win W   lvq3: -win W -epsilon E\n"
    "Optional:  -type lvq1|olvq1|lvq2|lvq3  -rand seed  -buffer N  -alpha_type linear|inverse_t\n"
    "           -snapfile name  -snapinterval N  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params params;
  struct snapshot_info snap = {0, NULL, 0};
  float winlen = 0.0f, epsilon = 0.0f;
  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }

  const char *progname = pak_progname(argv[0]);
  char *type_s = extract_parameter(argc, argv, "-type", OPTION);
  if (type_s) progname = type_s;
  int kind = !strcasecmp(progname, "lvq1") ? SOMHIP_LVQ1 : !strcasecmp(progname, "olvq1") ? SOMHIP_OLVQ1
           : !strcasecmp(progname, "lvq2") ? SOMHIP_LVQ2 : !strcasecmp(progname, "lvq3") ? SOMHIP_LVQ3 : 0;
  if (!kind) { fprintf(stderr, "Unknown LVQ type %s\n", progname); exit(1); }

  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  params.length = oatoi(extract_parameter(argc, argv, "-rlen", ALWAYS), 1);
  char *rand_s = extract_parameter(argc, argv, "-rand", OPTION);
  long buffer = oatoi(extract_parameter(argc, argv, "-buffer", OPTION), 0);
  char *alpha_s = extract_parameter(argc, argv, "-alpha_type", OPTION);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  char *snapshot_file = extract_parameter(argc, argv, "-snapfile", OPTION);
  long snapshot_interval = oatoi(extract_parameter(argc, argv, "-snapinterval", OPTION), 0);
  if (snapshot_interval) {
    if (!snapshot_file) {
      snapshot_file = out_code_file;
      fprintf(stderr, "snapshot file not specified, using '%s'", snapshot_file);
    }
    snap.interval = snapshot_interval; snap.filename = snapshot_file;
  }
  float alpha;
  switch (kind) {                                 /* lvqtrain.c:144-162 */
  case SOMHIP_OLVQ1: alpha = oatof(extract_parameter(argc, argv, "-alpha", OPTION), 0.0f); break;
  case SOMHIP_LVQ2:
    alpha = (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS));
    winlen = (float)atof(extract_parameter(argc, argv, "-win", ALWAYS));
    break;
  case SOMHIP_LVQ3:
    alpha = (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS));
    epsilon = (float)atof(extract_parameter(argc, argv, "-epsilon", ALWAYS));
    winlen = (float)atof(extract_parameter(argc, argv, "-win", ALWAYS));
    break;
  default: alpha = (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS)); break;
  }

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 1, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 1, 1);
  if (!codes) { fprintf(stderr, "Can't open code file '%s'\n", in_data_file); close_entries(data); exit(1); }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    close_entries(data); close_entries(codes); exit(1);
  }
  set_teach_params(&params, codes, data, funcname);
  params.alpha = alpha;
  params.snapshot = snapshot_interval ? &snap : NULL;
  init_random((int)oatoi(rand_s, 0));
  if (rand_s) {
    if (buffer > 0 && buffer < data->num_entries) { data->buffer = buffer; data->random_order = 1; }
    else randomize_entry_order(data);
  }
  params.alpha_func = alpha_func_by_name(alpha_s, &params.alpha_type);
  if (!params.alpha_func) {
    fprintf(stderr, "Unknown alpha type %s\n", alpha_s);
    close_entries(data); close_entries(codes); exit(1);
  }

  struct entries *codes2 = NULL;
  switch (kind) {
  case SOMHIP_LVQ1: codes2 = lvq1_training(&params); break;
  case SOMHIP_OLVQ1: codes2 = olvq1_training(&params, in_code_file, out_code_file); break;
  case SOMHIP_LVQ2: codes2 = lvq2_training(&params, winlen); break;
  case SOMHIP_LVQ3: codes2 = lvq3_training(&params, epsilon, winlen); break;
  }
  if (!codes2) {
    fprintf(stderr, "Teaching failed\n");
    close_entries(data); close_entries(codes); pak_shutdown(); exit(1);
  }
  ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
  invalidate_alphafile(out_code_file);            /* lvqtrain.c:249 */
  close_entries(data);
  close_entries(codes);
  pak_shutdown();
  return 0;
}
