/* vsom -- teach a self-organizing map on the MI355X engine.  Same flags, files and exit
 * behaviour as SOM_PAK's vsom (vsom.c:53-206); new: -batch B (default 1 = the reference's
 * online algorithm, bit-exact; B > 1 = mini-batch schedule). */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "vsom - teach self-organizing map (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file  -rlen N  -alpha A  -radius R\n"
    "Optional:  -rand seed  -fixed  -weights  -buffer N  -alpha_type linear|inverse_t\n"
    "           -snapfile name  -snapinterval N  -selfuncs hip  -batch B  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params params;
  struct entries *data = NULL, *codes = NULL;
  struct snapshot_info snap = {0, NULL, 0};
  int error = 0;

  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }

  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  long length = oatoi(extract_parameter(argc, argv, "-rlen", ALWAYS), 1);
  float alpha = (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS));
  float radius = (float)atof(extract_parameter(argc, argv, "-radius", ALWAYS));
  char *rand_s = extract_parameter(argc, argv, "-rand", OPTION);
  use_fixed_level = extract_parameter(argc, argv, "-fixed", OPTION2) != NULL;
  use_weights_level = extract_parameter(argc, argv, "-weights", OPTION2) != NULL;
  long buffer = oatoi(extract_parameter(argc, argv, "-buffer", OPTION), 0);
  char *alpha_s = extract_parameter(argc, argv, "-alpha_type", OPTION);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  char *snapshot_file = extract_parameter(argc, argv, "-snapfile", OPTION);
  long snapshot_interval = oatoi(extract_parameter(argc, argv, "-snapinterval", OPTION), 0);
  long batch = oatoi(extract_parameter(argc, argv, "-batch", OPTION), 1);

  if (snapshot_interval) {
    if (!snapshot_file) {
      snapshot_file = out_code_file;
      fprintf(stderr, "snapshot file not specified, using '%s'", snapshot_file);
    }
    snap.interval = snapshot_interval;
    snap.filename = snapshot_file;
  }

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  if ((data = open_entries(in_data_file, 0, 1)) == NULL) {
    fprintf(stderr, "cant open data file '%s'\n", in_data_file);
    error = 1; goto end;
  }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  if ((codes = open_entries(in_code_file, 0, 1)) == NULL) {
    fprintf(stderr, "Can't open code file '%s'\n", in_code_file);
    error = 1; goto end;
  }
  if (codes->topol < TOPOL_HEXA) {
    fprintf(stderr, "File %s is not a map file\n", in_code_file);
    error = 1; goto end;
  }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions");
    error = 1; goto end;
  }

  set_teach_params(&params, codes, data, funcname);
  set_som_params(&params);
  params.snapshot = snapshot_interval ? &snap : NULL;
  params.length = length; params.alpha = alpha; params.radius = radius; params.batch = batch;

  init_random((int)oatoi(rand_s, 0));
  if (rand_s) {                                  /* data vectors in random order (vsom.c:171) */
    if (buffer > 0 && buffer < data->num_entries) { data->buffer = buffer; data->random_order = 1; }   /* reshuffled per buffer */
    else randomize_entry_order(data);                                                               /* once, at load */
  }
  params.alpha_func = alpha_func_by_name(alpha_s, &params.alpha_type);
  if (!params.alpha_func) {
    fprintf(stderr, "Unknown alpha type %s\n", alpha_s);
    error = 1; goto end;
  }

  if (som_training(&params) == NULL) { error = 1; goto end; }   /* (the reference saves regardless) */

  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  save_entries(codes, out_code_file);
end:
  close_entries(data);
  close_entries(codes);
  pak_shutdown();
  return error;
}
