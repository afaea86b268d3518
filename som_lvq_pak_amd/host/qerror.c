/* qerror -- mean quantization error of a data set on a map (qerror.c:43-123), winners from
 * the MI355X engine, accumulation exactly the reference's (float sum of double roots). */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "qerror - quantization error of a map (MI355X engine)\n"
    "Required:  -cin file  -din file\n"
    "Optional:  -qetype 1 (neighbourhood-weighted error)  -radius r (for -qetype 1, default 1.0)\n"
    "           -buffer N  -selfuncs hip  -v level\n";

int main(int argc, char **argv)
{
  struct teach_params teach;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *in_code_file = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  int qmode = (int)oatoi(extract_parameter(argc, argv, "-qetype", OPTION), 0);
  teach.radius = oatof(extract_parameter(argc, argv, "-radius", OPTION), 1.0f);

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  struct entries *data = open_entries(in_data_file, 0, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are read from file %s\n", in_code_file);
  struct entries *codes = open_entries(in_code_file, 0, 1);
  if (!codes) { fprintf(stderr, "Can't open code file '%s'\n", in_code_file); close_entries(data); exit(1); }
  if (codes->topol < TOPOL_HEXA) {
    fprintf(stderr, "File %s is not a map file\n", in_code_file);
    close_entries(data); close_entries(codes); exit(1);
  }
  if (data->dimension != codes->dimension) {
    fprintf(stderr, "Data and codebook vectors have different dimensions (%d != %d)", data->dimension, codes->dimension);
    close_entries(data); close_entries(codes); exit(1);
  }
  set_teach_params(&teach, codes, data, funcname);
  set_som_params(&teach);
  float qerror = qmode > 0 ? find_qerror2(&teach) : find_qerror(&teach);
  long nod = data->num_entries;
  ifverbose(1)
    fprintf(stdout, "Quantization error of %s with map %s is %f per sample (%ld samples)\n",
            in_data_file, in_code_file, qerror / (float)nod, nod);
  else
    fprintf(stdout, "%f\n", qerror / (float)nod);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
