/* qerror -- mean quantization error of a data set on a map (qerror.c:43-123), winners from
 * the MI355X engine, accumulation exactly the reference's (float sum of double roots);
 * -qetype 1 = the neighbourhood-weighted error of find_qerror2 (som_rout.c:823). */
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
  struct pak_inputs io;
  memset(&teach, 0, sizeof teach);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  char *din = extract_parameter(argc, argv, "-din", ALWAYS), *cin = extract_parameter(argc, argv, "-cin", ALWAYS);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  int neighbourhood_weighted = oatoi(extract_parameter(argc, argv, "-qetype", OPTION), 0) > 0;
  teach.radius = oatof(extract_parameter(argc, argv, "-radius", OPTION), 1.0f);

  pak_gen_virtual_ok = 1;                                /* a gen: source is generated in HBM, never on the host */
  if (pak_open_inputs(din, 0, "Can't open data file '%s'\n", cin, 0, "Can't open code file '%s'\n", 2, &io)) exit(1);
  set_teach_params(&teach, io.codes, io.data, funcname);
  set_som_params(&teach);
  float total = neighbourhood_weighted ? find_qerror2(&teach) : find_qerror(&teach);
  long nod = io.data->num_entries;
  ifverbose(1)
    fprintf(stdout, "Quantization error of %s with map %s is %f per sample (%ld samples)\n", din, cin, total / (float)nod, nod);
  else
    fprintf(stdout, "%f\n", total / (float)nod);
  close_entries(io.data); close_entries(io.codes);
  pak_shutdown();
  return 0;
}
