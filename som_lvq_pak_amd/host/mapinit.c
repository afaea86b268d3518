/* mapinit / randinit / lininit -- initial map codebook (SOM_PAK mapinit.c:53-182): random in the
 * bounding box of the data (randinit_codes som_rout.c:34-162, host only) or laid out on the plane
 * of the two principal axes (lininit_codes som_rout.c:322-429: the two O(n dim^2) data passes run
 * on the MI355X engine, see paklib.c). */
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "pak.h"

static const char *usage =
    "randinit / lininit (mapinit -init rand|lin) - initialization of a SOM codebook (MI355X engine for lin)\n"
    "Required:  -din file  -cout file  -topol hexa|rect  -neigh bubble|gaussian  -xdim N  -ydim N\n"
    "Optional:  -rand seed  -init rand|lin  -v level\n";

int main(int argc, char **argv)
{
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  const char *progname = pak_progname(argv[0]);
  int init_rand = strcasecmp(progname, "randinit") == 0, init_lin = strcasecmp(progname, "lininit") == 0;
  char *in_data_file = extract_parameter(argc, argv, "-din", ALWAYS);
  char *out_code_file = extract_parameter(argc, argv, "-cout", ALWAYS);
  long randomize = (int)oatoi(extract_parameter(argc, argv, "-rand", OPTION), 0);
  char *s = extract_parameter(argc, argv, "-topol", ALWAYS);
  int topol = !strcasecmp(s, "hexa") ? TOPOL_HEXA : !strcasecmp(s, "rect") ? TOPOL_RECT : TOPOL_UNKNOWN;
  if (topol == TOPOL_UNKNOWN) { fprintf(stderr, "Unknown topology type %s\n", s); exit(1); }
  s = extract_parameter(argc, argv, "-neigh", ALWAYS);
  int neigh = !strcasecmp(s, "bubble") ? NEIGH_BUBBLE : !strcasecmp(s, "gaussian") ? NEIGH_GAUSSIAN : NEIGH_UNKNOWN;
  if (neigh == NEIGH_UNKNOWN) { fprintf(stderr, "Unknown neighborhood type %s\n", s); exit(1); }
  int xdim = (int)oatoi(extract_parameter(argc, argv, "-xdim", ALWAYS), 0);
  int ydim = (int)oatoi(extract_parameter(argc, argv, "-ydim", ALWAYS), 0);
  s = extract_parameter(argc, argv, "-init", OPTION);
  if (s) { init_lin = strcmp(s, "lin") == 0; init_rand = strcmp(s, "rand") == 0; }
  if (!init_rand && !init_lin) { fprintf(stderr, "Unknown initialization type %s\n", s ? s : progname); exit(1); }
  long noc = (long)xdim * ydim;
  if (noc <= 0 || xdim < 0) { fprintf(stderr, "Dimensions of map (%d %d) are incorrect\n", xdim, ydim); exit(1); }

  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  pak_gen_virtual_ok = 1;                              /* a gen: source stays on the device (bounding box / covariance passes) */
  struct entries *data = open_entries(in_data_file, 0, 1);
  if (!data) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); exit(1); }
  init_random((int)randomize);
  ifverbose(2) fprintf(stderr, "initializing codes (%s)\n", init_lin ? "linear" : "random");
  struct entries *codes = init_lin ? lininit_codes(data, topol, neigh, xdim, ydim)
                                   : randinit_codes(data, topol, neigh, xdim, ydim);
  if (!codes) { fprintf(stderr, "initialization failure\n"); exit(1); }
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", out_code_file);
  char comments[256];
  snprintf(comments, sizeof comments, "# random seed: %ld\n", randomize);
  save_entries_wcomments(codes, out_code_file, comments);
  close_entries(data); close_entries(codes);
  pak_shutdown();
  return 0;
}
