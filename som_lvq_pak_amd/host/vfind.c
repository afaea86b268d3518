/* vfind -- train several randomly initialised maps and keep the one with the smallest
 * quantization error (SOM_PAK vfind.c:110-330).  The answers are read from stdin in the
 * reference's order; every trial is randinit_codes (seed = trial number, counting down) ->
 * som_training (ordering part) -> som_training (fine tuning) -> find_qerror / find_qerror2 on the
 * test file, all on the MI355X engine. */
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "pak.h"

static long get_int(const char *q, long no)
{
  char str[100];
  printf("%s: ", q);
  if (!fgets(str, sizeof str, stdin)) return no;
  return oatoi(str, no);
}
static float get_float(const char *q, float no)
{
  char str[100];
  printf("%s: ", q);
  if (!fgets(str, sizeof str, stdin)) return no;
  return (float)atof(str);
}
static char *get_str(const char *q)
{
  char str[100];
  printf("%s: ", q);
  if (!fgets(str, sizeof str, stdin)) { printf("Can't read required data\n"); exit(1); }
  char *t = strdup(str), *e;
  if ((e = strchr(t, ' '))) *e = '\0';
  if ((e = strchr(t, '\n'))) *e = '\0';
  return t;
}

int main(int argc, char **argv)
{
  struct teach_params params;
  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  printf("vfind (MI355X engine): trains a number of randomly initialised maps in two parts each\n"
         "(ordering, fine tuning) and saves the one with the smallest quantization error on the\n"
         "test file.  Answer the questions below; training starts after the last one.\n\n");
  long not = get_int("Give the number of trials", 0);
  char *in_data_file = get_str("Give the input data file name");
  char *in_test_file = get_str("Give the input test file name");
  char *out_code_file = get_str("Give the output map file name");
  char *s = get_str("Give the topology type");
  int topol = !strcasecmp(s, "hexa") ? TOPOL_HEXA : !strcasecmp(s, "rect") ? TOPOL_RECT : TOPOL_UNKNOWN;
  if (topol == TOPOL_UNKNOWN) { ifverbose(2) fprintf(stderr, "Unknown topology type, using hexagonal\n"); topol = TOPOL_HEXA; }
  s = get_str("Give the neighborhood type");
  int neigh = !strcasecmp(s, "bubble") ? NEIGH_BUBBLE : !strcasecmp(s, "gaussian") ? NEIGH_GAUSSIAN : NEIGH_UNKNOWN;
  if (neigh == NEIGH_UNKNOWN) { ifverbose(2) fprintf(stderr, "Unknown neighborhood type, using bubble\n"); neigh = NEIGH_BUBBLE; }
  int xdim = (int)get_int("Give the x-dimension", 0);
  int ydim = (int)get_int("Give the y-dimension", 0);
  long length1 = get_int("Give the training length of first part", 0);
  float alpha1 = get_float("Give the training rate of first part", 0.0);
  float radius1 = get_float("Give the radius in first part", 0.0);
  long length2 = get_int("Give the training length of second part", 0);
  float alpha2 = get_float("Give the training rate of second part", 0.0);
  float radius2 = get_float("Give the radius in second part", 0.0);
  printf("\n");
  use_fixed_level = (int)oatoi(extract_parameter(argc, argv, "-fixed", OPTION), 0);
  use_weights_level = (int)oatoi(extract_parameter(argc, argv, "-weights", OPTION), 0);
  char *alpha_s = extract_parameter(argc, argv, "-alpha_type", OPTION);
  int qmode = (int)oatoi(extract_parameter(argc, argv, "-qetype", OPTION), 0);
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);

  int error = 0;
  struct entries *data = NULL, *testdata = NULL, *best = NULL;
  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", in_data_file);
  if (!(data = open_entries(in_data_file, 0, 1))) { fprintf(stderr, "Can't open data file '%s'\n", in_data_file); error = 1; goto end; }
  ifverbose(2) fprintf(stderr, "Test entries are read from file %s\n", in_test_file);
  if (!(testdata = open_entries(in_test_file, 0, 1))) { fprintf(stderr, "Can't open test data file '%s'\n", in_test_file); error = 1; goto end; }
  if ((long)xdim * ydim <= 0 || xdim < 0) { fprintf(stderr, "Dimensions of map (%d %d) are incorrect\n", xdim, ydim); error = 1; goto end; }
  params.alpha_func = alpha_func_by_name(alpha_s ? alpha_s : "linear", &params.alpha_type);
  if (!params.alpha_func) { fprintf(stderr, "Unknown alpha type %s\n", alpha_s); error = 1; goto end; }

  float qerrorb = FLT_MAX;
  long bnot = 0, nod = 0;
  while (not) {                                        /* vfind.c:244-306 */
    init_random((int)not);
    ifverbose(2) fprintf(stderr, "Initializing codebook\n");
    struct entries *codes = randinit_codes(data, topol, neigh, xdim, ydim);
    short at = params.alpha_type;
    ALPHA_FUNC *af = params.alpha_func;
    set_teach_params(&params, codes, NULL, funcname);
    params.alpha_type = at; params.alpha_func = af;
    set_som_params(&params);
    params.data = data;
    params.length = length1; params.alpha = alpha1; params.radius = radius1;
    ifverbose(2) fprintf(stderr, "Training map, first part, rlen: %ld alpha: %f\n", params.length, params.alpha);
    if (!som_training(&params)) { error = 1; goto end; }
    params.length = length2; params.alpha = alpha2; params.radius = radius2;
    ifverbose(2) fprintf(stderr, "Training map, second part, rlen: %ld alpha: %f\n", params.length, params.alpha);
    if (!som_training(&params)) { error = 1; goto end; }
    params.data = testdata;
    ifverbose(2) fprintf(stderr, "Calculating quantization error\n");
    float qerror = qmode > 0 ? find_qerror2(&params) : find_qerror(&params);
    nod = testdata->num_entries;
    if (qerror < qerrorb) {
      qerrorb = qerror;
      bnot = not;
      struct entries *tmp = best; best = codes; codes = tmp;
    }
    if (codes) close_entries(codes);
    ifverbose(1) fprintf(stderr, "%3ld: %f\n", not, qerror / (float)nod);
    not--;
  }
  if (best) {
    ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", out_code_file);
    save_entries(best, out_code_file);
    ifverbose(1) fprintf(stdout, "Smallest error with random seed %3ld: %f\n", bnot, qerrorb / (float)nod);
  }
end:
  if (best) close_entries(best);
  if (data) close_entries(data);
  if (testdata) close_entries(testdata);
  pak_shutdown();
  return error;
}
