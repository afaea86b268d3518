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

/* the fourteen questions, in the reference's order; answers as text (an empty stdin gives "") */
enum { Q_TRIALS, Q_DATA, Q_TEST, Q_OUT, Q_TOPOL, Q_NEIGH, Q_XDIM, Q_YDIM, Q_LEN1, Q_ALPHA1, Q_RADIUS1, Q_LEN2, Q_ALPHA2,
       Q_RADIUS2, Q_COUNT };
static const struct { const char *text; int is_name; } questions[Q_COUNT] = {
    {"Give the number of trials", 0},           {"Give the input data file name", 1},
    {"Give the input test file name", 1},       {"Give the output map file name", 1},
    {"Give the topology type", 1},              {"Give the neighborhood type", 1},
    {"Give the x-dimension", 0},                {"Give the y-dimension", 0},
    {"Give the training length of first part", 0},  {"Give the training rate of first part", 0},
    {"Give the radius in first part", 0},       {"Give the training length of second part", 0},
    {"Give the training rate of second part", 0},   {"Give the radius in second part", 0}};

static void ask(char answers[Q_COUNT][100])
{
  for (int q = 0; q < Q_COUNT; q++) {
    printf("%s: ", questions[q].text);
    if (!fgets(answers[q], 100, stdin)) {
      if (questions[q].is_name) { printf("Can't read required data\n"); exit(1); }
      answers[q][0] = '\0';
      continue;
    }
    if (questions[q].is_name) answers[q][strcspn(answers[q], " \n")] = '\0';
  }
}
static long as_long(const char *a) { return *a ? atol(a) : 0; }
static float as_float(const char *a) { return *a ? (float)atof(a) : 0.0f; }

/* one trial (vfind.c:247-300): randinit with the trial number as seed, the two training parts, the error on the test file */
struct vfind_job {
  struct entries *data, *testdata;
  int topol, neigh, xdim, ydim, weighted_error;
  short alpha_type;
  ALPHA_FUNC *alpha_func;
  const char *funcname, *out;
  struct { long length; float alpha, radius; const char *what; } part[2];
  long trials;
};
static struct entries *one_trial(struct vfind_job *j, long seed, float *qerror)
{
  struct teach_params params;
  memset(&params, 0, sizeof params);
  init_random((int)seed);
  ifverbose(2) fprintf(stderr, "Initializing codebook\n");
  struct entries *codes = randinit_codes(j->data, j->topol, j->neigh, j->xdim, j->ydim);
  if (!codes) return NULL;
  set_teach_params(&params, codes, NULL, j->funcname);
  params.alpha_type = j->alpha_type; params.alpha_func = j->alpha_func;
  set_som_params(&params);
  params.data = j->data;
  for (int p = 0; p < 2; p++) {
    params.length = j->part[p].length; params.alpha = j->part[p].alpha; params.radius = j->part[p].radius;
    ifverbose(2) fprintf(stderr, "Training map, %s part, rlen: %ld alpha: %f\n", j->part[p].what, params.length, params.alpha);
    if (!som_training(&params)) { close_entries(codes); return NULL; }
  }
  params.data = j->testdata;                           /* radius of the second part stays for -qetype 1 */
  ifverbose(2) fprintf(stderr, "Calculating quantization error\n");
  *qerror = j->weighted_error ? find_qerror2(&params) : find_qerror(&params);
  return codes;
}

/* vfind -gpus G: the trials are independent, so they run as G replicas, one process per GPU; trial
 * `seed` goes to rank (trials - seed) % G.  Every rank reports its errors and its best map to rank 0, which prints the
 * lines in the reference's order (seed counting down), keeps the first smallest error as the sequential loop does
 * (strict <, vfind.c:296) and saves that map: the output is the sequential run's, byte for byte. */
static int vfind_rank(int rank, int world, int *fds, void *arg)
{
  struct vfind_job *j = arg;
  if (pak_rank_device(rank) < 0) return 1;
  const long noc = (long)j->xdim * j->ydim, dim = j->data->dimension, nod = j->testdata->num_entries;
  float *err = malloc(sizeof(float) * (j->trials + 1));
  for (long s = 0; s <= j->trials; s++) err[s] = FLT_MAX;
  struct entries *best = NULL;
  float best_error = FLT_MAX;
  for (long seed = j->trials; seed > 0; seed--) {
    if ((j->trials - seed) % world != rank) continue;
    float q;
    struct entries *codes = one_trial(j, seed, &q);
    if (!codes) return 1;
    err[seed] = q;
    if (q < best_error) { best_error = q; struct entries *old = best; best = codes; codes = old; }
    if (codes) close_entries(codes);
  }
  float *rows = malloc(sizeof(float) * noc * dim);
  if (rank != 0) {
    if (best) memcpy(rows, best->points, sizeof(float) * noc * dim);
    return pak_sock_write(fds[0], err, sizeof(float) * (j->trials + 1)) || pak_sock_write(fds[0], rows, sizeof(float) * noc * dim);
  }
  /* rank 0: everybody's errors, then the best map of the rank that holds the winning trial */
  float *all = malloc(sizeof(float) * (j->trials + 1)), *theirs = malloc(sizeof(float) * (j->trials + 1));
  float **maps = calloc(world, sizeof(float *));
  memcpy(all, err, sizeof(float) * (j->trials + 1));
  for (int r = 1; r < world; r++) {
    maps[r] = malloc(sizeof(float) * noc * dim);
    if (pak_sock_read(fds[r - 1], theirs, sizeof(float) * (j->trials + 1)) || pak_sock_read(fds[r - 1], maps[r], sizeof(float) * noc * dim)) return 1;
    for (long s = 1; s <= j->trials; s++) if ((j->trials - s) % world == r) all[s] = theirs[s];
  }
  float win_error = FLT_MAX;
  long win_seed = 0;
  for (long seed = j->trials; seed > 0; seed--) {
    if (all[seed] < win_error) { win_error = all[seed]; win_seed = seed; }
    ifverbose(1) fprintf(stderr, "%3ld: %f\n", seed, all[seed] / (float)nod);
  }
  if (win_seed > 0) {
    const int owner = (int)((j->trials - win_seed) % world);
    /* a rank's best map is the one of its first smallest error: the winning trial is that rank's best */
    if (owner != 0) {
      if (!best) { init_random(1); best = randinit_codes(j->data, j->topol, j->neigh, j->xdim, j->ydim); }
      memcpy(best->points, maps[owner], sizeof(float) * noc * dim);
    }
    ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", j->out);
    save_entries(best, j->out);
    ifverbose(1) fprintf(stdout, "Smallest error with random seed %3ld: %f\n", win_seed, win_error / (float)nod);
  }
  return 0;
}

int main(int argc, char **argv)
{
  char ans[Q_COUNT][100];
  global_options(argc, argv);
  printf("vfind (MI355X engine): trains a number of randomly initialised maps in two parts each\n"
         "(ordering, fine tuning) and saves the one with the smallest quantization error on the\n"
         "test file.  Answer the questions below; training starts after the last one.\n\n");
  ask(ans);
  printf("\n");
  long trials = as_long(ans[Q_TRIALS]);
  int topol = !strcasecmp(ans[Q_TOPOL], "hexa") ? TOPOL_HEXA : !strcasecmp(ans[Q_TOPOL], "rect") ? TOPOL_RECT : TOPOL_UNKNOWN;
  if (topol == TOPOL_UNKNOWN) { ifverbose(2) fprintf(stderr, "Unknown topology type, using hexagonal\n"); topol = TOPOL_HEXA; }
  int neigh = !strcasecmp(ans[Q_NEIGH], "bubble") ? NEIGH_BUBBLE : !strcasecmp(ans[Q_NEIGH], "gaussian") ? NEIGH_GAUSSIAN : NEIGH_UNKNOWN;
  if (neigh == NEIGH_UNKNOWN) { ifverbose(2) fprintf(stderr, "Unknown neighborhood type, using bubble\n"); neigh = NEIGH_BUBBLE; }
  int xdim = (int)as_long(ans[Q_XDIM]), ydim = (int)as_long(ans[Q_YDIM]);
  const struct { long length; float alpha, radius; const char *what; } part[2] = {
      {as_long(ans[Q_LEN1]), as_float(ans[Q_ALPHA1]), as_float(ans[Q_RADIUS1]), "first"},
      {as_long(ans[Q_LEN2]), as_float(ans[Q_ALPHA2]), as_float(ans[Q_RADIUS2]), "second"}};

  use_fixed_level = (int)oatoi(extract_parameter(argc, argv, "-fixed", OPTION), 0);
  use_weights_level = (int)oatoi(extract_parameter(argc, argv, "-weights", OPTION), 0);
  char *alpha_s = extract_parameter(argc, argv, "-alpha_type", OPTION);
  int weighted_error = oatoi(extract_parameter(argc, argv, "-qetype", OPTION), 0) > 0;
  char *funcname = extract_parameter(argc, argv, "-selfuncs", OPTION);
  int gpus = (int)oatoi(extract_parameter(argc, argv, "-gpus", OPTION), 1);      /* new: trials as replicas over G GPUs */

  int error = 1;
  struct entries *data = NULL, *testdata = NULL, *best = NULL;
  ifverbose(2) fprintf(stderr, "Input entries are read from file %s\n", ans[Q_DATA]);
  if (!(data = open_entries(ans[Q_DATA], 0, 1))) { fprintf(stderr, "Can't open data file '%s'\n", ans[Q_DATA]); goto end; }
  ifverbose(2) fprintf(stderr, "Test entries are read from file %s\n", ans[Q_TEST]);
  if (!(testdata = open_entries(ans[Q_TEST], 0, 1))) { fprintf(stderr, "Can't open test data file '%s'\n", ans[Q_TEST]); goto end; }
  if ((long)xdim * ydim <= 0 || xdim < 0) { fprintf(stderr, "Dimensions of map (%d %d) are incorrect\n", xdim, ydim); goto end; }
  short alpha_type;
  ALPHA_FUNC *alpha_func = alpha_func_by_name(alpha_s ? alpha_s : "linear", &alpha_type);
  if (!alpha_func) { fprintf(stderr, "Unknown alpha type %s\n", alpha_s); goto end; }

  struct vfind_job job = { data, testdata, topol, neigh, xdim, ydim, weighted_error, alpha_type, alpha_func, funcname, ans[Q_OUT],
                           { {part[0].length, part[0].alpha, part[0].radius, part[0].what},
                             {part[1].length, part[1].alpha, part[1].radius, part[1].what} }, trials };
  if (gpus > 1) {                                       /* the trials as replicas, one process per GPU */
    error = pak_run_ranks(gpus, vfind_rank, &job);
    goto end;
  }
  float best_error = FLT_MAX;
  long best_seed = 0, nod = testdata->num_entries;
  for (long seed = trials; seed > 0; seed--) {         /* vfind.c:244-306: the seed is the trial counter */
    float qerror;
    struct entries *codes = one_trial(&job, seed, &qerror);
    if (!codes) goto end;
    if (qerror < best_error) {
      best_error = qerror; best_seed = seed;
      struct entries *old = best; best = codes; codes = old;
    }
    if (codes) close_entries(codes);
    ifverbose(1) fprintf(stderr, "%3ld: %f\n", seed, qerror / (float)nod);
  }
  if (best) {
    ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", ans[Q_OUT]);
    save_entries(best, ans[Q_OUT]);
    ifverbose(1) fprintf(stdout, "Smallest error with random seed %3ld: %f\n", best_seed, best_error / (float)nod);
  }
  error = 0;
end:
  if (best) close_entries(best);
  if (data) close_entries(data);
  if (testdata) close_entries(testdata);
  pak_shutdown();
  return error;
}
