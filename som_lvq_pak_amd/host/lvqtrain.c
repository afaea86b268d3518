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
    "           -snapfile name  -snapinterval N  -selfuncs hip  -gpus G  -v level\n";

/* which flags each algorithm insists on (lvqtrain.c:144-162) */
static const struct { const char *name; int kind, alpha_required, win, eps; } kinds[] = {
    {"lvq1", SOMHIP_LVQ1, 1, 0, 0}, {"olvq1", SOMHIP_OLVQ1, 0, 0, 0},
    {"lvq2", SOMHIP_LVQ2, 1, 1, 0}, {"lvq3", SOMHIP_LVQ3, 1, 1, 1}};

struct save_job { const char *cout; float *talpha; long noc; int olvq; };
static int save_trained(struct teach_params *teach, void *arg)          /* what follows a successful training (lvqtrain.c:245-249) */
{
  struct save_job *j = arg;
  if (j->olvq) alpha_write(j->talpha, j->noc, j->cout);                 /* lvq_rout.c:694 */
  ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", j->cout);
  save_entries(teach->codes, j->cout);
  invalidate_alphafile(j->cout);                                        /* lvqtrain.c:249 */
  return 0;
}

int main(int argc, char **argv)
{
  struct teach_params params;
  struct pak_train_cli cli;
  struct pak_inputs io;
  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }

  const char *progname = pak_progname(argv[0]);
  char *type_s = extract_parameter(argc, argv, "-type", OPTION);
  if (type_s) progname = type_s;
  int k = -1;
  for (int i = 0; i < 4; i++) if (strcasecmp(progname, kinds[i].name) == 0) k = i;
  if (k < 0) { fprintf(stderr, "Unknown LVQ type %s\n", progname); exit(1); }

  pak_train_cli(argc, argv, &cli);
  float alpha = kinds[k].alpha_required ? (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS))
                                        : oatof(extract_parameter(argc, argv, "-alpha", OPTION), 0.0f);
  float epsilon = kinds[k].eps ? (float)atof(extract_parameter(argc, argv, "-epsilon", ALWAYS)) : 0.0f;
  float winlen = kinds[k].win ? (float)atof(extract_parameter(argc, argv, "-win", ALWAYS)) : 0.0f;

  if (pak_open_inputs(cli.din, 1, "Can't open data file '%s'\n", cli.cin, 1, "Can't open code file '%s'\n", 0, &io)) exit(1);
  set_teach_params(&params, io.codes, io.data, cli.funcname);
  params.length = cli.length;
  params.alpha = alpha;
  params.snapshot = cli.want_snapshots ? &cli.snap : NULL;
  pak_apply_rand(io.data, cli.rand_s, cli.buffer);
  params.alpha_func = alpha_func_by_name(cli.alpha_s, &params.alpha_type);
  if (!params.alpha_func) {
    fprintf(stderr, "Unknown alpha type %s\n", cli.alpha_s);
    close_entries(io.data); close_entries(io.codes); exit(1);
  }

  int gpus = (int)oatoi(extract_parameter(argc, argv, "-gpus", OPTION), 1);     /* new: codebook rows sharded over G GPUs */
  if (gpus > 1 || getenv("SOMHIP_COMM")) {
    long noc = io.codes->num_entries;
    float *talpha = malloc(sizeof(float) * (noc + 1)), clamp = alpha;
    if (kinds[k].kind == SOMHIP_OLVQ1) {                          /* lvq_rout.c:614-627 */
      if (alpha == 0.0f) {
        if (!alpha_read(talpha, noc, cli.cin)) { clamp = 0.3f; for (long i = 0; i < noc; i++) talpha[i] = clamp; }
      } else {
        for (long i = 0; i < noc; i++) talpha[i] = alpha;
      }
    }
    struct save_job job = { cli.cout, talpha, noc, kinds[k].kind == SOMHIP_OLVQ1 };
    int bad = lvq_training_multi(&params, kinds[k].kind, winlen, epsilon, clamp, talpha, gpus, save_trained, &job);
    if (bad) fprintf(stderr, "Teaching failed\n");
    free(talpha);
    close_entries(io.data); close_entries(io.codes);
    pak_shutdown();
    return bad;
  }
  struct entries *trained = NULL;
  switch (kinds[k].kind) {
  case SOMHIP_LVQ1: trained = lvq1_training(&params); break;
  case SOMHIP_OLVQ1: trained = olvq1_training(&params, cli.cin, cli.cout); break;
  case SOMHIP_LVQ2: trained = lvq2_training(&params, winlen); break;
  case SOMHIP_LVQ3: trained = lvq3_training(&params, epsilon, winlen); break;
  }
  int error = trained == NULL;
  if (error) fprintf(stderr, "Teaching failed\n");
  else {
    ifverbose(2) fprintf(stdout, "Codebook entries are saved to file %s\n", cli.cout);
    save_entries(io.codes, cli.cout);
    invalidate_alphafile(cli.cout);               /* lvqtrain.c:249 */
  }
  close_entries(io.data);
  close_entries(io.codes);
  pak_shutdown();
  return error;
}
