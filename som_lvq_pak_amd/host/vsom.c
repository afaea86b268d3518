/* vsom -- teach a self-organizing map on the MI355X engine.  Same flags, files and exit
 * behaviour as SOM_PAK's vsom (vsom.c:53-206); new: -batch B (default 1 = the reference's
 * online algorithm, bit-exact; B > 1 = mini-batch schedule; auto = the engine's own batch sizes
 * along the schedule). */
#include <stdlib.h>
#include <string.h>
#include "pak.h"

static const char *usage =
    "vsom - teach self-organizing map (MI355X engine)\n"
    "Required:  -cin file  -din file  -cout file  -rlen N  -alpha A  -radius R\n"
    "Optional:  -rand seed  -fixed  -weights  -buffer N  -alpha_type linear|inverse_t\n"
    "           -snapfile name  -snapinterval N  -selfuncs hip  -batch B|auto  -gpus G  -v level\n"
    "Files:     text (.dat/.cod), raw fp32 (a name ending in .f32; see datconv), or -din gen:k=..,dim=..,n=..,seed=..\n";

static int save_codes(struct teach_params *teach, void *cout)
{
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", (char *)cout);
  return save_entries(teach->codes, (char *)cout) ? 1 : 0;
}

int main(int argc, char **argv)
{
  struct teach_params params;
  struct pak_train_cli cli;
  struct pak_inputs io = {NULL, NULL};
  int error = 1;

  memset(&params, 0, sizeof params);
  global_options(argc, argv);
  if (extract_parameter(argc, argv, "-help", OPTION2)) { fputs(usage, stdout); exit(0); }
  pak_train_cli(argc, argv, &cli);
  params.alpha = (float)atof(extract_parameter(argc, argv, "-alpha", ALWAYS));
  params.radius = (float)atof(extract_parameter(argc, argv, "-radius", ALWAYS));
  use_fixed_level = extract_parameter(argc, argv, "-fixed", OPTION2) != NULL;
  use_weights_level = extract_parameter(argc, argv, "-weights", OPTION2) != NULL;
  char *batch_s = extract_parameter(argc, argv, "-batch", OPTION);
  /* -batch auto: the engine's own batch sizes along the schedule (somhip_som_auto_batch) */
  long batch = batch_s && strcmp(batch_s, "auto") == 0 ? SOMHIP_BATCH_AUTO : oatoi(batch_s, 1);
  int gpus = (int)oatoi(extract_parameter(argc, argv, "-gpus", OPTION), 1);   /* new: one process per GPU, codebook sharded */

  pak_gen_virtual_ok = 1;                                /* a gen: source is generated in HBM, never on the host (unless -rand / -buffer) */
  if (pak_open_inputs(cli.din, 0, "cant open data file '%s'\n", cli.cin, 0, "Can't open code file '%s'\n", 1, &io)) goto end;
  set_teach_params(&params, io.codes, io.data, cli.funcname);
  set_som_params(&params);
  params.snapshot = cli.want_snapshots ? &cli.snap : NULL;
  params.length = cli.length;
  params.batch = batch;
  pak_apply_rand(io.data, cli.rand_s, cli.buffer);       /* data vectors in random order (vsom.c:171) */
  params.alpha_func = alpha_func_by_name(cli.alpha_s, &params.alpha_type);
  if (!params.alpha_func) { fprintf(stderr, "Unknown alpha type %s\n", cli.alpha_s); goto end; }

  if (gpus > 1 || getenv("SOMHIP_COMM")) {               /* the ranks train; rank 0 saves */
    if (gpus < 1 || gpus > 64) { fprintf(stderr, "-gpus %d?\n", gpus); goto end; }
    error = som_training_multi(&params, gpus, save_codes, cli.cout);
    goto end;
  }
  if (som_training(&params) == NULL) goto end;           /* (the reference saves regardless) */
  ifverbose(2) fprintf(stderr, "Codebook entries are saved to file %s\n", cli.cout);
  save_entries(io.codes, cli.cout);
  error = 0;
end:
  close_entries(io.data);
  close_entries(io.codes);
  pak_shutdown();
  return error;
}
