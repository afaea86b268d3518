/* somhip_glue.c -- INTEGRATION.md as one compilable file: what a maintainer of SOM_PAK / LVQ_PAK adds so that
 * `-selfuncs hip` sends the epoch loops to the MI355X engine (include/somhip.h) and everything else -- argument
 * parsing, .dat/.cod text I/O, labels, snapshots, output text -- stays the reference's own code.
 *
 * It is written against the reference's headers and is linked with the reference's UNMODIFIED objects.  Inside the
 * reference tree the four dispatch points below would be `#ifdef USE_SOMHIP` blocks at the top of the functions
 * (INTEGRATION.md section 4); from outside the tree the same effect is had by renaming five symbols of the compiled
 * objects (oracle/Makefile: objcopy --redefine-sym set_teach_params=ref_set_teach_params ...), which is how
 * tests/test_cli_tools.py::test_reference_tools_linked_with_the_glue proves this file against vsom / qerror / lvq1 /
 * olvq1 / lvq3 built from /root/reference.  Nothing of the product links this file.  (SOM_PAK and LVQ_PAK are two link
 * sets -- som_rout.o or lvq_rout.o -- so the two halves compile under -DSOMHIP_GLUE_SOM / -DSOMHIP_GLUE_LVQ.)
 *
 * Out of its scope (falls back to the reference's CPU loop, with a note at -v 2): -buffer N streaming of the data
 * (datafile.c:237-344; paklib.c's struct feed shows how to do it against the same ABI). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "lvq_pak.h"
#include "datafile.h"
#include "labels.h"
#include "som_rout.h"
#include "lvq_rout.h"
#include "somhip.h"

/* the reference's own definitions, under the names oracle/Makefile gives them */
int ref_set_teach_params(struct teach_params *params, struct entries *codes, struct entries *data, long dbuffer, char *name);
struct entries *ref_som_training(struct teach_params *teach);
float ref_find_qerror(struct teach_params *teach);
struct entries *ref_lvq1_training(struct teach_params *teach);
struct entries *ref_olvq1_training(struct teach_params *teach, char *infile, char *outfile);
struct entries *ref_lvq2_training(struct teach_params *teach, float winlen);
struct entries *ref_lvq3_training(struct teach_params *teach, float epsilon, float winlen);

static int hip_selected = 0;
static somhip_engine *eng = NULL;

/* datafile.c:1248-1282.  "hip" keeps the reference's per-sample functions of the "default" row (every tool that calls
 * teach->winner per sample keeps working) and switches the epoch-level functions below.  lvqtrain.c never extracts
 * -selfuncs (lvqtrain.c:90,188 pass NULL): SOMHIP_SELFUNCS=hip in the environment selects the row there. */
int set_teach_params(struct teach_params *params, struct entries *codes, struct entries *data, long dbuffer, char *name)
{
  const char *env = getenv("SOMHIP_SELFUNCS");
  hip_selected = (name && strcasecmp(name, "hip") == 0) || (!name && env && strcasecmp(env, "hip") == 0);
  return ref_set_teach_params(params, codes, data, dbuffer, hip_selected ? NULL : name);
}

static int engine_up(const char *who)
{
  if (!eng && somhip_engine_create(0, &eng)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); return 0; }
  return 1;
}

/* ---- list <-> dense (the reference keeps every row in its own heap block on a list, lvq_pak.h:73-113) ---- */
struct dense { long n; int dim; float *rows; unsigned char *mask; int32_t *label; int16_t *weight, *fixed; int any_mask, any_fixed; };

static int to_dense(struct entries *e, struct dense *d)
{
  eptr p;
  struct data_entry *t;
  long r = 0;
  memset(d, 0, sizeof *d);
  d->dim = e->dimension;
  for (t = rewind_entries(e, &p); t; t = next_entry(&p)) r++;      /* list order = presentation order (already shuffled if -rand) */
  if (r == 0) return 0;
  d->n = r;
  d->rows = malloc(sizeof(float) * r * d->dim);
  d->mask = calloc((size_t)r * d->dim, 1);
  d->label = malloc(sizeof(int32_t) * r);
  d->weight = malloc(sizeof(int16_t) * r);
  d->fixed = malloc(sizeof(int16_t) * 2 * r);
  for (r = 0, t = rewind_entries(e, &p); t; t = next_entry(&p), r++) {
    memcpy(d->rows + r * d->dim, t->points, sizeof(float) * d->dim);
    if (t->mask) { memcpy(d->mask + r * d->dim, t->mask, d->dim); d->any_mask = 1; }
    d->label[r] = get_entry_label(t);
    d->weight[r] = t->weight;
    d->fixed[2 * r] = t->fixed ? t->fixed->xfix : -1;
    d->fixed[2 * r + 1] = t->fixed ? t->fixed->yfix : -1;
    if (t->fixed) d->any_fixed = 1;
  }
  return 1;
}
static void rows_back(struct entries *e, const float *rows)
{
  eptr p;
  struct data_entry *t;
  long r = 0;
  for (t = rewind_entries(e, &p); t; t = next_entry(&p), r++) memcpy(t->points, rows + r * e->dimension, sizeof(float) * e->dimension);
}
static void free_dense(struct dense *d) { free(d->rows); free(d->mask); free(d->label); free(d->weight); free(d->fixed); }

static int buffered(struct entries *data) { return data->buffer > 0 || data->flags.loadmode == LOADMODE_BUFFER; }
static int any_masked(struct entries *e)               /* the LVQ loops of the engine take no masked samples */
{
  eptr p;
  struct data_entry *t;
  for (t = rewind_entries(e, &p); t; t = next_entry(&p)) if (t->mask) return 1;
  return 0;
}

#ifdef SOMHIP_GLUE_SOM
/* ---- som_training (som_rout.c:556-671) ---- */
struct entries *som_training(struct teach_params *teach)
{
  struct entries *codes = teach->codes, *data = teach->data, *ret = NULL;
  struct snapshot_info *snap = teach->snapshot;
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  long start, end;
  if (!hip_selected) return ref_som_training(teach);
  if (buffered(data)) {
    ifverbose(2) fprintf(stderr, "som_training: -buffer streaming is not bound to the engine here, using the CPU loop\n");
    return ref_som_training(teach);
  }
  if (set_som_params(teach)) { fprintf(stderr, "som_training: can't set SOM parameters\n"); return NULL; }   /* :576-580 */
  if (!to_dense(data, &x)) { fprintf(stderr, "som_training: can't get data\n"); return NULL; }                /* :584-588 */
  if (data->dimension != codes->dimension) {                                                                 /* :591-596 */
    fprintf(stderr, "code dimension (%d) != data dimension (%d)\n", codes->dimension, data->dimension);
    free_dense(&x);
    return NULL;
  }
  if (!to_dense(codes, &c)) { free_dense(&x); return NULL; }
  if (!engine_up("som_training")) goto done;
  if (somhip_codebook_create(eng, c.rows, NULL, c.n, c.dim, codes->topol, codes->neigh, codes->xdim, codes->ydim, 0, c.n, &cb) ||
      somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, x.weight, x.any_fixed ? x.fixed : NULL, &ds)) {
    fprintf(stderr, "som_training: %s\n", somhip_last_error());
    goto done;
  }
  /* iterations run in segments that end where the reference saves a snapshot: after iteration le when
   * le % interval == 0 and le > 0 (:650) */
  for (start = 0; start < teach->length; start = end) {
    end = teach->length;
    if (snap && snap->interval > 0) {
      long next = (start + snap->interval - 1) / snap->interval * snap->interval;
      if (next == 0) next = snap->interval;
      if (next + 1 < end) end = next + 1;
    }
    {
      somhip_som_params sp = { teach->length, teach->alpha, teach->radius, teach->alpha_type, use_fixed(-1), use_weights(-1),
                               1 /* the reference's online schedule; a -batch flag would go here */, start, end - start,
                               start % x.n };
      if (somhip_som_train(cb, ds, &sp, NULL, NULL)) { fprintf(stderr, "som_training: %s\n", somhip_last_error()); goto done; }
    }
    if (snap && snap->interval > 0 && end - 1 > 0 && (end - 1) % snap->interval == 0) {
      if (somhip_codebook_download(cb, c.rows)) goto done;
      rows_back(codes, c.rows);
      ifverbose(2) fprintf(stderr, "Saving snapshot, %ld iterations\n", end - 1);
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed, continuing teaching\n");
    }
  }
  if (somhip_codebook_download(cb, c.rows)) { fprintf(stderr, "som_training: %s\n", somhip_last_error()); goto done; }
  rows_back(codes, c.rows);
  ret = codes;
done:
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  free_dense(&c); free_dense(&x);
  return ret;
}

/* ---- find_qerror (som_rout.c:678-731): winners from the engine, the reference's own accumulation ---- */
float find_qerror(struct teach_params *teach)
{
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  float qerror = -1.0f;
  if (!hip_selected || buffered(teach->data)) return ref_find_qerror(teach);
  if (set_som_params(teach)) { fprintf(stderr, "find_qerror: can't set SOM parameters\n"); return -1; }
  if (!to_dense(teach->data, &x)) { fprintf(stderr, "find_qerror: can't get data\n"); return -1.0; }
  if (!to_dense(teach->codes, &c)) { free_dense(&x); return -1.0; }
  {
    int32_t *idx = malloc(sizeof(int32_t) * x.n), *ret = malloc(sizeof(int32_t) * x.n);
    float *diff = malloc(sizeof(float) * x.n);
    long i;
    if (engine_up("find_qerror") &&
        !somhip_codebook_create(eng, c.rows, NULL, c.n, c.dim, teach->codes->topol, teach->codes->neigh, teach->codes->xdim,
                                teach->codes->ydim, 0, c.n, &cb) &&
        !somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, NULL, NULL, &ds) &&
        !somhip_find_winners(cb, ds, 0, x.n, 1, SOMHIP_TIE_FIRST, idx, diff, ret)) {
      qerror = 0.0;
      for (i = 0; i < x.n; i++) {
        if (ret[i] == 0) continue;                    /* ignore empty vectors (:712) */
        qerror += sqrt((double) diff[i]);             /* float accumulator of double roots (:715) */
      }
    } else {
      fprintf(stderr, "find_qerror: %s\n", somhip_last_error());
    }
    free(idx); free(ret); free(diff);
  }
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  free_dense(&c); free_dense(&x);
  return qerror;
}

#endif /* SOMHIP_GLUE_SOM */

#ifdef SOMHIP_GLUE_LVQ
/* ---- lvq1 / olvq1 / lvq2 / lvq3_training (lvq_rout.c:498-916) ---- */
static struct entries *lvq_on_engine(struct teach_params *teach, int kind, float winlen, float epsilon, float *talpha, float clamp,
                                     const char *who)
{
  struct entries *codes = teach->codes, *data = teach->data, *ret = NULL;
  struct snapshot_info *snap = teach->snapshot;
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  long start, end;
  if (!to_dense(data, &x)) { fprintf(stderr, "%s: can't get data\n", who); return NULL; }
  if (!to_dense(codes, &c)) { free_dense(&x); return NULL; }
  if (!engine_up(who)) goto done;
  if (somhip_codebook_create(eng, c.rows, c.label, c.n, c.dim, SOMHIP_TOPOL_LVQ, 0, 0, 0, 0, c.n, &cb) ||
      somhip_dataset_create(eng, x.rows, x.n, x.dim, NULL, x.label, NULL, NULL, &ds)) {
    fprintf(stderr, "%s: %s\n", who, somhip_last_error());
    goto done;
  }
  for (start = 0; start < teach->length; start = end) {
    end = teach->length;
    if (snap && snap->interval > 0) {
      long next = (start + snap->interval - 1) / snap->interval * snap->interval;
      if (next == 0) next = snap->interval;
      if (next + 1 < end) end = next + 1;
    }
    {
      somhip_lvq_params lp = { kind, teach->length, clamp, teach->alpha_type, winlen, epsilon, start, end - start, start % x.n };
      if (somhip_lvq_train(cb, ds, &lp, talpha, NULL, NULL)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); goto done; }
    }
    if (snap && snap->interval > 0 && end - 1 > 0 && (end - 1) % snap->interval == 0) {
      if (somhip_codebook_download(cb, c.rows)) goto done;
      rows_back(codes, c.rows);
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed\n");
    }
  }
  if (somhip_codebook_download(cb, c.rows)) goto done;
  rows_back(codes, c.rows);
  ret = codes;
done:
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  free_dense(&c); free_dense(&x);
  return ret;
}

struct entries *lvq1_training(struct teach_params *teach)
{
  if (!hip_selected || buffered(teach->data) || any_masked(teach->data)) return ref_lvq1_training(teach);
  return lvq_on_engine(teach, SOMHIP_LVQ1, 0, 0, NULL, teach->alpha, "lvq1_training");
}
struct entries *lvq2_training(struct teach_params *teach, float winlen)
{
  if (!hip_selected || buffered(teach->data) || any_masked(teach->data)) return ref_lvq2_training(teach, winlen);
  return lvq_on_engine(teach, SOMHIP_LVQ2, winlen, 0, NULL, teach->alpha, "lvq2_training");
}
struct entries *lvq3_training(struct teach_params *teach, float epsilon, float winlen)
{
  if (!hip_selected || buffered(teach->data) || any_masked(teach->data)) return ref_lvq3_training(teach, epsilon, winlen);
  return lvq_on_engine(teach, SOMHIP_LVQ3, winlen, epsilon, NULL, teach->alpha, "lvq3_training");
}
struct entries *olvq1_training(struct teach_params *teach, char *infile, char *outfile)
{
  struct entries *ret;
  eptr p;
  long noc, i;
  float *talpha, alpha = teach->alpha;
  if (!hip_selected || buffered(teach->data) || any_masked(teach->data)) return ref_olvq1_training(teach, infile, outfile);
  rewind_entries(teach->codes, &p);                   /* make sure codes are loaded (:606) */
  noc = teach->codes->num_entries;
  talpha = (float *) oalloc(sizeof(float) * noc);     /* the rates: given, read from the .lra file, or 0.3 (:614-627) */
  if (alpha == 0.0) {
    if (!alpha_read(talpha, noc, infile)) {
      alpha = 0.3;
      for (i = 0; i < noc; i++) talpha[i] = alpha;
    }
  } else {
    for (i = 0; i < noc; i++) talpha[i] = alpha;
  }
  ret = lvq_on_engine(teach, SOMHIP_OLVQ1, 0, 0, talpha, alpha, "olvq1_training");
  if (ret) alpha_write(talpha, noc, outfile);         /* :694 */
  ofree(talpha);
  return ret;
}
#endif /* SOMHIP_GLUE_LVQ */
