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
 * Round 3: the per-sample surface as well.  The "hip" row's `winner` serves teach->winner(codes, sample, ...) -- what
 * compute_accuracy (accuracy.c:80-113), find_labels (vcal.c:106-129), visual, classify, knntest, cmatr, setlabel call once
 * per data vector -- from a table: on its first call for a (codes, data buffer, knn) it runs somhip_find_winners over the
 * whole data list in one go and then answers sample after sample from the results (codes are read-only in those
 * scanners; the row's `vector_adapt` marks the table stale, and so does any sample or code row whose bytes are no longer
 * the ones the table was made from).  -buffer N (datafile.c:237-344): the epoch-level functions walk the data buffer
 * by buffer through the reference's own rewind_entries / next_entry -- one somhip_dataset per loaded buffer, the same
 * loads in the same order as the reference's loop makes them, the -rand shuffle of every buffer included.  `-batch B` /
 * `-batch auto` (the engine's mini-batch schedules) is read from the command line with the
 * reference's own extract_parameter.  Not bound here: `-gpus` (one process per GPU needs the host program to fork before
 * any GPU use: som_lvq_pak_amd/host/paklib.c, pak_run_ranks). */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <float.h>

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
static struct entries *glue_data = NULL;     /* the data list of the last set_teach_params: what the winner table is made from */

static int hip_winner(struct entries *codes, struct data_entry *sample, struct winner_info *win, int knn);
static void hip_vector_adapt(struct data_entry *codetmp, struct data_entry *sample, int dim, float alpha);

/* datafile.c:1248-1282.  "hip" switches the epoch-level functions below and puts the table-serving winner and the
 * table-invalidating vector_adapt in the per-sample slots (the row's `dist` stays vector_dist_euc: it acts on two host
 * rows).  lvqtrain.c never extracts -selfuncs (lvqtrain.c:90,188 pass NULL): SOMHIP_SELFUNCS=hip in the environment
 * selects the row there. */
int set_teach_params(struct teach_params *params, struct entries *codes, struct entries *data, long dbuffer, char *name)
{
  const char *env = getenv("SOMHIP_SELFUNCS");
  int rc;
  hip_selected = (name && strcasecmp(name, "hip") == 0) || (!name && env && strcasecmp(env, "hip") == 0);
  rc = ref_set_teach_params(params, codes, data, dbuffer, hip_selected ? NULL : name);
  if (hip_selected) {
    glue_data = data;
    params->winner = hip_winner;
    params->vector_adapt = hip_vector_adapt;
  }
  return rc;
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

/* the rows that are in memory right now: the whole list, or -- with -buffer -- the buffer loaded last (read_entries,
 * datafile.c:237-344, chains them from e->entries).  No file access.  ptrs (may be NULL) receives the nodes. */
static int chain_dense(struct entries *e, struct dense *d, struct data_entry ***ptrs)
{
  struct data_entry *t;
  long r = 0;
  memset(d, 0, sizeof *d);
  d->dim = e->dimension;
  for (t = e->entries; t; t = t->next) r++;
  if (r == 0) return 0;
  d->n = r;
  d->rows = malloc(sizeof(float) * r * d->dim);
  d->mask = calloc((size_t)r * d->dim, 1);
  d->label = malloc(sizeof(int32_t) * r);
  d->weight = malloc(sizeof(int16_t) * r);
  d->fixed = malloc(sizeof(int16_t) * 2 * r);
  if (ptrs) *ptrs = malloc(sizeof(struct data_entry *) * r);
  for (r = 0, t = e->entries; t; t = t->next, r++) {
    memcpy(d->rows + r * d->dim, t->points, sizeof(float) * d->dim);
    if (t->mask) { memcpy(d->mask + r * d->dim, t->mask, d->dim); d->any_mask = 1; }
    d->label[r] = get_entry_label(t);
    d->weight[r] = t->weight;
    d->fixed[2 * r] = t->fixed ? t->fixed->xfix : -1;
    d->fixed[2 * r + 1] = t->fixed ? t->fixed->yfix : -1;
    if (t->fixed) d->any_fixed = 1;
    if (ptrs) (*ptrs)[r] = t;
  }
  return 1;
}

static int buffered(struct entries *data) { return data->flags.loadmode == LOADMODE_BUFFER; }

/* `-batch B` / `-batch auto` from the command line, with the reference's own extract_parameter (lvq_pak.c:583): the
 * tools' main() does not know the flag, so the glue reads the process's arguments itself */
static long glue_batch(void)
{
  static long batch = 0;
  if (batch == 0) {
    static char buf[65536];
    static char *argv[4096];
    int argc = 0;
    size_t n = 0, i;
    FILE *f = fopen("/proc/self/cmdline", "rb");
    batch = 1;
    if (f) {
      n = fread(buf, 1, sizeof buf - 1, f);
      fclose(f);
      for (i = 0; i < n && argc < 4095; i += strlen(buf + i) + 1) argv[argc++] = buf + i;
      if (argc > 0) {
        char *v = extract_parameter(argc, argv, "-batch", OPTION);
        if (v) batch = strcasecmp(v, "auto") == 0 ? SOMHIP_BATCH_AUTO : atol(v) > 1 ? atol(v) : 1;
      }
    }
  }
  return batch;
}

/* ---- the per-sample winner of the "hip" row, served from one scan of the whole data list ------------------------- */
static struct {
  int valid, dirty, knn, dim;
  struct entries *codes;
  long ncodes, n, cursor, calls;
  struct data_entry **ptr, **crow;     /* data nodes in list order; code rows in list order */
  float *xcopy, *ccopy;                /* the bytes the table was made from */
  unsigned char *mcopy;                /* masks of the data rows (all zero: none) */
  int32_t *idx, *ret;
  float *diff;
} tab;

static void tab_free(void)
{
  free(tab.ptr); free(tab.crow); free(tab.xcopy); free(tab.ccopy); free(tab.mcopy); free(tab.idx); free(tab.ret); free(tab.diff);
  memset(&tab, 0, sizeof tab);
}
static void hip_vector_adapt(struct data_entry *codetmp, struct data_entry *sample, int dim, float alpha)
{
  tab.dirty = 1;                       /* a code row changes: whatever was scanned before is history */
  adapt_vector(codetmp, sample, dim, alpha);
}
static int cpu_winner(struct entries *codes, struct data_entry *sample, struct winner_info *win, int knn)
{
  return knn == 1 ? find_winner_euc(codes, sample, win, knn) : find_winner_knn(codes, sample, win, knn);
}
static int tab_build(struct entries *codes, int knn)
{
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  eptr p;
  struct data_entry *t;
  long r;
  int ok = 0;
  static int gpu_failed = 0;           /* said once; from then on the reference's own per-sample functions answer */
  tab_free();
  if (gpu_failed || !glue_data || !chain_dense(glue_data, &x, &tab.ptr)) return 0;
  if (!to_dense(codes, &c)) { free_dense(&x); tab_free(); return 0; }
  tab.crow = malloc(sizeof(struct data_entry *) * c.n);
  for (r = 0, t = rewind_entries(codes, &p); t; t = next_entry(&p), r++) tab.crow[r] = t;
  tab.idx = malloc(sizeof(int32_t) * x.n * knn);
  tab.diff = malloc(sizeof(float) * x.n * knn);
  tab.ret = malloc(sizeof(int32_t) * x.n);
  if (engine_up("winner") &&
      !somhip_codebook_create(eng, c.rows, NULL, c.n, c.dim, SOMHIP_TOPOL_LVQ, 0, 0, 0, 0, c.n, &cb) &&
      !somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, NULL, NULL, &ds) &&
      !somhip_find_winners(cb, ds, 0, x.n, knn, knn == 1 ? SOMHIP_TIE_FIRST : SOMHIP_TIE_KNN, tab.idx, tab.diff, tab.ret)) {
    ok = 1;
  } else {
    fprintf(stderr, "winner: %s -- using the CPU functions\n", somhip_last_error());
    gpu_failed = 1;
  }
  if (cb) somhip_codebook_destroy(cb);
  if (ds) somhip_dataset_destroy(ds);
  if (!ok) { free_dense(&c); free_dense(&x); tab_free(); return 0; }
  tab.valid = 1; tab.knn = knn; tab.dim = x.dim; tab.codes = codes; tab.ncodes = c.n; tab.n = x.n;
  tab.xcopy = x.rows; x.rows = NULL;
  tab.mcopy = x.mask; x.mask = NULL;
  tab.ccopy = c.rows; c.rows = NULL;
  free_dense(&c); free_dense(&x);
  ifverbose(2) fprintf(stderr, "winner: %ld data vectors against %ld codes on the GPU (knn %d)\n", tab.n, tab.ncodes, knn);
  return 1;
}
static long tab_find(struct data_entry *sample)
{
  long i = tab.cursor, k;
  if (!(i < tab.n && tab.ptr[i] == sample)) {
    for (i = 0; i < tab.n && tab.ptr[i] != sample; i++) ;
    if (i == tab.n) return -1;
  }
  /* the node must still hold the bytes that were scanned (with -buffer the nodes are re-filled, datafile.c:245-282) */
  if (memcmp(sample->points, tab.xcopy + i * tab.dim, sizeof(float) * tab.dim)) return -2;
  for (k = 0; k < tab.dim; k++) if ((sample->mask ? sample->mask[k] != 0 : 0) != (tab.mcopy[i * tab.dim + k] != 0)) return -2;
  return i;
}
static int hip_winner(struct entries *codes, struct data_entry *sample, struct winner_info *win, int knn)
{
  long i, k;
  if (!hip_selected || knn < 1 || knn > 8) return cpu_winner(codes, sample, win, knn);
  if (tab.valid && !tab.dirty && tab.codes == codes && tab.ncodes == codes->num_entries && tab.knn == knn) {
    /* one code row per call is compared with what was scanned: a codebook changed behind vector_adapt's back is noticed */
    const long r = tab.calls++ % tab.ncodes;
    if (memcmp(tab.crow[r]->points, tab.ccopy + r * tab.dim, sizeof(float) * tab.dim)) tab.dirty = 1;
  }
  if (!(tab.valid && !tab.dirty && tab.codes == codes && tab.ncodes == codes->num_entries && tab.knn == knn)) {
    if (!tab_build(codes, knn)) return cpu_winner(codes, sample, win, knn);
  }
  i = tab_find(sample);
  if (i == -2 && tab_build(codes, knn)) i = tab_find(sample);      /* the buffer was re-filled: scan the new one */
  if (i < 0) return cpu_winner(codes, sample, win, knn);           /* not a vector of the data list (or no GPU answer) */
  tab.cursor = i + 1;
  for (k = 0; k < knn; k++) {
    const int32_t w = tab.idx[i * knn + k];
    win[k].index = w;
    win[k].winner = w >= 0 ? tab.crow[w] : NULL;
    win[k].diff = w >= 0 ? tab.diff[i * knn + k] : knn == 1 ? -1.0f : FLT_MAX;   /* what lvq_pak.c:52 / :168 leave when nothing beat FLT_MAX */
  }
  return tab.ret[i];
}

#ifdef SOMHIP_GLUE_SOM
/* ---- som_training (som_rout.c:556-671) ---- */
/* iterations [it0, it0 + cnt) of the schedule on data rows data_first ... of ds, in segments that end where the reference
 * saves a snapshot: after iteration le when le % interval == 0 and le > 0 (:650) */
static int som_segments(struct teach_params *teach, somhip_codebook *cb, somhip_dataset *ds, struct dense *c, long it0, long cnt,
                        long data_first)
{
  struct snapshot_info *snap = teach->snapshot;
  long start, end;
  for (start = it0; start < it0 + cnt; start = end) {
    end = it0 + cnt;
    if (snap && snap->interval > 0) {
      long next = (start + snap->interval - 1) / snap->interval * snap->interval;
      if (next == 0) next = snap->interval;
      if (next + 1 < end) end = next + 1;
    }
    {
      somhip_som_params sp = { teach->length, teach->alpha, teach->radius, teach->alpha_type, use_fixed(-1), use_weights(-1),
                               glue_batch() /* 1 = the reference's online schedule; -batch B / -batch auto: DESIGN section 2 */,
                               start, end - start, data_first + (start - it0) };
      if (somhip_som_train(cb, ds, &sp, NULL, NULL)) { fprintf(stderr, "som_training: %s\n", somhip_last_error()); return 1; }
    }
    if (snap && snap->interval > 0 && end - 1 > 0 && (end - 1) % snap->interval == 0) {
      if (somhip_codebook_download(cb, c->rows)) return 1;
      rows_back(teach->codes, c->rows);
      ifverbose(2) fprintf(stderr, "Saving snapshot, %ld iterations\n", end - 1);
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed, continuing teaching\n");
    }
  }
  return 0;
}

struct entries *som_training(struct teach_params *teach)
{
  struct entries *codes = teach->codes, *data = teach->data, *ret = NULL;
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  if (!hip_selected) return ref_som_training(teach);
  memset(&c, 0, sizeof c); memset(&x, 0, sizeof x);
  if (set_som_params(teach)) { fprintf(stderr, "som_training: can't set SOM parameters\n"); return NULL; }   /* :576-580 */
  if (data->dimension != codes->dimension) {                                                                 /* :591-596 */
    fprintf(stderr, "code dimension (%d) != data dimension (%d)\n", codes->dimension, data->dimension);
    return NULL;
  }
  if (!to_dense(codes, &c)) return NULL;
  if (!engine_up("som_training")) goto done;
  if (somhip_codebook_create(eng, c.rows, NULL, c.n, c.dim, codes->topol, codes->neigh, codes->xdim, codes->ydim, 0, c.n, &cb)) {
    fprintf(stderr, "som_training: %s\n", somhip_last_error());
    goto done;
  }
  if (!buffered(data)) {
    if (!to_dense(data, &x)) { fprintf(stderr, "som_training: can't get data\n"); goto done; }               /* :584-588 */
    if (somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, x.weight, x.any_fixed ? x.fixed : NULL, &ds)) {
      fprintf(stderr, "som_training: %s\n", somhip_last_error());
      goto done;
    }
    if (som_segments(teach, cb, ds, &c, 0, teach->length, 0)) goto done;
  } else {
    /* -buffer N (datafile.c:237-344): the data come N rows at a time into the same nodes.  The reference's loop takes
     * them through rewind_entries / next_entry (som_rout.c:600-610, wrapping at the end of the file); the same calls are
     * made here, one buffer's worth of iterations at a time, each buffer mirrored to the GPU while it is in memory. */
    eptr p;
    struct data_entry *sample = rewind_entries(data, &p);
    long it = 0, k;
    if (sample == NULL) { fprintf(stderr, "som_training: can't get data\n"); goto done; }
    while (it < teach->length) {
      long cnt;
      if (sample == NULL) {
        sample = rewind_entries(data, &p);
        if (sample == NULL) { fprintf(stderr, "som_training: couldn't rewind data (%ld/%ld iterations done)\n", it, teach->length); goto done; }
      }
      if (!chain_dense(data, &x, NULL)) goto done;
      cnt = x.n < teach->length - it ? x.n : teach->length - it;
      if (somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, x.weight, x.any_fixed ? x.fixed : NULL, &ds)) {
        fprintf(stderr, "som_training: %s\n", somhip_last_error());
        goto done;
      }
      if (som_segments(teach, cb, ds, &c, it, cnt, 0)) goto done;
      somhip_dataset_destroy(ds); ds = NULL;
      free_dense(&x); memset(&x, 0, sizeof x);
      it += cnt;
      for (k = 0; k < cnt && sample; k++) sample = next_entry(&p);     /* the last of these loads the next buffer, or ends the file */
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
  struct entries *data = teach->data;
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  float qerror = 0.0f;
  int failed = 0;
  eptr p;
  struct data_entry *sample;
  if (!hip_selected) return ref_find_qerror(teach);
  memset(&c, 0, sizeof c); memset(&x, 0, sizeof x);
  if (set_som_params(teach)) { fprintf(stderr, "find_qerror: can't set SOM parameters\n"); return -1; }
  if (!to_dense(teach->codes, &c)) return -1.0;
  if (!engine_up("find_qerror") ||
      somhip_codebook_create(eng, c.rows, NULL, c.n, c.dim, teach->codes->topol, teach->codes->neigh, teach->codes->xdim,
                             teach->codes->ydim, 0, c.n, &cb)) {
    fprintf(stderr, "find_qerror: %s\n", somhip_last_error());
    free_dense(&c);
    return -1.0;
  }
  /* one pass over the data, buffer by buffer (one "buffer" = the whole list without -buffer): :700-722 */
  if ((sample = rewind_entries(data, &p)) == NULL) { fprintf(stderr, "find_qerror: can't get data\n"); failed = 1; }
  while (sample && !failed) {
    long i, k;
    int32_t *idx, *ret;
    float *diff;
    if (!chain_dense(data, &x, NULL)) { failed = 1; break; }
    idx = malloc(sizeof(int32_t) * x.n); ret = malloc(sizeof(int32_t) * x.n); diff = malloc(sizeof(float) * x.n);
    if (!somhip_dataset_create(eng, x.rows, x.n, x.dim, x.any_mask ? x.mask : NULL, NULL, NULL, NULL, &ds) &&
        !somhip_find_winners(cb, ds, 0, x.n, 1, SOMHIP_TIE_FIRST, idx, diff, ret)) {
      for (i = 0; i < x.n; i++) {
        if (ret[i] == 0) continue;                    /* ignore empty vectors (:712) */
        qerror += sqrt((double) diff[i]);             /* float accumulator of double roots (:715) */
      }
    } else {
      fprintf(stderr, "find_qerror: %s\n", somhip_last_error());
      failed = 1;
    }
    free(idx); free(ret); free(diff);
    if (ds) { somhip_dataset_destroy(ds); ds = NULL; }
    k = x.n;
    free_dense(&x); memset(&x, 0, sizeof x);
    for (i = 0; i < k && sample; i++) sample = next_entry(&p);
  }
  if (cb) somhip_codebook_destroy(cb);
  free_dense(&c); free_dense(&x);
  return failed ? -1.0f : qerror;
}

#endif /* SOMHIP_GLUE_SOM */

#ifdef SOMHIP_GLUE_LVQ
/* ---- lvq1 / olvq1 / lvq2 / lvq3_training (lvq_rout.c:498-916) ---- */
/* iterations [it0, it0 + cnt) on data rows data_first ... of ds, cut where the reference saves snapshots */
static int lvq_segments(struct teach_params *teach, somhip_codebook *cb, somhip_dataset *ds, struct dense *c, int kind, float winlen,
                        float epsilon, float *talpha, float clamp, const char *who, long it0, long cnt, long data_first)
{
  struct snapshot_info *snap = teach->snapshot;
  long start, end;
  for (start = it0; start < it0 + cnt; start = end) {
    end = it0 + cnt;
    if (snap && snap->interval > 0) {
      long next = (start + snap->interval - 1) / snap->interval * snap->interval;
      if (next == 0) next = snap->interval;
      if (next + 1 < end) end = next + 1;
    }
    {
      somhip_lvq_params lp = { kind, teach->length, clamp, teach->alpha_type, winlen, epsilon, start, end - start, data_first + (start - it0) };
      if (somhip_lvq_train(cb, ds, &lp, talpha, NULL, NULL)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); return 1; }
    }
    if (snap && snap->interval > 0 && end - 1 > 0 && (end - 1) % snap->interval == 0) {
      if (somhip_codebook_download(cb, c->rows)) return 1;
      rows_back(teach->codes, c->rows);
      if (save_snapshot(teach, end - 1)) fprintf(stderr, "snapshot failed\n");
    }
  }
  return 0;
}

/* returns codes, NULL on an error, or (struct entries *)-1 when the data hold masked vectors (the LVQ loops of the engine
 * take none): the caller then runs the reference's loop -- only possible before anything was trained, i.e. without -buffer */
#define LVQ_MASKED ((struct entries *)-1)
static struct entries *lvq_on_engine(struct teach_params *teach, int kind, float winlen, float epsilon, float *talpha, float clamp,
                                     const char *who)
{
  struct entries *codes = teach->codes, *data = teach->data, *ret = NULL;
  struct dense c, x;
  somhip_codebook *cb = NULL;
  somhip_dataset *ds = NULL;
  memset(&c, 0, sizeof c); memset(&x, 0, sizeof x);
  if (!buffered(data)) {
    if (!to_dense(data, &x)) { fprintf(stderr, "%s: can't get data\n", who); return NULL; }
    if (x.any_mask) { free_dense(&x); return LVQ_MASKED; }
  }
  if (!to_dense(codes, &c)) { free_dense(&x); return NULL; }
  if (!engine_up(who)) goto done;
  if (somhip_codebook_create(eng, c.rows, c.label, c.n, c.dim, SOMHIP_TOPOL_LVQ, 0, 0, 0, 0, c.n, &cb)) {
    fprintf(stderr, "%s: %s\n", who, somhip_last_error());
    goto done;
  }
  if (!buffered(data)) {
    if (somhip_dataset_create(eng, x.rows, x.n, x.dim, NULL, x.label, NULL, NULL, &ds)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); goto done; }
    if (lvq_segments(teach, cb, ds, &c, kind, winlen, epsilon, talpha, clamp, who, 0, teach->length, 0)) goto done;
  } else {                                             /* -buffer N: as in som_training above (lvq_rout.c:527-541 wraps the same way) */
    eptr p;
    struct data_entry *sample = rewind_entries(data, &p);
    long it = 0, k;
    if (sample == NULL) { fprintf(stderr, "%s: can't get data\n", who); goto done; }
    while (it < teach->length) {
      long cnt;
      if (sample == NULL) {
        sample = rewind_entries(data, &p);
        if (sample == NULL) { fprintf(stderr, "%s: couldn't rewind data (%ld/%ld iterations done)\n", who, it, teach->length); goto done; }
      }
      if (!chain_dense(data, &x, NULL)) goto done;
      if (x.any_mask) { fprintf(stderr, "%s: masked data vectors with -buffer are not bound to the engine\n", who); goto done; }
      cnt = x.n < teach->length - it ? x.n : teach->length - it;
      if (somhip_dataset_create(eng, x.rows, x.n, x.dim, NULL, x.label, NULL, NULL, &ds)) { fprintf(stderr, "%s: %s\n", who, somhip_last_error()); goto done; }
      if (lvq_segments(teach, cb, ds, &c, kind, winlen, epsilon, talpha, clamp, who, it, cnt, 0)) goto done;
      somhip_dataset_destroy(ds); ds = NULL;
      free_dense(&x); memset(&x, 0, sizeof x);
      it += cnt;
      for (k = 0; k < cnt && sample; k++) sample = next_entry(&p);
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

/* the reference's own loop with the reference's own per-sample functions (the "hip" row's table-serving winner would
 * re-scan the whole data list after every adapt_vector) */
#define WITH_CPU_ROW(teach, call)                                                          \
  do {                                                                                     \
    WINNER_FUNCTION *w_ = (teach)->winner;                                                 \
    VECTOR_ADAPT *a_ = (teach)->vector_adapt;                                              \
    if (w_ == hip_winner) (teach)->winner = cpu_winner;                                    \
    if (a_ == hip_vector_adapt) (teach)->vector_adapt = adapt_vector;                      \
    ret = (call);                                                                          \
    (teach)->winner = w_; (teach)->vector_adapt = a_;                                      \
  } while (0)

struct entries *lvq1_training(struct teach_params *teach)
{
  struct entries *ret;
  if (hip_selected && (ret = lvq_on_engine(teach, SOMHIP_LVQ1, 0, 0, NULL, teach->alpha, "lvq1_training")) != LVQ_MASKED) return ret;
  WITH_CPU_ROW(teach, ref_lvq1_training(teach));
  return ret;
}
struct entries *lvq2_training(struct teach_params *teach, float winlen)
{
  struct entries *ret;
  if (hip_selected && (ret = lvq_on_engine(teach, SOMHIP_LVQ2, winlen, 0, NULL, teach->alpha, "lvq2_training")) != LVQ_MASKED) return ret;
  WITH_CPU_ROW(teach, ref_lvq2_training(teach, winlen));
  return ret;
}
struct entries *lvq3_training(struct teach_params *teach, float epsilon, float winlen)
{
  struct entries *ret;
  if (hip_selected && (ret = lvq_on_engine(teach, SOMHIP_LVQ3, winlen, epsilon, NULL, teach->alpha, "lvq3_training")) != LVQ_MASKED) return ret;
  WITH_CPU_ROW(teach, ref_lvq3_training(teach, epsilon, winlen));
  return ret;
}
struct entries *olvq1_training(struct teach_params *teach, char *infile, char *outfile)
{
  struct entries *ret;
  eptr p;
  long noc, i;
  float *talpha, alpha = teach->alpha;
  if (!hip_selected) return ref_olvq1_training(teach, infile, outfile);
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
  if (ret == LVQ_MASKED) {                            /* nothing was trained: the reference's loop from the start */
    ofree(talpha);
    WITH_CPU_ROW(teach, ref_olvq1_training(teach, infile, outfile));
    return ret;
  }
  if (ret) alpha_write(talpha, noc, outfile);         /* :694 */
  ofree(talpha);
  return ret;
}
#endif /* SOMHIP_GLUE_LVQ */
