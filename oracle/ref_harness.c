/* ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A small driver of our own that links against the reference's *unmodified*
 * objects (built by oracle/Makefile from /root/reference into oracle/_ref/)
 * and exposes them over flat arrays, so tests can
 *   - run the real som_training / lvq*_training / find_qerror on in-memory
 *     fp32 data (no "%g" round trip through .cod files, datafile.c:431), and
 *   - record the (index, diff) every winner call returned.
 * It contains no algorithm: every number comes out of the reference's own
 * find_winner_euc / find_winner_knn / adapt_vector / som_training / ...
 *
 * Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the resulting libref_harness.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "lvq_pak.h"
#include "datafile.h"
#include "labels.h"
#include "som_rout.h"
#include "lvq_rout.h"

/* ---------------- dense arrays -> struct entries list ---------------- */

static struct entries *list_from_dense(const float *rows, long n, int dim,
                                       const int *labels, const short *weight,
                                       const short *fixed_xy, const unsigned char *mask,
                                       int topol, int neigh, int xdim, int ydim)
{
  struct entries *e = alloc_entries();
  struct data_entry *prev = NULL;
  long r;
  int i;
  if (!e) return NULL;
  e->dimension = (short)dim;
  e->topol = (short)topol;
  e->neigh = (short)neigh;
  e->xdim = (short)xdim;
  e->ydim = (short)ydim;
  for (r = 0; r < n; r++) {
    struct data_entry *d = alloc_entry(e);
    if (!d) return NULL;
    memcpy(d->points, rows + r * (long)dim, sizeof(float) * dim);
    if (labels && labels[r] != LABEL_EMPTY) set_entry_label(d, labels[r]);
    if (weight) d->weight = weight[r];
    if (fixed_xy && fixed_xy[2 * r] >= 0) {
      d->fixed = malloc(sizeof(struct fixpoint));
      d->fixed->xfix = fixed_xy[2 * r];
      d->fixed->yfix = fixed_xy[2 * r + 1];
    }
    if (mask) {
      int any = 0;
      for (i = 0; i < dim; i++) any |= mask[r * (long)dim + i];
      if (any) {
        d->mask = malloc(dim);
        for (i = 0; i < dim; i++) d->mask[i] = mask[r * (long)dim + i] ? 1 : 0;
      }
    }
    if (prev) prev->next = d; else e->entries = d;
    prev = d;
  }
  e->num_entries = e->num_loaded = n;
  e->flags.totlen_known = 1;
  return e;
}

static void dense_from_list(struct entries *e, float *rows, int dim)
{
  struct data_entry *d;
  long r = 0;
  for (d = e->entries; d; d = d->next, r++)
    memcpy(rows + r * (long)dim, d->points, sizeof(float) * dim);
}

/* ---------------- winner tracing ---------------- */

static WINNER_FUNCTION *real_winner;
static long *tr_index;
static float *tr_diff;
static long tr_pos, tr_cap;

static int tracing_winner(struct entries *codes, struct data_entry *sample,
                          struct winner_info *w, int knn)
{
  int ret = real_winner(codes, sample, w, knn);
  int k;
  for (k = 0; k < knn; k++) {
    if (tr_index && tr_pos < tr_cap) {
      tr_index[tr_pos] = ret ? w[k].index : -2;
      if (tr_diff) tr_diff[tr_pos] = ret ? w[k].diff : -1.0f;
    }
    tr_pos++;
  }
  return ret;
}

static void trace_begin(WINNER_FUNCTION *f, long *idx, float *diff, long cap)
{
  real_winner = f; tr_index = idx; tr_diff = diff; tr_pos = 0; tr_cap = cap;
}

/* ---------------- exported entry points ---------------- */

static ALPHA_FUNC *alpha_by_type(int t)
{
  return (t == ALPHA_INVERSE_T) ? inverse_t_alpha : linear_alpha;
}

/* som_training (som_rout.c:556) through the "default" registry row. */
int ref_som_train(float *codes, long ncodes, int dim, int xdim, int ydim, int topol, int neigh,
                  const float *data, long ndata, const short *weight, const short *fixed_xy,
                  const unsigned char *mask,
                  long length, float alpha, float radius, int alpha_type,
                  int fixed_on, int weights_on,
                  long *trace_index, float *trace_diff, double *seconds)
{
  struct teach_params tp;
  struct entries *ce, *de;
  struct timespec t0, t1;
  memset(&tp, 0, sizeof tp);
  verbose(0);
  label_not_needed(1);
  use_fixed(fixed_on);
  use_weights(weights_on);
  ce = list_from_dense(codes, ncodes, dim, NULL, NULL, NULL, NULL, topol, neigh, xdim, ydim);
  de = list_from_dense(data, ndata, dim, NULL, weight, fixed_xy, mask, TOPOL_DATA, 0, 0, 0);
  if (!ce || !de) return 1;
  set_teach_params(&tp, ce, de, 0, NULL);
  set_som_params(&tp);
  trace_begin(tp.winner, trace_index, trace_diff, length);
  tp.winner = tracing_winner;
  tp.length = length; tp.alpha = alpha; tp.radius = radius;
  tp.alpha_type = alpha_type; tp.alpha_func = alpha_by_type(alpha_type);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (som_training(&tp) == NULL) return 2;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  dense_from_list(ce, codes, dim);
  close_entries(ce); close_entries(de);
  use_fixed(0); use_weights(0);
  return 0;
}

/* find_qerror (som_rout.c:678) / find_qerror2 (som_rout.c:823); returns the raw sum. */
float ref_find_qerror(const float *codes, long ncodes, int dim, int xdim, int ydim, int topol,
                      int neigh, const float *data, long ndata, const unsigned char *mask,
                      int qetype, float radius, long *trace_index, float *trace_diff)
{
  struct teach_params tp;
  struct entries *ce, *de;
  float q;
  memset(&tp, 0, sizeof tp);
  verbose(0);
  label_not_needed(1);
  ce = list_from_dense(codes, ncodes, dim, NULL, NULL, NULL, NULL, topol, neigh, xdim, ydim);
  de = list_from_dense(data, ndata, dim, NULL, NULL, NULL, mask, TOPOL_DATA, 0, 0, 0);
  set_teach_params(&tp, ce, de, 0, NULL);
  set_som_params(&tp);
  tp.radius = radius;
  trace_begin(tp.winner, trace_index, trace_diff, ndata);
  tp.winner = tracing_winner;
  q = qetype ? find_qerror2(&tp) : find_qerror(&tp);
  close_entries(ce); close_entries(de);
  return q;
}

/* kind: 1 lvq1, 2 olvq1, 3 lvq2, 4 lvq3 (lvq_rout.c:498,584,702,808).
 * lra_in / lra_out: paths whose "<first-dot-stripped>.lra" the reference reads / writes
 * for OLVQ1 (datafile.c:1030-1086); no dots in directory names. */
int ref_lvq_train(int kind, float *codes, const int *clabels, long ncodes, int dim,
                  const float *data, const int *dlabels, long ndata,
                  long length, float alpha, int alpha_type, float winlen, float epsilon,
                  char *lra_in, char *lra_out,
                  long *trace_index, float *trace_diff, double *seconds)
{
  struct teach_params tp;
  struct entries *ce, *de, *out = NULL;
  struct timespec t0, t1;
  int knn = (kind >= 3) ? 2 : 1;
  memset(&tp, 0, sizeof tp);
  verbose(0);
  label_not_needed(0);
  ce = list_from_dense(codes, ncodes, dim, clabels, NULL, NULL, NULL, TOPOL_LVQ, 0, 0, 0);
  de = list_from_dense(data, ndata, dim, dlabels, NULL, NULL, NULL, TOPOL_DATA, 0, 0, 0);
  if (!ce || !de) return 1;
  set_teach_params(&tp, ce, de, 0, NULL);
  if (knn == 2) tp.winner = find_winner_knn;          /* as lvqtrain.c:224,228 */
  trace_begin(tp.winner, trace_index, trace_diff, length * knn);
  tp.winner = tracing_winner;
  tp.length = length; tp.alpha = alpha;
  tp.alpha_type = alpha_type; tp.alpha_func = alpha_by_type(alpha_type);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  switch (kind) {
  case 1: out = lvq1_training(&tp); break;
  case 2: out = olvq1_training(&tp, lra_in, lra_out); break;
  case 3: out = lvq2_training(&tp, winlen); break;
  case 4: out = lvq3_training(&tp, epsilon, winlen); break;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  if (!out) return 2;
  dense_from_list(ce, codes, dim);
  close_entries(ce); close_entries(de);
  return 0;
}

/* find_winner_euc (knn=1, lvq_pak.c:41) / find_winner_knn (lvq_pak.c:152) over a data set.
 * index/diff are [ndata][knn]; ret[r] is the function's return value. */
int ref_winners(const float *codes, long ncodes, int dim, const float *data, long ndata,
                const unsigned char *mask, int knn, int use_knn_fn,
                long *index, float *diff, int *ret)
{
  struct entries *ce, *de;
  struct data_entry *d;
  struct winner_info *w = malloc(sizeof(*w) * (knn > 0 ? knn : 1));
  long r = 0;
  int k;
  label_not_needed(1);
  ce = list_from_dense(codes, ncodes, dim, NULL, NULL, NULL, NULL, TOPOL_LVQ, 0, 0, 0);
  de = list_from_dense(data, ndata, dim, NULL, NULL, NULL, mask, TOPOL_DATA, 0, 0, 0);
  for (d = de->entries; d; d = d->next, r++) {
    int rv = use_knn_fn ? find_winner_knn(ce, d, w, knn) : find_winner_euc(ce, d, w, knn);
    if (ret) ret[r] = rv;
    for (k = 0; k < knn; k++) {
      index[r * knn + k] = rv ? w[k].index : -2;
      diff[r * knn + k] = rv ? w[k].diff : -1.0f;
    }
  }
  free(w);
  close_entries(ce); close_entries(de);
  return 0;
}

/* vector_dist_euc (lvq_pak.c:291) for row pairs. */
float ref_vector_dist(const float *a, const unsigned char *ma, const float *b,
                      const unsigned char *mb, int dim)
{
  struct data_entry ea, eb;
  memset(&ea, 0, sizeof ea); memset(&eb, 0, sizeof eb);
  ea.points = (float *)a; ea.mask = (char *)ma;
  eb.points = (float *)b; eb.mask = (char *)mb;
  return vector_dist_euc(&ea, &eb, dim);
}

/* adapt_vector (lvq_pak.c:339) on one row. */
void ref_adapt_vector(float *c, const float *x, const unsigned char *mx, int dim, float alpha)
{
  struct data_entry ec, ex;
  memset(&ec, 0, sizeof ec); memset(&ex, 0, sizeof ex);
  ec.points = c; ex.points = (float *)x; ex.mask = (char *)mx;
  adapt_vector(&ec, &ex, dim, alpha);
}

float ref_alpha(int type, long iter, long length, float alpha)
{
  return alpha_by_type(type)(iter, length, alpha);
}

float ref_mapdist(int topol, int bx, int by, int tx, int ty)
{
  return get_mapdistf(topol)(bx, by, tx, ty);
}

/* the -rand shuffle: init_random(seed) then randomize_entry_order (datafile.c:1152) on a
 * list of n entries; perm[i] = original position of the row now at position i. */
int ref_shuffle_perm(long n, int seed, long *perm)
{
  struct entries *e = alloc_entries();
  struct data_entry *d, *prev = NULL;
  long r;
  e->dimension = 1;
  for (r = 0; r < n; r++) {
    d = alloc_entry(e);
    d->points[0] = (float)r;
    d->weight = 0;
    d->lab.label = 0;
    /* stash the original index where no rounding can touch it */
    d->fixed = malloc(sizeof(struct fixpoint));
    d->fixed->xfix = (short)(r & 0x7fff);
    d->fixed->yfix = (short)(r >> 15);
    if (prev) prev->next = d; else e->entries = d;
    prev = d;
  }
  init_random(seed);
  e->entries = randomize_entry_order(e->entries);
  for (r = 0, d = e->entries; d; d = d->next, r++)
    perm[r] = ((long)d->fixed->yfix << 15) | d->fixed->xfix;
  e->num_entries = e->num_loaded = n;
  close_entries(e);
  return 0;
}

/* the LCG itself (lvq_pak.c:459-473) */
void ref_rand_seq(int seed, long count, long *out)
{
  long i;
  osrand(seed);
  for (i = 0; i < count; i++) out[i] = orand();
}

/* randinit_codes (som_rout.c:34) on in-memory data, after init_random(seed). */
int ref_randinit(const float *data, long ndata, int dim, int topol, int neigh, int xdim, int ydim,
                 int seed, float *codes_out)
{
  struct entries *de, *ce;
  label_not_needed(1);
  de = list_from_dense(data, ndata, dim, NULL, NULL, NULL, NULL, TOPOL_DATA, 0, 0, 0);
  init_random(seed);
  ce = randinit_codes(de, topol, neigh, xdim, ydim);
  if (!ce) return 1;
  dense_from_list(ce, codes_out, dim);
  close_entries(ce); close_entries(de);
  return 0;
}
