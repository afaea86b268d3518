/* som_lvq_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A CPU restatement, over dense row-major arrays, of the arithmetic on the
 * SOM/LVQ training hot path of SOM_PAK/LVQ_PAK 3.2 (hynde/som_lvq_pak).  It is
 * the checker the HIP path is compared against; it is never linked into, called
 * from or shipped with the product (only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so).
 *
 * PARITY PINNED: every function here is checked bit-for-bit against the real
 * reference (oracle/_ref, built by oracle/Makefile from /root/reference) by
 * tests/test_oracle_vs_ref.py, and against the committed fixtures the reference
 * produced (tests/golden/, made by tests/golden/make_golden.py).
 *
 * Build with -ffp-contract=off: the reference's results are those of separate
 * fp32 mul and add.  Each function cites the reference lines it
 * follows.  Codebook row k here == list position k in the reference
 * (datafile.c:781,836), which for a map is unit (k % xdim, k / xdim)
 * (som_rout.c:641-642).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "som_lvq_oracle.h"

#define MASKED(m, i) ((m) != NULL && (m)[i] != 0)

/* ------------------------------------------------------------------ */
/* vector kernels                                                       */
/* ------------------------------------------------------------------ */

/* squared distance of one code row to x, summed left to right in fp32 over the
 * sample's unmasked components (lvq_pak.c:59-73).  The reference's early exit at
 * :72 cannot change which row wins or the winner's value (partial sums only
 * grow), so it is not restated.  *nmasked gets the count of skipped components. */
static float sqdist_row(const float *c, const float *x, const unsigned char *mask, int d,
                        int *nmasked)
{
  float acc = 0.0f;
  int skipped = 0;
  for (int i = 0; i < d; i++) {
    if (MASKED(mask, i)) { skipped++; continue; }
    float t = c[i] - x[i];
    acc += t * t;
  }
  if (nmasked) *nmasked = skipped;
  return acc;
}

/* find_winner_euc, lvq_pak.c:41-94.  Start value FLT_MAX, strict '<' so the
 * first (lowest-index) minimum stays (:56,:79).  Returns 0 when every component
 * of the sample is masked (:75), else 1; index -1 if nothing beat FLT_MAX. */
int orc_find_winner_euc(const float *codes, long n, int d, const float *x,
                        const unsigned char *mask, long *index, float *diff)
{
  float best = FLT_MAX;
  *index = -1;
  *diff = -1.0f;
  for (long k = 0; k < n; k++) {
    int skipped;
    float v = sqdist_row(codes + k * (long)d, x, mask, d, &skipped);
    if (skipped == d) return 0;
    if (v < best) { best = v; *index = k; *diff = v; }
  }
  return 1;
}

/* find_winner_knn, lvq_pak.c:152-221.  knn==1 delegates to find_winner_euc
 * (:160).  Otherwise a sorted list of the knn smallest; a new value advances past
 * entries it is strictly greater than (:197), so among equal values the LATER
 * row sits first -- the opposite tie rule to find_winner_euc. */
int orc_find_winner_knn(const float *codes, long n, int d, const float *x,
                        const unsigned char *mask, int knn, long *index, float *diff)
{
  if (knn == 1) return orc_find_winner_euc(codes, n, d, x, mask, index, diff);
  for (int j = 0; j < knn; j++) { index[j] = -1; diff[j] = FLT_MAX; }
  for (long k = 0; k < n; k++) {
    int skipped;
    float v = sqdist_row(codes + k * (long)d, x, mask, d, &skipped);
    if (skipped == d) return 0;
    int pos = 0;
    while (pos < knn && v > diff[pos]) pos++;
    if (pos < knn) {
      for (int j = knn - 1; j > pos; j--) { diff[j] = diff[j - 1]; index[j] = index[j - 1]; }
      diff[pos] = v;
      index[pos] = k;
    }
  }
  return knn;
}

/* vector_dist_euc, lvq_pak.c:291-316: both masks honoured, -1 if nothing left,
 * root taken in double and narrowed. */
float orc_vector_dist_euc(const float *a, const unsigned char *ma, const float *b,
                          const unsigned char *mb, int d)
{
  float acc = 0.0f;
  int skipped = 0;
  for (int i = 0; i < d; i++) {
    if (MASKED(ma, i) || MASKED(mb, i)) { skipped++; continue; }
    float t = a[i] - b[i];
    acc += t * t;
  }
  if (skipped == d) return -1.0f;
  return (float)sqrt((double)acc);
}

/* adapt_vector, lvq_pak.c:339-351: c += alpha*(x - c), three fp32 roundings. */
void orc_adapt_vector(float *c, const float *x, const unsigned char *mask, int d, float alpha)
{
  for (int i = 0; i < d; i++) {
    if (MASKED(mask, i)) continue;
    float step = alpha * (x[i] - c[i]);
    c[i] = c[i] + step;
  }
}

/* ------------------------------------------------------------------ */
/* lattice distance, schedules                                          */
/* ------------------------------------------------------------------ */

/* hexa_dist, som_rout.c:434-455: odd rows are shifted half a unit. */
float orc_hexa_dist(int bx, int by, int tx, int ty)
{
  float dx = (float)(bx - tx);
  if (((by - ty) % 2) != 0)
    dx = (float)((by % 2) == 0 ? (double)dx - 0.5 : (double)dx + 0.5);
  float r = dx * dx;
  float dy = (float)(by - ty);
  r = (float)((double)r + 0.75 * (double)dy * (double)dy);
  return (float)sqrt((double)r);
}

/* rect_dist, som_rout.c:457-468 */
float orc_rect_dist(int bx, int by, int tx, int ty)
{
  float dx = (float)(bx - tx);
  float r = dx * dx;
  float dy = (float)(by - ty);
  r = r + dy * dy;
  return (float)sqrt((double)r);
}

float orc_mapdist(int topol, int bx, int by, int tx, int ty)   /* get_mapdistf, som_rout.c:893 */
{
  return topol == ORC_TOPOL_RECT ? orc_rect_dist(bx, by, tx, ty) : orc_hexa_dist(bx, by, tx, ty);
}

/* linear_alpha lvq_pak.c:903-906, inverse_t_alpha lvq_pak.c:914-921 (constant 100.0 :909) */
float orc_alpha(int type, long iter, long length, float alpha)
{
  if (type == ORC_ALPHA_INVERSE_T) {
    float c = (float)length / 100.0f;
    return alpha * c / (c + (float)iter);
  }
  return alpha * (float)(length - iter) / (float)length;
}

/* som_rout.c:615 -- evaluated in double, narrowed on assignment */
float orc_som_radius(long iter, long length, float radius)
{
  return (float)(1.0 + ((double)radius - 1.0) * (double)(float)(length - iter)
                           / (double)(float)length);
}

/* som_rout.c:622-624 */
float orc_weighted_alpha(float talp, float weight)
{
  return (float)(1.0 - (double)(float)pow(1.0 - (double)talp, (double)weight));
}

/* gaussian_adapt's factor, som_rout.c:541-542: -dd*dd in fp32, the rest in double */
float orc_gaussian_h(float dd, float radius, float alpha)
{
  float neg = -dd * dd;
  return alpha * (float)exp((double)neg / (2.0 * (double)radius * (double)radius));
}

/* ------------------------------------------------------------------ */
/* SOM epoch loop                                                        */
/* ------------------------------------------------------------------ */

/* bubble_adapt som_rout.c:472-506 / gaussian_adapt som_rout.c:511-549 */
static void neighbourhood_adapt(float *codes, long n, int d, int xdim, int topol, int neigh,
                                const float *x, const unsigned char *mask, int bx, int by,
                                float radius, float alpha)
{
  for (long k = 0; k < n; k++) {
    int tx = (int)(k % xdim), ty = (int)(k / xdim);
    float dd = orc_mapdist(topol, bx, by, tx, ty);
    if (neigh == ORC_NEIGH_GAUSSIAN)
      orc_adapt_vector(codes + k * (long)d, x, mask, d, orc_gaussian_h(dd, radius, alpha));
    else if (dd <= radius)
      orc_adapt_vector(codes + k * (long)d, x, mask, d, alpha);
  }
}

/* som_training, som_rout.c:556-671.
 *
 * batch == 1 is the reference: sample t+1's winner search sees the codebook
 * already adapted by sample t.  batch > 1 is the *mini-batch schedule* of the HIP
 * engine's throughput mode, restated here so it has an exact oracle too: the
 * winners of samples [s, s+batch) are all found against the codebook as it stood
 * before sample s; their neighbourhood updates are then applied one after the
 * other in sample order with the reference's own per-sample radius and alpha.
 *
 * trace_index/trace_diff (may be NULL) receive one entry per iteration: the
 * winner index and squared distance, -2 for a skipped (fully masked) sample,
 * -3 for a fixed-point sample (no search, som_rout.c:628-632). */
int orc_som_training(float *codes, long n, int d, int xdim, int ydim, int topol, int neigh,
                     const float *data, long ndata, const short *weight, const short *fixed_xy,
                     const unsigned char *mask, long length, float alpha, float radius,
                     int alpha_type, int fixed_on, int weights_on, long batch,
                     long *trace_index, float *trace_diff)
{
  (void)ydim;
  if (batch < 1) batch = 1;
  long *bidx = (long *)malloc(sizeof(long) * (size_t)batch);
  float *bdiff = (float *)malloc(sizeof(float) * (size_t)batch);
  if (!bidx || !bdiff) return 1;

  for (long s = 0; s < length; s += batch) {
    long cnt = (length - s < batch) ? length - s : batch;
    /* winners against the frozen codebook */
    for (long j = 0; j < cnt; j++) {
      long le = s + j, row = le % ndata;                 /* data wraps, som_rout.c:602-610 */
      const unsigned char *m = mask ? mask + row * (long)d : NULL;
      if (fixed_on && fixed_xy && fixed_xy[2 * row] >= 0) {
        bidx[j] = -3; bdiff[j] = -1.0f;
      } else if (!orc_find_winner_euc(codes, n, d, data + row * (long)d, m, &bidx[j], &bdiff[j])) {
        bidx[j] = -2; bdiff[j] = -1.0f;
      }
      if (trace_index) trace_index[le] = bidx[j];
      if (trace_diff) trace_diff[le] = bdiff[j];
    }
    /* in-order updates */
    for (long j = 0; j < cnt; j++) {
      long le = s + j, row = le % ndata;
      const unsigned char *m = mask ? mask + row * (long)d : NULL;
      float trad = orc_som_radius(le, length, radius);
      float talp = orc_alpha(alpha_type, le, length, alpha);
      float w = weight ? (float)weight[row] : 0.0f;
      if (w > 0.0f && weights_on) talp = orc_weighted_alpha(talp, w);
      int bx, by;
      if (bidx[j] == -3) { bx = fixed_xy[2 * row]; by = fixed_xy[2 * row + 1]; }
      else if (bidx[j] == -2) continue;                  /* skip_teach, som_rout.c:635-640 */
      else { bx = (int)(bidx[j] % xdim); by = (int)(bidx[j] / xdim); }
      neighbourhood_adapt(codes, n, d, xdim, topol, neigh, data + row * (long)d, m, bx, by,
                          trad, talp);
    }
  }
  free(bidx); free(bdiff);
  return 0;
}

/* ------------------------------------------------------------------ */
/* LVQ epoch loops                                                       */
/* ------------------------------------------------------------------ */

/* lvq1_training lvq_rout.c:498-577, olvq1_training :584-697, lvq2_training
 * :702-803, lvq3_training :808-916.  Labels are the FIRST label of each entry
 * (labels.h:44).  talpha: OLVQ1's per-code rates, in/out, length n; the caller
 * initialises it the way :614-627 does (all = alpha, or .lra contents).
 * `alpha` is also OLVQ1's clamp (:671).  LVQ2/3 use find_winner_knn with k=2
 * (lvqtrain.c:224,228).  The trace has knn entries per iteration. */
int orc_lvq_training(int kind, float *codes, const int *clabels, long n, int d,
                     const float *data, const int *dlabels, long ndata, long length,
                     float alpha, int alpha_type, float winlen, float epsilon,
                     float *talpha, long *trace_index, float *trace_diff)
{
  int knn = (kind == ORC_LVQ2 || kind == ORC_LVQ3) ? 2 : 1;
  for (long le = 0; le < length; le++) {
    long row = le % ndata;
    const float *x = data + row * (long)d;
    int want = dlabels[row];
    long idx[2] = {-1, -1};
    float dist[2] = {-1.0f, -1.0f};
    orc_find_winner_knn(codes, n, d, x, NULL, knn, idx, dist);
    for (int k = 0; k < knn; k++) {
      if (trace_index) trace_index[le * knn + k] = idx[k];
      if (trace_diff) trace_diff[le * knn + k] = dist[k];
    }
    /* nothing beat FLT_MAX (codes pushed to infinity by a too large rate): the reference dereferences
     * a NULL winner here (lvq_rout.c:545); the checker stops and says so */
    if (idx[0] < 0 || (knn == 2 && idx[1] < 0)) return 2;
    if (kind == ORC_LVQ1) {
      float a = orc_alpha(alpha_type, le, length, alpha);
      float *c = codes + idx[0] * (long)d;
      orc_adapt_vector(c, x, NULL, d, clabels[idx[0]] == want ? a : -a);     /* :552-555 */
    } else if (kind == ORC_OLVQ1) {
      long k = idx[0];
      float *c = codes + k * (long)d;
      if (clabels[k] == want) {                                            /* :658-664 */
        orc_adapt_vector(c, x, NULL, d, talpha[k]);
        talpha[k] = talpha[k] / (1 + talpha[k]);
      } else {                                                             /* :665-673 */
        orc_adapt_vector(c, x, NULL, d, -talpha[k]);
        talpha[k] = talpha[k] / (1 - talpha[k]);
        if (talpha[k] > alpha) talpha[k] = alpha;
      }
    } else {
      float a = orc_alpha(alpha_type, le, length, alpha);
      long first = idx[0], second = idx[1];
      int l1 = clabels[first], l2 = clabels[second];
      if (l1 != l2) {
        if (l1 == want || l2 == want) {
          /* window on SQUARED distances, all fp32 (:770, :876) */
          if ((dist[0] / dist[1]) > ((1 - winlen) / (1 + winlen))) {
            if (l2 == want) { long t = first; first = second; second = t; }
            orc_adapt_vector(codes + first * (long)d, x, NULL, d, a);
            orc_adapt_vector(codes + second * (long)d, x, NULL, d, -a);
          }
        }
      } else if (kind == ORC_LVQ3 && l1 == want) {                          /* :890-895 */
        orc_adapt_vector(codes + first * (long)d, x, NULL, d, a * epsilon);
        orc_adapt_vector(codes + second * (long)d, x, NULL, d, a * epsilon);
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* read-only scans                                                       */
/* ------------------------------------------------------------------ */

int orc_winners(const float *codes, long n, int d, const float *data, long ndata,
                const unsigned char *mask, int knn, int use_knn_fn,
                long *index, float *diff, int *ret)
{
  for (long r = 0; r < ndata; r++) {
    const unsigned char *m = mask ? mask + r * (long)d : NULL;
    int rv = use_knn_fn
                 ? orc_find_winner_knn(codes, n, d, data + r * (long)d, m, knn, index + r * knn,
                                       diff + r * knn)
                 : orc_find_winner_euc(codes, n, d, data + r * (long)d, m, index + r * knn,
                                       diff + r * knn);
    if (!rv)
      for (int k = 0; k < knn; k++) { index[r * knn + k] = -2; diff[r * knn + k] = -1.0f; }
    if (ret) ret[r] = rv;
  }
  return 0;
}

/* the accumulation of find_qerror, som_rout.c:698-715: a FLOAT running sum of
 * double square roots, in data order, empty samples skipped (:712). */
float orc_qerror_from_diffs(const float *diff, const int *ret, long ndata)
{
  float q = 0.0f;
  for (long r = 0; r < ndata; r++) {
    if (ret && !ret[r]) continue;
    q = (float)((double)q + sqrt((double)diff[r]));
  }
  return q;
}

/* find_qerror, som_rout.c:678-731; returns the raw sum (qerror.c:118 divides). */
float orc_find_qerror(const float *codes, long n, int d, const float *data, long ndata,
                      const unsigned char *mask, long *index, float *diff)
{
  long *idx = index ? index : (long *)malloc(sizeof(long) * (size_t)ndata);
  float *df = diff ? diff : (float *)malloc(sizeof(float) * (size_t)ndata);
  int *rv = (int *)malloc(sizeof(int) * (size_t)ndata);
  orc_winners(codes, n, d, data, ndata, mask, 1, 0, idx, df, rv);
  float q = orc_qerror_from_diffs(df, rv, ndata);
  if (!index) free(idx);
  if (!diff) free(df);
  free(rv);
  return q;
}

/* find_qerror2, som_rout.c:823-891 with bubble_qerror :734-773 / gaussian_qerror
 * :776-819: neighbourhood-weighted sum of squared *rooted* distances. */
float orc_find_qerror2(const float *codes, long n, int d, int xdim, int topol, int neigh,
                       const float *data, long ndata, const unsigned char *mask, float radius)
{
  float total = 0.0f;
  for (long r = 0; r < ndata; r++) {
    const float *x = data + r * (long)d;
    const unsigned char *m = mask ? mask + r * (long)d : NULL;
    long bi; float bd;
    if (!orc_find_winner_euc(codes, n, d, x, m, &bi, &bd)) continue;
    int bx = (int)(bi % xdim), by = (int)(bi / xdim);
    float q = 0.0f;
    for (long k = 0; k < n; k++) {
      float dd = orc_mapdist(topol, bx, by, (int)(k % xdim), (int)(k / xdim));
      if (neigh == ORC_NEIGH_GAUSSIAN) {
        float neg = -dd * dd;
        float h = (float)exp((double)neg / (2.0 * (double)radius * (double)radius));
        float dv = orc_vector_dist_euc(codes + k * (long)d, NULL, x, m, d);
        q += h * dv * dv;
      } else if (dd <= radius) {
        float dv = orc_vector_dist_euc(codes + k * (long)d, NULL, x, m, d);
        q += dv * dv;
      }
    }
    total += q;
  }
  return total;
}

/* the count behind compute_accuracy, accuracy.c:80-113; correct[r] = the 1/0 of -cfout */
long orc_accuracy(const float *codes, const int *clabels, long n, int d, const float *data,
                  const int *dlabels, long ndata, unsigned char *correct)
{
  long hits = 0;
  for (long r = 0; r < ndata; r++) {
    long bi; float bd;
    orc_find_winner_euc(codes, n, d, data + r * (long)d, NULL, &bi, &bd);
    int ok = (clabels[bi] == dlabels[r]);
    if (correct) correct[r] = (unsigned char)ok;
    hits += ok;
  }
  return hits;
}

/* the per-unit label histogram of find_labels, vcal.c:99-129: hits[unit*nlabels+label];
 * label 0 (= LABEL_EMPTY, labels.h:26) and skipped samples add nothing. */
void orc_unit_hits(const float *codes, long n, int d, const float *data, const int *dlabels,
                   long ndata, const unsigned char *mask, int nlabels, long *hits)
{
  memset(hits, 0, sizeof(long) * (size_t)n * (size_t)nlabels);
  for (long r = 0; r < ndata; r++) {
    long bi; float bd;
    const unsigned char *m = mask ? mask + r * (long)d : NULL;
    if (!orc_find_winner_euc(codes, n, d, data + r * (long)d, m, &bi, &bd)) continue;
    if (dlabels[r] != 0) hits[bi * nlabels + dlabels[r]]++;
  }
}

/* ------------------------------------------------------------------ */
/* RNG, shuffle, randinit (the steps just before the path)              */
/* ------------------------------------------------------------------ */

static unsigned long lcg_state = 1;                         /* lvq_pak.c:459 */
void orc_srand(int seed) { lcg_state = (unsigned long)seed; }  /* osrand :465 */
long orc_rand(void)                                          /* orand :470 */
{
  lcg_state = (lcg_state * 23UL) % 100000001UL;
  return (long)(int)(lcg_state % 32767UL);
}

/* randomize_entry_order, datafile.c:1152-1188, after init_random(seed) lvq_pak.c:478
 * (seed 0 means time() there; not supported here).  perm[i] = original row now at i. */
void orc_shuffle_perm(long n, int seed, long *perm)
{
  orc_srand(seed);
  for (long i = 0; i < n; i++) perm[i] = i;
  for (long i = 0; i < n; i++) {
    long j = orc_rand() % n;
    long t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
}

/* randinit_codes, som_rout.c:34-162 (no masks): uniform in the data's bounding box.
 * Note the reference seeds its running maximum with FLT_MIN (smallest positive), :109. */
int orc_randinit(const float *data, long ndata, int d, int xdim, int ydim, int seed,
                 float *codes_out)
{
  float *lo = (float *)malloc(sizeof(float) * (size_t)d), *hi = (float *)malloc(sizeof(float) * (size_t)d);
  if (!lo || !hi) return 1;
  for (int i = 0; i < d; i++) { hi[i] = FLT_MIN; lo[i] = FLT_MAX; }
  for (long r = 0; r < ndata; r++)
    for (int i = 0; i < d; i++) {
      float v = data[r * (long)d + i];
      if (hi[i] < v) hi[i] = v;
      if (lo[i] > v) lo[i] = v;
    }
  orc_srand(seed);
  for (long k = 0; k < (long)xdim * ydim; k++)
    for (int i = 0; i < d; i++) {
      double u = (double)(float)orc_rand() / 32768.0;
      codes_out[k * (long)d + i] = (float)((double)lo[i] + (double)(hi[i] - lo[i]) * u);
    }
  free(lo); free(hi);
  return 0;
}
