/* somhip.h -- C ABI of the MI355X (gfx950) SOM/LVQ training engine.
 *
 * This is the drop-in boundary for the hot path of SOM_PAK/LVQ_PAK 3.2
 * (hynde/som_lvq_pak): best-matching-unit search + codebook update inside the
 * vsom / lvqtrain epoch loops, and the read-only winner scans of qerror /
 * accuracy / vcal.  Plain C: opaque handles, plain pointers and sizes, int status
 * (0 = ok, nonzero = error; text via somhip_last_error()).  Nothing here aborts and no
 * C++ exception crosses this boundary; a failing call leaves a message and returns
 * nonzero, the way the reference's training functions return NULL and print to stderr
 * (som_rout.c:576-596).
 *
 * One host thread per engine (the reference is single-threaded and not re-entrant,
 * SURVEY.md 8b); one engine per process per GPU.
 *
 * Each entry point cites the reference interface it replaces (file:line in
 * hynde/som_lvq_pak).  INTEGRATION.md shows the reference-side binding.
 */
#ifndef SOMHIP_H
#define SOMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOMHIP_VERSION 1

/* ids equal to the reference's (lvq_pak.h:209-224) */
enum { SOMHIP_TOPOL_LVQ = 2, SOMHIP_TOPOL_HEXA = 3, SOMHIP_TOPOL_RECT = 4 };
enum { SOMHIP_NEIGH_BUBBLE = 1, SOMHIP_NEIGH_GAUSSIAN = 2 };
enum { SOMHIP_ALPHA_LINEAR = 1, SOMHIP_ALPHA_INVERSE_T = 2 };
enum { SOMHIP_LVQ1 = 1, SOMHIP_OLVQ1 = 2, SOMHIP_LVQ2 = 3, SOMHIP_LVQ3 = 4 };

/* winner tie rules: FIRST = find_winner_euc (lowest index among equal distances,
 * lvq_pak.c:79); KNN = find_winner_knn with knn >= 2 (later row first, lvq_pak.c:197) */
enum { SOMHIP_TIE_FIRST = 0, SOMHIP_TIE_KNN = 1 };

typedef struct somhip_engine somhip_engine;
typedef struct somhip_codebook somhip_codebook;   /* replaces the `codes` entries list, lvq_pak.h:89-113 */
typedef struct somhip_dataset somhip_dataset;     /* replaces the `data` entries list (one -buffer worth) */

const char *somhip_last_error(void);
int somhip_version(void);

/* GPUs visible to this process (a multi-GPU host: one process per GPU, rank r on device r % count) */
int  somhip_device_count(int *count);

/* ---- engine ---- */
int  somhip_engine_create(int device, somhip_engine **out);
/* Destroying an engine releases the device memory of every codebook / data set created on it; those handles
 * stay valid for their own destroy call (any order of the destroy calls is fine), every other call on them
 * fails with "the engine of this handle was destroyed". */
void somhip_engine_destroy(somhip_engine *e);
/* the HIP stream all work of this engine is enqueued on (a hipStream_t) */
void *somhip_engine_stream(somhip_engine *e);
int  somhip_engine_sync(somhip_engine *e);

/* Winner-search implementation for runs of samples (mini-batch training, find_winners k=1):
 *   SOMHIP_SCAN_DIRECT  direct-form fp32 scan on the vector ALU (the reference's arithmetic)
 *   SOMHIP_SCAN_MFMA    fp32-MFMA distance GEMM as a pre-filter with a rigorous error bound +
 *                       exact re-rank of the surviving rows by the direct-form arithmetic
 *   SOMHIP_SCAN_MFMA_BF16  the same with the GEMM on the bf16 matrix pipe (operands split
 *                       hi + lo, three MFMAs per K-step; wider bound, same exact re-rank)
 * All return bit-identical winners; the MFMA forms apply where there are no masks (default:
 * SOMHIP_SCAN_MFMA_BF16). */
enum { SOMHIP_SCAN_DIRECT = 0, SOMHIP_SCAN_MFMA = 1, SOMHIP_SCAN_MFMA_BF16 = 2 };
int  somhip_engine_set_scan_mode(somhip_engine *e, int mode);
/* How a mini-batch (batch > 1) applies its neighbourhood updates:
 *   SOMHIP_UPDATE_EXACT  adapt_vector's arithmetic, hit after hit (lvq_pak.c:348-349: sub, mul, add, each rounded):
 *                        bit-identical to orc_som_training(batch) in oracle/ -- the default
 *   SOMHIP_UPDATE_GEMM   the same affine map of every unit, c' = P c + sum_j w_j x_j, evaluated as a matrix product
 *                        on the fp32 matrix pipe (kernels/som_update_gemm.hpp); same result up to fp32 rounding of a
 *                        sum instead of a chain, hits whose weight has decayed below 2^-24 skipped; no masked
 *                        components, dim a multiple of 128; gaussian neighbourhoods (maps up to 1024 x 1024): every
 *                        unit's rate per sample as a dense weight matrix, the rates through the fp32 exp
 *                        (otherwise the exact kernels run)
 * batch == 1 (the reference's online algorithm) is not affected.  Environment: SOMHIP_UPDATE_MODE=gemm|exact. */
enum { SOMHIP_UPDATE_EXACT = 0, SOMHIP_UPDATE_GEMM = 1 };
int  somhip_engine_set_update_mode(somhip_engine *e, int mode);
/* cumulative re-rank statistics of the MFMA path since engine creation:
 * out[0] = row groups re-ranked, out[1] = rows re-ranked, out[2] = max groups for one sample,
 * out[3] = samples searched, out[4] = (row, iteration) updates applied by mini-batch runs,
 * out[5] = (row group, iteration) pairs with at least one update, out[6] = list entries ((row group, iteration)
 * pairs) the GEMM-form update walked (it stops where the weights have decayed away), out[7] = (row group, sample) pairs
 * that survived level 1 of the two-level pre-filter and went through the three-product level 2 */
int  somhip_scan_stats(somhip_engine *e, uint64_t out[8]);

/* diagnostics (tests): run only the pre-filter of the current scan mode on data rows
 * [first, first+count) and return its raw outputs: wmin[ngroups][*bpad] (group minima of
 * s~ = ||c||^2 - 2<c,x>) and tau[count].  wmin must hold ngroups * (count rounded up to 32). */
int  somhip_debug_prefilter(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                            float *wmin, float *tau, int64_t *bpad);

/* ---- codebook mirror -------------------------------------------------------
 * rows: host, row-major [n_rows][dim] fp32, row k = list position k of the
 * reference's codebook (datafile.c:781,836) = map unit (k % xdim, k / xdim)
 * (som_rout.c:641-642).  labels: first label of each row (labels.h:44) or NULL.
 * For a row-sharded codebook (multi-GPU) this process holds global rows
 * [row_offset, row_offset + n_rows) of n_global; single GPU: row_offset 0,
 * n_global == n_rows. */
int  somhip_codebook_create(somhip_engine *e, const float *rows, const int32_t *labels,
                            int64_t n_rows, int dim, int topol, int neigh, int xdim, int ydim,
                            int64_t row_offset, int64_t n_global, somhip_codebook **out);
/* Interleaved shards of a map (multi-GPU SOM training): the map is cut into 8x8-unit patches numbered
 * row-major and shard s of S holds patches s, s+S, s+2S, ...  A neighbourhood (som_rout.c:472-549)
 * of any radius around any winner then covers about 1/S of its units on every shard, so the update
 * work of a batch is the same on all ranks; contiguous blocks of rows load the ranks that hold the
 * middle of the map up to 1.4x the mean.  Map sides must be multiples of 8.
 * somhip_shard_units: the global unit index (k of the comment above) of every row of the shard, in the
 * order in which create_interleaved / download / upload exchange its rows (units == NULL: count only).
 * Keys, traces and winners carry global unit indices as with contiguous shards. */
int  somhip_shard_units(int xdim, int ydim, int shard_index, int shard_count, int64_t *units, int64_t *n_units);
int  somhip_codebook_create_interleaved(somhip_engine *e, const float *rows, int64_t n_rows, int dim,
                                        int topol, int neigh, int xdim, int ydim, int shard_index,
                                        int shard_count, somhip_codebook **out);
/* device -> host gather of the rows (what save_entries / save_snapshot need,
 * datafile.c:353, lvq_pak.c:665) */
int  somhip_codebook_download(somhip_codebook *cb, float *rows);
int  somhip_codebook_upload(somhip_codebook *cb, const float *rows);
void somhip_codebook_destroy(somhip_codebook *cb);

/* ---- data mirror -----------------------------------------------------------
 * rows row-major [n_rows][dim]; optional per-row arrays (NULL = absent):
 *   mask    [n_rows][dim] nonzero = component ignored (data_entry.mask, lvq_pak.h:84)
 *   labels  first label per row                      (labels.h:44)
 *   weight  data_entry.weight                        (lvq_pak.h:81)
 *   fixed_xy [n_rows][2], -1 = none                  (struct fixpoint, lvq_pak.h:65) */
int  somhip_dataset_create(somhip_engine *e, const float *rows, int64_t n_rows, int dim,
                           const uint8_t *mask, const int32_t *labels, const int16_t *weight,
                           const int16_t *fixed_xy, somhip_dataset **out);
/* same, but `dev_rows` already lives in this GPU's memory (row-major fp32) and is
 * used in place, not copied (bench / streaming ingest) */
int  somhip_dataset_wrap_device(somhip_engine *e, const float *dev_rows, int64_t n_rows, int dim,
                                somhip_dataset **out);
/* a data set generated in place: rows [first_row, first_row + n_rows) of the seeded Gaussian-mixture stream
 * (k_centres centres 4z, row = centre[k(row)] + z, z = sum of twelve 16-bit uniforms - 6: counter-based
 * and exact, so it equals the host form bit for bit -- `-din gen:k=..,dim=..,n=..,seed=..` in the C tools,
 * pak_gen_row in som_lvq_pak_amd/host/paklib.c).  centres (host, may be NULL) receives each row's mixture id,
 * which also becomes the data set's labels.  No host copy, no PCIe: what the C4/C5-sized runs need. */
int  somhip_dataset_generate(somhip_engine *e, uint64_t seed, int k_centres, int dim, int64_t first_row,
                             int64_t n_rows, int32_t *centres, somhip_dataset **out);
/* rows [first, first + count) of the mirror back to the host (generated data sets: picking initial codes) */
int  somhip_dataset_download_rows(somhip_dataset *ds, int64_t first, int64_t count, float *rows);
void somhip_dataset_destroy(somhip_dataset *ds);

/* ---- winner scans: WINNER_FUNCTION over a run of samples (lvq_pak.h:146) -----
 * find_winner_euc (lvq_pak.c:41) when tie == SOMHIP_TIE_FIRST and knn == 1,
 * find_winner_knn (lvq_pak.c:152) when tie == SOMHIP_TIE_KNN (1 <= knn <= 8).
 * Samples are data rows [first, first+count).  Outputs are host arrays
 * [count][knn]: index = global row (or -1: nothing beat FLT_MAX), diff = SQUARED
 * distance exactly as the reference's fp32 left-to-right sum gives it; ret[i] = the
 * function's return value (knn, or 0 = every component masked). ret may be NULL. */
int  somhip_find_winners(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                         int knn, int tie, int32_t *index, float *diff, int32_t *ret);

/* ---- som_training (som_rout.c:556-671) --------------------------------------
 * Runs iterations [start_iter, start_iter+count) of a schedule of `length`
 * iterations; iteration le uses data row (data_first + (le - start_iter)) % n_rows.
 * batch == 1: the reference's strictly online algorithm, bit-exact.
 * batch  > 1: mini-batch schedule -- the winners of each run of `batch` iterations
 *             are found against the codebook as it stood before the run, then the
 *             neighbourhood updates are applied in iteration order (exact oracle:
 *             orc_som_training(batch) in oracle/).
 * batch == SOMHIP_BATCH_AUTO: the mini-batch schedule with the engine's own batch sizes (somhip_som_auto_batch):
 *             long batches while the end state still forgets them, short ones after -- or batch 1 where that rule
 *             has not been measured against the online engine (small maps, short runs).  Measured at configs[3]'s
 *             real length on three seed pairs (profiles/r03_conformity_*.jsonl).
 * trace_index/trace_diff (host, [count], may be NULL): winner of every iteration;
 * -2 = skipped (sample fully masked), -3 = fixed-point sample (no search). */
#define SOMHIP_BATCH_AUTO (-1)
typedef struct somhip_som_params {
  int64_t length;        /* teach_params.length  (lvq_pak.h:198) */
  float   alpha;         /* teach_params.alpha   (lvq_pak.h:197) */
  float   radius;        /* teach_params.radius  (lvq_pak.h:196) */
  int32_t alpha_type;    /* teach_params.alpha_type (lvq_pak.h:189) */
  int32_t use_fixed;     /* use_fixed(-1)   (lvq_pak.c:508) */
  int32_t use_weights;   /* use_weights(-1) (lvq_pak.c:519) */
  int64_t batch;
  int64_t start_iter, count, data_first;
} somhip_som_params;
int  somhip_som_train(somhip_codebook *cb, somhip_dataset *ds, const somhip_som_params *p,
                      int32_t *trace_index, float *trace_diff);
/* The batch of the SOMHIP_BATCH_AUTO schedule that holds iteration `iter`: [*batch_start, *batch_start + *batch_len).
 * p: length, alpha, alpha_type, radius of the run (teach_params, lvq_pak.h:186-204); n_units = xdim * ydim of the WHOLE map
 * (also on a shard), topol / neigh as in the codebook.  The schedule is a rule in (n_units, radius(t), alpha(t)) --
 * csrc/host_som.inc, som_auto_plan: 32768 iterations per batch while what a batch leaves in the map is forgotten again by
 * the end of the run (the sum F(t) of alpha(t') x the share of the map one sample teaches over the rest of the run is
 * >= 64), 8192 after (configs[3]: 32768 up to iteration 8 486 912 of 10 M, then 8192; where the line lies was measured
 * on three seed pairs).  Where the rule is not vouched for by a measurement
 * against the online engine -- maps below 16384 units, runs whose long phase holds fewer than 64 long batches -- every
 * batch is ONE iteration: `auto` is then the reference's own online schedule (bit-exact).
 * A host that drives somhip_batch_winner_keys / somhip_som_batch_update itself (one process per GPU) asks this
 * function for its batch boundaries, so that every rank cuts the run the same way.  Plain host arithmetic: no GPU needed. */
int  somhip_som_auto_batch(const somhip_som_params *p, int64_t n_units, int topol, int neigh, int64_t iter,
                           int64_t *batch_start, int64_t *batch_len);

/* ---- lvq1/olvq1/lvq2/lvq3_training (lvq_rout.c:498,584,702,808) --------------
 * kind = SOMHIP_LVQ1..LVQ3.  talpha (host, [n_rows], in/out) = OLVQ1's per-code
 * rates, initialised by the caller the way lvq_rout.c:614-627 does; `alpha` is also
 * OLVQ1's clamp (:671).  The result is the online (one sample at a time) result of the
 * reference, bit for bit; internally the loop runs as exact speculative batches (one
 * frozen-codebook scan per batch, samples certified and applied in order; see
 * kernels.hpp K6) unless SOMHIP_LVQ_ONLINE=1 or a row does not fit the on-chip cache
 * (dim > 2048), in which case every iteration is its own launch.  The trace has knn
 * entries per iteration (knn = 2 for LVQ2/LVQ3, else 1). */
typedef struct somhip_lvq_params {
  int32_t kind;
  int64_t length;
  float   alpha;
  int32_t alpha_type;
  float   winlen;        /* -win     (lvq_rout.c:702) */
  float   epsilon;       /* -epsilon (lvq_rout.c:808) */
  int64_t start_iter, count, data_first;
} somhip_lvq_params;
int  somhip_lvq_train(somhip_codebook *cb, somhip_dataset *ds, const somhip_lvq_params *p,
                      float *talpha, int32_t *trace_index, float *trace_diff);
/* out[0] = codebook rescans (batches) done by somhip_lvq_train so far, out[1] = samples,
 * out[2] / out[3] = batches cut short because a sample's candidate list was exhausted /
 * the on-chip row cache was full, out[4..7] = 100 MHz ticks the in-order kernel spent in
 * its phases (inputs, cached-row distances, decision, correction; summed over the components of a batch),
 * out[8] = independent components walked, out[9] = sum over the batches of the largest component's size
 * (the length of the longest serial walk), out[10] = (sample, 64-row group) pairs the exact top-k re-rank behind the
 * MFMA pre-filter evaluated, out[11] reserved */
int  somhip_lvq_stats(somhip_engine *e, uint64_t out[12]);

/* ---- lininit's data passes (find_eigenvectors, som_rout.c:211-289) ----------------
 * sum[i] / count[i]: fp32 sum and number of the unmasked values of component i over all rows in
 * order (:243-254).  r[i*dim + j], j >= i: fp32 sum over the rows, in order, of
 * (x_i - mean_i) * (x_j - mean_j) for rows where neither component is masked (:266-284);
 * elements with j < i are left 0.  The eigenvector iteration itself is O(dim^2) host work. */
int  somhip_column_sums(somhip_dataset *ds, float *sum, int64_t *count);
int  somhip_centered_products(somhip_dataset *ds, const float *mean, float *r);

/* ---- randinit's data pass (randinit_codes, som_rout.c:98-131) --------------------
 * lo[i] / hi[i] / count[i]: smallest and largest unmasked value of component i over all rows, and their
 * number (count may be NULL).  The reference seeds its maximum with FLT_MIN and its minimum with FLT_MAX
 * (:108-111) -- apply that on the result; a component without data returns lo = FLT_MAX, hi = -FLT_MAX. */
int  somhip_column_minmax(somhip_dataset *ds, float *lo, float *hi, int64_t *count);

/* ---- find_qerror2 (som_rout.c:823-885; bubble_qerror :734-772, gaussian_qerror :775-818):
 * out[i] = sum over the neighbourhood of sample first+i's winner of (h *) d*d, d =
 * vector_dist_euc (lvq_pak.c:291-316), accumulated in fp32 in unit order exactly as the
 * reference does; ret[i] = 0 for samples with every component masked (the reference skips
 * them, :858).  The caller adds out[] in data order (the reference's float accumulator). */
int  somhip_qerror2(somhip_codebook *cb, somhip_dataset *ds, float radius, int64_t first,
                    int64_t count, float *out, int32_t *ret);

/* ---- two-phase mini-batch primitives (what somhip_som_train(batch>1) is made of;
 * exposed so a multi-GPU host can put its collective between them) -------------
 * keys: DEVICE array [count] of uint64 = (fp32 bits of squared distance << 32) |
 * global row index.  All distances are >= 0, so unsigned order == (distance, index)
 * order and an element-wise MIN across shards is exactly find_winner_euc over the
 * whole codebook, lowest index winning ties (lvq_pak.c:79).  "No winner in this shard"
 * is 0x7FFFFFFFFFFFFFFF, so every key is non-negative as int64 and a SIGNED 64-bit MIN
 * (all that torch.distributed / RCCL offer) orders them the same way. */
int  somhip_batch_winner_keys(somhip_codebook *cb, somhip_dataset *ds, int64_t first,
                              int64_t count, uint64_t *dev_keys);
int  somhip_som_batch_update(somhip_codebook *cb, somhip_dataset *ds, const somhip_som_params *p,
                             int64_t batch_start_iter, int64_t count, int64_t data_first,
                             const uint64_t *dev_keys);
/* The same search over a ROW-SHARDED codebook with the pre-filter's bounds exchanged between the shards (find_winner_euc,
 * lvq_pak.c:41-96, over rows that live on several GPUs).  somhip_batch_winner_keys on a shard keeps every row group
 * within a window of the SHARD's smallest pre-filter value, so N shards together re-rank N times what one GPU would;
 * with the smallest bound of all shards in its place they re-rank what the whole codebook's search would:
 *     somhip_shard_winner_begin    level 1 of the pre-filter; dev_bound[count] (DEVICE floats) <- an upper bound on
 *                                  (squared distance to the nearest row of this shard) - ||x||^2; presets dev_keys
 *     host: all-reduce(MIN) of dev_bound as floats
 *     somhip_shard_winner_refine   level 2 for the groups within the exchanged bound; dev_bound <- the tighter bound
 *     host: all-reduce(MIN) of dev_bound
 *     somhip_shard_winner_finish   exact re-rank of the rows within the bound; dev_keys as somhip_batch_winner_keys
 *                                  (a shard that holds no candidate for a sample leaves 0x7FFFFFFFFFFFFFFF)
 *     host: all-reduce(MIN) of dev_keys, then somhip_som_batch_update
 * After the last all-reduce the keys are bit-identical to those of somhip_batch_winner_keys + all-reduce (the bounds only
 * remove rows that provably cannot win).  The three calls of one search share the engine's scratch memory: nothing else
 * may run on the engine between them (a call that is not the continuation of the search begun -- out of order, another
 * codebook, data set or range, a whole search in between -- is refused with an error).  somhip_shard_exchange_available: 1 if this shape takes the path (bf16 pre-filter,
 * dim a multiple of 32, 225 <= count <= 65504, no masks), else 0 -- every rank must take the same path, so a host
 * all-reduces the answer (MIN) once per run. */
int  somhip_shard_exchange_available(somhip_codebook *cb, somhip_dataset *ds, int64_t count);
int  somhip_shard_winner_begin(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                               uint64_t *dev_keys, float *dev_bound);
int  somhip_shard_winner_refine(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count, float *dev_bound);
int  somhip_shard_winner_finish(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                                const float *dev_bound, uint64_t *dev_keys);
/* k-NN over a row-sharded codebook (the X2 exchange): dev_keys[count][knn] (knn = 1, 2, 4 or 8) =
 * this shard's knn best rows per sample, ascending; tag = global row index (SOMHIP_TIE_FIRST) or its
 * bitwise complement (SOMHIP_TIE_KNN: on equal distances the LATER row sorts first, lvq_pak.c:197).
 * Missing entries (shard smaller than knn) are all-ones.  All-gather the lists and keep the knn
 * smallest keys per sample: that is find_winner_knn (lvq_pak.c:152-221) over the whole codebook. */
int  somhip_batch_topk_keys(somhip_codebook *cb, somhip_dataset *ds, int64_t first, int64_t count,
                            int knn, int tie, uint64_t *dev_keys);
/* The lists of all shards, all-gathered into dev_gathered[n_shards][count][knn] -> dev_keys[count][knn], the knn
 * smallest keys per sample, ascending: exactly the merge the comment above describes, on the device. */
int  somhip_merge_topk_keys(somhip_engine *e, const uint64_t *dev_gathered, int n_shards, int64_t count, int knn,
                            uint64_t *dev_keys);

/* ---- lvq1/olvq1/lvq2/lvq3_training (lvq_rout.c:498-916) over a ROW-SHARDED codebook -----------------------------
 * Every rank holds rows [row_offset, row_offset + n_rows) of the codebook (with their labels; OLVQ1: their rates,
 * somhip_lvq_rates_upload) and sees the same data.  One batch of count <= 1024 iterations is three calls, the
 * host's two collectives between them; the result is the reference's online result bit for bit, as with
 * somhip_lvq_train:
 *   1. somhip_batch_topk_keys(knn 8, tie = KNN for LVQ2/LVQ3 else FIRST)   this shard's 8 nearest rows per sample
 *        host: all-gather the lists, somhip_merge_topk_keys                 -> the frozen global top 8 (every rank)
 *   2. somhip_lvq_batch_candidates   labels / OLVQ1 rates of the 8 listed rows and tile copies of the `xrows`
 *        nearest ones, filled in for the rows this shard owns, zero elsewhere:
 *          dev_lab[count][8] int32, dev_ta[count][8] float (OLVQ1; else may be NULL),
 *          dev_rows[count][xrows][4 * ceil(dim / 4)] float
 *        host: all-reduce(SUM) of the three buffers as 32-bit INTEGERS (exact: every other rank adds 0)
 *   3. somhip_lvq_batch_apply   every rank walks the batch (kernels/lvq_batch.hpp: the walk is deterministic, so
 *        all ranks take the same decisions) and commits the corrected rows it owns.  *consumed <= count iterations
 *        were applied -- the same number on every rank; the next batch starts there.  A winner beyond the `xrows`
 *        exchanged rows ends the batch early (a larger xrows trades exchange volume for longer batches; 8 = never).
 * trace_index / trace_diff: as somhip_lvq_train, for the consumed iterations. */
int  somhip_lvq_rates_upload(somhip_codebook *cb, const float *talpha);     /* [n_rows] local rows, lvq_rout.c:614-627 */
int  somhip_lvq_rates_download(somhip_codebook *cb, float *talpha);
int  somhip_lvq_batch_candidates(somhip_codebook *cb, int64_t count, int kind, const uint64_t *dev_keys, int xrows,
                                 int32_t *dev_lab, float *dev_ta, float *dev_rows);
int  somhip_lvq_batch_apply(somhip_codebook *cb, somhip_dataset *ds, const somhip_lvq_params *p,
                            int64_t batch_start_iter, int64_t count, int64_t data_first, const uint64_t *dev_keys,
                            const int32_t *dev_lab, const float *dev_ta, const float *dev_rows, int xrows,
                            int64_t *consumed, int32_t *trace_index, float *trace_diff);

/* ---- collectives for a C host with one process per GPU ---------------------------------------------
 * somhip_comm wraps an RCCL communicator (librccl.so is opened on first use, never linked): the all-reduce of packed
 * winner keys is ncclAllReduce(ncclUint64, ncclMin) enqueued on the engine's own stream between
 * somhip_batch_winner_keys and somhip_som_batch_update -- no host synchronisation in the step.
 *   rank 0: somhip_comm_unique_id(id)  ->  the host hands the 128 bytes to the other ranks (pipe, file, MPI, ...)
 *   all:    somhip_comm_create(engine, id, rank, world, &comm)
 * somhip_comm_create_sockets: the same operations over host sockets in a star around rank 0 (fds: rank 0 passes the
 * world-1 descriptors of ranks 1.., every other rank its one descriptor to rank 0) -- for rehearsals in which several
 * ranks share a GPU (RCCL refuses duplicate devices); each operation then synchronises and stages through the host.
 * allreduce_sum_u32 / allgather serve the LVQ exchange (candidate rows as integers; per-shard top-8 lists). */
typedef struct somhip_comm somhip_comm;
int  somhip_comm_unique_id(void *id128);
int  somhip_comm_create(somhip_engine *e, const void *id128, int rank, int world, somhip_comm **out);
int  somhip_comm_create_sockets(somhip_engine *e, int rank, int world, const int *fds, somhip_comm **out);
int  somhip_comm_allreduce_min_keys(somhip_comm *c, uint64_t *dev_keys, int64_t count);
int  somhip_comm_allreduce_min_f32(somhip_comm *c, float *dev_values, int64_t count);   /* the bounds of somhip_shard_winner_* */
int  somhip_comm_allreduce_sum_u32(somhip_comm *c, uint32_t *dev_words, int64_t count);
int  somhip_comm_allgather(somhip_comm *c, const void *dev_send, void *dev_recv, int64_t bytes_per_rank);
void somhip_comm_destroy(somhip_comm *c);

/* device scratch helpers for hosts without their own allocator */
int  somhip_device_alloc(somhip_engine *e, int64_t bytes, void **dev_ptr);
int  somhip_device_free(somhip_engine *e, void *dev_ptr);
int  somhip_copy_to_host(somhip_engine *e, void *host_dst, const void *dev_src, int64_t bytes);
int  somhip_copy_to_device(somhip_engine *e, void *dev_dst, const void *host_src, int64_t bytes);

/* ---- instrumentation ---------------------------------------------------------
 * Per-kernel launch statistics measured with HIP events on the engine's stream
 * (used by bench.py for the roofline line).  name = one of the kernel names
 * returned by somhip_kernel_name(i), i in [0, somhip_kernel_count()). */
int  somhip_timing_enable(somhip_engine *e, int on);
/* restrict the events to the kernels whose bit (1 << kernel id) is set; default all */
int  somhip_timing_select(somhip_engine *e, uint64_t kernel_mask);
int  somhip_timing_reset(somhip_engine *e);
int  somhip_kernel_count(void);
const char *somhip_kernel_name(int i);
int  somhip_timing_get(somhip_engine *e, int kernel, int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* SOMHIP_H */
