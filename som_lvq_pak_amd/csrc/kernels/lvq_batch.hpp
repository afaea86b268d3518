// kernels/lvq_batch.hpp -- K6: exact batched LVQ, in independent components
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "lvq.hpp"

namespace somhip {

// =====================================================================================
// K6: EXACT batched LVQ ("speculate on a frozen codebook, repair in order").
//
// An LVQ iteration corrects one or two code rows (lvq_rout.c:552-555, 658-673, 779-780,
// 890-895), so between iteration t and t+j only <= 2j rows differ from the codebook the
// batch started with.  Phase 1 (top-8 scan) finds, for every sample of the batch, the
// LVQ_K0 = 8 nearest rows of the FROZEN codebook as exact keys.  Phase 2 walks samples in
// iteration order, keeps every row corrected so far in an LDS cache (lane = cache slot)
// and, per sample,
//   * recomputes the distance of each cached row with the reference's arithmetic,
//   * takes the 2 smallest keys over  {cached rows}  U  {frozen candidates not cached},
//   * accepts them only if the last one is <= the sample's 8th frozen key -- every row
//     outside the list that was not corrected still has its frozen key, which is larger --
//     so the accepted winners ARE find_winner_euc / find_winner_knn on the codebook as
//     iteration t sees it; otherwise (or when the cache is full) the batch ends here,
//   * applies the LVQ1 / OLVQ1 / LVQ2.1 / LVQ3 decision to the cached copies.
//
// Round 2: phase 2 no longer is ONE serial walk.  Two samples can only influence each other
// through a row that is (or becomes) a winner of both.  A winner w of sample j that was
// accepted lies within R_j = sqrt(8th frozen distance of j) of x_j; after its correction it lies
// within rho_j = (1 + max|alpha|) R_j (+ rounding slack) of x_j (c' = c + a (x - c):
// x - c' = (1 - a)(x - c)).  Hence a row can reach sample j from sample i only if
//        ||x_i - x_j||  <=  rho_i + rho_j                                  (*)
// (triangle inequality on the real vectors; every computed distance is within gamma_{d+2} of the
// real one and the factors below leave 2^-10 of room for that).  The connected components of
// relation (*) are therefore independent: no row is ever read or written by two of them, and
// each component's walk sees exactly the rows the serial walk would show it (rows touched by
// other components have keys above the component's samples' bounds, so they change neither
// its winners nor its accept/stop decisions -- DESIGN.md section 4 has the argument in full).
// One workgroup walks one component, all components of a batch at once (k_lvq_pair_adj ->
// k_lvq_components -> k_lvq_batch_apply, grid = components); corrected rows go to a staging
// area, not to the codebook, so that when some component has to stop at sample J (candidate
// list exhausted / cache full) the walk is repeated with the batch cut at the smallest such J
// and only then committed (k_lvq_commit).  Data in one tight blob is one component: the walk is
// then the serial one of round 1.
// =====================================================================================
constexpr int LVQ_K0 = 8;
constexpr int LVQ_BT = 512;          // threads = maximum number of cache slots
constexpr int LVQ_BMAX = 1024;       // samples per batch (= threads of the component kernel)
constexpr int LVQ_AW = LVQ_BMAX / 32;   // adjacency words per sample
constexpr int LVQ_DYN_LDS = 150 * 1024; // dynamic LDS of the walk (row cache); its static arrays take ~8.5 KiB of the 160

struct LvqBatchOut {                 // device -> host summary of one walk
  int32_t ncomp, pad;
  int32_t start[LVQ_BMAX + 1];       // component c owns comp_samples[start[c] .. start[c+1])
  int32_t stop[LVQ_BMAX];            // batch-relative sample at which component c stopped (INT_MAX: walked to the end)
  int32_t reason[LVQ_BMAX];          // 1 candidate list exhausted, 2 cache full, 3 row not at hand (sharded exchange)
  int32_t nslots[LVQ_BMAX];          // rows the component corrected (staged)
  int64_t cycles[4];                 // s_memtime ticks spent in phases A..D, summed over components
};

// Control block of the batch loop that does not wait for the host (host_lvq.inc: lvq_train_batched).  The host
// launches batch after batch assuming that each walks to its end; a batch in which a component stopped early does
// not commit, raises `poison`, and from then on no later batch changes the codebook, the rates or the bf16 copies --
// until the host has seen the flag, redone that batch the careful way and cleared it.
struct LvqCtl {
  int32_t poison;                    // != 0: batch `batch` stopped at sample `limit` for `reason`; nothing later was applied
  int32_t batch, limit, reason;
  unsigned long long batches, comps, largest, cycles[4];   // statistics of the batches that were committed
};

__device__ __forceinline__ float lvq_sq(float c, float x) { const float t = c - x; return t * t; }

// ---- rho_j of relation (*): one wave per sample -------------------------------------------------
// rho = ((1 + amax) sqrt(d8) ) (1 + 2^-10) + 2^-18 (||x|| + sqrt(d8)), evaluated in double and rounded up;
// +inf when the sample's list is not full (tiny codebooks: everything interacts).
__global__ __launch_bounds__(256) void k_lvq_sample_rho(const float *__restrict__ rows, int64_t n_rows, int d,
                                                        int64_t first, int count, const uint64_t *__restrict__ cand,
                                                        float amax_host, const float *__restrict__ amax_dev,
                                                        float *__restrict__ rho, float *__restrict__ xnorm = nullptr) {
  const float amax = amax_dev ? *amax_dev : amax_host;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= count) return;
  const float *x = rows + ((first + j) % n_rows) * static_cast<int64_t>(d);
  float s = 0.0f;
  for (int i = lane; i < d; i += WAVE) s += x[i] * x[i];
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, WAVE);
  if (lane == 0) {
    const uint64_t k8 = cand[static_cast<int64_t>(j) * LVQ_K0 + LVQ_K0 - 1];
    const uint32_t bits = static_cast<uint32_t>(k8 >> 32);
    float r = __builtin_inff();
    if (k8 != KEY_NONE && bits < FLT_MAX_BITS && amax >= 0.0f) {
      const double r8 = sqrt(static_cast<double>(__uint_as_float(bits)));
      const double v = (1.0 + static_cast<double>(amax)) * r8 * (1.0 + 1.0 / 1024.0) +
                       (sqrt(static_cast<double>(s) * 1.001) + r8) / 262144.0;
      r = static_cast<float>(v);
      if (static_cast<double>(r) < v) r = __uint_as_float(__float_as_uint(r) + 1);
    }
    rho[j] = r;
    if (xnorm) xnorm[j] = s;                               // ||x_j||^2 as summed here (k_lvq_pair_adj_mfma)
  }
}

// OLVQ1: the rates live per row.  A correction uses the winner's rate, a winner is a listed candidate (or a row
// corrected earlier in the batch, whose rate then is ta/(1+ta) < ta or min(ta/(1-ta), clamp): lvq_rout.c:663,
// :670-672), so max(clamp, largest listed rate) bounds every rate of the batch -- provided all of them lie in
// [0, 1); anything else (a hand-made .lra file) gives -1 = "unknown": one component.
__global__ __launch_bounds__(256) void k_lvq_amax(const uint64_t *__restrict__ cand, int64_t total, const float *__restrict__ cand_ta,
                                                  float clamp, float *__restrict__ amax) {
  __shared__ float s_m[256];
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  float m = 0.0f;
  for (int64_t e = threadIdx.x; e < total; e += 256)
    if (cand[e] != KEY_NONE) {
      const float t = cand_ta[e];
      if (!(t >= 0.0f && t < 1.0f)) s_bad = 1;
      m = fmaxf(m, t);
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, WAVE));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; i++) m = fmaxf(m, s_m[i]);
    *amax = (s_bad || !(clamp >= 0.0f && clamp < 1.0f)) ? -1.0f : fmaxf(m, clamp);
  }
}

// ---- relation (*) for every pair of the batch: adjacency bit rows adj[j][i / 32] ---------------
// Tiles of 64 x 64 pairs (lower triangle + diagonal; the mirror image is written too); each thread
// forms 4 x 4 squared distances over the dims in chunks of 32 staged in LDS.  Any summation order
// will do here (the bound only needs |s - S| <= gamma S).
__global__ __launch_bounds__(256) void k_lvq_pair_adj(const float *__restrict__ rows, int64_t n_rows, int d,
                                                      int64_t first, int count, const float *__restrict__ rho,
                                                      uint32_t *__restrict__ adj) {
  const int tj = blockIdx.y, ti = blockIdx.x;
  if (ti > tj) return;
  // chunk-major LDS tiles [k / 4][row] of float4: a thread's 4 + 4 rows are 8 ds_read_b128 per 4 dims (192 flop)
  __shared__ float4 sa[8][64], sb[8][64];
  __shared__ uint8_t sadj[64][65];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const bool vec = (d & 3) == 0;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0f;
  for (int k0 = 0; k0 < d; k0 += 32) {
    for (int e = tid; e < 64 * 8; e += 256) {                 // e -> (row, chunk of 4 dims); 8 consecutive chunks = 128 B of one row
      const int r = e >> 3, q = e & 7;
      const int ja = tj * 64 + r, ib = ti * 64 + r, k = k0 + 4 * q;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (k < d) {
        const float *pa = rows + ((first + ja) % n_rows) * static_cast<int64_t>(d) + k;
        const float *pb = rows + ((first + ib) % n_rows) * static_cast<int64_t>(d) + k;
        if (vec) {
          if (ja < count) va = *reinterpret_cast<const float4 *>(pa);
          if (ib < count) vb = *reinterpret_cast<const float4 *>(pb);
        } else {
          float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
          for (int u = 0; u < 4 && k + u < d; u++) { if (ja < count) ta[u] = pa[u]; if (ib < count) tb[u] = pb[u]; }
          va = make_float4(ta[0], ta[1], ta[2], ta[3]); vb = make_float4(tb[0], tb[1], tb[2], tb[3]);
        }
      }
      sa[q][r] = va; sb[q][r] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) {
      float4 va[4], vb[4];
#pragma unroll
      for (int a = 0; a < 4; a++) { va[a] = sa[q][ty * 4 + a]; vb[a] = sb[q][tx * 4 + a]; }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const float t0 = va[a].x - vb[b].x, t1 = va[a].y - vb[b].y, t2 = va[a].z - vb[b].z, t3 = va[a].w - vb[b].w;
          acc[a][b] += t0 * t0; acc[a][b] += t1 * t1; acc[a][b] += t2 * t2; acc[a][b] += t3 * t3;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int j = tj * 64 + ty * 4 + a, i = ti * 64 + tx * 4 + b;
      bool on = false;
      if (j < count && i < count && j != i) {
        const double lhs = sqrt(static_cast<double>(acc[a][b])) * (1.0 - 1.0 / 4096.0);
        on = lhs <= static_cast<double>(rho[j]) + static_cast<double>(rho[i]);
      }
      sadj[ty * 4 + a][tx * 4 + b] = on ? 1 : 0;
    }
  __syncthreads();
  if (tid < 128) {
    const int r = tid >> 1, h = tid & 1;
    uint32_t w = 0, wt = 0;
#pragma unroll 8
    for (int b = 0; b < 32; b++) { w |= static_cast<uint32_t>(sadj[r][32 * h + b]) << b; wt |= static_cast<uint32_t>(sadj[32 * h + b][r]) << b; }
    if (tj * 64 + r < LVQ_BMAX) adj[static_cast<int64_t>(tj * 64 + r) * LVQ_AW + ti * 2 + h] = w;
    if (ti != tj && ti * 64 + r < LVQ_BMAX) adj[static_cast<int64_t>(ti * 64 + r) * LVQ_AW + tj * 2 + h] = wt;
  }
}

// The same relation with the pairwise distances in Gram form on the fp32 matrix pipe: d^2(j, i) = ||x_j||^2 +
// ||x_i||^2 - 2 <x_j, x_i>, the inner products of a 64 x 64 tile of pairs by v_mfma_f32_32x32x2_f32 (one 32 x 32
// sub-tile per wave, operands straight from the row-major data: per 8 dims one float4 per lane and operand, element
// e of it feeding MFMA e -- lanes 0-31 carry dims k0 + e, lanes 32-63 dims k0 + 4 + e, the same for both operands).
// (*) only has to be decided CONSERVATIVELY (an extra edge merges two components, which costs parallelism, never
// exactness): with u = 2^-24 the computed value differs from the real d^2 by at most (3 d + 16) u (n_j + n_i)
// (fp32 products, fp32 accumulation in any order, the two wave-summed norms), so "computed <= (rho_j + rho_i)^2
// (1 + 2^-10) + 8 (d + 8) u (n_j + n_i)" holds for every pair that satisfies (*).  The inner products are
// bit-symmetric in (j, i) (same products, same order), hence so is the matrix; every tile of it is computed
// (no mirror writes).  Needs d % 8 == 0; other shapes use k_lvq_pair_adj.
__global__ __launch_bounds__(256) void k_lvq_pair_adj_mfma(const float *__restrict__ rows, int64_t n_rows, int d,
                                                           int64_t first, int count, const float *__restrict__ rho,
                                                           const float *__restrict__ xnorm, uint32_t *__restrict__ adj) {
  const int tj = blockIdx.y, ti = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;
  const int j = tj * 64 + wr * 32 + l31, i = ti * 64 + wc * 32 + l31;      // this lane's row of operand A / B
  const float *pa = rows + ((first + (j < count ? j : count - 1)) % n_rows) * static_cast<int64_t>(d) + 4 * half;
  const float *pb = rows + ((first + (i < count ? i : count - 1)) % n_rows) * static_cast<int64_t>(d) + 4 * half;
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; v++) acc[v] = 0.0f;
  int k0 = 0;
  for (; k0 + 32 <= d; k0 += 32) {                         // 32 dims: eight loads in flight, then 16 MFMAs
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = *reinterpret_cast<const float4 *>(pa + k0 + 8 * u);
      b[u] = *reinterpret_cast<const float4 *>(pb + k0 + 8 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
  for (; k0 < d; k0 += 8) {
    const float4 a = *reinterpret_cast<const float4 *>(pa + k0), b = *reinterpret_cast<const float4 *>(pb + k0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  // register v of the tile: row (v / 4) * 8 + half * 4 + v % 4 (a sample j), column l31 (sample i)
  const bool i_ok = i < count;
  const double ni = i_ok ? static_cast<double>(xnorm[i]) : 0.0, ri = i_ok ? static_cast<double>(rho[i]) : 0.0;
  const double slack_unit = 8.0 * (d + 8) * 5.9604644775390625e-08;
  const int word = ti * 2 + wc;
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int jr = tj * 64 + wr * 32 + (v >> 2) * 8 + half * 4 + (v & 3);
    bool on = false;
    if (jr < count && i_ok && jr != i) {
      const double nj = static_cast<double>(xnorm[jr]), rj = static_cast<double>(rho[jr]);
      const double d2 = nj + ni - 2.0 * static_cast<double>(acc[v]);
      const double r = rj + ri;
      on = d2 <= r * r * (1.0 + 1.0 / 1024.0) + slack_unit * (nj + ni);      // (rho = +inf: always)
    }
    const unsigned long long bal = __ballot(on);
    const int j0 = tj * 64 + wr * 32 + (v >> 2) * 8 + (v & 3);               // the row of half 0; half 1: + 4
    if (lane == 0) {
      if (j0 < LVQ_BMAX) adj[static_cast<int64_t>(j0) * LVQ_AW + word] = static_cast<uint32_t>(bal);
      if (j0 + 4 < LVQ_BMAX) adj[static_cast<int64_t>(j0 + 4) * LVQ_AW + word] = static_cast<uint32_t>(bal >> 32);
    }
  }
}

// ---- connected components of the batch under (*) ----------------------------------------------
// One workgroup, thread = sample.  Labels converge by min-propagation over the adjacency rows (held
// in LDS) with pointer jumping; then components are numbered by decreasing size (the longest walk
// starts first) and their samples listed in iteration order.  single != 0: one component (the
// serial walk: SOMHIP_LVQ_SERIAL, rates that cannot be bounded).
__global__ __launch_bounds__(LVQ_BMAX) void k_lvq_components(const uint32_t *__restrict__ adj, int count, int single,
                                                            int32_t *__restrict__ comp_samples, LvqBatchOut *__restrict__ out) {
  extern __shared__ uint32_t s_adjrows[];                 // [count][LVQ_AW]
  __shared__ int32_t s_label[LVQ_BMAX], s_size[LVQ_BMAX], s_rank[LVQ_BMAX], s_start[LVQ_BMAX + 1];
  __shared__ int s_changed, s_ncomp;
  const int j = threadIdx.x;
  const int nw = (count + 31) >> 5;
  for (int e = j; e < count * LVQ_AW; e += LVQ_BMAX) s_adjrows[e] = single ? 0u : adj[e];
  s_label[j] = single ? 0 : j;
  s_size[j] = 0;
  if (j == 0) s_ncomp = 0;
  __syncthreads();
  const uint32_t *my = s_adjrows + j * LVQ_AW;
  for (int round = 0; round < LVQ_BMAX && !single; round++) {
    if (j == 0) s_changed = 0;
    __syncthreads();
    int m = j < count ? s_label[j] : 0;
    if (j < count)
      for (int w = 0; w < nw; w++) {
        uint32_t bits = my[w];
        while (bits) {
          const int i = (w << 5) + __ffs(bits) - 1;
          bits &= bits - 1;
          const int li = s_label[i];
          m = li < m ? li : m;
        }
      }
    __syncthreads();
    if (j < count && m != s_label[j]) { s_label[j] = m; s_changed = 1; }
    __syncthreads();
    for (int hop = 0; hop < 4; hop++) {                   // pointer jumping: label <- label[label]
      const int l = j < count ? s_label[s_label[j]] : 0;
      __syncthreads();
      if (j < count) s_label[j] = l;
      __syncthreads();
    }
    const int changed = s_changed;
    __syncthreads();                                      // everyone has read the flag before the next round clears it
    if (!changed) break;
  }
  // ---- sizes; the roots in sample order; rank of each root by (size descending, root ascending); start offsets;
  // position of each sample inside its component (samples of a component stay in iteration order).  All of it in
  // wave-parallel steps: the serial versions (one thread's prefix sum over the components, a 1024-trip loop per
  // root and per sample) were most of this kernel's time.
  const int lane = j & 63, wave = j >> 6;
  __shared__ int32_t s_wtot[LVQ_BMAX / 64 + 1], s_rid[LVQ_BMAX], s_rsize[LVQ_BMAX];
  const int l = j < count ? s_label[j] : -1;
  if (j < count) atomicAdd(&s_size[l], 1);
  __syncthreads();
  const bool root = j < count && l == j;
  {
    const unsigned long long bal = __ballot(root);
    if (lane == 0) s_wtot[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += s_wtot[w];
    if (j == 0) { int t = 0; for (int w = 0; w < LVQ_BMAX / 64; w++) t += s_wtot[w]; s_ncomp = t; }
    const int r = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (root) { s_rid[r] = j; s_rsize[r] = s_size[j]; }
  }
  __syncthreads();
  const int nc = s_ncomp;
  // thread r < nc: the r-th root; its rank among the roots
  int my_rank = 0;
  if (j < nc) {
    const int mine = s_rsize[j];
    for (int k = 0; k < nc; k++) {
      const int sk = s_rsize[k];
      my_rank += (sk > mine || (sk == mine && k < j)) ? 1 : 0;
    }
    s_rank[s_rid[j]] = my_rank;
  }
  __syncthreads();
  // start[]: exclusive prefix sum of the sizes in rank order (thread = rank)
  {
    if (j < nc) s_start[1 + my_rank] = s_rsize[j];       // scatter, then scan in place
    if (j == 0) s_start[0] = 0;
    __syncthreads();
    int v = (j < nc) ? s_start[1 + j] : 0;                // size of the component of rank j
    int inc = v;                                          // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += s_wtot[w];
    if (j < nc) s_start[1 + j] = before + inc;
  }
  // position inside the component: earlier samples with the same label = those of earlier waves (counted per wave
  // and label into the LDS that held the adjacency rows) + those of lower lanes in this wave
  {
    int32_t *s_cnt = reinterpret_cast<int32_t *>(s_adjrows);      // [wave][label], 16 x count ints (<= 64 KiB of the 128)
    __syncthreads();
    for (int e = j; e < (LVQ_BMAX / 64) * count; e += LVQ_BMAX) s_cnt[e] = 0;
    __syncthreads();
    int prior = 0, wave_total = 0;                         // lower lanes / all lanes of this wave with my label
    for (int k = 0; k < 64; k++) {                         // (in uniform control flow: every lane's label is read)
      const int lk = __builtin_amdgcn_readlane(l, k);
      wave_total += lk == l ? 1 : 0;
      prior += (lk == l && k < lane) ? 1 : 0;
    }
    if (j < count && prior == 0) s_cnt[wave * count + l] = wave_total;   // the first lane of each label
    __syncthreads();
    if (j < count) {
      int pos = prior;
      for (int w = 0; w < wave; w++) pos += s_cnt[w * count + l];
      comp_samples[s_start[s_rank[l]] + pos] = j;
    }
  }
  __syncthreads();
  if (j == 0) { out->ncomp = nc; out->pad = 0; for (int c = 0; c < 4; c++) out->cycles[c] = 0; }
  if (j <= nc) out->start[j] = s_start[j];
}

// ---- phase 2: one workgroup walks one component ------------------------------------------------
// cand_rows (sharded codebooks; else null): tile-form copies [count][xc][d4] float4 of the xc nearest
// frozen candidates of every sample, exchanged between the ranks; cand_lab / cand_ta [count][8]: label and
// OLVQ1 rate of every listed candidate.  Row ids inside keys are GLOBAL rows; local tiles are used when
// cand_rows is null.  limit: samples >= limit are not walked (the repeat after a stop).
__global__ __launch_bounds__(LVQ_BT) void k_lvq_batch_apply(CbView cb, const float *__restrict__ rows,
                                                            int64_t n_rows, int64_t first, int limit,
                                                            const int32_t *__restrict__ cand_lab,
                                                            const float *__restrict__ cand_ta,
                                                            const float4 *__restrict__ cand_rows, int xc,
                                                            const uint64_t *__restrict__ cand,
                                                            const LvqStep *__restrict__ st, int knn,
                                                            int slots, const int32_t *__restrict__ comp_samples,
                                                            uint64_t *__restrict__ fin,
                                                            float4 *__restrict__ stage_rows, int32_t *__restrict__ stage_rowid,
                                                            float *__restrict__ stage_ta, LvqBatchOut *__restrict__ out) {
  const int comp = blockIdx.x;
  if (comp >= out->ncomp) return;
  extern __shared__ float4 lvq_dyn[];
  const int d4 = cb.d4;
  float4 *cache = lvq_dyn;                         // [d4][slots]  lane = slot
  float4 *s_x = lvq_dyn + static_cast<size_t>(d4) * slots;   // [d4]
  float4 *s_pre = s_x + d4;                        // [2][d4]  tile rows of the two nearest frozen candidates
  constexpr int NW = LVQ_BT / WAVE;
  __shared__ int32_t s_slot_row[LVQ_BT];
  __shared__ int32_t s_slot_lab[LVQ_BT];
  __shared__ float s_slot_ta[LVQ_BT];
  __shared__ uint64_t s_ck[LVQ_K0];
  __shared__ int32_t s_clab[LVQ_K0];
  __shared__ float s_cta[LVQ_K0];
  __shared__ uint64_t s_wtop[NW][2];
  __shared__ int32_t s_wslot[NW][2];
  __shared__ uint32_t s_flags;
  __shared__ int s_stop, s_m, s_nupd, s_uslot[2], s_usrc[2];
  __shared__ float s_ua[2];
  __shared__ int32_t s_urow[2];
  __shared__ uint16_t s_list[LVQ_BMAX];           // this component's samples, iteration order

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool vec = (cb.d & 3) == 0;
  const bool knn2 = knn == 2;
  const int base = out->start[comp];
  int count = out->start[comp + 1] - base;
  for (int e = tid; e < count; e += LVQ_BT) s_list[e] = comp_samples[base + e];
  if (tid == 0) { s_m = 0; s_stop = 0; }
  __syncthreads();
  while (count > 0 && s_list[count - 1] >= limit) count--;      // ascending: only a tail can lie beyond the cut
  const int64_t stage0 = 2 * static_cast<int64_t>(base);        // this component's staging slots: [2 base, 2 base + 2 size)

  // Per-sample inputs are fetched one sample ahead into registers, so their global-memory latency
  // hides behind the phases of the sample before; candidate keys and step scalars two ahead, so the
  // loads that depend on them need no wait.  A candidate's row fetched early is only used if
  // that row is still uncorrected when its sample is decided, i.e. still frozen.
  float4 nx = make_float4(0.f, 0.f, 0.f, 0.f), np0 = nx, np1 = nx;
  uint64_t nck = KEY_NONE, kc0 = KEY_NONE, kc1 = KEY_NONE, kck = KEY_NONE;   // k*: keys of the sample after next
  int32_t nlab = 0;
  float nta = 0.0f;
  LvqStep nst = {}, cst = {};
  auto row_chunk = [&](int js, int c, uint32_t r, int q) -> float4 {   // chunk q of frozen candidate c (row r) of sample js
    if (cand_rows) return cand_rows[(static_cast<int64_t>(js) * xc + c) * d4 + q];
    return *tile_ptr(cb, r >> 6, q, r & 63);
  };
  auto fetch_keys = [&](int t) {
    if (t < count) {
      const int64_t js = s_list[t];
      kc0 = cand[js * LVQ_K0 + 0];
      kc1 = cand[js * LVQ_K0 + 1];
      if (tid < LVQ_K0) kck = cand[js * LVQ_K0 + tid];
    }
  };
  auto fetch = [&](int t) {                  // needs fetch_keys(t) issued one sample earlier
    const int js = s_list[t];
    const float *xr = rows + ((first + js) % n_rows) * static_cast<int64_t>(cb.d);
    const uint64_t c0 = kc0, c1 = kc1;
    nck = kck;
    if (tid < d4) {
      nx = vec ? reinterpret_cast<const float4 *>(xr)[tid] : load_x4<false>(xr, tid, cb.d);
      if (c0 != KEY_NONE) {
        const uint32_t r = knn2 ? ~static_cast<uint32_t>(c0) : static_cast<uint32_t>(c0);
        np0 = row_chunk(js, 0, r, tid);
      }
      if (c1 != KEY_NONE && (!cand_rows || xc > 1)) {
        const uint32_t r = knn2 ? ~static_cast<uint32_t>(c1) : static_cast<uint32_t>(c1);
        np1 = row_chunk(js, 1, r, tid);
      }
    }
    if (tid < LVQ_K0 && nck != KEY_NONE) {
      nlab = cand_lab[static_cast<int64_t>(js) * LVQ_K0 + tid];
      nta = cand_ta ? cand_ta[static_cast<int64_t>(js) * LVQ_K0 + tid] : 0.0f;
    }
    if (tid == 0) nst = st[js];
    fetch_keys(t + 1);
  };
  fetch_keys(0);
  if (count > 0) fetch(0);

  int m = 0, j = 0, reason = 0, stop_at = 0x7FFFFFFF;
  int64_t cyc[4] = {0, 0, 0, 0};
  int64_t tick = static_cast<int64_t>(__builtin_readcyclecounter());
  auto lap = [&](int ph) {
    const int64_t now = static_cast<int64_t>(__builtin_readcyclecounter());
    cyc[ph] += now - tick;
    tick = now;
  };
  for (; j < count; j++) {
    const int js = s_list[j];
    // ---- A: this sample's inputs -> LDS, next sample's loads issued
    if (tid < d4) { s_x[tid] = nx; s_pre[tid] = np0; s_pre[d4 + tid] = np1; }
    if (tid < LVQ_K0) { s_ck[tid] = nck; s_clab[tid] = nlab; s_cta[tid] = nta; }
    if (tid == 0) s_flags = 0;
    cst = nst;
    __syncthreads();
    lap(0);
    if (j + 1 < count) fetch(j + 1);
    // ---- B: exact distance of every cached row (dims in order, sub / mul / add) ----
    uint64_t key = KEY_NONE;
    if (tid < m) {
      // The sum is one dependent chain of d adds; everything else is arranged to stay off it: the
      // LDS reads run two blocks (of 4 chunks = 16 dims) ahead and the sub/mul of the next block are
      // independent work the ALU can issue between the chain's adds.
      float acc = 0.0f;
      const float4 *cp = cache + tid;
      const int nblk = d4 >> 2;
      float4 rAc[4], rAx[4], rBc[4], rBx[4];
      float pA[16], pB[16];
#define LVQ_LOAD(RC, RX, BLK)                                                             \
      {                                                                                   \
        const int q_ = ((BLK) < nblk ? (BLK) : nblk - 1) * 4;                              \
        _Pragma("unroll") for (int u = 0; u < 4; u++) { RC[u] = cp[(q_ + u) * slots]; RX[u] = s_x[q_ + u]; } \
      }
#define LVQ_PROD(P, RC, RX)                                                               \
      _Pragma("unroll") for (int u = 0; u < 4; u++) {                                     \
        const f32x2 t0_ = f32x2{RC[u].x, RC[u].y} - f32x2{RX[u].x, RX[u].y};              \
        const f32x2 t1_ = f32x2{RC[u].z, RC[u].w} - f32x2{RX[u].z, RX[u].w};              \
        const f32x2 p0_ = t0_ * t0_, p1_ = t1_ * t1_;   /* v_pk_*: each half rounded like the scalar op */ \
        P[4 * u + 0] = p0_.x; P[4 * u + 1] = p0_.y; P[4 * u + 2] = p1_.x; P[4 * u + 3] = p1_.y; \
      }
#define LVQ_SUM(P) _Pragma("unroll") for (int i = 0; i < 16; i++) acc = acc + P[i];
      if (nblk > 0) {
        LVQ_LOAD(rAc, rAx, 0)
        LVQ_LOAD(rBc, rBx, 1)
        LVQ_PROD(pA, rAc, rAx)
        int b = 0;
        for (; b + 2 <= nblk; b += 2) {
          LVQ_LOAD(rAc, rAx, b + 2)
          LVQ_PROD(pB, rBc, rBx)
          LVQ_SUM(pA)
          LVQ_LOAD(rBc, rBx, b + 3)
          LVQ_PROD(pA, rAc, rAx)
          LVQ_SUM(pB)
        }
        if (nblk & 1) { LVQ_SUM(pA) }
      }
#undef LVQ_LOAD
#undef LVQ_PROD
#undef LVQ_SUM
      for (int q = nblk * 4; q < d4; q++) {
        const float4 c = cp[q * slots];
        const float4 x = s_x[q];
        acc = sq_acc(acc, c.x, x.x);
        acc = sq_acc(acc, c.y, x.y);
        acc = sq_acc(acc, c.z, x.z);
        acc = sq_acc(acc, c.w, x.w);
      }
      const uint32_t r = static_cast<uint32_t>(s_slot_row[tid]);
      const uint32_t tag = knn2 ? ~r : r;
      key = make_key(acc, tag);
      uint32_t f = 0;
#pragma unroll
      for (int c = 0; c < LVQ_K0; c++) f |= (static_cast<uint32_t>(s_ck[c]) == tag && s_ck[c] != KEY_NONE) ? (1u << c) : 0u;
      if (f) atomicOr(&s_flags, f);
    }
    if (wave * WAVE < m) {
      const uint64_t b0 = wave_min_u64_dpp(key);
      const uint64_t rest = key == b0 ? KEY_NONE : key;
      const uint64_t b1 = wave_min_u64_dpp(rest);
      const unsigned long long w0 = __ballot(key == b0), w1 = __ballot(rest == b1);
      if (lane == 0) {
        s_wtop[wave][0] = b0; s_wtop[wave][1] = b1;
        s_wslot[wave][0] = b0 != KEY_NONE ? wave * WAVE + __ffsll(w0) - 1 : -1;
        s_wslot[wave][1] = b1 != KEY_NONE ? wave * WAVE + __ffsll(w1) - 1 : -1;
      }
    }
    __syncthreads();
    lap(1);
    // ---- C: merge, validate, decide, allocate cache slots: wave 0, payloads carried in registers ----
    if (wave == 0) {
      uint64_t v = KEY_NONE;
      int32_t plab = 0, pslot = -1, psrc = -1;
      float pta = 0.0f;
      if (lane < 2 * NW) {
        if ((lane >> 1) * WAVE < m) {
          v = s_wtop[lane >> 1][lane & 1];
          pslot = s_wslot[lane >> 1][lane & 1];
          if (pslot >= 0) { plab = s_slot_lab[pslot]; pta = s_slot_ta[pslot]; }
        }
      } else if (lane < 2 * NW + LVQ_K0) {
        const int c = lane - 2 * NW;
        if (!((s_flags >> c) & 1u)) { v = s_ck[c]; plab = s_clab[c]; pta = s_cta[c]; psrc = c; }
      }
      const uint64_t bound = s_ck[LVQ_K0 - 1];
      const uint64_t k0 = wave_min_u64_dpp(v);
      const uint64_t rest = v == k0 ? KEY_NONE : v;            // keys are unique (tag = row)
      const uint64_t k1 = wave_min_u64_dpp(rest);
      const int l0 = k0 != KEY_NONE ? __ffsll(__ballot(v == k0)) - 1 : 0;
      const int l1 = k1 != KEY_NONE ? __ffsll(__ballot(rest == k1)) - 1 : 0;
      const int l0u = __builtin_amdgcn_readfirstlane(l0), l1u = __builtin_amdgcn_readfirstlane(l1);
      const int32_t wlab[2] = {__builtin_amdgcn_readlane(plab, l0u), __builtin_amdgcn_readlane(plab, l1u)};
      const int32_t wslot0[2] = {__builtin_amdgcn_readlane(pslot, l0u), __builtin_amdgcn_readlane(pslot, l1u)};
      const int32_t wsrc[2] = {__builtin_amdgcn_readlane(psrc, l0u), __builtin_amdgcn_readlane(psrc, l1u)};
      const float wta[2] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pta), l0u)),
                            __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pta), l1u))};
      int32_t wslot[2] = {wslot0[0], wslot0[1]};
      const uint64_t last = knn2 ? k1 : k0;
      int stop = 0, nupd = 0;
      if (last > bound) {
        stop = 1;                                  // a row outside the list could be nearer: rescan
      } else {
        const LvqStep sp = {__builtin_amdgcn_readfirstlane(cst.kind),
                            __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst.alpha))),
                            __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst.alpha_clamp))),
                            __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst.win_ratio))),
                            __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cst.epsilon))),
                            __builtin_amdgcn_readfirstlane(cst.label)};
        const bool ok0 = static_cast<uint32_t>(k0 >> 32) < FLT_MAX_BITS;
        const bool ok1 = static_cast<uint32_t>(k1 >> 32) < FLT_MAX_BITS;
        const uint32_t t0 = static_cast<uint32_t>(k0), t1 = static_cast<uint32_t>(k1);
        const int32_t wrow[2] = {static_cast<int32_t>(knn2 ? ~t0 : t0), static_cast<int32_t>(knn2 ? ~t1 : t1)};
        int uidx[2] = {0, 0};
        float ua[2] = {0.f, 0.f};
        bool olvq_correct = false;
        if (sp.kind == 1 && ok0) {                                  // LVQ1, lvq_rout.c:552-555
          nupd = 1; uidx[0] = 0; ua[0] = (wlab[0] == sp.label) ? sp.alpha : -sp.alpha;
        } else if (sp.kind == 2 && ok0) {                           // OLVQ1, lvq_rout.c:658-673
          olvq_correct = wlab[0] == sp.label;
          nupd = 1; uidx[0] = 0; ua[0] = olvq_correct ? wta[0] : -wta[0];
        } else if (sp.kind >= 3 && ok0 && ok1) {                    // LVQ2.1 / LVQ3
          const float d0 = __uint_as_float(static_cast<uint32_t>(k0 >> 32));
          const float d1 = __uint_as_float(static_cast<uint32_t>(k1 >> 32));
          if (wlab[0] != wlab[1]) {
            if (wlab[0] == sp.label || wlab[1] == sp.label) {
              if ((d0 / d1) > sp.win_ratio) {                       // lvq_rout.c:770 / :876
                const int best = (wlab[1] == sp.label) ? 1 : 0;
                nupd = 2; uidx[0] = best; ua[0] = sp.alpha; uidx[1] = best ^ 1; ua[1] = -sp.alpha;
              }
            }
          } else if (sp.kind == 4 && wlab[0] == sp.label) {         // lvq_rout.c:890-895
            const float ae = sp.alpha * sp.epsilon;
            nupd = 2; uidx[0] = 0; ua[0] = ae; uidx[1] = 1; ua[1] = ae;
          }
        }
        int mm = m;
        bool missing = false;                      // a frozen winner whose row was not exchanged (sharded codebooks)
        for (int u = 0; u < nupd; u++)
          if (wslot[uidx[u]] < 0) { mm++; missing |= cand_rows != nullptr && wsrc[uidx[u]] >= xc; }
        if (mm > slots) {
          stop = 2;                                // cache full: nothing of this sample is applied
        } else if (missing) {
          stop = 3;
        } else if (lane == 0) {
          mm = m;
          for (int u = 0; u < nupd; u++) {
            const int w = uidx[u];
            s_urow[u] = wrow[w];
            s_ua[u] = ua[u];
            if (wslot[w] < 0) {                    // winner came from the frozen list: not cached yet
              const int sl = mm++;
              s_slot_row[sl] = wrow[w];
              s_slot_lab[sl] = wlab[w];
              s_slot_ta[sl] = wta[w];
              s_uslot[u] = sl;
              s_usrc[u] = wsrc[w];                 // 0 / 1: prefetched row, >= 2: fetch it now
              wslot[w] = sl;
            } else {
              s_uslot[u] = wslot[w];
              s_usrc[u] = -1;                      // already cached
            }
          }
          if (sp.kind == 2 && nupd == 1) {         // the corrected row advances its rate (lvq_rout.c:663, :670-672)
            float ta = wta[0];
            if (olvq_correct) {
              ta = ta / (1 + ta);
            } else {
              ta = ta / (1 - ta);
              if (ta > sp.alpha_clamp) ta = sp.alpha_clamp;
            }
            s_slot_ta[wslot[0]] = ta;
          }
          s_m = mm;
          fin[2 * static_cast<int64_t>(js)] = k0;
          fin[2 * static_cast<int64_t>(js) + 1] = k1;
        }
      }
      if (lane == 0) { s_stop = stop; s_nupd = nupd; }
    }
    __syncthreads();
    lap(2);
    if (s_stop) { reason = s_stop; stop_at = js; break; }
    m = s_m;
    // ---- D: adapt_vector on the cached copies (lvq_pak.c:339-351) ----
    const int nupd = s_nupd;
    if (tid < d4) {
      const float4 x = s_x[tid];
      for (int u = 0; u < nupd; u++) {
        const int sl = s_uslot[u], src = s_usrc[u];
        float4 c;
        if (src < 0) c = cache[tid * slots + sl];
        else if (src < 2) c = s_pre[src * d4 + tid];
        else c = row_chunk(js, src, static_cast<uint32_t>(s_urow[u]), tid);
        cache[tid * slots + sl] = adapt4(c, x, s_ua[u]);
      }
    }
    __syncthreads();
    lap(3);
  }
  // ---- stage the corrected rows (and OLVQ1 rates); k_lvq_commit writes them to the codebook ----
  for (int e = tid; e < m * d4; e += LVQ_BT) {
    const int sl = e / d4, q = e - sl * d4;
    stage_rows[(stage0 + sl) * d4 + q] = cache[q * slots + sl];
  }
  for (int sl = tid; sl < m; sl += LVQ_BT) {
    stage_rowid[stage0 + sl] = s_slot_row[sl];
    stage_ta[stage0 + sl] = s_slot_ta[sl];
  }
  if (tid == 0) {
    out->stop[comp] = stop_at; out->reason[comp] = reason; out->nslots[comp] = m;
    for (int k = 0; k < 4; k++) atomicAdd(reinterpret_cast<unsigned long long *>(&out->cycles[k]), static_cast<unsigned long long>(cyc[k]));
  }
}

// ---- commit: staged rows -> codebook tiles (rows of this shard only), OLVQ1 rates, list of written rows ----
__global__ __launch_bounds__(256) void k_lvq_commit(CbView cb, const LvqBatchOut *__restrict__ out,
                                                    const float4 *__restrict__ stage_rows, const int32_t *__restrict__ stage_rowid,
                                                    const float *__restrict__ stage_ta, float *__restrict__ talpha,
                                                    int32_t *__restrict__ mod_rows, int32_t *__restrict__ mod_count,
                                                    LvqCtl *__restrict__ ctl = nullptr, int batch_id = 0, int batch_len = 0) {
  const int comp = blockIdx.x;
  if (comp >= out->ncomp) return;
  if (ctl) {                                             // the loop that does not wait for the host: commit only a batch that ran through
    __shared__ int s_first[256], s_why[256], s_big[256];
    if (*reinterpret_cast<volatile int32_t *>(&ctl->poison)) return;
    int mine = 0x7FFFFFFF, why = 0, big = 0;
    for (int k = threadIdx.x; k < out->ncomp; k += 256) {
      if (out->stop[k] < mine) { mine = out->stop[k]; why = out->reason[k]; }
      big = max(big, out->start[k + 1] - out->start[k]);
    }
    s_first[threadIdx.x] = mine; s_why[threadIdx.x] = why; s_big[threadIdx.x] = big;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
      if (threadIdx.x < off) {
        if (s_first[threadIdx.x + off] < s_first[threadIdx.x]) {
          s_first[threadIdx.x] = s_first[threadIdx.x + off]; s_why[threadIdx.x] = s_why[threadIdx.x + off];
        }
        s_big[threadIdx.x] = max(s_big[threadIdx.x], s_big[threadIdx.x + off]);
      }
      __syncthreads();
    }
    const int first = s_first[0];
    if (first < batch_len) {                             // every workgroup of this launch comes to the same verdict
      if (comp == 0 && threadIdx.x == 0) {
        ctl->batch = batch_id; ctl->limit = first; ctl->reason = s_why[0];
        __threadfence();
        *reinterpret_cast<volatile int32_t *>(&ctl->poison) = 1;
      }
      return;
    }
    if (comp == 0 && threadIdx.x == 0) {
      ctl->batches += 1ull; ctl->comps += static_cast<unsigned long long>(out->ncomp);
      ctl->largest += static_cast<unsigned long long>(s_big[0]);
      for (int k = 0; k < 4; k++) ctl->cycles[k] += static_cast<unsigned long long>(out->cycles[k]);
    }
  }
  const int ns = out->nslots[comp];
  const int64_t stage0 = 2 * static_cast<int64_t>(out->start[comp]);
  for (int sl = 0; sl < ns; sl++) {
    const int64_t g = static_cast<int64_t>(stage_rowid[stage0 + sl]) - cb.row_offset;      // local row
    if (g < 0 || g >= cb.n) continue;                                                       // another shard's row
    for (int q = threadIdx.x; q < cb.d4; q += 256) *tile_ptr_w(cb, g >> 6, q, static_cast<int>(g & 63)) = stage_rows[(stage0 + sl) * cb.d4 + q];
    if (threadIdx.x == 0) {
      if (talpha) talpha[g] = stage_ta[stage0 + sl];
      mod_rows[atomicAdd(mod_count, 1)] = static_cast<int32_t>(g);
    }
  }
}

// labels / OLVQ1 rates of every listed candidate (rows of this shard; others stay 0 and come by all-reduce)
__global__ void k_lvq_cand_meta(CbView cb, const uint64_t *__restrict__ cand, int64_t total, int knn,
                                const int32_t *__restrict__ clabels, const float *__restrict__ talpha,
                                int32_t *__restrict__ cand_lab, float *__restrict__ cand_ta) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint64_t k = cand[e];
  int32_t lab = 0;
  float ta = 0.0f;
  if (k != KEY_NONE) {
    const uint32_t t = static_cast<uint32_t>(k);
    const int64_t g = static_cast<int64_t>(knn == 2 ? ~t : t) - cb.row_offset;
    if (g >= 0 && g < cb.n) { lab = clabels[g]; ta = talpha ? talpha[g] : 0.0f; }
  }
  cand_lab[e] = lab;
  if (cand_ta) cand_ta[e] = ta;
}

// X2 merge: gathered[g][count][K] = every shard's K best keys per sample -> the K smallest per sample
// (keys are unique: tag = global row or its complement), ascending, missing entries all-ones.  One thread per sample.
template <int K>
__global__ void k_merge_shard_topk(const uint64_t *__restrict__ gathered, int nshards, int64_t count, uint64_t *__restrict__ out) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= count) return;
  uint64_t top[K];
#pragma unroll
  for (int t = 0; t < K; t++) top[t] = KEY_NONE;
  for (int g = 0; g < nshards; g++)
    for (int t = 0; t < K; t++) {
      uint64_t v = gathered[(static_cast<int64_t>(g) * count + j) * K + t];
      if (v >= top[K - 1]) break;                       // each shard's list is ascending
#pragma unroll
      for (int u = 0; u < K; u++) {                     // sorted insertion
        const uint64_t lo = top[u] < v ? top[u] : v;
        v = top[u] < v ? v : top[u];
        top[u] = lo;
      }
    }
#pragma unroll
  for (int t = 0; t < K; t++) out[j * K + t] = top[t];
}

// tile-form copies of the xc nearest frozen candidates of every sample (rows of this shard; others stay 0)
__global__ __launch_bounds__(256) void k_lvq_cand_rows(CbView cb, const uint64_t *__restrict__ cand, int count, int xc, int knn,
                                                       float4 *__restrict__ cand_rows) {
  const int64_t e = blockIdx.x;                         // (sample, candidate)
  const int js = static_cast<int>(e / xc), c = static_cast<int>(e % xc);
  if (js >= count) return;
  const uint64_t k = cand[static_cast<int64_t>(js) * LVQ_K0 + c];
  int64_t g = -1;
  if (k != KEY_NONE) { const uint32_t t = static_cast<uint32_t>(k); g = static_cast<int64_t>(knn == 2 ? ~t : t) - cb.row_offset; }
  const bool mine = g >= 0 && g < cb.n;
  for (int q = threadIdx.x; q < cb.d4; q += 256)
    cand_rows[e * cb.d4 + q] = mine ? *tile_ptr(cb, g >> 6, q, static_cast<int>(g & 63)) : make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace somhip
