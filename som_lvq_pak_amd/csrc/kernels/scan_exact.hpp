// kernels/scan_exact.hpp -- K1: exact direct-form winner scan and top-k merge
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "layout.hpp"

namespace somhip {

// =====================================================================================
// K1: exact winner scan of a tile of S samples against 64*R code rows per wave.
//
// find_winner_euc (lvq_pak.c:41-94) / find_winner_knn (lvq_pak.c:152-221) for a whole
// run of samples at once.  One lane = one code row (R rows when R > 1), S running sums
// per row kept in registers; every (row, sample) sum is formed in dim order with
// separate sub/mul/add roundings, so each value equals the reference's bit for bit.
// The reference's early exit (lvq_pak.c:72) is result-neutral and not reproduced.
//
// grid.x = sample tiles (fastest: consecutive workgroups share the code tile in L2 and
// each XCD keeps seeing the same sample tiles), grid.y = code-row blocks of 4*R groups.
//
// TOPK == 1: the winner per sample is folded into keys[sample] with a 64-bit atomic
//            min of (distance bits, tag); tag = global row (FIRST tie rule) or
//            ~global row (KNN tie rule: later row first).
// TOPK  > 1: every workgroup writes its TOPK best keys per sample to
//            partial[sample][gridDim.y][TOPK]; k_merge_topk finishes.
// =====================================================================================
template <int S, int R, int TOPK>
__global__ __launch_bounds__(256) void k_scan_exact(CbView cb, const float4 *__restrict__ xt,
                                                    int64_t count, int tie_knn,
                                                    uint64_t *__restrict__ keys,
                                                    uint64_t *__restrict__ partial) {
  __shared__ uint64_t red[4][S][TOPK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t sb = blockIdx.x;
  const int64_t g0 = (static_cast<int64_t>(blockIdx.y) * 4 + wave) * R;
  const float4 *xtile = xt + sb * cb.d4 * S;

  float acc[R][S];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int s = 0; s < S; s++) acc[r][s] = 0.0f;

  if (g0 < cb.ngroups) {
    for (int q = 0; q < cb.d4; q++) {
      float4 c[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        int64_t g = g0 + r < cb.ngroups ? g0 + r : cb.ngroups - 1;
        c[r] = *tile_ptr(cb, g, q, lane);
      }
#pragma unroll
      for (int s = 0; s < S; s++) {
        float4 x = xtile[q * S + s];          // wave-uniform address
#pragma unroll
        for (int r = 0; r < R; r++) {
          float a = acc[r][s];
          a = sq_acc(a, c[r].x, x.x);
          a = sq_acc(a, c[r].y, x.y);
          a = sq_acc(a, c[r].z, x.z);
          a = sq_acc(a, c[r].w, x.w);
          acc[r][s] = a;
        }
      }
    }
  }

  // per-sample reduction over this wave's rows
#pragma unroll
  for (int s = 0; s < S; s++) {
    uint64_t k[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int64_t row = (g0 + r) * WAVE + lane;
      bool live = (g0 + r) < cb.ngroups && row < cb.n;
      uint32_t grow = unit_of_row(cb, row);
      k[r] = live ? make_key(acc[r][s], tie_knn ? ~grow : grow) : KEY_NONE;
    }
#pragma unroll
    for (int t = 0; t < TOPK; t++) {
      uint64_t mine = k[0];
#pragma unroll
      for (int r = 1; r < R; r++) mine = k[r] < mine ? k[r] : mine;
      uint64_t best = wave_min_u64(mine);
      if (TOPK > 1) {
#pragma unroll
        for (int r = 0; r < R; r++)
          if (k[r] == best) k[r] = KEY_NONE;   // keys are unique (tag = row)
      }
      if (lane == 0) red[wave][s][t] = best;
    }
  }
  __syncthreads();
  // merge the 4 waves: thread (s, t-th smallest)
  for (int e = threadIdx.x; e < S; e += blockDim.x) {
    int64_t smp = sb * S + e;
    if (smp >= count) continue;
    uint64_t cand[4 * TOPK];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int t = 0; t < TOPK; t++) cand[w * TOPK + t] = red[w][e][t];
    if (TOPK == 1) {
      uint64_t b = cand[0];
#pragma unroll
      for (int w = 1; w < 4; w++) b = cand[w] < b ? cand[w] : b;
      atomicMin(reinterpret_cast<unsigned long long *>(keys + smp),
                static_cast<unsigned long long>(b));
    } else {
      for (int t = 0; t < TOPK; t++) {
        int arg = 0;
        uint64_t b = cand[0];
        for (int j = 1; j < 4 * TOPK; j++)
          if (cand[j] < b) { b = cand[j]; arg = j; }
        cand[arg] = KEY_NONE;
        partial[(smp * gridDim.y + blockIdx.y) * TOPK + t] = b;
      }
    }
  }
}

// merge partial[sample][nblk][K] -> keys_out[sample][K]; one wave per sample
template <int K>
__global__ void k_merge_topk(const uint64_t *__restrict__ partial, int nblk, int64_t count,
                             uint64_t *__restrict__ keys_out) {
  int64_t smp = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (smp >= count) return;
  const uint64_t *p = partial + smp * nblk * K;
  int total = nblk * K;
  uint64_t prev = 0;
  bool first = true;
  for (int t = 0; t < K; t++) {
    uint64_t mine = KEY_NONE;
    for (int j = lane; j < total; j += WAVE) {
      uint64_t v = p[j];
      if ((first || v > prev) && v < mine) mine = v;   // keys are unique
    }
    uint64_t best = wave_min_u64(mine);
    if (lane == 0) keys_out[smp * K + t] = best;
    prev = best;
    first = false;
    if (best == KEY_NONE) { for (int u = t + 1; u < K; u++) if (lane == 0) keys_out[smp * K + u] = KEY_NONE; break; }
  }
}

}  // namespace somhip
