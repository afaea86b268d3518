// kernels/som_online.hpp -- K3: fused online SOM iteration, pipelined row stream
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "som_update.hpp"

namespace somhip {

// =====================================================================================
// K3: one ONLINE SOM iteration, fused: apply iteration t-1's neighbourhood update to
// each code row and, in the same pass over the row, accumulate its distance to sample
// t (som_training's inner loop, som_rout.c:600-662, with find_winner_euc and
// bubble/gaussian_adapt).  The codebook is therefore read once per iteration instead of
// twice, and written only where it changed.  The winner of iteration t is folded into
// slot[t] with a 64-bit atomic min; the next launch (stream order) reads it.
//   has_prev / has_cur select prologue (no update yet) and flush (no search left).
// A wave streams its 64 rows through two register buffers of U chunks (U KiB) each: the
// next buffer's loads are issued before the current one is consumed, and the loop body
// is branch-free (UPD / SEARCH / MASKED / VEC are compile-time) so the waits the
// compiler places are counted, not drains.
// =====================================================================================
template <bool UPD, bool SEARCH, bool MASKED, bool VEC>
__device__ __forceinline__ void online_chunk(const CbView &cb, int64_t g, int lane, int q, float4 c,
                                             bool upd, float a, const float *__restrict__ xp,
                                             const float *__restrict__ xc,
                                             const uint8_t *__restrict__ mp,
                                             const uint8_t *__restrict__ mc, float &acc) {
  if (UPD) {
    const float4 x = load_x4<VEC>(xp, q, cb.d);
    if (upd) {
      const float4 n = adapt4(c, x, a);
      if (MASKED) {
        if (q * 4 + 0 < cb.d && mp[q * 4 + 0] == 0) c.x = n.x;
        if (q * 4 + 1 < cb.d && mp[q * 4 + 1] == 0) c.y = n.y;
        if (q * 4 + 2 < cb.d && mp[q * 4 + 2] == 0) c.z = n.z;
        if (q * 4 + 3 < cb.d && mp[q * 4 + 3] == 0) c.w = n.w;
      } else {
        c = n;
      }
      *tile_ptr_w(cb, g, q, lane) = c;
    }
  }
  if (SEARCH) {
    const float4 x = load_x4<VEC>(xc, q, cb.d);
    if (MASKED) {
      if (q * 4 + 0 < cb.d && mc[q * 4 + 0] == 0) acc = sq_acc(acc, c.x, x.x);
      if (q * 4 + 1 < cb.d && mc[q * 4 + 1] == 0) acc = sq_acc(acc, c.y, x.y);
      if (q * 4 + 2 < cb.d && mc[q * 4 + 2] == 0) acc = sq_acc(acc, c.z, x.z);
      if (q * 4 + 3 < cb.d && mc[q * 4 + 3] == 0) acc = sq_acc(acc, c.w, x.w);
    } else {
      acc = sq_acc(acc, c.x, x.x);
      acc = sq_acc(acc, c.y, x.y);
      acc = sq_acc(acc, c.z, x.z);
      acc = sq_acc(acc, c.w, x.w);
    }
  }
}

template <bool UPD, bool SEARCH, bool MASKED, bool VEC, int U>
__device__ __forceinline__ float online_stream(const CbView &cb, int64_t g, int lane, bool upd, float a,
                                               const float *__restrict__ xp,
                                               const float *__restrict__ xc,
                                               const uint8_t *__restrict__ mp,
                                               const uint8_t *__restrict__ mc) {
  float acc = 0.0f;
  float4 bufA[U], bufB[U];
  const int nfull = (cb.d4 / (2 * U)) * (2 * U);
  const int last = cb.d4 - 1;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) bufA[u] = *tile_ptr(cb, g, u, lane);
    for (int qb = 0; qb < nfull; qb += 2 * U) {
#pragma unroll
      for (int u = 0; u < U; u++) bufB[u] = *tile_ptr(cb, g, qb + U + u, lane);
#pragma unroll
      for (int u = 0; u < U; u++)
        online_chunk<UPD, SEARCH, MASKED, VEC>(cb, g, lane, qb + u, bufA[u], upd, a, xp, xc, mp, mc, acc);
#pragma unroll
      for (int u = 0; u < U; u++) {            // prefetch for the next trip (clamped at the end)
        int q = qb + 2 * U + u;
        bufA[u] = *tile_ptr(cb, g, q < last ? q : last, lane);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        online_chunk<UPD, SEARCH, MASKED, VEC>(cb, g, lane, qb + U + u, bufB[u], upd, a, xp, xc, mp, mc, acc);
    }
  }
  for (int q = nfull; q < cb.d4; q++)          // tail chunks
    online_chunk<UPD, SEARCH, MASKED, VEC>(cb, g, lane, q, *tile_ptr(cb, g, q, lane), upd, a, xp, xc, mp, mc, acc);
  return acc;
}

template <bool GAUSS, bool MASKED, int U>
__global__ __launch_bounds__(256) void k_som_online_step(CbView cb, const float *__restrict__ rows,
                                                         const uint8_t *__restrict__ mask,
                                                         const int64_t *__restrict__ prev_row_p,
                                                         const int64_t *__restrict__ cur_row_p,
                                                         int has_prev, int has_cur,
                                                         const uint64_t *__restrict__ prev_slot,
                                                         uint64_t *__restrict__ cur_slot,
                                                         const StepScalars *__restrict__ prev_sc,
                                                         const StepScalars *__restrict__ cur_sc) {
  // every per-iteration quantity arrives through device arrays, so one captured launch
  // sequence (hipGraph) can be replayed for every chunk of the run
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (g >= cb.ngroups) return;
  const int64_t prev_row = *prev_row_p, cur_row = *cur_row_p;
  const int64_t row = g * WAVE + lane;
  const uint32_t grow = unit_of_row(cb, row);
  const uint32_t xdim = static_cast<uint32_t>(cb.xdim);
  const bool live = row < cb.n;

  bool upd = false;
  float a = 0.0f;
  if (has_prev) {
    const StepScalars s = *prev_sc;
    int bx = -1, by = -1;
    if (s.reach >= 0) {
      if (s.fixed >= 0) { bx = fixed_x(s.fixed); by = fixed_y(s.fixed); }
      else {
        uint64_t k = *prev_slot;
        if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) {
          const uint32_t widx = static_cast<uint32_t>(k);
          bx = static_cast<int>(widx % xdim); by = static_cast<int>(widx / xdim);
        }
      }
    }
    if (bx >= 0) {
      int tx, ty;
      txty_of_row(cb, row, tx, ty);
      const float lsq = lattice_sq(cb.topol, bx, by, tx, ty);
      if (GAUSS) { a = gaussian_alpha(lsq, s.thresh, s.alpha); upd = live; }
      else { a = s.alpha; upd = live && (lsq <= s.thresh); }
    }
  }
  bool search = has_cur;
  if (has_cur) { const StepScalars s = *cur_sc; if (s.reach < 0 || s.fixed >= 0) search = false; }
  const bool any_upd = __any(upd);
  if (!any_upd && !search) return;

  const float *xp = rows + prev_row * cb.d;
  const float *xc = rows + cur_row * cb.d;
  const uint8_t *mp = MASKED ? mask + prev_row * cb.d : nullptr;
  const uint8_t *mc = MASKED ? mask + cur_row * cb.d : nullptr;
  const bool vec = (cb.d & 3) == 0;
  float acc;
#define ONLINE_GO(UU, SS)                                                                               \
  acc = vec ? online_stream<UU, SS, MASKED, true, U>(cb, g, lane, upd, a, xp, xc, mp, mc)               \
            : online_stream<UU, SS, MASKED, false, U>(cb, g, lane, upd, a, xp, xc, mp, mc)
  if (any_upd && search) { ONLINE_GO(true, true); }
  else if (search) { ONLINE_GO(false, true); }
  else { ONLINE_GO(true, false); }
#undef ONLINE_GO
  if (search) {
    uint64_t k = live ? make_key(acc, grow) : KEY_NONE;
    k = wave_min_u64(k);
    if (lane == 0)
      atomicMin(reinterpret_cast<unsigned long long *>(cur_slot), static_cast<unsigned long long>(k));
  }
}

}  // namespace somhip
