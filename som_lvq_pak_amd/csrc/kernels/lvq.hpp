// kernels/lvq.hpp -- K5/K6: online LVQ iteration, exact batched LVQ
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "rerank.hpp"

namespace somhip {

// =====================================================================================
// K5: one ONLINE LVQ iteration, fused the same way: apply iteration t-1's LVQ1 / OLVQ1 /
// LVQ2.1 / LVQ3 correction (lvq_rout.c:542-555, 650-673, 750-783, 855-896) to the one or
// two rows it touches, then accumulate every row's distance to sample t and leave this
// workgroup's two best keys in part[blockIdx.x][2] (knn = 2 uses find_winner_knn's tie
// order, lvq_pak.c:197).  Every workgroup of the next launch merges all partials
// itself (a few hundred 8-byte words) -- no extra launch, no grid barrier.
// =====================================================================================
struct LvqStep {
  int32_t kind;        // SOMHIP_LVQ1..3
  float alpha;         // schedule value for this iteration (unused by OLVQ1)
  float alpha_clamp;   // OLVQ1: initial alpha (lvq_rout.c:671)
  float win_ratio;     // (1-w)/(1+w) in fp32 (lvq_rout.c:770)
  float epsilon;
  int32_t label;       // the sample's first label
};

// insert v into the running two smallest (k0 <= k1), branch-free and by value
#define TOP2_INSERT(k0, k1, v)                         \
  do {                                                 \
    const uint64_t v_ = (v);                           \
    const uint64_t hi_ = v_ > (k0) ? v_ : (k0);        \
    (k0) = v_ < (k0) ? v_ : (k0);                      \
    (k1) = hi_ < (k1) ? hi_ : (k1);                    \
  } while (0)

__global__ __launch_bounds__(256) void k_lvq_online_step(CbView cb, const float *__restrict__ rows,
                                                         const int32_t *__restrict__ clabels,
                                                         float *__restrict__ talpha,
                                                         int64_t prev_row, int64_t cur_row,
                                                         int has_prev, int has_cur, int knn,
                                                         const uint64_t *__restrict__ prev_part,
                                                         int prev_nblk,
                                                         uint64_t *__restrict__ cur_part,
                                                         uint64_t *__restrict__ prev_final,
                                                         const LvqStep *__restrict__ prev_st) {
  __shared__ uint64_t sh[4][2];
  __shared__ uint64_t shw[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int64_t row = g * WAVE + lane;
  const bool live = g < cb.ngroups && row < cb.n;

  // --- merge the previous iteration's partial top-2 (identical in every workgroup) ---
  int64_t u_row0 = -1, u_row1 = -1;        // the (at most two) rows iteration t-1 corrects
  float u_a0 = 0.f, u_a1 = 0.f;
  if (has_prev) {
    if (wave == 0) {
      uint64_t k0 = KEY_NONE, k1 = KEY_NONE;
      for (int j = lane; j < prev_nblk * 2; j += WAVE) TOP2_INSERT(k0, k1, prev_part[j]);
      uint64_t b0 = wave_min_u64(k0);
      uint64_t mine = (k0 == b0) ? k1 : k0;
      uint64_t b1 = wave_min_u64(mine);
      if (lane == 0) { shw[0] = b0; shw[1] = b1; }
    }
    __syncthreads();
    const uint64_t b0 = shw[0], b1 = shw[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { prev_final[0] = b0; prev_final[1] = b1; }
    const LvqStep st = *prev_st;
    const uint32_t t0 = static_cast<uint32_t>(b0), t1 = static_cast<uint32_t>(b1);
    const int64_t i0 = knn == 2 ? static_cast<int64_t>(~t0) : static_cast<int64_t>(t0);
    const int64_t i1 = static_cast<int64_t>(~t1);
    if (st.kind == 1) {                                   // LVQ1, lvq_rout.c:552-555
      u_row0 = i0;
      u_a0 = (clabels[i0] == st.label) ? st.alpha : -st.alpha;
    } else if (st.kind == 2) {                            // OLVQ1, lvq_rout.c:658-673
      u_row0 = i0;
      float ta = talpha[i0];
      u_a0 = (clabels[i0] == st.label) ? ta : -ta;
    } else {                                              // LVQ2.1 / LVQ3
      const int l0 = clabels[i0], l1 = clabels[i1];
      const float d0 = __uint_as_float(static_cast<uint32_t>(b0 >> 32));
      const float d1 = __uint_as_float(static_cast<uint32_t>(b1 >> 32));
      if (l0 != l1) {
        if (l0 == st.label || l1 == st.label) {
          if ((d0 / d1) > st.win_ratio) {                 // lvq_rout.c:770 / :876
            int64_t best = i0, nbest = i1;
            if (l1 == st.label) { best = i1; nbest = i0; }
            u_row0 = best;  u_a0 = st.alpha;
            u_row1 = nbest; u_a1 = -st.alpha;
          }
        }
      } else if (st.kind == 4 && l0 == st.label) {        // lvq_rout.c:890-895
        float ae = st.alpha * st.epsilon;
        u_row0 = i0; u_a0 = ae;
        u_row1 = i1; u_a1 = ae;
      }
    }
  }
  const int64_t grow = row + cb.row_offset;
  // which (if any) correction applies to this lane's row; if both name the same row
  // (cannot happen: two distinct neighbours) the first wins
  int which = -1;
  if (live) { if (grow == u_row0) which = 0; else if (grow == u_row1) which = 1; }
  const bool upd = which >= 0;
  const float a = which == 1 ? u_a1 : u_a0;
  const bool any_upd = __any(upd);
  const float *xp = rows + prev_row * cb.d;
  const float *xc = rows + cur_row * cb.d;
  const bool vec = (cb.d & 3) == 0;
  float acc = 0.0f;
  if (g < cb.ngroups && (any_upd || has_cur)) {
    // same pipelined, branch-free row stream as the SOM step (two register buffers per wave)
#define LVQ_GO(UU, SS)                                                                                  \
    acc = vec ? online_stream<UU, SS, false, true, 8>(cb, g, lane, upd, a, xp, xc, nullptr, nullptr)     \
              : online_stream<UU, SS, false, false, 8>(cb, g, lane, upd, a, xp, xc, nullptr, nullptr)
    if (any_upd && has_cur) { LVQ_GO(true, true); }
    else if (has_cur) { LVQ_GO(false, true); }
    else { LVQ_GO(true, false); }
#undef LVQ_GO
  }
  // OLVQ1: the owner of the corrected row advances its rate (lvq_rout.c:663, :670-672)
  if (has_prev && upd && which == 0) {
    const LvqStep st = *prev_st;
    if (st.kind == 2) {
      float ta = talpha[grow];
      if (clabels[grow] == st.label) {
        ta = ta / (1 + ta);
      } else {
        ta = ta / (1 - ta);
        if (ta > st.alpha_clamp) ta = st.alpha_clamp;
      }
      talpha[grow] = ta;
    }
  }
  if (has_cur) {
    uint32_t tag = static_cast<uint32_t>(grow);
    uint64_t k = live ? make_key(acc, knn == 2 ? ~tag : tag) : KEY_NONE;
    uint64_t b0 = wave_min_u64(k);
    uint64_t mine = (k == b0) ? KEY_NONE : k;
    uint64_t b1 = wave_min_u64(mine);
    if (lane == 0) { sh[wave][0] = b0; sh[wave][1] = b1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t k0 = KEY_NONE, k1 = KEY_NONE;
      for (int w = 0; w < 4; w++) { TOP2_INSERT(k0, k1, sh[w][0]); TOP2_INSERT(k0, k1, sh[w][1]); }
      cur_part[blockIdx.x * 2 + 0] = k0;
      cur_part[blockIdx.x * 2 + 1] = k1;
    }
  }
}

}  // namespace somhip
