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

// =====================================================================================
// K6: EXACT batched LVQ ("speculate on a frozen codebook, repair in order").
//
// An LVQ iteration corrects one or two code rows (lvq_rout.c:552-555, 658-673, 779-780,
// 890-895), so between iteration t and t+j only <= 2j rows differ from the codebook the
// batch started with.  Phase 1 (k_scan_exact<.., 8> + k_merge_topk) finds, for every
// sample of the batch, the LVQ_K0 = 8 nearest rows of the FROZEN codebook as exact keys.
// Phase 2 (this kernel, one workgroup, samples strictly in iteration order) keeps every
// row corrected so far in an LDS cache (lane = cache slot) and, per sample,
//   * recomputes the distance of each cached row with the reference's arithmetic,
//   * takes the 2 smallest keys over  {cached rows}  U  {frozen candidates not cached},
//   * accepts them only if the last one is <= the sample's 8th frozen key -- every row
//     outside the list that was not corrected still has its frozen key, which is larger --
//     so the accepted winners ARE find_winner_euc / find_winner_knn on the codebook as
//     iteration t sees it; otherwise (or when the cache is full) the batch ends here and
//     the host starts the next one at this sample,
//   * applies the LVQ1 / OLVQ1 / LVQ2.1 / LVQ3 decision to the cached copies.
// The result is therefore bit-identical to the online loop for every batch size; the
// batch only sets how often the whole codebook is re-scanned.
// =====================================================================================
constexpr int LVQ_K0 = 8;
constexpr int LVQ_BT = 512;          // threads = maximum number of cache slots

struct LvqBatchCtl {
  int32_t consumed;    // samples applied (the next batch starts at first + consumed)
  int32_t nmod;        // distinct rows corrected (= rows written back)
  int32_t reason;      // 0 whole batch, 1 candidate list exhausted, 2 cache full
  int32_t pad;
  int64_t cycles[4];   // s_memtime ticks (100 MHz) spent in phases A, B, C, D (wave 0)
};

__device__ __forceinline__ float lvq_sq(float c, float x) { const float t = c - x; return t * t; }

__global__ __launch_bounds__(LVQ_BT) void k_lvq_batch_apply(CbView cb, const float *__restrict__ rows,
                                                            int64_t n_rows, int64_t first, int count,
                                                            const int32_t *__restrict__ clabels,
                                                            float *__restrict__ talpha,
                                                            const uint64_t *__restrict__ cand,
                                                            const LvqStep *__restrict__ st, int knn,
                                                            int slots, uint64_t *__restrict__ fin,
                                                            int32_t *__restrict__ mod_rows,
                                                            LvqBatchCtl *__restrict__ ctl) {
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

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool vec = (cb.d & 3) == 0;
  const bool knn2 = knn == 2;
  if (tid == 0) { s_m = 0; s_stop = 0; }
  __syncthreads();

  // Per-sample inputs are fetched one sample ahead into registers, so their global-memory latency
  // hides behind the phases of the sample before; candidate keys and step scalars two ahead, so the
  // loads that depend on them need no wait.  A candidate's tile row fetched early is only used if
  // that row is still uncorrected when its sample is decided, i.e. still frozen.
  float4 nx = make_float4(0.f, 0.f, 0.f, 0.f), np0 = nx, np1 = nx;
  uint64_t nck = KEY_NONE, kc0 = KEY_NONE, kc1 = KEY_NONE, kck = KEY_NONE;   // k*: keys of the sample after next
  int32_t nlab = 0;
  float nta = 0.0f;
  LvqStep nst = {}, cst = {};
  int64_t frow = first % n_rows;            // data row of the sample being fetched (wraps like the reference)
  auto fetch_keys = [&](int jj) {
    if (jj < count) {
      kc0 = cand[static_cast<int64_t>(jj) * LVQ_K0 + 0];
      kc1 = cand[static_cast<int64_t>(jj) * LVQ_K0 + 1];
      if (tid < LVQ_K0) kck = cand[static_cast<int64_t>(jj) * LVQ_K0 + tid];
    }
  };
  auto fetch = [&](int jj) {                 // needs fetch_keys(jj) issued one sample earlier
    const float *xr = rows + frow * static_cast<int64_t>(cb.d);
    frow = frow + 1 == n_rows ? 0 : frow + 1;
    const uint64_t c0 = kc0, c1 = kc1;
    nck = kck;
    if (tid < d4) {
      nx = vec ? reinterpret_cast<const float4 *>(xr)[tid] : load_x4<false>(xr, tid, cb.d);
      if (c0 != KEY_NONE) {
        const uint32_t r = knn2 ? ~static_cast<uint32_t>(c0) : static_cast<uint32_t>(c0);
        np0 = *tile_ptr(cb, r >> 6, tid, r & 63);
      }
      if (c1 != KEY_NONE) {
        const uint32_t r = knn2 ? ~static_cast<uint32_t>(c1) : static_cast<uint32_t>(c1);
        np1 = *tile_ptr(cb, r >> 6, tid, r & 63);
      }
    }
    if (tid < LVQ_K0 && nck != KEY_NONE) {
      const uint32_t r = knn2 ? ~static_cast<uint32_t>(nck) : static_cast<uint32_t>(nck);
      nlab = clabels[r];
      nta = talpha ? talpha[r] : 0.0f;
    }
    if (tid == 0) nst = st[jj];
    fetch_keys(jj + 1);
  };
  fetch_keys(0);
  if (count > 0) fetch(0);

  int m = 0, j = 0, reason = 0;
  int64_t cyc[4] = {0, 0, 0, 0};
  int64_t tick = static_cast<int64_t>(__builtin_readcyclecounter());
  auto lap = [&](int ph) {
    const int64_t now = static_cast<int64_t>(__builtin_readcyclecounter());
    cyc[ph] += now - tick;
    tick = now;
  };
  for (; j < count; j++) {
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
        if (!((s_flags >> c) & 1u)) { v = s_ck[c]; plab = s_clab[c]; pta = s_cta[c]; psrc = c < 2 ? c : 2; }
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
        for (int u = 0; u < nupd; u++) if (wslot[uidx[u]] < 0) mm++;
        if (mm > slots) {
          stop = 2;                                // cache full: nothing of this sample is applied
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
              s_usrc[u] = wsrc[w];                 // 0 / 1: prefetched row, 2: fetch from the tiles
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
          fin[2 * static_cast<int64_t>(j)] = k0;
          fin[2 * static_cast<int64_t>(j) + 1] = k1;
        }
      }
      if (lane == 0) { s_stop = stop; s_nupd = nupd; }
    }
    __syncthreads();
    lap(2);
    if (s_stop) { reason = s_stop; break; }
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
        else { const uint32_t r = static_cast<uint32_t>(s_urow[u]); c = *tile_ptr(cb, r >> 6, tid, r & 63); }
        cache[tid * slots + sl] = adapt4(c, x, s_ua[u]);
      }
    }
    __syncthreads();
    lap(3);
  }
  // ---- write the corrected rows (and OLVQ1 rates) back ----
  for (int e = tid; e < m * d4; e += LVQ_BT) {
    const int sl = e / d4, q = e - sl * d4;
    const uint32_t r = static_cast<uint32_t>(s_slot_row[sl]);
    *tile_ptr_w(cb, r >> 6, q, r & 63) = cache[q * slots + sl];
  }
  if (talpha)
    for (int sl = tid; sl < m; sl += LVQ_BT) talpha[s_slot_row[sl]] = s_slot_ta[sl];
  if (mod_rows)
    for (int sl = tid; sl < m; sl += LVQ_BT) mod_rows[sl] = s_slot_row[sl];
  if (tid == 0) {
    ctl->consumed = j; ctl->nmod = m; ctl->reason = reason; ctl->pad = 0;
    for (int k = 0; k < 4; k++) ctl->cycles[k] = cyc[k];
  }
}

}  // namespace somhip
