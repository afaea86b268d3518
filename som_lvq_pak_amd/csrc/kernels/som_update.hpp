// kernels/som_update.hpp -- K4a-K4: members, launch order, in-order mini-batch update
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "scan_masked.hpp"

namespace somhip {

// =====================================================================================
// K4a: winners of a run -> lattice coordinates.  bxy[b] = (bx, by) of iteration b's
// best-matching unit (som_rout.c:641-642), from its key or its fixed point
// (som_rout.c:628-632); bx = -1 when the iteration teaches nothing (skipped sample, or
// no row beat FLT_MAX).
// =====================================================================================
__global__ void k_decode_winners(const uint64_t *__restrict__ keys, const StepScalars *__restrict__ sc,
                                 int64_t count, int xdim, int2 *__restrict__ bxy) {
  int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= count) return;
  const StepScalars s = sc[b];
  int2 o = make_int2(-1, -1);
  if (s.reach >= 0) {
    if (s.fixed >= 0) o = make_int2(fixed_x(s.fixed), fixed_y(s.fixed));
    else {
      uint64_t k = keys[b];
      if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) {
        const uint32_t widx = static_cast<uint32_t>(k);
        o = make_int2(static_cast<int>(widx % static_cast<uint32_t>(xdim)), static_cast<int>(widx / static_cast<uint32_t>(xdim)));
      }
    }
  }
  bxy[b] = o;
}

// =====================================================================================
// K4b: who updates whom.  For every row group (64 code rows) the samples of the run whose
// neighbourhood reaches it, in iteration order, each with the 64-bit mask of the member
// rows: hexa_dist/rect_dist <= radius (som_rout.c:496) decided per (row, sample) with the
// exact lattice arithmetic, after a cheap reach test on lattice rows.  One workgroup per
// row group, one thread per sample (256 at a time), ordered compaction by ballot/prefix.
// Gaussian neighbourhoods touch every row, so their list is every taught sample.
//   cnt[g]                 number of entries
//   ent[g*count + k]       {sample index in the run, member mask}
// =====================================================================================
struct MemberEntry { uint32_t sample; float alpha; unsigned long long mask; };   // alpha: the iteration's rate
// a group's list has room for every sample of the run plus LIST_PAD null entries (mask 0, offset 0) that K4b writes
// behind the real ones: K4s walks the list four entries at a time and prefetches three ahead without a bound check
constexpr int LIST_PAD = 8;
__host__ __device__ __forceinline__ int64_t list_stride(int64_t count) { return count + LIST_PAD; }

template <bool GAUSS, int NT, int RRT = 4>   // NT threads: 256, or 1024 when a small shard has few row groups to spread / the lists
                                             // will not be cut short; RRT samples per thread and trip
__global__ __launch_bounds__(NT) void k_som_members(CbView cb, int64_t count,
                                                     const int2 *__restrict__ bxy,
                                                     const uint64_t *__restrict__ keys,
                                                     const StepScalars *__restrict__ sc,
                                                     uint32_t *__restrict__ cnt,
                                                     MemberEntry *__restrict__ ent,
                                                     unsigned long long *__restrict__ stats,
                                                     int64_t xoff_first = -1, int64_t xoff_rows = 0,
                                                     int xoff_scale = 0, uint32_t tail_need = 0,
                                                     uint32_t *__restrict__ lstart = nullptr, int gauss_gemm = 0,
                                                     int reach_max = -1) {
  // reach_max >= 0 (with decoded winners, bubble): the largest `reach` of the run's iterations.  A winner whose box of
  // that reach misses the group is dropped on its 8 bytes of coordinates alone -- the 16 bytes of its step scalars are
  // only fetched for the samples that pass.  (With small neighbourhoods every workgroup otherwise reads the scalars of
  // every sample of the run to reject nearly all of them: 1024 groups x 32768 samples x 24 bytes through L2 per launch.)
  // gauss_gemm (GAUSS only; the consumer is K4m's gaussian form): the entry's mask field carries the winner's lattice
  // coordinates (x | y << 10, 10 bits each), 8 bits of remainder of, and in its upper word the float log2(e) / (2 radius^2) of the iteration; samples whose
  // rate would be below 2^-40 of alpha for every unit of the group are left out
  // xoff_first >= 0 (the consumer is K4s): the entry carries, instead of the sample's index in the run, where its
  // row starts in the data array in float4 units -- ((xoff_first + index) mod xoff_rows) * d / 4 -- so that the
  // update kernel's scalar unit adds instead of wrapping and multiplying per entry
  constexpr int RR = RRT;               // samples per thread and trip: their loads are issued together
  constexpr int NW = NT / 64;
  constexpr int SC = (RR * NW + 63) / 64;   // (round, wave) counts per lane of the wavefront that scans them
  static_assert(RR * NW <= 256, "the (round, wave) counts are scanned by one wavefront, up to four per lane");
  __shared__ uint32_t s_wcount[RR * NW + 1];             // counts, then exclusive prefix (+ total) per trip
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t g = blockIdx.x;
  const uint32_t xdim = static_cast<uint32_t>(cb.xdim);
  const int64_t r0 = g * WAVE;
  const int64_t r_last = (r0 + WAVE < cb.n ? r0 + WAVE : cb.n) - 1;
  const int nlive = static_cast<int>(r_last - r0 + 1);
  int g_tx0, g_ty0, g_txl, g_ty1;
  txty_of_row(cb, r0, g_tx0, g_ty0);
  txty_of_row(cb, r_last, g_txl, g_ty1);
  // x extent of the group's units: a patch is 8 wide; a linear group inside one map row spans
  // [first, last]; one that wraps covers everything
  const int g_tx1 = cb.patch_w ? g_tx0 + 7 : (g_ty0 == g_ty1 ? g_txl : static_cast<int>(xdim) - 1);
  const int g_txa = cb.patch_w ? g_tx0 : (g_ty0 == g_ty1 ? g_tx0 : 0);
  const bool small_map = cb.xdim <= 1024 && g_ty1 < 1024;
  const unsigned long long live_mask = nlive >= 64 ? ~0ull : ((1ull << nlive) - 1);
  MemberEntry *out = ent + g * list_stride(count);
  uint32_t base = 0;
  unsigned long long rows_total = 0, pairs_total = 0;
  // tail_need > 0 (the consumer is K4m, which walks a list from its END and stops once every unit's decay is below its
  // cut): only the tail of the list is made -- the trips run from the end of the batch, the entries land at the end
  // of the group's region (lstart[g] says where the list begins), and the walk stops after the trip in which the
  // tail has collected tail_need entries that hold EVERY live unit (the host sizes tail_need so that K4m stops
  // inside them, whole chunks included: the result is the one the full list gives, bit for bit).  At radius 128 that
  // is the last ~500 of a batch's 16 384 samples instead of all of them.
  __shared__ uint32_t s_full;
  if (tid == 0) s_full = 0u;
  // Two phases per trip (bubble, decoded winners).  Deciding membership is ~200 vector instructions per sample (eight
  // lattice rows, an integer square root each), and a wavefront pays them whenever ONE of its 64 samples is near the group:
  // with the samples in stream order that is nearly every wavefront from radius ~10 up, although only a tenth of the
  // samples are near (step_probe.py: 0.48 ms per 32768 vectors at radius 20-33, 0.27 ms at radius 2).  So the trip's
  // samples first go through the cheap box test and the near ones are queued in LDS, in sample order; the membership
  // arithmetic then runs over the queue with every lane busy.  The order of the entries -- queue order = sample order --
  // and every value in them are unchanged.
  __shared__ uint16_t s_q[NT * RR];
  const bool prepass = !GAUSS && !keys && reach_max >= 0;
  const bool tail = tail_need != 0u;
  uint32_t pos_end = static_cast<uint32_t>(count);
  const int64_t ntrips = (count + NT * RR - 1) / (NT * RR);

  for (int64_t trip = 0; trip < ntrips; trip++) {
    const int64_t b0 = tail ? count - (trip + 1) * (NT * RR) : trip * (NT * RR);   // (tail: may start below 0)
    unsigned long long mm[RR];
    float al[RR];
    int64_t bb[RR];
    uint32_t n_near = NT * RR;
    if (prepass) {
      // phase A: winners whose box of reach_max (>= every iteration's reach) touches the group, queued in sample order
      bool nr[RR];
      unsigned long long balq[RR];
#pragma unroll
      for (int r = 0; r < RR; r++) {
        const int64_t b = b0 + NT * r + tid;
        nr[r] = false;
        if (b >= 0 && b < count) {
          const int2 w = bxy[b];
          nr[r] = w.x >= 0 && w.y + reach_max >= g_ty0 && w.y - reach_max <= g_ty1 && w.x + reach_max >= g_txa && w.x - reach_max <= g_tx1;
        }
      }
#pragma unroll
      for (int r = 0; r < RR; r++) {
        balq[r] = __ballot(nr[r]);
        if (lane == 0) s_wcount[r * NW + wave] = __popcll(balq[r]);
      }
      __syncthreads();
      if (wave == 0) {
        uint32_t cc[SC], c = 0;                          // lane l: counts SC l .. SC l + SC - 1
#pragma unroll
        for (int k = 0; k < SC; k++) { cc[k] = SC * lane + k < RR * NW ? s_wcount[SC * lane + k] : 0u; c += cc[k]; }
        uint32_t inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = __shfl_up(inc, off, WAVE);
          if (lane >= off) inc += o;
        }
        uint32_t run = inc - c;
#pragma unroll
        for (int k = 0; k < SC; k++) { if (SC * lane + k < RR * NW) s_wcount[SC * lane + k] = run; run += cc[k]; }
        if (lane == 63) s_wcount[RR * NW] = inc;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RR; r++)
        if (nr[r]) s_q[s_wcount[r * NW + wave] + __popcll(balq[r] & ((1ull << lane) - 1))] = static_cast<uint16_t>(NT * r + tid);
      n_near = s_wcount[RR * NW];
      __syncthreads();                                   // the queue is whole; s_wcount is free for phase B
    }
#pragma unroll
    for (int r = 0; r < RR; r++) {
      // phase B: queue position NT r + tid (without the queue: the trip's sample of that number)
      const uint32_t qpos = static_cast<uint32_t>(NT * r + tid);
      const int64_t b = b0 + (prepass ? static_cast<int64_t>(qpos < n_near ? s_q[qpos] : 0) : static_cast<int64_t>(qpos));
      bb[r] = b;
      unsigned long long m = 0;
      float alpha_b = 0.f;
      bool near = qpos < n_near && b >= 0 && b < count;
      int2 w = make_int2(-1, -1);
      if (near) {
      const StepScalars s = sc[b];
      if (keys) {                                        // winners decoded here (K4a's rule), no extra launch
        w = make_int2(-1, -1);
        if (s.reach >= 0) {
          if (s.fixed >= 0) w = make_int2(fixed_x(s.fixed), fixed_y(s.fixed));
          else {
            const uint64_t k = keys[b];
            if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) {
              const uint32_t widx = static_cast<uint32_t>(k);
              w = make_int2(static_cast<int>(widx % xdim), static_cast<int>(widx / xdim));
            }
          }
        }
      } else {
        w = bxy[b];
      }
      alpha_b = s.alpha;
      // reach (rows) >= radius/0.866 + 1 also bounds the x extent (unit spacing 1, half-unit shifts)
      if (w.x >= 0 && w.y + s.reach >= g_ty0 && w.y - s.reach <= g_ty1 &&
          (GAUSS || (w.x + s.reach >= g_txa && w.x - s.reach <= g_tx1))) {
        if (GAUSS && gauss_gemm) {
          // nearest the winner can be to a unit of this group's 8 x 8 patch (or row run), conservatively: half a unit of
          // hexagonal shift taken off the x distance
          const int ddx = w.x < g_txa ? g_txa - w.x : (w.x > g_tx1 ? w.x - g_tx1 : 0);
          const int ddy = w.y < g_ty0 ? g_ty0 - w.y : (w.y > g_ty1 ? w.y - g_ty1 : 0);
          const double fx = ddx > 0 ? ddx - 0.5 : 0.0;
          const double lat_min = fx * fx + 0.75 * static_cast<double>(ddy) * ddy;
          const double coef = 1.4426950408889634 / (2.0 * static_cast<double>(s.thresh) * static_cast<double>(s.thresh));
          if (lat_min * coef <= 40.0) {
            // coef as a float + the remainder in units of 2^-32 coef (8 signed bits): the exponent reaches ~40, and a
            // float's 2^-24 on it would put 2e-6 into every rate -- a bias that adds up over thousands of hits
            const float cf = static_cast<float>(coef);
            int rq = static_cast<int>(rint((coef - static_cast<double>(cf)) / (static_cast<double>(cf) * 2.3283064365386963e-10)));
            rq = rq < -128 ? -128 : rq > 127 ? 127 : rq;
            m = (static_cast<unsigned long long>(__float_as_uint(cf)) << 32) | (static_cast<unsigned long long>(rq & 0xFF) << 20) |
                (static_cast<unsigned long long>(static_cast<uint32_t>(w.y) & 0x3FFu) << 10) | (static_cast<uint32_t>(w.x) & 0x3FFu);
          }
        } else if (GAUSS) m = live_mask;
        else if (cb.patch_w && small_map) {
          // 8x8 patch, exact integer form: with every lattice quantity a multiple of 1/4,
          //   lattice_sq <= thresh  <=>  (2dx)^2 + 3 dy^2 <= floor(4 thresh)   (hexa)
          //                              dx^2 + dy^2     <= floor(thresh)     (rect)
          // and in one lattice row the members are a contiguous run of tx.
          const bool rect = cb.topol == 4;
          const int K = static_cast<int>(floor(static_cast<double>(s.thresh) * (rect ? 1.0 : 4.0)));
          if (K >= 0) {
            // branch-free per lattice row: W = floor(sqrt(rem)) from the hardware's 1-ulp sqrt, corrected by
            // integer comparisons (rem < 2^24: exact in fp32); ceil / floor of the halves by arithmetic shifts;
            // the row's run of members as 8 bits of one of two 32-bit words
            const int c3 = rect ? 1 : 3;
            uint32_t mw0 = 0, mw1 = 0;
#pragma unroll
            for (int iy = 0; iy < 8; iy++) {
              const int dy = w.y - (g_ty0 + iy);
              const int rem = K - c3 * dy * dy;
              int W = static_cast<int>(__builtin_amdgcn_sqrtf(static_cast<float>(rem < 0 ? 0 : rem)));
              W += ((W + 1) * (W + 1) <= rem);
              W += ((W + 1) * (W + 1) <= rem);
              W -= (W * W > rem);
              W -= (W * W > rem);
              // rect: |bx - tx| <= W.  hexa: |2(bx - tx) + o| <= W, o = 0 on same-parity rows,
              // -1 when by is even, +1 when by is odd (som_rout.c:440-447):  2 bx + o - W <= 2 tx <= 2 bx + o + W
              int lo, hi;
              if (rect) { lo = w.x - W; hi = w.x + W; }
              else {
                const int cx = 2 * w.x + ((dy & 1) ? ((w.y & 1) ? 1 : -1) : 0);
                lo = (cx - W + 1) >> 1;                    // ceil((cx - W) / 2)
                hi = (cx + W) >> 1;                        // floor((cx + W) / 2)
              }
              lo = (lo < g_tx0 ? g_tx0 : lo) - g_tx0;
              hi = (hi > g_tx0 + 7 ? g_tx0 + 7 : hi) - g_tx0;
              const int len = hi - lo + 1;                 // <= 8
              const uint32_t run = (rem >= 0 && len > 0) ? (((1u << len) - 1u) << lo) : 0u;
              if (iy < 4) mw0 |= run << (8 * iy); else mw1 |= run << (8 * (iy - 4));
            }
            m = (static_cast<unsigned long long>(mw1) << 32) | mw0;
          }
        } else if (cb.patch_w) {
          for (int u = 0; u < 64; u++) {                 // maps wider than 1024: per-unit test
            const int tx = g_tx0 + (u & 7), ty = g_ty0 + (u >> 3);
            if (lattice_sq(cb.topol, w.x, w.y, tx, ty) <= s.thresh) m |= 1ull << u;
          }
        } else {
          int tx = g_tx0, ty = g_ty0;
          for (int u = 0; u < nlive; u++) {
            const float lsq = small_map ? lattice_sq_small(cb.topol, w.x, w.y, tx, ty)
                                        : lattice_sq(cb.topol, w.x, w.y, tx, ty);
            if (lsq <= s.thresh) m |= 1ull << u;
            if (++tx == static_cast<int>(xdim)) { tx = 0; ty++; }
          }
        }
      }
    }
      mm[r] = m;
      al[r] = alpha_b;
    }
    // ordered compaction of the trip: ballots -> (round, wave) counts -> one wavefront scans them
    unsigned long long bal[RR];
    uint32_t nfull = 0;
#pragma unroll
    for (int r = 0; r < RR; r++) {
      bal[r] = __ballot(mm[r] != 0);
      if (lane == 0) s_wcount[r * NW + wave] = __popcll(bal[r]);
      if (tail) nfull += __popcll(__ballot((mm[r] & live_mask) == live_mask));
    }
    if (tail && lane == 0 && nfull) atomicAdd(&s_full, nfull);
    __syncthreads();
    if (wave == 0) {
      uint32_t cc[SC], c = 0;                            // lane l: counts SC l .. SC l + SC - 1
#pragma unroll
      for (int k = 0; k < SC; k++) { cc[k] = SC * lane + k < RR * NW ? s_wcount[SC * lane + k] : 0u; c += cc[k]; }
      uint32_t inc = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, WAVE);
        if (lane >= off) inc += o;
      }
      uint32_t run = inc - c;
#pragma unroll
      for (int k = 0; k < SC; k++) { if (SC * lane + k < RR * NW) s_wcount[SC * lane + k] = run; run += cc[k]; }
      if (lane == 63) s_wcount[RR * NW] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RR; r++) {
      const unsigned long long m = mm[r];
      if (m != 0) {
        MemberEntry e;
        const int64_t bidx = bb[r];
        e.sample = xoff_first < 0 ? static_cast<uint32_t>(bidx)
                                  : static_cast<uint32_t>(((xoff_first + bidx) % xoff_rows) * xoff_scale);   // scale: d/4 (float4 units) or 4 d (bytes)
        e.alpha = al[r]; e.mask = m;
        const uint32_t at = tail ? pos_end - s_wcount[RR * NW] : base;
        out[at + s_wcount[r * NW + wave] + __popcll(bal[r] & ((1ull << lane) - 1))] = e;
        rows_total += (GAUSS && gauss_gemm) ? nlive : __popcll(m);
        pairs_total += 1;
      }
    }
    if (tail) pos_end -= s_wcount[RR * NW]; else base += s_wcount[RR * NW];
    const bool enough = tail && s_full >= tail_need;
    __syncthreads();
    if (enough) break;
  }
  if (tail) base = static_cast<uint32_t>(count) - pos_end;
  if (tid == 0) { cnt[g] = base; if (lstart) lstart[g] = tail ? pos_end : 0u; }
  if (tid < LIST_PAD) { MemberEntry z; z.sample = 0u; z.alpha = 0.0f; z.mask = 0ull; out[(tail ? static_cast<uint32_t>(count) : base) + tid] = z; }
  // instrumentation: (row, iteration) updates and (row group, iteration) pairs of this run
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    rows_total += __shfl_xor(rows_total, off, WAVE);
    pairs_total += __shfl_xor(pairs_total, off, WAVE);
  }
  if (lane == 0 && stats) {     // 64 counter pairs (summed by the host): one pair took ~8 000 same-address atomics
    unsigned long long *st = stats + 8 + 2 * (g & 63);
    if (rows_total) atomicAdd(st, rows_total);
    if (pairs_total) atomicAdd(st + 1, pairs_total);
  }
}

// K4c: launch order for K4 -- row groups by member count, heaviest first, so the long
// workgroups start early and the tail of the launch is made of short ones (rank by counting;
// ties by index).  order[rank] = group.
__global__ __launch_bounds__(256) void k_order_groups(const uint32_t *__restrict__ cnt, int ngroups,
                                                      uint32_t *__restrict__ order) {
  __shared__ uint32_t s_cnt[8192];                     // host guarantees ngroups <= 8192
  for (int k = threadIdx.x; k < ngroups; k += blockDim.x) s_cnt[k] = cnt[k];
  __syncthreads();
  // 8 lanes share one group's ranking (an eighth of the comparisons each)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = t >> 3, part = t & 7;
  const bool live = g < ngroups;
  const uint32_t mine = live ? s_cnt[g] : 0u;
  uint32_t rank = 0;
  if (live)
    for (int k = part; k < ngroups; k += 8) {
      const uint32_t c = s_cnt[k];
      rank += (c > mine) || (c == mine && k < g);
    }
  rank += __shfl_xor(rank, 1, WAVE);
  rank += __shfl_xor(rank, 2, WAVE);
  rank += __shfl_xor(rank, 4, WAVE);
  if (live && part == 0) order[rank] = static_cast<uint32_t>(g);
}

// =====================================================================================
// K4: in-order neighbourhood update of a run of samples, driven by K4b's member lists.
//
// bubble_adapt (som_rout.c:472-506) / gaussian_adapt (:511-549) + adapt_vector
// (lvq_pak.c:339-351) for iterations batch_start .. batch_start+count-1, applied to
// every code row in iteration order.  One lane = one code row, QW chunks (4*QW dims) of
// it held in registers across the whole run, so each touched row is read and written
// once per run whatever the batch size.  A workgroup = ONE row group x 4 consecutive dim
// slices (one per wave): the four waves see the same member list, so they stay balanced
// between barriers.  The list is walked in tiles of TB entries: entry scalars -> LDS, the
// tile's sample slices staged into LDS (one coalesced pass), then every wave applies the
// tile's updates in order (member lanes from the entry's mask, x as broadcast reads).
// =====================================================================================
//
// PIPE = true is the form for small shards (few waves per SIMD, nothing to hide LDS latency behind):
// the lane's membership bits of the whole tile are gathered first, then the entries are walked in
// unrolled groups of 8 with the next entry's x chunks and alpha already in flight while the current
// one is applied.  Same operations on the same values in the same order as PIPE = false.
template <int QW, int TB, bool GAUSS, bool MASKED, bool PIPE = false>
__global__ __launch_bounds__(256) void k_som_update_run(CbView cb, const float *__restrict__ rows,
                                                        const uint8_t *__restrict__ mask,
                                                        int64_t n_rows, int64_t data_first,
                                                        int64_t count,
                                                        const int2 *__restrict__ bxy,
                                                        const StepScalars *__restrict__ sc,
                                                        const uint32_t *__restrict__ cnt,
                                                        const MemberEntry *__restrict__ ent,
                                                        const uint32_t *__restrict__ order) {
  constexpr int BQ = 4 * QW;                          // chunks per workgroup
  static_assert(TB % 8 == 0, "PIPE walks the tile in groups of 8 entries");
  __shared__ float4 xs[TB + (PIPE ? 1 : 0)][BQ];              // PIPE: one row of slack for the last prefetch
  __shared__ uint32_t ms[MASKED ? TB : 1][MASKED ? BQ : 1];   // 4 mask bits per chunk
  __shared__ float s_ga[GAUSS ? TB + 1 : 1][GAUSS ? WAVE : 1];    // gaussian: per-lane alpha
  __shared__ unsigned long long s_mask[TB];
  __shared__ long long s_xoff[TB];
  __shared__ float s_alpha[TB + 1], s_thr[TB];
  __shared__ int s_bx[TB], s_by[TB];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Work items = (row group, slice of 4*QW chunks), heaviest groups first (K4c).  The launch is one-dimensional
  // and walks the item list boustrophedon in blocks of 256 (one per CU): a small shard's workgroups are all
  // resident at once, placed round-robin, and would otherwise stack the 8 or 16 slices of the heaviest groups
  // on the same CUs; a big launch still starts its heaviest items first.
  const uint32_t nslices = static_cast<uint32_t>((cb.d4 + BQ - 1) / BQ);
  const uint32_t total = static_cast<uint32_t>(cb.ngroups) * nslices;
  const uint32_t blk = blockIdx.x / 256u, pos = blockIdx.x % 256u;
  const uint32_t bsize = total - blk * 256u < 256u ? total - blk * 256u : 256u;
  const uint32_t item = blk * 256u + ((blk & 1u) ? bsize - 1u - pos : pos);
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const uint32_t n_ent = cnt[g];
  if (n_ent == 0) return;                             // nothing in this run touches the group
  const int qblk = static_cast<int>(item % nslices) * BQ;
  const int q0 = qblk + wave * QW;
  const bool vec = (cb.d & 3) == 0;
  const MemberEntry *list = ent + g * list_stride(count);
  int tx, ty;
  txty_of_row(cb, g * WAVE + lane, tx, ty);

  float4 c[QW];
#pragma unroll
  for (int j = 0; j < QW; j++)
    c[j] = (q0 + j < cb.d4) ? *tile_ptr(cb, g, q0 + j, lane) : make_float4(0.f, 0.f, 0.f, 0.f);

  for (uint32_t k0 = 0; k0 < n_ent; k0 += TB) {
    const int tb = static_cast<int>(n_ent - k0 < TB ? n_ent - k0 : TB);
    // ---- entry scalars
    if (tid < tb) {
      const MemberEntry e = list[k0 + tid];
      const StepScalars s = sc[e.sample];
      s_mask[tid] = e.mask;
      s_alpha[tid] = s.alpha;
      s_xoff[tid] = ((data_first + e.sample) % n_rows) * cb.d;
      if (GAUSS) { const int2 w = bxy[e.sample]; s_bx[tid] = w.x; s_by[tid] = w.y; s_thr[tid] = s.thresh; }
    } else if (PIPE && tid < TB) {
      s_mask[tid] = 0ull;                             // PIPE gathers the bits of all TB slots
    }
    __syncthreads();
    // ---- sample slices -> LDS (+ gaussian: per-lane alpha, once per workgroup)
    for (int e = tid; e < tb * BQ; e += 256) {
      const int i = e / BQ, j = e % BQ, q = qblk + j;
      if (q < cb.d4) {
        const long long xo = s_xoff[i];
        const float *xr = rows + xo;
        xs[i][j] = vec ? reinterpret_cast<const float4 *>(xr)[q] : load_x4<false>(xr, q, cb.d);
        if (MASKED) {
          const uint8_t *mk = mask + xo;
          uint32_t mm = 0;
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (q * 4 + u >= cb.d || mk[q * 4 + u] != 0) mm |= 1u << u;
          ms[i][j] = mm;
        }
      }
    }
    if (GAUSS) {
      for (int i = wave; i < tb; i += 4)
        s_ga[i][lane] = gaussian_alpha(lattice_sq(cb.topol, s_bx[i], s_by[i], tx, ty), s_thr[i], s_alpha[i]);
    }
    __syncthreads();
    // ---- the tile's updates, in iteration order
    auto apply = [&](int i, const float4 *x, float a) {
      if (!MASKED && PIPE) {                          // packed fp32: two elements per instruction, each half rounded like the scalar op
        const f32x2 a2 = {a, a};
#pragma unroll
        for (int j = 0; j < QW; j++) {
          f32x2 lo = {c[j].x, c[j].y}, hi = {c[j].z, c[j].w};
          const f32x2 tl = f32x2{x[j].x, x[j].y} - lo, th = f32x2{x[j].z, x[j].w} - hi;
          const f32x2 sl = a2 * tl, sh = a2 * th;
          lo = lo + sl; hi = hi + sh;
          c[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < QW; j++) {
        const float4 n = adapt4(c[j], x[j], a);
        if (MASKED) {
          const uint32_t mm = ms[i][wave * QW + j];
          if (!(mm & 1u)) c[j].x = n.x;
          if (!(mm & 2u)) c[j].y = n.y;
          if (!(mm & 4u)) c[j].z = n.z;
          if (!(mm & 8u)) c[j].w = n.w;
        } else {
          c[j] = n;      // padding dims: x pad = 0, c pad = 0 -> stays 0
        }
      }
    };
    if (PIPE) {
      if (q0 < cb.d4) {
        const uint32_t *m32 = reinterpret_cast<const uint32_t *>(s_mask) + (lane >> 5);
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < TB; i++) bits |= ((m32[2 * i] >> (lane & 31)) & 1u) << i;
        float4 xa[QW], xb[QW];
        float aa, ab;
        auto fetch = [&](int i, float4 *x, float &a) {
#pragma unroll
          for (int j = 0; j < QW; j++) x[j] = xs[i][wave * QW + j];
          a = GAUSS ? s_ga[i][lane] : s_alpha[i];
        };
        fetch(0, xa, aa);
        for (int i0 = 0; i0 < tb; i0 += 8) {
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            fetch(i0 + u + 1, xb, ab);
            if (bits & (1u << u)) apply(i0 + u, xa, aa);
            fetch(i0 + u + 2, xa, aa);
            if (bits & (2u << u)) apply(i0 + u + 1, xb, ab);
          }
          bits >>= 8;
        }
      }
    } else if (q0 < cb.d4) {
      for (int i = 0; i < tb; i++) {
        if ((s_mask[i] >> lane) & 1ull) apply(i, &xs[i][wave * QW], GAUSS ? s_ga[i][lane] : s_alpha[i]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < QW; j++)
    if (q0 + j < cb.d4) *tile_ptr_w(cb, g, q0 + j, lane) = c[j];
}

// =====================================================================================
// K4s: the same in-order update for the common case (bubble neighbourhood, no masks, dim % 4 == 0), with
// everything that is uniform over a wavefront kept out of the vector unit and out of LDS.
//
// In K4 every entry costs each wave a broadcast ds_read_b128 per chunk (the LDS pipe, shared by the CU's four
// SIMDs, saturates before the vector ALUs do), a staging pass and two barriers per tile, and ~8 vector
// instructions to turn the entry's mask into an exec mask.  Here a wave is on its own: the member entry
// {sample, alpha, mask} and the sample's 4*QW dims arrive by *scalar* loads (constant address space: the
// list and the data are not written by this kernel), the mask becomes the exec mask directly
// (inverse ballot), alpha and x are SGPR operands of the sub/mul/add, and the next entry's loads are in
// flight while the current one is applied.  No LDS, no barriers; arithmetic and order exactly K4's.
// =====================================================================================
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// The scalar loads and the waits are inline assembly: the compiler would otherwise sink each load to its
// first use (inside the next entry's exec-masked block), i.e. issue it and wait for it at once.  Scalar
// loads return out of order, so a wait is always lgkmcnt(0); a buffer is only read after it has been passed
// through k4s_wait (in/out operand), which is what orders its uses behind the wait.
__device__ __forceinline__ const float *k4s_uniform(const float *p) {     // the same address in every lane -> an SGPR pair
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(u >> 32));
  return reinterpret_cast<const float *>((static_cast<uint64_t>(hi) << 32) | lo);
}
// x (SGPR pair) - c (VGPR pair), packed: written out because the compiler, given a VGPR rate, copies x into
// VGPRs first (14 v_mov per entry) instead of using the scalar operand
__device__ __forceinline__ f32x2 k4s_pk_sub(f32x2 x_sgpr, f32x2 c) {
  f32x2 t;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "s"(x_sgpr), "v"(c));
  return t;
}
template <int NF> struct K4sX;
template <> struct K4sX<8> {
  f32x8_t v;
  __device__ __forceinline__ void load(const float *p) { asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p)); }
  __device__ __forceinline__ void load_off(const float *base, uint32_t off) { asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(base), "s"(off)); }
  __device__ __forceinline__ void wait(u32x4_t &e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(e)); }
  __device__ __forceinline__ float4 chunk(int j) const { return make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]); }
};
template <> struct K4sX<16> {
  f32x16_t v;
  __device__ __forceinline__ void load(const float *p) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p)); }
  __device__ __forceinline__ void load_off(const float *base, uint32_t off) { asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "s"(off)); }
  __device__ __forceinline__ void wait(u32x4_t &e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(e)); }
  // (the VGPR operand goes through its own, empty statement: an asm with a vector output makes all its outputs divergent)
  __device__ __forceinline__ void wait(u32x4_t &e, float &a) { wait(e); asm volatile("" : "+v"(a)); }
  __device__ __forceinline__ float4 chunk(int j) const { return make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]); }
};
template <> struct K4sX<32> {
  f32x16_t v, w;
  __device__ __forceinline__ void load(const float *p) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(v), "=&s"(w) : "s"(p));
  }
  __device__ __forceinline__ void wait(u32x4_t &e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(w), "+s"(e)); }
  __device__ __forceinline__ void wait(u32x4_t &e, float &a) { wait(e); asm volatile("" : "+v"(a)); }
  __device__ __forceinline__ float4 chunk(int j) const {
    return j < 4 ? make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3])
                 : make_float4(w[4 * j - 16], w[4 * j - 15], w[4 * j - 14], w[4 * j - 13]);
  }
};
__device__ __forceinline__ void k4s_load_entry(u32x4_t &e, const MemberEntry *p) {
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(e) : "s"(p));
}
__device__ __forceinline__ void k4s_load_entry_off(u32x4_t &e, const MemberEntry *base, uint32_t off) {
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(e) : "s"(base), "s"(off));
}

// OFF32 (data set smaller than 4 GiB): the entries carry the BYTE offset of the sample's row and both kinds of
// scalar load take their address as base + register offset, so an entry costs the scalar unit three instructions
// of address work instead of ten (8 scalar instructions per entry in all; the measured time did not change --
// neither on the whole map, where the vector ALU is the bound, nor on an eighth of it).
template <int QW, bool PK = false, bool OFF32 = false>
__global__ __launch_bounds__(256) void k_som_update_bubble_s(CbView cb, const float *__restrict__ rows,
                                                             int64_t n_rows, int64_t data_first, int64_t count,
                                                             const uint32_t *__restrict__ cnt,
                                                             const MemberEntry *__restrict__ ent,
                                                             const uint32_t *__restrict__ order) {
  static_assert(sizeof(MemberEntry) == 16, "one s_load_dwordx4 per entry");
  constexpr int BQ = 4 * QW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // (row group, slice) items in the CU-balanced order of K4
  const uint32_t nslices = static_cast<uint32_t>(cb.d4 / BQ);          // host: d4 % BQ == 0
  const uint32_t total = static_cast<uint32_t>(cb.ngroups) * nslices;
  const uint32_t blk = blockIdx.x / 256u, pos = blockIdx.x % 256u;
  const uint32_t bsize = total - blk * 256u < 256u ? total - blk * 256u : 256u;
  const uint32_t item = blk * 256u + ((blk & 1u) ? bsize - 1u - pos : pos);
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const uint32_t n_ent = cnt[g];
  if (n_ent == 0) return;
  const int q0 = static_cast<int>(item % nslices) * BQ + wave * QW;

  float4 c[QW];
#pragma unroll
  for (int j = 0; j < QW; j++) c[j] = *tile_ptr(cb, g, q0 + j, lane);
  // (waited for here, once: otherwise the compiler keeps one s_waitcnt vmcnt per chunk inside every phase of the loop)
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)

  const MemberEntry *list = ent + g * list_stride(count);
  const float *xbase = rows + 4 * q0;
  const uint32_t last = n_ent - 1u;
  // the entry's first word is the row's start in float4 units (K4b's xoff mode; host: n_rows * d / 4 < 2^32)
  auto xrow = [&](const u32x4_t &e) -> const float * {
    return xbase + (static_cast<uint64_t>(e.x) << 2);
  };
  auto entry = [&](uint32_t k) -> const MemberEntry * {      // clamped: past the end the last entry is re-read, never applied
    return list + __builtin_amdgcn_readfirstlane(static_cast<int>(k < last ? k : last));
  };
  auto apply = [&](const u32x4_t &e, const K4sX<4 * QW> &x) {
    if (__builtin_amdgcn_inverse_ballot_w64((static_cast<unsigned long long>(e.w) << 32) | e.z)) {
      const float a = __uint_as_float(e.y);
      if (PK) {                                        // v_pk_add/mul_f32: two elements per instruction, each half rounded like the scalar op
        const f32x2 a2 = {a, a};
#pragma unroll
        for (int j = 0; j < QW; j++) {
          const float4 xv = x.chunk(j);
          f32x2 lo = {c[j].x, c[j].y}, hi = {c[j].z, c[j].w};
          const f32x2 tl = f32x2{xv.x, xv.y} - lo, th = f32x2{xv.z, xv.w} - hi;
          const f32x2 sl = a2 * tl, sh = a2 * th;
          lo = lo + sl; hi = hi + sh;
          c[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      } else {
#pragma unroll
        for (int j = 0; j < QW; j++) c[j] = adapt4(c[j], x.chunk(j), a);
      }
    }
  };
  // Phase k: wait -> x_k and entry k+2 have landed; issue x_{k+1} and entry k+3; apply entry k from registers.
  // Four entry buffers and two x buffers in pure rotation (the loop is unrolled over the four phases), so no
  // value ever has to be copied between a load and the wait that makes it readable -- the compiler knows nothing
  // about that window (tests/test_build.py checks the ISA for it).
  u32x4_t e0, e1, e2, e3;
  K4sX<4 * QW> xA, xB;
  if constexpr (OFF32) {
    // the list is followed by LIST_PAD null entries: no clamping of the prefetches, one exit test per four entries
    uint32_t ko = 32u;                                    // byte offset of the entry loaded next
    k4s_load_entry(e0, list);
    k4s_load_entry_off(e1, list, 16u);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
    xA.load_off(xbase, e0.x);
    k4s_load_entry_off(e2, list, ko);
#define K4S_PHASE(XC, XN, EK, EK1, EK2, EK3)        \
    XC.wait(EK2);                                     \
    XN.load_off(xbase, EK1.x);                        \
    ko += 16u;                                        \
    k4s_load_entry_off(EK3, list, ko);                \
    apply(EK, XC);
    for (uint32_t k = 0; k < n_ent; k += 4u) {
      K4S_PHASE(xA, xB, e0, e1, e2, e3)
      K4S_PHASE(xB, xA, e1, e2, e3, e0)
      K4S_PHASE(xA, xB, e2, e3, e0, e1)
      K4S_PHASE(xB, xA, e3, e0, e1, e2)
    }
#undef K4S_PHASE
  } else {
  k4s_load_entry(e0, entry(0));
  k4s_load_entry(e1, entry(1));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
  xA.load(xrow(e0));
  k4s_load_entry(e2, entry(2));
#define K4S_PHASE(XC, XN, EK, EK1, EK2, EK3) \
  XC.wait(EK2);                              \
  XN.load(xrow(EK1));                        \
  k4s_load_entry(EK3, entry(k + 3u));        \
  apply(EK, XC);                             \
  if (++k >= n_ent) break;
  for (uint32_t k = 0;;) {
    K4S_PHASE(xA, xB, e0, e1, e2, e3)
    K4S_PHASE(xB, xA, e1, e2, e3, e0)
    K4S_PHASE(xA, xB, e2, e3, e0, e1)
    K4S_PHASE(xB, xA, e3, e0, e1, e2)
  }
#undef K4S_PHASE
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nothing of ours may still be landing in SGPRs at exit
#pragma unroll
  for (int j = 0; j < QW; j++) *tile_ptr_w(cb, g, q0 + j, lane) = c[j];
}

// =====================================================================================
// K4g: gaussian neighbourhoods (gaussian_adapt, som_rout.c:511-549) in the same scalar-operand form.  Every
// unit is updated by every sample with its own rate alpha * exp(-dist^2 / (2 radius^2)) -- two fp64
// transcendentals per (unit, sample), as expensive as the update of 300 dims.  One workgroup of up to 16 waves
// holds ALL the dims of a row group (wave w: chunks [w*QW, (w+1)*QW)), so a tile's 32 x 64 rates are computed
// once, two entries per wave, into LDS (double-buffered: one barrier per tile) instead of once per dim slice;
// then every wave applies the tile with x from scalar loads, the lane's rate from one ds_read_b32 (issued an
// entry ahead, under the same wait as the scalar loads) and packed fp32 arithmetic.
// =====================================================================================
typedef const __attribute__((address_space(4))) u32x4_t konst_u32x4;
typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) i32x2_t konst_i32x2;

__device__ __forceinline__ double k4g_readlane(double v, uint32_t l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), static_cast<int>(l));
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), static_cast<int>(l));
  return __hiloint2double(hi, lo);
}

template <int QW>
__global__ __launch_bounds__(1024) void k_som_update_gauss_s(CbView cb, const float *__restrict__ rows,
                                                             int64_t n_rows, int64_t data_first, int64_t count,
                                                             const int2 *__restrict__ bxy,
                                                             const StepScalars *__restrict__ sc,
                                                             const uint32_t *__restrict__ cnt,
                                                             const MemberEntry *__restrict__ ent,
                                                             const uint32_t *__restrict__ order) {
  static_assert(sizeof(StepScalars) == 16 && sizeof(MemberEntry) == 16, "scalar loads of 16 bytes");
  constexpr int TB = 32;
  __shared__ float s_ga[2][TB][WAVE];
  const int nw = static_cast<int>(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t nslices = static_cast<uint32_t>(cb.d4 / (QW * nw));          // host: d4 % (QW * nw) == 0
  const uint32_t total = static_cast<uint32_t>(cb.ngroups) * nslices;
  const uint32_t blk = blockIdx.x / 256u, pos = blockIdx.x % 256u;
  const uint32_t bsize = total - blk * 256u < 256u ? total - blk * 256u : 256u;
  const uint32_t item = blk * 256u + ((blk & 1u) ? bsize - 1u - pos : pos);
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const uint32_t n_ent = cnt[g];
  if (n_ent == 0) return;                              // the whole workgroup
  const int q0 = static_cast<int>(item % nslices) * QW * nw + wave * QW;
  int tx, ty;
  txty_of_row(cb, g * WAVE + lane, tx, ty);

  float4 c[QW];
#pragma unroll
  for (int j = 0; j < QW; j++) c[j] = *tile_ptr(cb, g, q0 + j, lane);
  // (waited for here, once: otherwise the compiler keeps one s_waitcnt vmcnt per chunk inside every phase of the loop)
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)

  const MemberEntry *list = ent + g * list_stride(count);
  const float *xbase = rows + 4 * q0;
  const uint32_t last = n_ent - 1u;
  const uint32_t nr = static_cast<uint32_t>(n_rows), df = static_cast<uint32_t>(data_first);
  auto xrow = [&](const u32x4_t &e) -> const float * {
    uint32_t r = df + e.x;
    if (r >= nr) r -= nr;
    return k4s_uniform(xbase + static_cast<int64_t>(r) * cb.d);
  };
  auto entry = [&](uint32_t k) -> const MemberEntry * {
    return list + __builtin_amdgcn_readfirstlane(static_cast<int>(k < last ? k : last));
  };
  const uint32_t lds_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float *)&s_ga[0][0][lane]));
  auto rate_read = [&](float &a, uint32_t k) {         // rate of entry k for this lane: tile (k / TB) & 1, slot k % TB
    const uint32_t addr = lds_lane + ((k & (2u * TB - 1u)) * WAVE) * 4u;
    asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(addr) : "memory");
  };
  auto tile_top = [&](uint32_t k) {                    // rates of entries [k, k + TB): two (or more) entries per wave
    const uint32_t tb = n_ent - k < TB ? n_ent - k : TB;
    const int buf = (k / TB) & 1;
    for (uint32_t i = wave; i < tb; i += nw) {
      const u32x4_t e = reinterpret_cast<konst_u32x4 *>((const __attribute__((address_space(4))) MemberEntry *)list)[k + i];
      const u32x4_t s = ((konst_u32x4 *)sc)[e.x];      // {alpha, thresh (= radius), fixed, reach}
      const i32x2_t w = ((konst_i32x2 *)bxy)[e.x];
      s_ga[buf][i][lane] = gaussian_alpha(lattice_sq(cb.topol, w.x, w.y, tx, ty), __uint_as_float(s.y), __uint_as_float(s.x));
    }
    __syncthreads();
  };
  auto apply = [&](const u32x4_t &e, const K4sX<4 * QW> &x, float a) {
    if (__builtin_amdgcn_inverse_ballot_w64((static_cast<unsigned long long>(e.w) << 32) | e.z)) {
      const f32x2 a2 = {a, a};
#pragma unroll
      for (int j = 0; j < QW; j++) {
        const float4 xv = x.chunk(j);
        f32x2 lo = {c[j].x, c[j].y}, hi = {c[j].z, c[j].w};
        const f32x2 tl = k4s_pk_sub(f32x2{xv.x, xv.y}, lo), th = k4s_pk_sub(f32x2{xv.z, xv.w}, hi);
        const f32x2 sl = a2 * tl, sh = a2 * th;
        lo = lo + sl; hi = hi + sh;
        c[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    }
  };
  u32x4_t e0, e1, e2, e3;
  K4sX<4 * QW> xA, xB;
  float aA = 0.f, aB = 0.f;
  k4s_load_entry(e0, entry(0));
  k4s_load_entry(e1, entry(1));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
  xA.load(xrow(e0));
  k4s_load_entry(e2, entry(2));
  // phase k as in K4s; the rate of entry k+1 is read together with x_{k+1} unless k+1 opens a tile (its rates do
  // not exist yet: the tile top computes them, passes the barrier and reads the first one itself)
#define K4G_PHASE(XC, XN, AC, AN, EK, EK1, EK2, EK3, FIRST, LAST4)             \
  if (FIRST && (k & (TB - 1u)) == 0u) { tile_top(k); rate_read(AC, k); }       \
  XC.wait(EK2, AC);                                                            \
  XN.load(xrow(EK1));                                                          \
  k4s_load_entry(EK3, entry(k + 3u));                                          \
  if (!(LAST4 && ((k + 1u) & (TB - 1u)) == 0u) && k + 1u < n_ent) rate_read(AN, k + 1u); \
  apply(EK, XC, AC);                                                           \
  if (++k >= n_ent) break;
  for (uint32_t k = 0;;) {
    K4G_PHASE(xA, xB, aA, aB, e0, e1, e2, e3, true, false)
    K4G_PHASE(xB, xA, aB, aA, e1, e2, e3, e0, false, false)
    K4G_PHASE(xA, xB, aA, aB, e2, e3, e0, e1, false, false)
    K4G_PHASE(xB, xA, aB, aA, e3, e0, e1, e2, false, true)
  }
#undef K4G_PHASE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < QW; j++) *tile_ptr_w(cb, g, q0 + j, lane) = c[j];
}

// =====================================================================================
// K4h: K4g for runs that lie in one piece in the data set (no wrap inside the batch, the batch smaller than 4 GiB),
// dims in whole 32s; the rate in gauss_rate.hpp's short form (FAST false: the library chain, SOMHIP_GAUSS_LIBM=1).
// K4g costs the CU's ONE scalar unit ~36 instructions per (wave, entry) -- list entry, clamped index, 64-bit row
// address, tile tests, four branches -- against 24 packed vector instructions: with 32 waves on a CU the scalar unit
// is the bound (36 x 32 = 1152 issue cycles per round of entries against 768 of vector work per SIMD).  Here:
//   * a wave holds 32 dims of its 64 rows (8 chunks) and applies an entry in two halves of 16 dims, each with its own
//     16-SGPR operand buffer: the half being applied and the half in flight -- 48 vector instructions per entry;
//   * a tile's 32 sample indices come with ONE vector load, lane l = entry l, and reach
//     the scalar side by v_readlane; the row address is base + index * row bytes in 32 bits;
//   * no list entry, no mask: under the gaussian neighbourhood every listed sample teaches every live row, so the
//     dead rows of a last, partial group just compute along and are not stored;
//   * rates two tiles deep in LDS: the tile being applied and the one being written (K4g: written, barrier, applied --
//     every wave waits for the slowest rate), one barrier per tile of 64, no test inside a tile: 2 entries per loop trip; a
//     last partial tile goes entry by entry without look-ahead;
//   * the scalars of a tile's entries (rate, radius, winner) by vector loads, lane l = entry l, handed to the two
//     entries a wave computes the rates of by v_readlane.
// ~12 scalar instructions per (wave, entry).  Bits: the same three roundings per element in the same order.
// =====================================================================================
// (a call, not inlined: the library chain wants ~40 registers of its own, which a wave of 8 per SIMD does not have next to
// its 32 of row data -- around the rare call they go to scratch)
__device__ __attribute__((noinline)) float gaussian_alpha_call(float lat_sq, float radius, float alpha) {
  return gaussian_alpha(lat_sq, radius, alpha);
}
struct K4hX {                                              // 16 dims of a sample's row as scalar operands
  f32x16_t v;
  __device__ __forceinline__ void load(const float *p) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p)); }
  __device__ __forceinline__ void load_40(const float *p) { asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(v) : "s"(p)); }
  __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }
  __device__ __forceinline__ void wait(float &a) { wait(); asm volatile("" : "+v"(a)); }
  __device__ __forceinline__ float4 chunk(int j) const { return make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]); }
};

template <bool FAST = true>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_som_update_gauss_h(CbView cb, const float *__restrict__ xrun, int64_t count,
                                                             const int2 *__restrict__ bxy,
                                                             const StepScalars *__restrict__ sc,
                                                             const uint32_t *__restrict__ cnt,
                                                             const MemberEntry *__restrict__ ent,
                                                             const uint32_t *__restrict__ order) {
#ifndef K4H_TB
#define K4H_TB 64            // entries per tile = per barrier (32: 8.19 ms per 4096 vectors at configs[3], 64: 8.00)
#endif
  constexpr int TB = K4H_TB, QW = 8;
  __shared__ float s_ga[2 * TB][WAVE];
  const int nw = static_cast<int>(blockDim.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t nslices = static_cast<uint32_t>(cb.d4 / (QW * nw));          // host: d4 % (QW * nw) == 0
  const uint32_t total = static_cast<uint32_t>(cb.ngroups) * nslices;
  const uint32_t blk = blockIdx.x / 256u, pos = blockIdx.x % 256u;
  const uint32_t bsize = total - blk * 256u < 256u ? total - blk * 256u : 256u;
  const uint32_t item = blk * 256u + ((blk & 1u) ? bsize - 1u - pos : pos);
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const uint32_t n_ent = cnt[g];
  if (n_ent == 0) return;                              // the whole workgroup
  const int q0 = static_cast<int>(item % nslices) * QW * nw + wave * QW;
  int tx, ty;
  txty_of_row(cb, g * WAVE + lane, tx, ty);
  const bool live = g * WAVE + lane < cb.n;

  float4 c[QW];
#pragma unroll
  for (int j = 0; j < QW; j++) c[j] = *tile_ptr(cb, g, q0 + j, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), once

  const MemberEntry *list = ent + g * list_stride(count);
  const float *xbase = k4s_uniform(xrun + 4 * q0);
  const uint32_t rowbytes = static_cast<uint32_t>(cb.d) * 4u;
  const uint32_t ntiles = (n_ent + TB - 1u) / TB;
  const uint32_t lds_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float *)&s_ga[0][lane]));

  // lane l (< TB): the sample of entry t * TB + l
  auto tile_samples = [&](uint32_t t) -> uint32_t {
    const uint32_t k = t * TB + static_cast<uint32_t>(lane);
    return (lane < TB && k < n_ent) ? list[k].sample : 0u;
  };
  // rates of tile t into its third of s_ga (smp: tile_samples(t))
  auto tile_rates = [&](uint32_t t, uint32_t smp) {
    if (t >= ntiles) return;
    const uint32_t tb = n_ent - t * TB < TB ? n_ent - t * TB : TB;
    float alpha_l = 0.f, radius_l = 1.f;
    int wx_l = 0, wy_l = 0;
    double den_l = 1.0, rcp_l = 1.0;
    if (static_cast<uint32_t>(lane) < tb) {
      const StepScalars s = sc[smp];
      const int2 w = bxy[smp];
      alpha_l = s.alpha; radius_l = s.thresh; wx_l = w.x; wy_l = w.y;
      gauss_rate_den(radius_l, &den_l, &rcp_l);
    }
    float *dst = &s_ga[(t & 1u) * TB][lane];
    for (uint32_t i = wave; i < tb; i += nw) {
      const int il = static_cast<int>(i);
      const float alpha = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha_l), il));
      const int wx = __builtin_amdgcn_readlane(wx_l, il), wy = __builtin_amdgcn_readlane(wy_l, il);
      const double den = k4g_readlane(den_l, i), rcp = k4g_readlane(rcp_l, i);
      const float lat = lattice_sq(cb.topol, wx, wy, tx, ty);
      float h, a;
      if (FAST && gauss_rate_fast(lat, den, rcp, &h)) a = alpha * h;
      else a = gaussian_alpha_call(lat, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(radius_l), il)), alpha);
      dst[i * WAVE] = a;
    }
  };
  auto xptr = [&](uint32_t smp_v, uint32_t l) -> const float * {   // row of the sample lane l of smp_v holds
    const uint32_t idx = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(smp_v), static_cast<int>(l)));
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(xbase) + idx * rowbytes);
  };
  auto rate_read = [&](float &a, uint32_t addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(addr) : "memory"); };
  auto apply = [&](const K4hX &x, float a, int half) {
    const f32x2 a2 = {a, a};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float4 xv = x.chunk(j);
      float4 &cc = c[4 * half + j];
      f32x2 lo = {cc.x, cc.y}, hi = {cc.z, cc.w};
      const f32x2 tl = k4s_pk_sub(f32x2{xv.x, xv.y}, lo), th = k4s_pk_sub(f32x2{xv.z, xv.w}, hi);
      const f32x2 sl = a2 * tl, sh = a2 * th;
      lo = lo + sl; hi = hi + sh;
      cc = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
  };

  K4hX x0, x1;                                            // dims [0, 16) / [16, 32) of the wave's slice
  float aA = 0.f, aB = 0.f;
  uint32_t smp_cur = 0u;
  // step u: (barrier) the rates of tile u are made, tile u - 1 is applied.  Tile u's rates go where tile u - 2's were:
  // those were read during step u - 1, before the barrier; tile u - 1's were written during step u - 1, before it.
  for (uint32_t u = 0; u <= ntiles; u++) {
    if (u > 0u) __syncthreads();
    const uint32_t smp_new = tile_samples(u);
    tile_rates(u, smp_new);
    if (u > 0u) {
      const uint32_t t = u - 1u;
      const uint32_t tb = n_ent - t * TB < TB ? n_ent - t * TB : TB;
      const uint32_t base = lds_lane + (t & 1u) * (TB * WAVE * 4u);
      if (tb == TB) {
        const float *p = xptr(smp_cur, 0u);
        x0.load(p);
        rate_read(aA, base);
        // an entry in two halves: half 0 waits for x0 and the rate and sends for half 1; half 1 waits for x1 and sends for
        // the next entry's half 0 and its rate (the tile's last entry sends for nothing: no request is in flight across the
        // barrier and the rate computation).  sched_barrier: the vector work of a half stays between the request it
        // follows and the wait it precedes -- the scheduler otherwise sinks it below the next wait, which then stands
        // right behind its own request
#define K4H_HALF0(AC)                                                  \
        x0.wait(AC);                                                   \
        x1.load_40(p);                                                 \
        __builtin_amdgcn_sched_barrier(0);                             \
        apply(x0, AC, 0);                                              \
        __builtin_amdgcn_sched_barrier(0);                             \
        x1.wait();
#define K4H_ENTRY(AC, AN, E)                                           \
        K4H_HALF0(AC)                                                  \
        p = xptr(smp_cur, (E) + 1u);                                   \
        x0.load(p);                                                    \
        rate_read(AN, base + ((E) + 1u) * (WAVE * 4u));                \
        __builtin_amdgcn_sched_barrier(0);                             \
        apply(x1, AC, 1);                                              \
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t e = 0; e < TB - 2u; e += 2u) {
          K4H_ENTRY(aA, aB, e)
          K4H_ENTRY(aB, aA, e + 1u)
        }
        K4H_ENTRY(aA, aB, TB - 2u)
        K4H_HALF0(aB)
        apply(x1, aB, 1);
#undef K4H_ENTRY
#undef K4H_HALF0
      } else {
        for (uint32_t e = 0; e < tb; e++) {               // the list's last, partial tile: entry by entry, no look-ahead
          const float *p = xptr(smp_cur, e);
          x0.load(p);
          x1.load_40(p);
          rate_read(aA, base + e * (WAVE * 4u));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x0.v), "+s"(x1.v));
          asm volatile("" : "+v"(aA));
          apply(x0, aA, 0);
          apply(x1, aA, 1);
        }
      }
    }
    smp_cur = smp_new;
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < QW; j++) *tile_ptr_w(cb, g, q0 + j, lane) = c[j];
  }
}


}  // namespace somhip
