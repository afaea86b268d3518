// kernels/rerank.hpp -- K2r/K2k/K2m/K2s/K2p: exact re-rank behind the pre-filter (top-1 and top-k)
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "som_online.hpp"

namespace somhip {

// =====================================================================================
// K2r: exact re-rank.  One wave per sample: global minimum of the group minima, then
// for every group within tau of it, the masked rows' distances with the reference's
// arithmetic (lane = row, dims in order, sub/mul/add), exact (distance, index) minimum.
// stats[0] += groups re-ranked, stats[1] += rows re-ranked, stats[2] = max groups/sample.
// =====================================================================================
__global__ __launch_bounds__(256) void k_rerank(CbView cb, const float *__restrict__ rows,
                                                int64_t n_rows, int64_t first, int64_t count,
                                                int64_t bpad, const float *__restrict__ wmin,
                                                const uint64_t *__restrict__ wmask,
                                                const float *__restrict__ tau,
                                                const uint32_t *__restrict__ pair_count, uint32_t cap,
                                                uint64_t *__restrict__ keys,
                                                unsigned long long *__restrict__ stats) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= count) return;
  if (pair_count && *pair_count <= cap) return;          // the pair path handled the run
  float m = 3.4e38f;
  for (int64_t g = lane; g < cb.ngroups; g += WAVE) m = fminf(m, wmin[g * bpad + b]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, WAVE));
  const float thr = m + tau[b];
  const float *x = rows + ((first + b) % n_rows) * cb.d;
  const bool vec = (cb.d & 3) == 0;
  uint64_t best = KEY_NONE;
  unsigned ngroups_done = 0, nrows_done = 0;
  for (int64_t gb = 0; gb < cb.ngroups; gb += WAVE) {
    const int64_t gl = gb + lane;
    const float wv = gl < cb.ngroups ? wmin[gl * bpad + b] : 3.4e38f;
    const bool q = wv <= thr && wv < 3.0e38f;              // (3.4e38: left out by k_l2_select under a shard-exchange bound)
    uint64_t ball = __ballot(q);
    while (ball) {
      const int t = __builtin_ctzll(ball);
      ball &= ball - 1;
      const int64_t g = gb + t;
      const uint64_t mask = wmask[g * bpad + b];
      // lane -> code row of the group: the mask bit of row rr is
      //   half = (rr>>2)&1, i = rr>>5, r = (rr&3) + 4*((rr&31)>>3)  -> bit 32*half + 16*i + r
      const int rr = lane;
      const int hbit = (rr >> 2) & 1, ib = rr >> 5, rb = (rr & 3) + 4 * ((rr & 31) >> 3);
      const bool mine = (mask >> (32 * hbit + 16 * ib + rb)) & 1ull;
      const int64_t row = g * WAVE + lane;
      // same arithmetic and order as k_scan_exact / k_som_online_step, loads pipelined
      const float acc = vec ? online_stream<false, true, false, true, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr)
                            : online_stream<false, true, false, false, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr);
      const bool ok = mine && row < cb.n;
      const uint64_t k = ok ? make_key(acc, unit_of_row(cb, row)) : KEY_NONE;
      best = k < best ? k : best;
      ngroups_done++;
      nrows_done += __popcll(mask);
    }
  }
  best = wave_min_u64(best);
  if (lane == 0) {
    atomicMin(reinterpret_cast<unsigned long long *>(keys + b), static_cast<unsigned long long>(best));
    atomicAdd(stats + 0, static_cast<unsigned long long>(ngroups_done));
    atomicAdd(stats + 1, static_cast<unsigned long long>(nrows_done));
    atomicMax(stats + 2, static_cast<unsigned long long>(ngroups_done));
  }
}

// =====================================================================================
// K2k: exact k nearest rows per sample behind the same pre-filter (the frozen candidate lists of
// the batched LVQ engine, K6).  With m_K = the K-th smallest group minimum of s~ and delta = tau/2
// the bound on |s~ + ||x||^2 - d|:  the K rows that realise the K smallest group minima have exact
// distances <= m_K + ||x||^2 + delta, so the exact K-th distance is at most that, and every row of
// the exact top K has s~ <= m_K + 2 delta -- its group's minimum is <= m_K + tau.  One wave per
// sample: each lane keeps the K smallest minima of its strided groups, K extraction rounds give
// m_K; then ALL 64 rows of every group with minimum <= m_K + tau get the reference's arithmetic
// (lane = row, dims in order) and are merged into the running K best keys (tag = row, or ~row for
// the k-NN tie order).
// =====================================================================================
template <int K>
__global__ __launch_bounds__(256) void k_rerank_topk(CbView cb, const float *__restrict__ rows,
                                                     int64_t n_rows, int64_t first, int64_t count,
                                                     int64_t bpad, const float *__restrict__ wmin,
                                                     const float *__restrict__ tau, int tie_knn,
                                                     uint64_t *__restrict__ keys_out,
                                                     const uint32_t *__restrict__ only_if = nullptr) {
  if (only_if && *only_if == 0u) return;                 // launched behind the pair-list kernels: runs only if their list overflowed
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= count) return;
  // ---- m_K: K-th smallest group minimum
  float mine[K];
#pragma unroll
  for (int t = 0; t < K; t++) mine[t] = 3.4e38f;
  for (int64_t g = lane; g < cb.ngroups; g += WAVE) {
    float v = wmin[g * bpad + b];
#pragma unroll
    for (int t = 0; t < K; t++) {                        // sorted insertion
      const float lo = fminf(mine[t], v);
      v = fmaxf(mine[t], v);
      mine[t] = lo;
    }
  }
  float mk = 3.4e38f;
  for (int t = 0; t < K; t++) {                          // K rounds: smallest head, its lane pops
    float h = mine[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) h = fminf(h, __shfl_xor(h, off, WAVE));
    mk = h;
    const unsigned long long who = __ballot(mine[0] == h);
    if (lane == __builtin_ctzll(who)) {
#pragma unroll
      for (int u = 0; u + 1 < K; u++) mine[u] = mine[u + 1];
      mine[K - 1] = 3.4e38f;
    }
  }
  const float thr = (mk >= 3.0e38f) ? 3.4e38f : mk + tau[b];      // fewer than K groups: take them all
  // ---- exact distances of every row of the surviving groups, running K best
  const float *x = rows + ((first + b) % n_rows) * cb.d;
  const bool vec = (cb.d & 3) == 0;
  uint64_t top[K];
#pragma unroll
  for (int t = 0; t < K; t++) top[t] = KEY_NONE;
  for (int64_t gb = 0; gb < cb.ngroups; gb += WAVE) {
    const int64_t gl = gb + lane;
    const bool q = gl < cb.ngroups && wmin[gl * bpad + b] <= thr;
    uint64_t ball = __ballot(q);
    while (ball) {
      const int t = __builtin_ctzll(ball);
      ball &= ball - 1;
      const int64_t g = gb + t;
      const int64_t row = g * WAVE + lane;
      const float acc = vec ? online_stream<false, true, false, true, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr)
                            : online_stream<false, true, false, false, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr);
      const uint32_t grow = unit_of_row(cb, row);
      uint64_t k = row < cb.n ? make_key(acc, tie_knn ? ~grow : grow) : KEY_NONE;
      for (int it = 0; it < K; it++) {                   // at most K rows of one group can enter
        const uint64_t best = wave_min_u64_dpp(k);
        if (best >= top[K - 1]) break;                   // wave-uniform
        uint64_t v = best;
#pragma unroll
        for (int u = 0; u < K; u++) {                    // sorted insertion (keys are unique)
          const uint64_t lo = top[u] < v ? top[u] : v;
          v = top[u] < v ? v : top[u];
          top[u] = lo;
        }
        if (k == best) k = KEY_NONE;
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int t = 0; t < K; t++) keys_out[b * K + t] = top[t];
  }
}

// K2k in three launches for big codebooks, where one wave per sample (above) would stream megabytes
// alone: (1) k_topk_select -- per sample m_K, then the (sample, group) pairs of the surviving groups
// as one contiguous block of a shared list; (2) k_topk_pairs -- one wave per pair: exact distances of
// the group's 64 rows, its K smallest keys; (3) k_topk_merge -- per sample the K smallest of its
// pairs' keys.  A full list (*overflow != 0) sends the run to the one-wave kernel instead.
struct TopkSpan { uint32_t start, n; };

// (round 3: workgroup = TOPK_NB samples x TOPK_NS slices of the groups, as k_group_kth -- one wave per sample with
// lane = group read wmin across its rows, a cache line per lane, three times: 51 us at configs[4].)
constexpr int TOPK_NB = 16, TOPK_NS = 1024 / TOPK_NB;
template <int K>
__global__ __launch_bounds__(1024) void k_topk_select(CbView cb, int64_t count, int64_t bpad,
                                                      const float *__restrict__ wmin,
                                                      const float *__restrict__ tau, uint32_t cap,
                                                      uint2 *__restrict__ pairs, TopkSpan *__restrict__ span,
                                                      uint32_t *__restrict__ counter /* [0] fill, [1] overflow */,
                                                      uint32_t *__restrict__ gcnt = nullptr, uint2 *__restrict__ glist = nullptr,
                                                      uint32_t cap_g = 0) {
  constexpr int NB = TOPK_NB, NS = TOPK_NS;
  __shared__ float s_k[NS][K][NB];
  __shared__ uint32_t s_n[NS][NB];
  __shared__ uint32_t s_start[NB], s_total[NB];
  __shared__ float s_thr[NB];
  const int tid = threadIdx.x, bx = tid % NB, gy = tid / NB;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * NB + bx;
  const bool valid = b < count;
  float mine[K];
#pragma unroll
  for (int t = 0; t < K; t++) mine[t] = 3.4e38f;
  auto insert = [&](float v) {                             // sorted insertion
#pragma unroll
    for (int t = 0; t < K; t++) {
      const float lo = fminf(mine[t], v);
      v = fmaxf(mine[t], v);
      mine[t] = lo;
    }
  };
  if (valid) {
    int64_t g = gy;
    for (; g + 3 * NS < cb.ngroups; g += 4 * NS) {
      const float v0 = wmin[g * bpad + b], v1 = wmin[(g + NS) * bpad + b], v2 = wmin[(g + 2 * NS) * bpad + b], v3 = wmin[(g + 3 * NS) * bpad + b];
      insert(v0); insert(v1); insert(v2); insert(v3);
    }
    for (; g < cb.ngroups; g += NS) insert(wmin[g * bpad + b]);
  }
#pragma unroll
  for (int t = 0; t < K; t++) s_k[gy][t][bx] = mine[t];
  __syncthreads();
  if (gy < 8) {
    for (int k = gy + 8; k < NS; k += 8)
#pragma unroll
      for (int u = 0; u < K; u++) insert(s_k[k][u][bx]);
#pragma unroll
    for (int t = 0; t < K; t++) s_k[gy][t][bx] = mine[t];      // (slices 0-7 are read by nobody but slice 0, below)
  }
  __syncthreads();
  if (gy == 0) {
    for (int k = 1; k < 8; k++)
#pragma unroll
      for (int u = 0; u < K; u++) insert(s_k[k][u][bx]);
    const float mk = mine[K - 1];
    s_thr[bx] = (mk >= 3.0e38f || !valid) ? 3.4e38f : mk + tau[b];
  }
  __syncthreads();
  const float thr = s_thr[bx];
  uint32_t n = 0;
  if (valid)
    for (int64_t g = gy; g < cb.ngroups; g += NS) n += wmin[g * bpad + b] <= thr ? 1u : 0u;
  s_n[gy][bx] = n;
  __syncthreads();
  if (gy == 0 && valid) {
    uint32_t total = 0;
    for (int k = 0; k < NS; k++) total += s_n[k][bx];
    const uint32_t start = atomicAdd(counter, total);
    if (start + total > cap) atomicMax(counter + 1, 1u);
    span[b].start = start; span[b].n = total;
    s_start[bx] = start; s_total[bx] = total;
  }
  __syncthreads();
  if (!valid || s_start[bx] + s_total[bx] > cap) return;
  uint32_t at = s_start[bx];
  for (int k = 0; k < gy; k++) at += s_n[k][bx];
  for (int64_t g = gy; g < cb.ngroups; g += NS) {
    if (wmin[g * bpad + b] <= thr) {
      const uint32_t p = at++;
      pairs[p] = make_uint2(static_cast<uint32_t>(b), static_cast<uint32_t>(g));
      if (gcnt) {                                          // the same pair filed under its group (k_topk_pairs_bygroup)
        const uint32_t slot = atomicAdd(&gcnt[g], 1u);
        if (slot < cap_g) glist[static_cast<size_t>(g) * cap_g + slot] = make_uint2(static_cast<uint32_t>(b), p);
        else atomicMax(counter + 1, 1u);
      }
    }
  }
}

// (2) by row group: the samples filed under a group are taken four at a time -- the group's tile (64 rows x d, 256 KiB
// at d = 1024) is streamed once per four samples instead of once per pair.  One WAVE per workgroup and no LDS (round 3):
// as waves of one 256-thread workgroup with the sample rows in LDS, the passes held 64 KiB each -- two workgroups per
// CU, three rounds of them at configs[4], where most groups have work for one wave only (270 us per batch of 1024).
// The distance of every (row, sample) is the reference's sum in the reference's order, as in k_topk_pairs; the results go
// to the same per-pair slots, so (3) does not care which of the two ran.
// The passes as a list: (group, first sample of the pass) for every S samples filed under a group.  (A grid of
// groups x passes is mostly empty workgroups -- at configs[4] 1600 of 100 000 had work, and starting the others was
// what the kernel's 175 us were.)
template <int S>
__global__ __launch_bounds__(256) void k_topk_worklist(int64_t ngroups, const uint32_t *__restrict__ gcnt, uint32_t cap_g,
                                                       const uint32_t *__restrict__ counter, uint32_t *__restrict__ wcount,
                                                       uint2 *__restrict__ work, uint32_t wcap) {
  if (counter[1]) return;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ngroups; g += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const uint32_t n = gcnt[g] < cap_g ? gcnt[g] : cap_g;
    const uint32_t np = (n + S - 1u) / S;
    if (np) {
      const uint32_t b = atomicAdd(wcount, np);
      for (uint32_t i = 0; i < np && b + i < wcap; i++) work[b + i] = make_uint2(static_cast<uint32_t>(g), i * S);
    }
  }
}

template <int K, int S = 4>       // S samples per pass over the group's tile
__global__ __launch_bounds__(64) void k_topk_pairs_bygroup(CbView cb, const float *__restrict__ rows, int64_t n_rows,
                                                            int64_t first, int tie_knn, const uint32_t *__restrict__ gcnt,
                                                            const uint2 *__restrict__ glist, uint32_t cap_g,
                                                            const uint32_t *__restrict__ counter,
                                                            const uint32_t *__restrict__ wcount, const uint2 *__restrict__ work, uint32_t wcap,
                                                            uint64_t *__restrict__ partial /* [pair][K] */) {
  if (counter[1]) return;
  const uint32_t nwork = *wcount < wcap ? *wcount : wcap;
  const int lane = threadIdx.x & 63;
  const int64_t first0 = first % n_rows;
#ifndef TOPK_U
#define TOPK_U 4            // (configs[4] shape, k_rerank per batch: 2: 210 us, 4: 178, 8: 188, 16: 379)
#endif
  constexpr int U = TOPK_U;                              // tile chunks per register buffer (two buffers, as in K3's row stream)
  for (uint32_t w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int64_t g = work[w].x;
    const uint32_t base = work[w].y;
    const uint32_t n = gcnt[g] < cap_g ? gcnt[g] : cap_g;
    const int64_t row = g * WAVE + lane;
    const uint32_t grow = unit_of_row(cb, row);
    const uint32_t m = n - base < static_cast<uint32_t>(S) ? n - base : static_cast<uint32_t>(S);
    uint32_t slot[S];
    // the pass's sample rows stay in REGISTERS, lane l of xs[s][j] = chunk 64 j + l of sample s (host: d4 <= 256), and a
    // chunk's four floats reach the arithmetic as scalar operands by v_readlane: through LDS (the first form) every chunk
    // cost the lone wave four exposed ds_read latencies
    float4 xs[S][4];
#pragma unroll
    for (int s = 0; s < S; s++) {
      const uint2 en = glist[static_cast<size_t>(g) * cap_g + base + (static_cast<uint32_t>(s) < m ? s : 0)];
      slot[s] = en.y;
      int64_t r = first0 + en.x;
      if (r >= n_rows) r %= n_rows;                        // (the run wraps inside the data set: rare)
      const float4 *x = reinterpret_cast<const float4 *>(rows + r * cb.d);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = lane + j * WAVE;
        xs[s][j] = x[q < cb.d4 ? q : cb.d4 - 1];
      }
    }
    float acc[S];
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = 0.0f;
    auto bcast = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    auto chunk = [&](auto jc, int q, const float4 c) {     // jc: the chunk's block of 64 as a compile-time constant
      constexpr int j = decltype(jc)::value;
      const int l = q - 64 * j;
      // a lone wave per SIMD: written so that the 4 S squares of a chunk are independent of one another and the S sums advance
      // side by side (sample after sample, every instruction waited for the one before it: 900 cycles per chunk);
      // each sum still takes its dims in order, one rounding per operation (sq_acc's arithmetic)
      float sq[S][4];
#pragma unroll
      for (int s = 0; s < S; s++) {
        const float d0 = c.x - bcast(xs[s][j].x, l), d1 = c.y - bcast(xs[s][j].y, l);
        const float d2 = c.z - bcast(xs[s][j].z, l), d3 = c.w - bcast(xs[s][j].w, l);
        sq[s][0] = d0 * d0; sq[s][1] = d1 * d1; sq[s][2] = d2 * d2; sq[s][3] = d3 * d3;
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int s = 0; s < S; s++) acc[s] = acc[s] + sq[s][e];
    };
    float4 bufA[U], bufB[U];
    const int nfull = (cb.d4 / (2 * U)) * (2 * U), last = cb.d4 - 1;
    if (nfull > 0) {
#pragma unroll
      for (int u = 0; u < U; u++) bufA[u] = *tile_ptr(cb, g, u, lane);
    }
    auto block = [&](auto jc) {                            // chunks [64 j, 64 j + 64) of the rows
      constexpr int j = decltype(jc)::value;
      const int q_hi = cb.d4 < 64 * j + 64 ? cb.d4 : 64 * j + 64;
      const int full_hi = nfull < q_hi ? nfull : q_hi;     // (2 U divides 64: a double step never straddles two blocks)
      for (int qb = 64 * j; qb < full_hi; qb += 2 * U) {
#pragma unroll
        for (int u = 0; u < U; u++) bufB[u] = *tile_ptr(cb, g, qb + U + u, lane);
#pragma unroll
        for (int u = 0; u < U; u++) chunk(jc, qb + u, bufA[u]);
#pragma unroll
        for (int u = 0; u < U; u++) { const int q = qb + 2 * U + u; bufA[u] = *tile_ptr(cb, g, q < last ? q : last, lane); }
#pragma unroll
        for (int u = 0; u < U; u++) chunk(jc, qb + U + u, bufB[u]);
      }
      for (int q = full_hi > 64 * j ? full_hi : 64 * j; q < q_hi; q++) chunk(jc, q, *tile_ptr(cb, g, q, lane));
    };
    block(std::integral_constant<int, 0>());
    if (cb.d4 > 64) block(std::integral_constant<int, 1>());
    if (cb.d4 > 128) block(std::integral_constant<int, 2>());
    if (cb.d4 > 192) block(std::integral_constant<int, 3>());
#pragma unroll
    for (int s = 0; s < S; s++) {
      if (static_cast<uint32_t>(s) >= m) break;
      uint64_t k = row < cb.n ? make_key(acc[s], tie_knn ? ~grow : grow) : KEY_NONE;
#pragma unroll
      for (int t = 0; t < K; t++) {
        const uint64_t best = wave_min_u64_dpp(k);
        if (lane == 0) partial[static_cast<size_t>(slot[s]) * K + t] = best;
        if (k == best) k = KEY_NONE;
      }
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void k_topk_pairs(CbView cb, const float *__restrict__ rows, int64_t n_rows,
                                                    int64_t first, int tie_knn, const uint2 *__restrict__ pairs,
                                                    const uint32_t *__restrict__ counter,
                                                    uint64_t *__restrict__ partial /* [pair][K] */) {
  if (counter[1]) return;
  const uint32_t np = counter[0];
  const int lane = threadIdx.x & 63;
  const bool vec = (cb.d & 3) == 0;
  const uint32_t nw = gridDim.x * 4;
  for (uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < np; p += nw) {
    const uint2 pr = pairs[p];
    const float *x = rows + ((first + pr.x) % n_rows) * cb.d;
    const int64_t g = pr.y, row = g * WAVE + lane;
    const float acc = vec ? online_stream<false, true, false, true, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr)
                          : online_stream<false, true, false, false, 8>(cb, g, lane, false, 0.f, x, x, nullptr, nullptr);
    const uint32_t grow = unit_of_row(cb, row);
    uint64_t k = row < cb.n ? make_key(acc, tie_knn ? ~grow : grow) : KEY_NONE;
#pragma unroll
    for (int t = 0; t < K; t++) {
      const uint64_t best = wave_min_u64_dpp(k);
      if (lane == 0) partial[static_cast<size_t>(p) * K + t] = best;
      if (k == best) k = KEY_NONE;
    }
  }
}

template <int K>
__global__ __launch_bounds__(256) void k_topk_merge(int64_t count, const TopkSpan *__restrict__ span,
                                                    const uint64_t *__restrict__ partial,
                                                    const uint32_t *__restrict__ counter,
                                                    uint64_t *__restrict__ keys_out,
                                                    unsigned long long *__restrict__ pairs_total = nullptr) {
  if (pairs_total && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(pairs_total, static_cast<unsigned long long>(counter[0]));
  if (counter[1]) return;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= count) return;
  const TopkSpan sp = span[b];
  const uint64_t *p = partial + static_cast<size_t>(sp.start) * K;
  const int total = static_cast<int>(sp.n) * K;
  uint64_t prev = 0;
  bool firstround = true;
  for (int t = 0; t < K; t++) {
    uint64_t mine = KEY_NONE;
    for (int j = lane; j < total; j += WAVE) {
      const uint64_t v = p[j];
      if ((firstround || v > prev) && v < mine) mine = v;   // keys are unique
    }
    const uint64_t best = wave_min_u64_dpp(mine);
    if (lane == 0) keys_out[b * K + t] = best;
    prev = best;
    firstround = false;
    if (best == KEY_NONE) { for (int u = t + 1; u < K; u++) if (lane == 0) keys_out[b * K + u] = KEY_NONE; break; }
  }
}

// =====================================================================================
// K2s / K2p: the usual case of the re-rank, row-granular.  K2s (one wave per sample) finds
// the groups within tau of the global minimum and appends the masked rows as
// (sample, row) pairs to one list; K2p (one lane per pair) recomputes each pair's distance
// with the reference's arithmetic (dims in order, sub/mul/add) and folds the key into
// keys[sample] with a 64-bit atomic min.  If the list overflows (pathological codebooks: huge
// numbers of near-ties) K2p does nothing and K2r above re-ranks the whole run group by group.
// =====================================================================================
// order-preserving float <-> uint32 (for atomicMin on values of either sign)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

// K2x: the winner search of a row-sharded codebook with the pre-filter's bounds exchanged between the shards
// (somhip_shard_winner_begin / _refine / _finish).  A shard on its own keeps every group within a window of ITS smallest
// level-1 value, so N shards together keep N times what one codebook would; with the smallest upper bound of all shards
// instead, the shards together keep what the whole codebook would keep.
//   ub[b] = (this shard's smallest pre-filter value) + delta: the exact nearest row of the WHOLE codebook has s <= ub for
//   every shard's ub (s = distance - ||x||^2, the same on every shard), hence <= the MIN over the shards; in its own shard
//   its pre-filter value is <= that MIN + delta(shard).  The deltas are each shard's own (its largest row norm).
__global__ void k_shard_bound(int64_t count, const uint32_t *__restrict__ gmin, const float *__restrict__ delta,
                              float *__restrict__ ub) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= count) return;
  const uint32_t o = gmin[b];
  float v = 3.4e38f;
  if (o != 0xFFFFFFFFu) {
    const float m = __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);      // (= ordered_to_float)
    if (m < 3.0e38f) {
      v = m + delta[b];
      v += fabsf(v) * 1.2e-7f;                           // the sum rounded up
    }
  }
  ub[b] = v;
}
// K2m: gmin[b] = min over the shard's row groups of the group minima (ordered-uint encoding; the
// host presets 0xFFFFFFFF).  Workgroup = 32 consecutive samples (one 128-byte line of wmin) x one
// chunk of groups, 8 interleaved group phases; one atomicMin per (sample, chunk).
__global__ __launch_bounds__(256) void k_group_min(int64_t ngroups, int64_t bpad, int64_t chunk,
                                                   const float *__restrict__ wmin, uint32_t *__restrict__ gmin) {
  __shared__ float s_min[8][32];
  const int tid = threadIdx.x, bx = tid & 31, gy = tid >> 5;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 32 + bx;
  const int64_t g_lo = static_cast<int64_t>(blockIdx.y) * chunk;
  const int64_t g_hi = g_lo + chunk < ngroups ? g_lo + chunk : ngroups;
  float m = 3.4e38f;
  if (b < bpad)
    for (int64_t g = g_lo + gy; g < g_hi; g += 8) m = fminf(m, wmin[g * bpad + b]);
  s_min[gy][bx] = m;
  __syncthreads();
  if (gy == 0 && b < bpad) {
#pragma unroll
    for (int k = 1; k < 8; k++) m = fminf(m, s_min[k][bx]);
    atomicMin(gmin + b, float_to_ordered(m));
  }
}

// The K-th smallest group minimum per sample (same output convention as k_group_min: the ordered image of the
// float): the two-level pre-filter of a top-K search measures its level-1 window from this value instead of the
// minimum.  Workgroup = 32 samples x 32 slices of the groups (coalesced over samples); each thread keeps the K
// smallest of its slice; the lists are merged in two steps.  Fewer than K groups: 3.4e38 (all pass).
template <int K>
__global__ __launch_bounds__(1024) void k_group_kth(int64_t ngroups, int64_t bpad, const float *__restrict__ wmin,
                                                    uint32_t *__restrict__ gkth) {
  // 32 samples x 32 slices of the groups (round 3; 8 slices of 256 threads: 195 dependent trips per thread at
  // configs[4], 74 us on 32 workgroups); the slices' lists are merged 8 -> 1 by four threads per sample, then 4 -> 1
  __shared__ float s_k[32][K][32];
  const int tid = threadIdx.x, bx = tid & 31, gy = tid >> 5;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 32 + bx;
  float mine[K];
#pragma unroll
  for (int t = 0; t < K; t++) mine[t] = 3.4e38f;
  auto insert = [&](float v) {                             // sorted insertion
#pragma unroll
    for (int t = 0; t < K; t++) {
      const float lo = fminf(mine[t], v);
      v = fmaxf(mine[t], v);
      mine[t] = lo;
    }
  };
  if (b < bpad) {
    int64_t g = gy;
    for (; g + 96 < ngroups; g += 128) {                   // four loads in flight
      const float v0 = wmin[g * bpad + b], v1 = wmin[(g + 32) * bpad + b], v2 = wmin[(g + 64) * bpad + b], v3 = wmin[(g + 96) * bpad + b];
      insert(v0); insert(v1); insert(v2); insert(v3);
    }
    for (; g < ngroups; g += 32) insert(wmin[g * bpad + b]);
  }
#pragma unroll
  for (int t = 0; t < K; t++) s_k[gy][t][bx] = mine[t];
  __syncthreads();
  if (gy < 4) {
    for (int k = gy + 4; k < 32; k += 4)
#pragma unroll
      for (int u = 0; u < K; u++) insert(s_k[k][u][bx]);
  }
  __syncthreads();                                         // (everyone has read what it merges before the four write)
  if (gy < 4) {
#pragma unroll
    for (int t = 0; t < K; t++) s_k[gy][t][bx] = mine[t];
  }
  __syncthreads();
  if (gy == 0 && b < bpad) {
    for (int k = 1; k < 4; k++)
#pragma unroll
      for (int u = 0; u < K; u++) insert(s_k[k][u][bx]);
    gkth[b] = float_to_ordered(mine[K - 1]);
  }
}

// K2s: rows of every group within tau of the sample's global minimum -> (sample, row) pairs.
// Same workgroup shape as K2m (32 samples x a chunk of groups), so the whole wmin matrix is
// read by thousands of workgroups at once instead of 128 long-running ones.
__global__ __launch_bounds__(256) void k_rerank_select(CbView cb, int64_t count, int64_t bpad, int64_t chunk,
                                                       const float *__restrict__ wmin,
                                                       const uint64_t *__restrict__ wmask,
                                                       const float *__restrict__ tau,
                                                       const uint32_t *__restrict__ gmin,
                                                       uint32_t *__restrict__ gcount,
                                                       uint32_t cap, uint32_t cap_col,
                                                       uint2 *__restrict__ pairs,
                                                       uint32_t *__restrict__ col_count,
                                                       uint32_t *__restrict__ pair_count,
                                                       unsigned long long *__restrict__ stats,
                                                       const float *__restrict__ xub = nullptr) {
  // xub (shard exchange): the bound agreed between the shards takes the place of this shard's own minimum
  // The pair list is cut into one segment of cap_col entries per 32-sample column (blockIdx.x), each
  // with its own counter: a single list counter took ~6 500 same-address atomics per launch and
  // that serialisation, not the 16 MiB of wmin, was this kernel's time.  A full segment raises
  // *pair_count above cap, which sends the whole run to the group-granular K2r.
  __shared__ uint32_t s_cnt[8][32];
  const int tid = threadIdx.x, bx = tid & 31, gy = tid >> 5, lane = tid & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 32 + bx;
  const int64_t g_lo = static_cast<int64_t>(blockIdx.y) * chunk;
  const int64_t g_hi = g_lo + chunk < cb.ngroups ? g_lo + chunk : cb.ngroups;
  const bool live = b < count;
  const float thr = live ? (xub ? xub[b] : ordered_to_float(gmin[b])) + tau[b] : -3.4e38f;
  unsigned ngr = 0, nrow = 0;
  uint2 *seg = pairs + static_cast<size_t>(blockIdx.x) * cap_col;
  // blocks of 64 groups (8 per thread): all the loads of a block are issued together and the wave reserves
  // its room in the segment once per block -- the atomic's round trip, taken once per 8 groups, was most
  // of this kernel's time
  for (int64_t g0 = g_lo; g0 < g_hi; g0 += 64) {
    float wv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int64_t g = g0 + 8 * k + gy;
      wv[k] = (live && g < g_hi) ? wmin[g * bpad + b] : 3.4e38f;
    }
    unsigned long long mk[8];
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int64_t g = g0 + 8 * k + gy;
      unsigned long long mask = 0;
      if (wv[k] <= thr && live && g < g_hi) {
        mask = wmask[g * bpad + b];
        // drop padding rows of the last group: bit 32h+16i+r is row 32i + (r&3) + 8(r>>2) + 4h
        if ((g + 1) * WAVE > cb.n) {
          unsigned long long keep = 0;
          for (int t = 0; t < 64; t++) {
            const int h = t >> 5, i = (t >> 4) & 1, r = t & 15;
            if (g * WAVE + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h < cb.n) keep |= 1ull << t;
          }
          mask &= keep;
        }
      }
      mk[k] = mask;
      const unsigned c = __popcll(mask);
      n += c;
      ngr += c != 0;
    }
    if (__ballot(n != 0) == 0) continue;                 // wave-uniform
    // wave-aggregated reservation in the pair list
    unsigned pre = n;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const unsigned v = __shfl_up(pre, off, WAVE);
      if (lane >= off) pre += v;
    }
    const unsigned wave_total = __shfl(pre, WAVE - 1, WAVE);
    unsigned base = 0;
    if (lane == 0) {
      base = atomicAdd(col_count + blockIdx.x, wave_total);
      if (base + wave_total > cap_col) atomicMax(pair_count, cap + 1);
    }
    unsigned at = __shfl(base, 0, WAVE) + pre - n;
    nrow += n;
    if (n) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int64_t g = g0 + 8 * k + gy;
        unsigned long long mm = mk[k];
        while (mm) {
          const int t = __builtin_ctzll(mm);
          mm &= mm - 1;
          const int h = t >> 5, i = (t >> 4) & 1, r = t & 15;
          if (at < cap_col)
            seg[at] = make_uint2(static_cast<uint32_t>(b),
                                 static_cast<uint32_t>(g * WAVE + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h));
          at++;
        }
      }
    }
  }
  // statistics: groups re-ranked per sample (summed over the chunks through gcount), totals
  s_cnt[gy][bx] = ngr;
  __syncthreads();
  if (gy == 0 && live) {
    unsigned tot = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) tot += s_cnt[k][bx];
    if (tot) {
      const unsigned before = atomicAdd(gcount + b, tot);
      atomicMax(col_count + gridDim.x * 3 + blockIdx.x, before + tot);      // per-column maximum
    }
  }
  unsigned a0 = ngr, a1 = nrow;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { a0 += __shfl_xor(a0, off, WAVE); a1 += __shfl_xor(a1, off, WAVE); }
  if (lane == 0 && a1) {                                                    // per-column totals; K2p folds them
    atomicAdd(col_count + gridDim.x * 1 + blockIdx.x, a0);
    atomicAdd(col_count + gridDim.x * 2 + blockIdx.x, a1);
  }
  (void)stats;
}

constexpr int PAIR_MAX_COLS = 4096;     // 32-sample columns per run (131 072 samples)

__global__ __launch_bounds__(256) void k_rerank_pairs(CbView cb, const float *__restrict__ rows,
                                                      int64_t n_rows, int64_t first, uint32_t cap,
                                                      uint32_t cap_col, int ncols,
                                                      const uint2 *__restrict__ pairs,
                                                      const uint32_t *__restrict__ col_count,
                                                      const uint32_t *__restrict__ pair_count,
                                                      uint64_t *__restrict__ keys,
                                                      unsigned long long *__restrict__ stats,
                                                      const float *__restrict__ rowmajor = nullptr) {
  // rowmajor (may be null; d % 4 == 0): the row-major copy of the rows k_prep_codes_bf16 made -- the four lanes of a quad
  // then read 64 consecutive bytes of the row per step instead of 16 bytes out of each of four KiB (PMC, round 2: 2.3
  // times the bytes of the rows fetched from the tiles).  Same values, same order of operations.
  // A fixed, small grid (workgroup launches cost ~50 ns each: a grid sized for the worst case was
  // the whole cost of this kernel).  Every workgroup builds the same table of 64-pair chunks per
  // column (prefix sums of the segment fills) and takes chunks round-robin.
  //
  // Four lanes share a pair.  The sum over the dims has to be one sequential chain, but the subtract and
  // the square are element-wise: lane j of the quad loads chunks q = j (mod 4) of the row (16 bytes out of
  // every KiB -- a quarter of the scattered reads per lane, four times the waves) and of the sample,
  // squares its differences, and every lane of the quad then adds the products in dim order, fetching
  // each through a quad-broadcast DPP operand.  Same roundings in the same order as sq_acc.
  __shared__ uint32_t s_pref[PAIR_MAX_COLS + 1];
  __shared__ uint32_t s_scan[4];
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && *pair_count <= cap) {            // the columns' statistics, once
    for (int c = tid; c < ncols; c += 256) {
      const uint32_t g = col_count[ncols * 1 + c], r = col_count[ncols * 2 + c];
      if (r) { atomicAdd(stats + 0, static_cast<unsigned long long>(g)); atomicAdd(stats + 1, static_cast<unsigned long long>(r)); }
      atomicMax(stats + 2, static_cast<unsigned long long>(col_count[ncols * 3 + c]));
    }
  }
  if (*pair_count > cap) return;                         // a segment overflowed: K2r does the whole run
  const int per = (ncols + 255) / 256;
  uint32_t mine = 0;
  for (int c = tid * per; c < (tid + 1) * per && c < ncols; c++) mine += (col_count[c] + 63u) >> 6;
  uint32_t inc = mine;                                   // inclusive scan: in the wave by shuffles, then the 4 wave totals
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, WAVE);
    if ((tid & 63) >= off) inc += v;
  }
  if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
  __syncthreads();
  for (int w = 0; w < (tid >> 6); w++) inc += s_scan[w];
  {
    uint32_t run = inc - mine;                           // exclusive prefix of this thread's columns
    for (int c = tid * per; c < (tid + 1) * per && c < ncols; c++) { s_pref[c] = run; run += (col_count[c] + 63u) >> 6; }
    if (tid == 255) s_pref[ncols] = inc;
  }
  __syncthreads();
  const uint32_t total = s_pref[ncols];
  const bool vec = (cb.d & 3) == 0;
  const int j = tid & 3;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    int lo = 0, hi = ncols;                              // column with s_pref[col] <= id < s_pref[col + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= id) lo = mid; else hi = mid; }
    const uint32_t slot = (id - s_pref[lo]) * 64u + (tid >> 2);
    if (slot >= col_count[lo]) continue;                 // quad-uniform
    const uint2 pr = pairs[static_cast<size_t>(lo) * cap_col + slot];
    const int64_t row = pr.y;
    const int64_t cstep = rowmajor ? 1 : WAVE;           // float4 units between consecutive chunks of a row
    const float4 *crow = rowmajor ? reinterpret_cast<const float4 *>(rowmajor + row * cb.d)
                                  : reinterpret_cast<const float4 *>(cb.tiles) + ((row >> 6) * cb.d4) * WAVE + (row & 63);
    const float *x = rows + ((first + pr.x) % n_rows) * cb.d;
    float acc = 0.0f;
    constexpr int UP = 8;                                // chunks per lane and round: 32 chunks of the pair in flight
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q0 = 0; q0 < cb.d4; q0 += 4 * UP) {
      float4 cc[UP], xx[UP];
      if (vec && q0 + 4 * UP <= cb.d4) {                 // whole round in range: 16 unconditional loads in flight
        const float4 *cq = crow + static_cast<int64_t>(q0 + j) * cstep;
        const float4 *xq = reinterpret_cast<const float4 *>(x) + q0 + j;
#pragma unroll
        for (int u = 0; u < UP; u++) { cc[u] = cq[static_cast<int64_t>(4 * u) * cstep]; xx[u] = xq[4 * u]; }
      } else {
#pragma unroll
        for (int u = 0; u < UP; u++) {
          const int q = q0 + 4 * u + j;
          const bool in = q < cb.d4;                     // past the end: (0 - 0)^2 = +0, exact to add
          cc[u] = in ? crow[static_cast<int64_t>(q) * cstep] : zero4;
          xx[u] = !in ? zero4 : vec ? reinterpret_cast<const float4 *>(x)[q] : load_x4<false>(x, q, cb.d);
        }
      }
#pragma unroll
      for (int u = 0; u < UP; u++) {
        float t;
        t = cc[u].x - xx[u].x; cc[u].x = t * t;
        t = cc[u].y - xx[u].y; cc[u].y = t * t;
        t = cc[u].z - xx[u].z; cc[u].z = t * t;
        t = cc[u].w - xx[u].w; cc[u].w = t * t;
      }
#define QB(V, CTRL) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(V), CTRL, 0xF, 0xF, false))
#define QADD(U, CTRL) acc = acc + QB(cc[U].x, CTRL); acc = acc + QB(cc[U].y, CTRL); acc = acc + QB(cc[U].z, CTRL); acc = acc + QB(cc[U].w, CTRL);
#pragma unroll
      for (int u = 0; u < UP; u++) {                     // chunks q0 + 4u .. q0 + 4u + 3, held by quad lanes 0..3
        QADD(u, 0x00) QADD(u, 0x55) QADD(u, 0xAA) QADD(u, 0xFF)
      }
#undef QADD
#undef QB
    }
    if (j == 0) {
      const uint64_t k = make_key(acc, unit_of_row(cb, row));
      atomicMin(reinterpret_cast<unsigned long long *>(keys + pr.x), static_cast<unsigned long long>(k));
    }
  }
}

}  // namespace somhip
