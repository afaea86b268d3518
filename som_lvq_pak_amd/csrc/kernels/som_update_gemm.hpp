// kernels/som_update_gemm.hpp -- K4m: the mini-batch neighbourhood update as a GEMM on the fp32 matrix pipe
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "som_update.hpp"

namespace somhip {

// =====================================================================================
// K4m.  The in-order update of a run, c <- c + a_j (x_j - c) for every sample j whose neighbourhood holds the
// unit (bubble_adapt som_rout.c:472-506 + adapt_vector lvq_pak.c:339-351), unrolled over the run's k hits of
// one unit is the affine map
//        c'  =  P_0 c  +  sum_j  w_j x_j ,      w_j = a_j prod_{i > j} (1 - a_i) ,   P_0 = prod_i (1 - a_i)
// (a_j = the iteration's rate where the unit is a member, 0 elsewhere).  For the 64 units of a row group that is
// a [64 x k] x [k x d] matrix product: W from the group's member list (K4b), X = the listed samples' rows.  It
// runs on v_mfma_f32_32x32x2_f32 -- fp32 operands, fp32 accumulation: two flops per (unit, hit, dim) on the
// matrix pipe instead of three dependent, separately rounded ones on the vector ALU.
//
// This is the UPDATE MODE "gemm" (somhip_engine_set_update_mode): the same real-valued result as the exact
// kernels, rounded differently (a sum of k products instead of a chain of k three-op steps), so it is not
// bit-identical to orc_som_training(batch) -- it is held to that within fp32 rounding noise by the parity
// tests and, end to end, to the reference's online result by bench.py's qerror check.
//
// The list is walked BACKWARDS in chunks of KT entries: the weights then come from one running product P (the
// decay every earlier hit will suffer), and once P < 2^-32 for every live unit of the group the rest of the list
// cannot move the result by a thousandth of an fp32 ulp of the later terms (the weights sum to 1 - P_0 <= 1) and
// is skipped: at alpha = 0.05 that is ~440 hits however many thousand the batch holds for the group.
//
// Workgroup = one row group x (128 NTW)-dim slice, 4 waves of 32 NTW dims each.  Per chunk of KT list entries:
// wave 0 forms the weights (lane = unit) into LDS, all waves stage the chunk's sample rows into LDS; both for the
// NEXT chunk, whose global loads (rows, and the member entries of the chunk after it) were issued before this
// chunk's MFMAs -- 2 x NTW MFMAs per pair of entries and wave.
// =====================================================================================
constexpr int GEMM_KT = 16;            // entries per chunk (8 K-steps of the 32x32x2 MFMA)
constexpr float GEMM_CUT = 2.3283064365386963e-10f;   // 2^-32
constexpr int GEMM_MAX_RUN = 8192;     // samples per run this kernel takes (the host sends longer runs to the exact kernels)
constexpr int LIST_CHUNKS_MAX = GEMM_MAX_RUN / GEMM_KT + 1;

template <int NTW>                     // 32-dim tiles per wave: the workgroup covers 128 * NTW dims
__global__ __launch_bounds__(256, 2) void k_som_update_gemm(CbView cb, const float *__restrict__ rows, int64_t n_rows,
                                                            int64_t data_first, int64_t count,
                                                            const uint32_t *__restrict__ cnt,
                                                            const MemberEntry *__restrict__ ent,
                                                            const uint32_t *__restrict__ order,
                                                            unsigned long long *__restrict__ stats) {
  constexpr int DW = 128 * NTW;                         // dims per workgroup
  constexpr int KT = GEMM_KT;
  // LDS: sample rows [2][KT][DW], weights [2][KT][64], the member entries of three chunks, final decays.  After
  // the walk the first 32 x DW floats are reused to turn the accumulators into the codebook's tile order.
  __shared__ float s_x[2 * KT * DW];
  __shared__ float s_w[2][KT][WAVE];
  __shared__ MemberEntry s_ent[3][KT];
  __shared__ float s_p[WAVE];
  __shared__ int s_more[2];                             // per buffer: another chunk follows
  __shared__ unsigned long long s_earlier[LIST_CHUNKS_MAX + 1];   // units that still have a hit in a chunk before chunk c
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t nslices = static_cast<uint32_t>(cb.d / DW);
  const uint32_t item = blockIdx.x;
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const int d0 = static_cast<int>(item % nslices) * DW;  // first dim of the slice
  const int n_ent = static_cast<int>(cnt[g]);
  if (n_ent == 0) return;
  const MemberEntry *list = ent + g * list_stride(count);
  const bool live = g * WAVE + lane < cb.n;

  f32x16 acc[2][NTW];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int n = 0; n < NTW; n++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[t][n][v] = 0.0f;

  constexpr int F4 = DW / 4;                            // float4 per row of the slice
  constexpr int PER = KT * F4 / 256;                    // float4 per thread and chunk
  static_assert(KT * F4 % 256 == 0, "whole float4 per thread");
  float4 xr[PER];
  MemberEntry er, ec;                                   // wave 0, lanes < KT: one entry of the chunk after next / of the next chunk
  ec.sample = 0u; ec.alpha = 0.0f; ec.mask = 0ull;
  float P = 1.0f;                                       // wave 0, lane = unit: decay of everything processed so far
  unsigned long long processed = 0;

  // chunk c (c = 0 is the END of the list) holds entries [n_ent - (c + 1) KT, n_ent - c KT); indices < 0: nothing
  auto fetch_entries = [&](int c) {                      // global -> register (wave 0, one entry per lane)
    if (wave == 0 && lane < KT) {
      const int idx = n_ent - (c + 1) * KT + lane;
      er.sample = 0u; er.alpha = 0.0f; er.mask = 0ull;
      if (idx >= 0) er = list[idx];
    }
  };
  auto put_entries = [&](int c) { if (wave == 0 && lane < KT) s_ent[c % 3][lane] = er; };
  auto load_rows = [&](int c) {                          // sample rows of chunk c (entries already in LDS) -> registers
    uint32_t smp[PER];
    bool ok[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {                      // all entry reads first, then all row loads: nothing waits on a round trip
      const MemberEntry m = s_ent[c % 3][(tid + 256 * u) / F4];
      smp[u] = m.sample;
      ok[u] = m.mask != 0ull;
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {                      // the entry carries where its sample's row starts, in float4 units (K4b)
      const int q = (tid + 256 * u) % F4;
      const float4 v = reinterpret_cast<const float4 *>(rows)[static_cast<size_t>(ok[u] ? smp[u] : 0u) + (d0 >> 2) + q];
      xr[u] = ok[u] ? v : make_float4(0.f, 0.f, 0.f, 0.f);   // (row 0 is always there; its values are dropped)
    }
  };
  auto store_rows = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int e = (tid + 256 * u) / F4, q = (tid + 256 * u) % F4;
      *reinterpret_cast<float4 *>(&s_x[(buf * KT + e) * DW + 4 * q]) = xr[u];
    }
  };
  const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane >= 32 ? 1u << (lane - 32) : 0u;
  auto weights = [&](int c, int buf) {                   // wave 0: w of chunk c (its entries: lanes < KT of `ec`), walking backwards
    if (wave != 0) return;
#pragma unroll
    for (int e = KT - 1; e >= 0; e--) {                  // entry e broadcast from lane e's registers: no memory on this chain
      const uint32_t mlo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(ec.mask)), e));
      const uint32_t mhi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(ec.mask >> 32)), e));
      const float ae = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ec.alpha), e));
      const float a = ((mlo & bit_lo) | (mhi & bit_hi)) ? ae : 0.0f;
      s_w[buf][e][lane] = a * P;
      P = P * (1.0f - a);
    }
    // go on while some live unit both has weight left to give (P >= 2^-32) and an earlier hit to give it to
    const bool more = n_ent - (c + 1) * KT > 0 && __any(live && P >= GEMM_CUT && ((s_earlier[c] >> lane) & 1ull));
    if (lane == 0) s_more[buf] = more ? 1 : 0;
  };
  // s_earlier[c] = OR of the masks of every entry before chunk c (chunks c + 1, c + 2, ...): lane = chunk, then a suffix scan
  if (wave == 0) {
    const int nch = (n_ent + KT - 1) / KT;
    unsigned long long carry = 0ull;                     // OR of all chunks beyond the ones handled so far
    for (int c0 = ((nch - 1) / WAVE) * WAVE; c0 >= 0; c0 -= WAVE) {   // blocks of 64 chunks, from the far end of the walk
      const int c = c0 + lane;
      unsigned long long m = 0ull;
      if (c < nch)
        for (int e = 0; e < KT; e++) { const int idx = n_ent - (c + 1) * KT + e; if (idx >= 0) m |= list[idx].mask; }
      unsigned long long inc = m;                        // inclusive suffix OR inside the block: lanes above
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        const unsigned long long o = __shfl_down(inc, off, WAVE);
        if (lane + off < WAVE) inc |= o;
      }
      const unsigned long long above = __shfl_down(inc, 1, WAVE);
      if (c < nch) s_earlier[c] = (lane + 1 < WAVE ? above : 0ull) | carry;
      carry |= __shfl(inc, 0, WAVE);
    }
  }

  fetch_entries(0); put_entries(0); ec = er;
  fetch_entries(1); put_entries(1);
  __syncthreads();
  load_rows(0);
  weights(0, 0);
  ec = er;                                               // chunk 1's entries
  store_rows(0);
  __syncthreads();
  for (int c = 0;; c++) {
    const int buf = c & 1;
    const bool more = s_more[buf] != 0;
    if (more) { load_rows(c + 1); fetch_entries(c + 2); }   // next chunk's rows and the entries after it: in flight during the MFMAs
    // ---- C += W_chunk * X_chunk: every operand of the chunk read from LDS first (one latency, not eight)
    float a0[KT / 2], a1[KT / 2], bx[KT / 2][NTW];
#pragma unroll
    for (int s = 0; s < KT / 2; s++) {
      const int kk = 2 * s + (lane >> 5);
      a0[s] = s_w[buf][kk][lane & 31];
      a1[s] = s_w[buf][kk][32 + (lane & 31)];
#pragma unroll
      for (int n = 0; n < NTW; n++) bx[s][n] = s_x[(buf * KT + kk) * DW + wave * 32 * NTW + 32 * n + (lane & 31)];
    }
#pragma unroll
    for (int s = 0; s < KT / 2; s++)
#pragma unroll
      for (int n = 0; n < NTW; n++) {
        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bx[s][n], acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bx[s][n], acc[1][n], 0, 0, 0);
      }
    processed += KT;
    if (!more) break;
    weights(c + 1, buf ^ 1);
    ec = er;                                             // the entries fetched during the MFMAs: chunk c + 2
    store_rows(buf ^ 1);
    put_entries(c + 2);
    __syncthreads();
  }
  if (wave == 0) s_p[lane] = P;
  // ---- c' = P c + acc.  Register v of a 32x32 accumulator tile holds row (v / 4) * 8 + (lane / 32) * 4 + v % 4,
  // column lane % 32 (rows = units, columns = dims).  One half of the units at a time goes through LDS in the
  // codebook's tile order [chunk][unit][4], so that the read-modify-write of the tiles is whole float4s, 32 units in a row.
#pragma unroll
  for (int t = 0; t < 2; t++) {
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NTW; n++) {
      const int dl = wave * 32 * NTW + 32 * n + (lane & 31);       // dim inside the slice
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int ul = (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);   // unit inside the half
        s_x[((dl >> 2) * 32 + ul) * 4 + (dl & 3)] = acc[t][n][v];
      }
    }
    __syncthreads();
    for (int e = tid; e < F4 * 32; e += 256) {
      const int q = e >> 5, ul = e & 31, unit = 32 * t + ul;
      const float4 a = *reinterpret_cast<const float4 *>(&s_x[e * 4]);
      float4 *p = tile_ptr_w(cb, g, (d0 >> 2) + q, unit);
      const float4 c = *p;
      const float pu = s_p[unit];
      *p = make_float4(pu * c.x + a.x, pu * c.y + a.y, pu * c.z + a.z, pu * c.w + a.w);
    }
  }
  if (stats && tid == 0 && d0 == 0) atomicAdd(stats + 8 + 2 * 64 + (g & 7), processed);   // entries walked for this group
}

}  // namespace somhip
