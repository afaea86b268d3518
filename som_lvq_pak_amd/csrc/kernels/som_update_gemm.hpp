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
// decay every earlier hit will suffer), and once P < 2^-24 for every live unit of the group that still has an
// earlier hit, the rest of the list is skipped: the weights of everything skipped sum to less than P (all weights
// sum to 1 - P_0 <= 1), so the skipped part of c' is below 2^-24 of the data's magnitude -- half an ulp of a value
// of that magnitude, a thirtieth of the rounding noise the sum of the ~330 kept products carries anyway (measured
// against the exact kernels: 2e-6 of the scale, tools/gemm_check.py).  At alpha = 0.05 that is ~330 hits however
// many thousand the batch holds for the group.
//
// Workgroup = one row group x (128 NTW)-dim slice, 4 waves of 32 NTW dims each.  The product is formed TRANSPOSED,
// D[dims x units] = X^T[dims x k] W^T[k x units]: an accumulator register then holds four consecutive dims of one
// unit over 32 consecutive units per half wave -- whole float4s of the codebook's tiles, so c' = P c + D is read,
// formed and written straight from the accumulators (512 contiguous bytes per half wave), no pass through LDS.
//
// On gfx950 the fp32 MFMA does not run in the shadow of anything the vector ALU does -- VALU cycles of any wave of
// the SIMD add to its matrix time (tools/micro/mfma_indep.hip, profiles/r02_mfma_valu_overlap_microbench.txt) --
// so the loop is built to issue as few vector instructions as possible next to its 64 MFMAs per chunk and wave:
//   * the member entries come in by SCALAR loads, four entries (64 bytes) per s_load_dwordx16 into two SGPR
//     buffers in rotation, a quarter chunk ahead;
//   * weights: every wave runs the chain for the NEXT chunk itself, lane = unit, three vector instructions per
//     entry -- w = alpha P (scalar operand), w = member ? w : 0 (the entry's 64-bit mask as the select's SGPR pair),
//     P = P - w -- and one v_permlane32_swap per pair of entries to turn two weight registers into the two B
//     operands (units 0-31 / 32-63 x entries 2s, 2s+1).  No wave waits for another one's weights;
//   * sample rows of a chunk: LDS-DMA (global_load_lds, 1 KiB pieces), two chunk buffers, requested a chunk ahead;
//     operand A = a float4 / float2 / float of a row per lane and K-step
//     (M tile mt, row m  <->  dim NTW m + mt of the wave's slice).  One barrier per chunk hands over the buffer.
// (Round 2, first form: wave 0 made the weights into LDS, ~13 vector instructions per entry, and every wave staged
// rows through registers: 0.39 ms for 448 entries x 1024 groups, of which the MFMAs were 0.25 ms.)
// =====================================================================================
constexpr int GEMM_KT = 16;            // entries per chunk (8 K-steps of the 32x32x2 MFMA)
constexpr float GEMM_CUT = 5.9604644775390625e-08f;    // 2^-24
constexpr int GEMM_MAX_RUN = 65504;    // samples per run this kernel takes (the host sends longer runs to the exact kernels)
constexpr int GEMM_FRONT_PAD = 16;     // entries of readable memory in front of the first list (host: scratch layout)

template <int NTW> struct GemmA;                         // NTW consecutive floats of a staged row
template <> struct GemmA<4> { typedef float4 T; };
template <> struct GemmA<2> { typedef float2 T; };
template <> struct GemmA<1> { typedef float T; };
__device__ __forceinline__ float gemm_a_elem(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ float gemm_a_elem(const float2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float gemm_a_elem(const float &v, int) { return v; }
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
// four consecutive entries -> 16 SGPRs.  Inline assembly for the reason given at K4s: the compiler would sink the
// load to its first use.  A buffer is read only after it went through gemm_wait_quarter.
__device__ __forceinline__ void gemm_load_quarter(u32x16_t &q, const MemberEntry *p) {
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(q) : "s"(p));
}
__device__ __forceinline__ void gemm_wait_quarter(u32x16_t &q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q)); }
// One 1 KiB piece global -> LDS (lane l: 16 bytes from src to dst + 16 l).  Inline assembly and not
// __builtin_amdgcn_global_load_lds: the compiler cannot tell the two chunk buffers apart and puts s_waitcnt vmcnt(0)
// between a request and the next ds_read of ANY LDS address -- the requests for the next chunk would be waited for
// before the first MFMA of this one (measured: 0.05 of 0.48 ms).  The kernel's own vmcnt(0) + barrier at the top of
// a chunk is the wait that counts.
__device__ __forceinline__ void gemm_dma_piece(const float4 *src, const float *lds_dst) {
  typedef const __attribute__((address_space(3))) float lds_float;
  const uint32_t dst = (uint32_t)(uintptr_t)((lds_float *)lds_dst);      // the LDS byte address
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
}  // (m0 is a reserved register: the compiler keeps nothing in it across statements and cannot be told about the write)

// GAUSS: gaussian neighbourhoods (gaussian_adapt som_rout.c:511-549) in the same form.  Every sample updates every
// unit with its own rate a = alpha exp(-lattice_sq / (2 radius^2)), so W is dense: the chain computes the rate per
// (unit, entry) -- lattice_sq from the lane's unit and the entry's winner (fp32, exact for maps up to 1024 x 1024),
// v_exp_f32 with the iteration's log2(e) / (2 radius^2) from the entry -- and there is no early stop (a unit far from
// every winner keeps P = 1): the whole list is walked, 2 * 64 * d flop per entry instead of the 3 * 64 * d of the exact
// kernels' vector arithmetic.  K4b leaves out the samples whose rate is below 2^-40 alpha for the whole group.
template <int NTW, bool GAUSS = false> // 32-dim tiles per wave: the workgroup covers 128 * NTW dims
__global__ __launch_bounds__(256, 2) void k_som_update_gemm(CbView cb, const float *__restrict__ rows, int64_t n_rows,
                                                            int64_t data_first, int64_t count,
                                                            const uint32_t *__restrict__ cnt,
                                                            const MemberEntry *__restrict__ ent,
                                                            const uint32_t *__restrict__ order,
                                                            unsigned long long *__restrict__ stats,
                                                            const uint32_t *__restrict__ lstart = nullptr) {
  static_assert(sizeof(MemberEntry) == 16, "four entries per s_load_dwordx16");
  constexpr int DW = 128 * NTW;                         // dims per workgroup
  constexpr int KT = GEMM_KT;
  typedef typename GemmA<NTW>::T AT;
  __shared__ __attribute__((aligned(16))) float s_x[2 * KT * DW];              // sample rows of two chunks, [KT][DW] each
  __shared__ int s_hc[WAVE];                                       // per unit: the furthest chunk (from the list's end) with a hit
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const uint32_t nslices = static_cast<uint32_t>(cb.d / DW);
  const uint32_t item = blockIdx.x;
  const uint32_t rank = item / nslices;
  const int64_t g = order ? order[rank] : rank;
  const int d0 = static_cast<int>(item % nslices) * DW;  // first dim of the slice
  const int n_ent = __builtin_amdgcn_readfirstlane(static_cast<int>(cnt[g]));
  if (n_ent == 0) return;
  // (K4b in tail mode makes only the end of the list, at the end of the group's region: lstart says where it begins)
  const MemberEntry *list = ent + g * list_stride(count) + (lstart ? __builtin_amdgcn_readfirstlane(static_cast<int>(lstart[g])) : 0);
  const bool live = g * WAVE + lane < cb.n;

  f32x16 acc[NTW][2];                                   // [dim tile][unit half]
#pragma unroll
  for (int m = 0; m < NTW; m++)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int v = 0; v < 16; v++) acc[m][t][v] = 0.0f;

  // chunk c (c = 0 is the END of the list) holds entries [n_ent - (c + 1) KT, n_ent - c KT); quarter j of it the four
  // entries from first_of(c, j).  Indices < 0: nothing (mask forced to 0; the load is clamped to the 16 readable
  // entries in front of the list -- the previous group's, or the scratch buffer's front pad).
  auto first_of = [&](int c, int j) { return n_ent - (c + 1) * KT + 4 * j; };
  auto load_quarter = [&](u32x16_t &q, int c, int j) {
    const int i0 = first_of(c, j);
    gemm_load_quarter(q, list + (i0 > -GEMM_FRONT_PAD ? i0 : -GEMM_FRONT_PAD));
  };
  // where the rows of a chunk's 16 samples start: lane e < 16 of every wave fetches entry e's offset (a vector load:
  // the requests for a whole chunk's rows go out together, a chunk ahead, before the scalar quarters have arrived)
  struct Offs { uint32_t sample, mlo, mhi; };            // raw: looked at only when the rows are requested, a chunk later
  auto fetch_offsets = [&](int c) {
    int idx = n_ent - (c + 1) * KT + (lane & (KT - 1));  // (every lane loads: no exec-masked block, no wait at the load)
    idx = idx >= 0 ? idx : n_ent;                        // a null entry behind the list (LIST_PAD)
    const MemberEntry *p = list + idx;
    Offs o;
    o.sample = __builtin_nontemporal_load(&p->sample);
    const uint32_t *m = reinterpret_cast<const uint32_t *>(&p->mask);
    o.mlo = __builtin_nontemporal_load(m);
    o.mhi = __builtin_nontemporal_load(m + 1);
    return o;
  };
  constexpr int PW = KT * DW / 256 / 4;                 // 1 KiB DMA pieces per wave and chunk
  auto dma_rows = [&](int c, const Offs &o) {            // rows of chunk c -> LDS buffer c & 1, this wave's pieces
    float *buf = s_x + (c & 1) * KT * DW;
    const uint32_t offs = (o.mlo | o.mhi) != 0u ? o.sample : 0u;   // (row 0 is always there; its weight is 0)
#pragma unroll
    for (int u = 0; u < PW; u++) {
      const int p = wave * PW + u;                       // piece: floats [256 p, 256 p + 256) of the buffer
      const int f = p * 256 + lane * 4;
      // (a piece is a part of one row, or two whole rows when the slice is 128 dims wide)
      const uint32_t smp = DW >= 256 ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(offs), p * 256 / DW))
                                     : static_cast<uint32_t>(__shfl(static_cast<int>(offs), f / DW, WAVE));
      // the entry carries where its sample's row starts, in float4 units (K4b)
      const float4 *src = reinterpret_cast<const float4 *>(rows) + static_cast<size_t>(smp) + ((d0 + f % DW) >> 2);
      gemm_dma_piece(src, buf + p * 256);
    }
  };
  // the weights of K-step s of a chunk (entries 2s + 1, then 2s: the walk goes backwards) as the two B operands:
  // b0 = units 0-31 (lanes 0-31: entry 2s, lanes 32-63: entry 2s + 1), b1 = units 32-63 likewise.  q = the quarter
  // that holds the two entries, i0 = the list index of its first entry.
  int u_tx = 0, u_ty = 0;                                // GAUSS: this lane's unit on the lattice
  if (GAUSS) txty_of_row(cb, g * WAVE + lane, u_tx, u_ty);
  auto weight_pair = [&](const u32x16_t &q, int i0, int s, float &Pr, float &b0, float &b1) {
    float w[2];
#pragma unroll
    for (int k = 1; k >= 0; k--) {
      const int i = (2 * s + k) & 3;
      if (GAUSS) {
        // gaussian_adapt's rate alpha exp(-dd dd / (2 radius^2)), dd = (float) sqrt(lattice_sq) (som_rout.c:539-542): the
        // numerator in the reference's form (a float square root -- v_sqrt_f32, 1 ulp --, squared in float); the exponent in
        // base 2 with the coefficient's remainder carried along (t + lo is exact to 2^-32 of the exponent), the
        // exponential by v_exp_f32 (1 ulp) instead of a double exp
        const uint32_t xy = q[4 * i + 2];
        const float lat = lattice_sq_small(cb.topol, static_cast<int>(xy & 0x3FFu), static_cast<int>((xy >> 10) & 0x3FFu), u_tx, u_ty);
        const float dd = __builtin_amdgcn_sqrtf(lat);   // v_sqrt_f32, 1 ulp
        const float n2 = dd * dd;
        const float chi = __uint_as_float(q[4 * i + 3]);
        const float clo = chi * (static_cast<float>(static_cast<int>(xy << 4) >> 24) * 2.3283064365386963e-10f);
        const float t = n2 * chi;
        const float lo = __builtin_fmaf(n2, chi, -t) + n2 * clo;
        const float h = __builtin_amdgcn_exp2f(-t);
        // (below the start of the list the fields are whatever memory holds: the rate is SELECTED to 0, not multiplied)
        const float r = __uint_as_float(q[4 * i + 1]) * __builtin_fmaf(-0.6931471805599453f * lo, h, h);
        const float a = (i0 + i >= 0 && live) ? r : 0.0f;
        w[k] = a * Pr;
      } else {
        const unsigned long long mask = i0 + i >= 0 ? (static_cast<unsigned long long>(q[4 * i + 3]) << 32) | q[4 * i + 2] : 0ull;
        const float ap = __uint_as_float(q[4 * i + 1]) * Pr;
        w[k] = __builtin_amdgcn_inverse_ballot_w64(mask) ? ap : 0.0f;
      }
      Pr = Pr - w[k];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[0]), __float_as_uint(w[1]), false, false);
    b0 = __uint_as_float(r[0]);
    b1 = __uint_as_float(r[1]);
  };

  // s_hc[u] = the chunk furthest from the list's end that still holds a hit of unit u (-1: none): unit u has an EARLIER
  // hit than chunk c iff s_hc[u] > c.  Blocks of 64 chunks from the far end of the walk (lane = chunk: the OR of its 16
  // masks), a ballot per unit; over as soon as every live unit has been seen (at a big radius: the first block).
  if (wave == 0 && !GAUSS) {
    const int nch = (n_ent + KT - 1) / KT;
    int hc = -1;
    for (int c0 = ((nch - 1) / WAVE) * WAVE; c0 >= 0; c0 -= WAVE) {
      const int c = c0 + lane;
      unsigned long long m = 0ull;
      if (c < nch)
        for (int e = 0; e < KT; e++) { const int idx = n_ent - (c + 1) * KT + e; if (idx >= 0) m |= list[idx].mask; }
      for (int u = 0; u < WAVE; u++) {
        const unsigned long long bal = __ballot((m >> u) & 1ull);
        if (lane == u && hc < 0 && bal) hc = c0 + 63 - __clzll(bal);
      }
      if (__all(!live || hc >= 0)) break;
    }
    s_hc[lane] = hc;
  }
  auto more_after = [&](int c, float Pr) {               // is chunk c + 1 needed, P being the decay after chunk c
    // go on while some live unit both has weight left to give (P >= 2^-24) and an earlier hit to give it to
    if (GAUSS) return n_ent - (c + 1) * KT > 0;          // dense weights: the whole list
    return n_ent - (c + 1) * KT > 0 && __any(live && Pr >= GEMM_CUT && s_hc[lane] > c);
  };

  float P = 1.0f;                                        // lane = unit: decay of everything processed so far
  unsigned long long processed = 0;
  float b0[KT / 2], b1[KT / 2];                          // operand B of the current chunk
  u32x16_t qa, qb;                                       // two quarters of entries in SGPRs, in rotation
  load_quarter(qa, 0, 3);
  dma_rows(0, fetch_offsets(0));
  Offs on = fetch_offsets(1);                            // row offsets of the next chunk
#pragma unroll
  for (int j = 3; j >= 0; j--) {                         // chunk 0: rows requested, weights formed, nothing to overlap with
    u32x16_t &q = (j & 1) ? qa : qb, &qn = (j & 1) ? qb : qa;
    gemm_wait_quarter(q);
    if (j > 0) load_quarter(qn, 0, j - 1); else load_quarter(qn, 1, 3);
    weight_pair(q, first_of(0, j), 2 * j + 1, P, b0[2 * j + 1], b1[2 * j + 1]);
    weight_pair(q, first_of(0, j), 2 * j, P, b0[2 * j], b1[2 * j]);
  }
  __syncthreads();                                       // s_hc
  bool more = more_after(0, P);
  for (int c = 0;; c++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of chunk c; the offsets of chunk c + 1
    __syncthreads();                                     // everybody's pieces; everybody is done with the other buffer
    dma_rows(c + 1, on);
    on = fetch_offsets(c + 2);
    const float *buf = s_x + (c & 1) * KT * DW + wave * 32 * NTW + NTW * l31;
    float Pn = P, nb0[KT / 2], nb1[KT / 2];
    AT ax[2];
    ax[1] = *reinterpret_cast<const AT *>(buf + (KT - 2 + half) * DW);
    // K-steps 7..0 of chunk c; under them, quarter by quarter, chunk c + 1: its entries (qa holds quarter 3 on entry),
    // the request for its rows, its weights.  (Past the start of the list all of that is idle work on null entries.)
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      u32x16_t &q = (j & 1) ? qa : qb, &qn = (j & 1) ? qb : qa;
      gemm_wait_quarter(q);
      if (j > 0) load_quarter(qn, c + 1, j - 1); else load_quarter(qn, c + 2, 3);
#pragma unroll
      for (int s = 2 * j + 1; s >= 2 * j; s--) {
        if (s > 0) ax[(s - 1) & 1] = *reinterpret_cast<const AT *>(buf + (2 * (s - 1) + half) * DW);
        weight_pair(q, first_of(c + 1, j), s, Pn, nb0[s], nb1[s]);
#pragma unroll
        for (int m = 0; m < NTW; m++) {
          const float a = gemm_a_elem(ax[s & 1], m);
          acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[s], acc[m][0], 0, 0, 0);
          acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[s], acc[m][1], 0, 0, 0);
        }
      }
    }
    processed += KT;
    if (!more) break;
    P = Pn;
#pragma unroll
    for (int s = 0; s < KT / 2; s++) { b0[s] = nb0[s]; b1[s] = nb1[s]; }
    more = more_after(c + 1, P);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of ours may still be landing in LDS or SGPRs
  // ---- c' = P c + D.  Register v of a 32x32 accumulator tile holds row (v / 4) * 8 + (lane / 32) * 4 + v % 4, column
  // lane % 32: rows = dims NTW row + mt of the wave's slice, columns = units.  For i = v / 4 the rows 8 i + 4 half + j,
  // j = 0..3, of the NTW tiles are 4 NTW consecutive dims: NTW float4 of the unit's tile row.
  float pu[2];
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(P), false, false);
    pu[0] = __uint_as_float(r[0]);                       // lanes l, l + 32: P of unit l
    pu[1] = __uint_as_float(r[1]);                       //                  P of unit 32 + l
  }
#pragma unroll
  for (int t = 0; t < 2; t++) {
    float4 *tp[4][NTW];
    float4 cv[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < NTW; f++) {
        const int q = ((d0 + wave * 32 * NTW + NTW * (8 * i + 4 * half)) >> 2) + f;
        tp[i][f] = tile_ptr_w(cb, g, q, 32 * t + l31);
        cv[i][f] = *tp[i][f];
      }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int f = 0; f < NTW; f++) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {                    // element 4 f + k of the 4 NTW dims: row j, tile mt
          const int el = 4 * f + k, j = el / NTW, mt = el % NTW;
          o[k] = acc[mt][t][4 * i + j];
        }
        const float4 c = cv[i][f];
        *tp[i][f] = make_float4(pu[t] * c.x + o[0], pu[t] * c.y + o[1], pu[t] * c.z + o[2], pu[t] * c.w + o[3]);
      }
  }
  if (stats && tid == 0 && d0 == 0) atomicAdd(stats + 8 + 2 * 64 + (g & 7), processed);   // entries walked for this group
}

}  // namespace somhip
