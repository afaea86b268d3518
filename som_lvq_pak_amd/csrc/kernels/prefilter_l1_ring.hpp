// kernels/prefilter_l1_ring.hpp -- level 1 of the two-level pre-filter as a PERSISTENT kernel over an LDS ring (round 3)
// (part of kernels.hpp; see the notes at the top of that file and of prefilter_mfma.hpp, K2c)
//
// Same arithmetic, same tile and the same LDS image per stage as k_dist_mfma_bf16_l1w16 (256 codes x 256 samples per
// workgroup of 8 waves, a stage = one K-step of 32 dims of the hi arrays = 32 KiB, v_mfma_f32_16x16x32_bf16, wave
// (wr, wc) multiplies code group wr by sample tiles 4 wc .. 4 wc + 3), hence the same group minima bit for bit.  What
// changes is everything around the MFMAs (rocprof of round 2: 35 % of that kernel was not matrix work):
//   * a RING of NS = 4 stage slots instead of two buffers: the LDS-DMA requests of two stages are in flight while a
//     third is read (the fourth is the one just read, whose last reads may still be under way), and a stage is waited
//     for with a COUNTED s_waitcnt vmcnt (the younger stage stays in flight) -- the code tiles come from the Infinity Cache (64 MiB of hi tiles do not fit an XCD's L2), 500-900
//     cycles away, and one stage of compute (~1000 cycles) did not cover that;
//   * the fragments of stage q + 1 are read from LDS WHILE the 32 MFMAs of stage q issue (code fragments into a second
//     register set, a sample fragment into its own registers once its column of MFMAs is out), by ds_read_b128 in
//     inline assembly with counted s_waitcnt lgkmcnt: hipcc puts lgkmcnt(0) in front of the first use behind any
//     control-flow join -- a loop's back edge included -- however old the read it needs
//     (before: 10 ds_read_b128, s_waitcnt lgkmcnt(0), 16 MFMAs, 2 reads, wait, 16 MFMAs -- both waves of a SIMD in
//     lockstep, so the matrix pipe stood still during every wait);
//   * one workgroup per CU walks its tiles in one launch and the ring runs on ACROSS tile boundaries: the requests of
//     the next tile's first stages are made during the last stages of this one (no pipeline fill per tile);
//   * bare s_barrier (no fence: __syncthreads() would drain vmcnt) -- what orders LDS is stated where it is used;
//   * epilogue through LDS: the four groups' minima of a tile are stored as 512 contiguous bytes per wave (before: 64
//     bytes per store instruction), and the per-sample minimum over the workgroup's tiles of a sample column goes into
//     gmin1 with one atomicMin per (workgroup, column, sample) -- k_group_min's pass over the whole wmin matrix (110 MB at 65536 x 32768) is gone.
// Tiles are dealt to the workgroups in the order of k_dist_mfma_bf16_l1w16 (super-columns of 64 sample columns);
// workgroup w takes tiles w, w + G, w + 2G, ...: with G a multiple of 8 a workgroup stays on the columns of its XCD.
#pragma once
#include "rerank.hpp"

namespace somhip {

// L1R_ABLATE (measurement builds only: make lib EXTRA=-DL1R_ABLATE=n LIB=...; results are then wrong by design):
//   1 no epilogue arithmetic, 2 no LDS-DMA requests inside the stages, 3 no fragment reads inside the stages,
//   4 no MFMAs -- what the kernel costs without each part (tools/l1_probe.py --quick, profiles/r03_l1_ablation.txt)
#ifndef L1R_ABLATE
#define L1R_ABLATE 0
#endif
#ifndef L1R_SPREAD
#define L1R_SPREAD 0
#endif
#ifndef L1R_LATE
#define L1R_LATE 5        // waves 4-7 make their requests behind this column of MFMAs (3: +1.5 %, profiles/r03_l1_variants.txt)
#endif
#ifndef L1R_PRIO_LEVEL
#define L1R_PRIO_LEVEL 3
#endif
#ifndef L1R_PRIO
#define L1R_PRIO 1        // waves 4-7 at s_setprio 1: -5 % (profiles/r03_l1_variants.txt); L1R_SPREAD (one request per column): +4 %
#endif
constexpr int L1R_NS = 4;                   // ring slots
constexpr int L1R_TOT = 2048;               // uint4 per slot: 4 groups x 4 k-blocks x 64 rows + 8 sample tiles x 4 k-blocks x 32
constexpr int L1R_RED = 16 * 256;           // floats: [4 groups][4 row quarters][256 samples]
constexpr int L1R_CN = 4 * 4 * 64;          // floats: squared norms of the tile's rows, four tiles in rotation (the requests run up to two tiles ahead when a tile has two stages)
constexpr size_t L1R_LDS_BYTES = sizeof(uint4) * L1R_NS * L1R_TOT + sizeof(float) * (L1R_RED + L1R_CN);

typedef uint32_t l1r_u32x4 __attribute__((ext_vector_type(4)));

// a kernel argument held in a scalar register from here on (the compiler otherwise re-loads arguments inside the loop)
template <typename T>
__device__ __forceinline__ T l1r_pin(T v) {
  asm volatile("" : "+s"(v));
  return v;
}
// one ds_read_b128 the compiler does not see as a load: NOTHING but the counted waits below orders its result
template <int OFF>
__device__ __forceinline__ void l1r_read(l1r_u32x4 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

__global__ __launch_bounds__(512, 2) void k_dist_mfma_bf16_l1r(CbView cb, int d8_, const uint4 *__restrict__ chi_,
                                                               const uint4 *__restrict__ xhi_, const float *__restrict__ cn_,
                                                               int64_t bpad_, float *__restrict__ wmin_,
                                                               uint32_t *__restrict__ gmin1_, int ntx_, int nty_) {
  constexpr int BD_KB = 4, NS = L1R_NS, TOT = L1R_TOT;
  constexpr int CH = 0, XH = 4 * BD_KB * 64;
  static_assert(XH + 8 * BD_KB * 32 == TOT, "slot size");
  extern __shared__ uint4 l1r_lds[];
  uint4 *lds = l1r_lds;
  float *s_red = reinterpret_cast<float *>(l1r_lds + NS * TOT);        // [4 groups][4 row quarters][256 samples]
  float *s_cn = s_red + L1R_RED;                                       // [tile & 3][4 groups][64 rows]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int d8 = l1r_pin(d8_), ntx = l1r_pin(ntx_), nty = l1r_pin(nty_);
  const int64_t bpad = l1r_pin(bpad_), ngroups = l1r_pin(cb.ngroups);
  const uint4 *chi = l1r_pin(chi_), *xhi = l1r_pin(xhi_);
  const float *cn = l1r_pin(cn_);
  float *wmin = l1r_pin(wmin_);
  uint32_t *gmin1 = l1r_pin(gmin1_);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                // this wave multiplies code group wr x sample tiles 4 wc .. 4 wc + 3
  const int kg = lane >> 4, l15 = lane & 15;
  const int arr = wave & 1, sel = wave >> 1;              // what this wave brings in: k-blocks 2 arr, 2 arr + 1 of group sel and of sample tiles 2 sel, 2 sel + 1
  const int64_t nst = bpad / 32;
  const int nstage = d8 / BD_KB;                          // even (host: d8 % 8 == 0)
  const int ntiles = ntx * nty;
  const int G = static_cast<int>(gridDim.x), wg = static_cast<int>(blockIdx.x);
  const int my_tiles = wg < ntiles ? (ntiles - wg + G - 1) / G : 0;
  if (my_tiles == 0) return;
  constexpr int L1_SC = 64;
  auto tile_xy = [&](int lin, int &tx, int &ty) {
    const int per_sc = L1_SC * nty;
    const int sc = lin / per_sc, rem = lin - sc * per_sc;
    const int cols = ntx - sc * L1_SC < L1_SC ? ntx - sc * L1_SC : L1_SC;   // the last super-column may be narrower
    ty = rem / cols;
    tx = sc * L1_SC + rem - ty * cols;
  };
  // ---- the issuing side: runs three stages ahead of the multiplying side, on into the next tile.  Behind the last
  // real stage it goes on with three PHANTOM stages (the last tile's operands again, into slots nobody reads any more):
  // every stage of the loop then makes one set of requests and waits for one -- no branches, one constant in the
  // s_waitcnt.  With a tile's first stage the waves with arr == 0 also request the squared norms of their code group
  // (256 bytes, one dword per lane) into s_cn[tile & 3]: one request more in front of that stage's four, which
  // makes a wait that counts four per stage stricter by one request, never laxer. ----
  const uint4 *pc = nullptr, *px0 = nullptr, *px1 = nullptr;
  int is_tile = 0, is_stage = 0, q_issue = 0;
  auto issue_tile = [&](int k) {
    int tx, ty;
    tile_xy(wg + (k < my_tiles ? k : my_tiles - 1) * G, tx, ty);
    const int64_t g0 = static_cast<int64_t>(ty) * 4, st0 = static_cast<int64_t>(tx) * 8;
    const int64_t gsrc = g0 + sel < ngroups ? g0 + sel : ngroups - 1;
    pc = chi + (gsrc * d8 + 2 * arr) * 64 + lane;
    const int64_t t0 = st0 + 2 * sel < nst ? st0 + 2 * sel : nst - 1;
    const int64_t t1 = st0 + 2 * sel + 1 < nst ? st0 + 2 * sel + 1 : nst - 1;
    px0 = xhi + (t0 * d8 + 2 * arr) * 32 + lane;
    px1 = xhi + (t1 * d8 + 2 * arr) * 32 + lane;
    if (arr == 0)
      __builtin_amdgcn_global_load_lds((glb_void *)(cn + gsrc * 64 + lane), (lds_void *)(s_cn + (k & 3) * 256 + sel * 64), 4, 0, 0);
  };
  const int dc = CH + (sel * BD_KB + 2 * arr) * 64;
  const int dx = XH + ((2 * sel) * BD_KB + 2 * arr) * 32;             // + t * BD_KB * 32
  auto issue_piece = [&](int k) {                         // piece k of the next stage of the flattened (tile, stage) sequence
    if (k == 0 && is_stage == 0) issue_tile(is_tile);
    uint4 *buf = lds + (q_issue & (NS - 1)) * TOT;
    const int kb0 = is_stage * BD_KB;
    if (k == 0) __builtin_amdgcn_global_load_lds((glb_void *)(pc + kb0 * 64), (lds_void *)(buf + dc), 16, 0, 0);
    if (k == 1) __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + 1) * 64), (lds_void *)(buf + dc + 64), 16, 0, 0);
    if (k == 2) __builtin_amdgcn_global_load_lds((glb_void *)(px0 + kb0 * 32), (lds_void *)(buf + dx), 16, 0, 0);
    if (k == 3) {
      __builtin_amdgcn_global_load_lds((glb_void *)(px1 + kb0 * 32), (lds_void *)(buf + dx + BD_KB * 32), 16, 0, 0);
      q_issue++;
      if (++is_stage == nstage) { is_stage = 0; is_tile++; }
    }
  };
  auto issue = [&]() { issue_piece(0); issue_piece(1); issue_piece(2); issue_piece(3); };
  // ---- the multiplying side ----
  f32x4v acc[4][8];                                       // [16-row block of the group][16-sample block of the wave's 128]
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[i][j][r] = 0.0f;
  };
  // byte addresses in LDS of this lane's fragments inside a slot: code fragment i at fa_b + 256 i, sample fragment j at
  // fb_b + 2048 (j >> 1) + 256 (j & 1)
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint4 *)lds));
  const uint32_t fa_b = lds0 + 16u * static_cast<uint32_t>(CH + (wr * BD_KB + kg) * 64 + l15);
  const uint32_t fb_b = lds0 + 16u * static_cast<uint32_t>(XH + ((wc * 4) * BD_KB + kg) * 32 + l15);
  l1r_u32x4 bfr[8];                                       // sample fragments: ONE set, refilled column by column (below)
  int q = 0;                                              // the stage being multiplied, over all tiles
#define L1R_COL(J)                                                                                                     \
  do {                                                                                                                 \
    if (L1R_ABLATE != 4) {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                    \
        acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ca[i]), __builtin_bit_cast(bf16x8, bfr[J]), acc[i][J], 0, 0, 0); \
    }                                                                                                                  \
    if (L1R_ABLATE != 3) {                                                                                             \
      l1r_read<2048 * ((J) >> 1) + 256 * ((J) & 1)>(bfr[J], nb);                                                       \
      if ((J) < 4) l1r_read<256 * ((J) & 3)>(na[(J) & 3], nab);                                                        \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
  auto stage = [&](l1r_u32x4 (&ca)[4], l1r_u32x4 (&na)[4]) {
    // On entry: the reads of stage q's fragments were made during stage q - 1, in the order b0 a0 b1 a1 b2 a2 b3 a3 b4 b5
    // b6 b7; requests of stages q + 1 and q + 2 made (real or phantom).
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // this wave's pieces of stage q + 1 have landed (q + 2 may be under way)
    // everybody's pieces of stage q + 1 have landed, and slot (q - 1) % NS is free: a wave is here only after it issued
    // the MFMAs of stage q - 1, for which every one of its reads of that stage had to be back
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // stage q + 3 into slot (q + 3) % NS = (q - 1) % NS.  The two waves of a SIMD (w and w + 4) leave the barrier together:
    // if both made their four requests now (60 cycles and more each, MI355X_MICROARCH.md) the matrix pipe of their SIMD
    // would stand still for that long -- waves 0 .. 3 make them here, waves 4 .. 7 behind a later column of their MFMAs,
    // so that each wave's requests go out under its partner's MFMAs (L1R_LATE: behind the sixth column measured best)
    if (L1R_ABLATE != 2 && !L1R_SPREAD && wave < 4) issue();
    const uint32_t slot_b = static_cast<uint32_t>((q + 1) & (NS - 1)) * (TOT * 16u);
    const uint32_t nab = fa_b + slot_b, nb = fb_b + slot_b;
    // b0 .. b3 and a0 .. a3 of stage q are back once at most the four reads behind them (b4 .. b7) are outstanding
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ca[0]), "+v"(ca[1]), "+v"(ca[2]), "+v"(ca[3]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]));
    // fragments of stage q + 1 while the MFMAs of stage q issue: behind column j of the MFMAs (4 of them) the sample
    // fragment of that column is refilled in place, and -- behind the first four columns -- one code fragment goes into
    // the other set (sched_barrier: the order written here is the order issued)
    // (L1R_SPREAD: one request behind each of a wave's columns 0-3 (waves 0-3) / 4-7 (waves 4-7) instead of four in a row)
    L1R_COL(0); if (L1R_SPREAD && wave < 4) issue_piece(0);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 0 && wave >= 4) issue();
    L1R_COL(1); if (L1R_SPREAD && wave < 4) issue_piece(1);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 1 && wave >= 4) issue();
    L1R_COL(2); if (L1R_SPREAD && wave < 4) issue_piece(2);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 2 && wave >= 4) issue();
    L1R_COL(3); if (L1R_SPREAD && wave < 4) issue_piece(3);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 3 && wave >= 4) issue();
    // b4 of stage q: behind it b5 b6 b7 and the eight reads just made
    asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(bfr[4]));
    L1R_COL(4); if (L1R_SPREAD && wave >= 4) issue_piece(0);
    asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(bfr[5]));  // b6 b7 + 9
    L1R_COL(5); if (L1R_SPREAD && wave >= 4) issue_piece(1);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 5 && wave >= 4) issue();
    asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(bfr[6]));  // b7 + 10
    L1R_COL(6); if (L1R_SPREAD && wave >= 4) issue_piece(2);
    if (L1R_ABLATE != 2 && !L1R_SPREAD && L1R_LATE == 6 && wave >= 4) issue();
    asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(bfr[7]));  // 11
    L1R_COL(7); if (L1R_SPREAD && wave >= 4) issue_piece(3);
    q++;
  };
  zero_acc();
  issue(); issue(); issue();                              // stages 0, 1, 2 into slots 0, 1, 2
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // stage 0 has landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                           // (everybody's)
  asm volatile("" ::: "memory");
  l1r_u32x4 a0[4], a1[4];
  {                                                       // stage 0's fragments, in the order stage() expects
    const uint32_t nab = fa_b, nb = fb_b;
    l1r_read<0>(bfr[0], nb); l1r_read<0>(a0[0], nab);
    l1r_read<256>(bfr[1], nb); l1r_read<256>(a0[1], nab);
    l1r_read<2048>(bfr[2], nb); l1r_read<512>(a0[2], nab);
    l1r_read<2304>(bfr[3], nb); l1r_read<768>(a0[3], nab);
    l1r_read<4096>(bfr[4], nb); l1r_read<4352>(bfr[5], nb); l1r_read<6144>(bfr[6], nb); l1r_read<6400>(bfr[7], nb);
  }
  // from here on: stage() waits for stage q + 1, requests stage q + 3 into the slot stage q - 1 was read from, and
  // reads stage q + 1 while it multiplies stage q: two stages in flight, one being read, one just read
  if (L1R_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(L1R_PRIO_LEVEL);   // (MI355X_MICROARCH.md, two waves per SIMD, item 4: static priority for the younger half)
  float run_min = 3.4e38f;                                // (threads 0 .. 255) minimum over this workgroup's tiles of sample column run_st0
  int64_t run_st0 = -1;
  for (int t = 0; t < my_tiles; t++) {
    int tx, ty;
    tile_xy(wg + t * G, tx, ty);
    const int64_t g0 = static_cast<int64_t>(ty) * 4, st0 = static_cast<int64_t>(tx) * 8;
    const bool gok = g0 + wr < ngroups;
    for (int p = 0; p < nstage / 2; p++) {
      stage(a0, a1);
      stage(a1, a0);
    }
    // ---- epilogue: minimum over the group's 64 rows of ||c||^2 - 2 <c_hi, x_hi> per sample.  Rows 16 i + 4 kg + v are in
    // this lane: its minimum goes to s_red[group][kg][sample], the four row quarters are folded after the barrier.
    // (cn - 2 acc as one fma: 2 acc is exact, so the value is that of the separate multiply and subtract.)  The norms
    // came into s_cn with this tile's first stage: every wave has since waited for a younger request of its own and
    // passed a barrier.
    // (read by ds_read_b128 in assembly as well: in front of a read of LDS that the compiler can see, it drains vmcnt
    // -- the two stages in flight -- because an LDS-DMA request might have written there)
    float4 cnv[4];
    {
      l1r_u32x4 c0, c1, c2, c3;
      const uint32_t ca_b = lds0 + static_cast<uint32_t>(sizeof(uint4) * NS * TOT + sizeof(float) * L1R_RED) +
                            4u * static_cast<uint32_t>((t & 3) * 256 + wr * 64 + 4 * kg);
      l1r_read<0>(c0, ca_b); l1r_read<64>(c1, ca_b); l1r_read<128>(c2, ca_b); l1r_read<192>(c3, ca_b);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      cnv[0] = __builtin_bit_cast(float4, c0); cnv[1] = __builtin_bit_cast(float4, c1);
      cnv[2] = __builtin_bit_cast(float4, c2); cnv[3] = __builtin_bit_cast(float4, c3);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float m = 3.4e38f;
#pragma unroll
      for (int i = 0; i < (L1R_ABLATE == 1 ? 1 : 4); i++) {
        m = fminf(m, fminf(__builtin_fmaf(-2.0f, acc[i][j][0], cnv[i].x), __builtin_fmaf(-2.0f, acc[i][j][1], cnv[i].y)));
        if (L1R_ABLATE == 1) break;
        m = fminf(m, fminf(__builtin_fmaf(-2.0f, acc[i][j][2], cnv[i].z), __builtin_fmaf(-2.0f, acc[i][j][3], cnv[i].w)));
      }
      s_red[(wr * 4 + kg) * 256 + (wc * 4 + (j >> 1)) * 32 + 16 * (j & 1) + l15] = gok ? m : 3.4e38f;
    }
    if (L1R_ABLATE != 1) zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's minima are in LDS ...
    __builtin_amdgcn_s_barrier();                         // ... and so are everybody's
    asm volatile("" ::: "memory");
    {
      const int r4 = tid >> 7, pair = tid & 127;          // wave w stores group w >> 1, samples 128 (w & 1) ... + 127: 512 contiguous bytes
      const float *sr = s_red + r4 * 1024 + 2 * pair;
      const float2 v0 = *reinterpret_cast<const float2 *>(sr), v1 = *reinterpret_cast<const float2 *>(sr + 256);
      const float2 v2 = *reinterpret_cast<const float2 *>(sr + 512), v3 = *reinterpret_cast<const float2 *>(sr + 768);
      float2 v;
      v.x = fminf(fminf(v0.x, v1.x), fminf(v2.x, v3.x));
      v.y = fminf(fminf(v0.y, v1.y), fminf(v2.y, v3.y));
      const int64_t b = st0 * 32 + 2 * pair;
      if (g0 + r4 < ngroups && b < bpad) *reinterpret_cast<float2 *>(wmin + (g0 + r4) * bpad + b) = v;
    }
    if (gmin1 && tid < 256) {                             // the tile's minimum per sample (groups beyond the codebook hold 3.4e38)
      // a workgroup's consecutive tiles lie in one sample column while it stays inside a super-column (tile_xy: tiles w,
      // w + G, ... with G a multiple of 64 keep tx and step ty): the minimum runs on in a register and goes to memory
      // with ONE atomicMin per (workgroup, column) instead of one per tile -- 64 times fewer at configs[3]
      if (st0 != run_st0) {
        const int64_t b = run_st0 * 32 + tid;
        if (run_st0 >= 0 && b < bpad) atomicMin(gmin1 + b, float_to_ordered(run_min));
        run_min = 3.4e38f;
        run_st0 = st0;
      }
      float m = s_red[tid];
#pragma unroll
      for (int k = 1; k < 16; k++) m = fminf(m, s_red[k * 256 + tid]);
      run_min = fminf(run_min, m);
    }
    // (s_red is written again nstage barriers later at the earliest; the LDS reads of the epilogue are the compiler's
    // own: its lgkmcnt(0) in front of their use also covers the fragment reads under way, which is harmless)
  }
#undef L1R_COL
  if (gmin1 && tid < 256 && run_st0 >= 0) {
    const int64_t b = run_st0 * 32 + tid;
    if (b < bpad) atomicMin(gmin1 + b, float_to_ordered(run_min));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the phantom requests and reads must be done before the workgroup's LDS is given away
}

}  // namespace somhip
