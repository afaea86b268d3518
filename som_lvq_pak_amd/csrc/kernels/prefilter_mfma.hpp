// kernels/prefilter_mfma.hpp -- K2/K2b: MFMA distance pre-filter (fp32 and split-bf16 GEMMs, norms, tau)
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "scan_exact.hpp"

namespace somhip {

// =====================================================================================
// K2: the throughput path of the winner search -- an fp32-MFMA distance GEMM used as a
// PRE-FILTER, followed by an exact re-rank (K2r) of the few rows it cannot rule out.
//
//   s~[n,b] = ||c_n||^2 - 2 <c_n, x_b>          (MFMA v_mfma_f32_32x32x2_f32, fp32 fma chain)
//
// differs from the reference's direct-form value d[n,b] = sum_i fl(fl(c_i-x_i)^2) by
// rounding only; with u = 2^-24, g_k = k*u/(1-k*u):
//   |s~ + ||x||^2 - d|  <=  2 g_{d+2} (||x|| + ||c||)^2          
// so every row that can be the exact winner (or tie with it) satisfies
//   s~[n,b] <= min_n s~[n,b] + tau_b,   tau_b = 4 g_{d+2} (||x_b|| + max_n ||c_n||)^2 .
// Per (row group of 64 codes, sample) the kernel keeps the group minimum and a 64-bit
// mask of rows within tau_b of it; K2r recomputes the masked rows of the groups within
// tau_b of the global minimum with the reference's own arithmetic and takes the exact
// (distance, index) minimum.  Result: bit-identical to k_scan_exact / find_winner_euc.
//
// Workgroup = 4 waves as 2 (row groups) x 2 (pairs of 32-sample tiles): a 128 x 128 tile
// of the distance matrix, K = d in stages of QB chunks (4*QB dims) through LDS, two
// stages (register-staged prefetch of the next while the current feeds the MFMAs).
// Codes are the A operand (rows -> accumulator registers), samples the B operand
// (column -> lane), so a sample's minimum over codes is an in-register reduction.
// =====================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MF_QB = 8;                  // chunks per stage: 32 dims

__global__ void k_row_norms(CbView cb, float *__restrict__ cn, unsigned int *__restrict__ cn_max_bits) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (g >= cb.ngroups) return;
  float acc = 0.0f;
  for (int q = 0; q < cb.d4; q++) {
    const float4 c = *tile_ptr(cb, g, q, lane);
    acc += c.x * c.x; acc += c.y * c.y; acc += c.z * c.z; acc += c.w * c.w;
  }
  const int64_t row = g * WAVE + lane;
  cn[row] = row < cb.n ? acc : 3.0e38f;            // padding rows can never be candidates
  float m = row < cb.n ? acc : 0.0f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, WAVE));
  if (lane == 0) atomicMax(cn_max_bits, __float_as_uint(m));   // values >= 0: bit order = value order
}

// tau[b] for the samples of a run (one wave per sample).  The same launch presets the per-run
// scratch of the re-rank (a handful of separate memsets cost more than this whole kernel): the
// keys (all ones), the per-sample global minima (all ones) and group counts, the per-column
// counters and the overflow word.
struct RerankInit {
  uint64_t *keys;        // [count]
  uint32_t *gmin;        // [bpad] then gcount [bpad] then col counters [4 * ncols]
  uint32_t *pair_count;  // overflow word
  int64_t bpad;
  int ncols;
  uint64_t key_init;     // KEY_NONE, or INT64_MAX when the keys go straight into a signed MIN all-reduce
  uint32_t *l2_gmin1;    // two-level pre-filter: per-sample level-1 minimum (preset to the largest ordered value) ...
  uint32_t *l2_cnt;      // ... and the survivor count of every row group (preset to 0); null when one level runs
  int64_t l2_ngroups;
};
__global__ void k_sample_tau(const float *__restrict__ rows, int64_t n_rows, int d, int64_t first,
                             int64_t count, const unsigned int *__restrict__ cn_max_bits,
                             double err_prod, double err_sq, float *__restrict__ tau, RerankInit init,
                             double l1_prod = 0.0, double l1_sq = 0.0, float *__restrict__ tau1 = nullptr,
                             float *__restrict__ xw = nullptr) {
  // xw (shard exchange, K2x below): [0, bpad) delta1, [bpad, 2 bpad) max(delta1, 3 delta3), [2 bpad, 3 bpad) delta3, each rounded up
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (init.gmin) {                                       // 64 * count threads >= bpad + 4 * ncols
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < count && init.keys) init.keys[t] = init.key_init;
    if (t < init.bpad) { init.gmin[t] = 0xFFFFFFFFu; init.gmin[init.bpad + t] = 0u; }
    if (t < 4 * static_cast<int64_t>(init.ncols)) init.gmin[2 * init.bpad + t] = 0u;
    if (t == 0) *init.pair_count = 0u;
    if (init.l2_cnt) {
      if (t < init.bpad) init.l2_gmin1[t] = 0xFFFFFFFFu;
      const int64_t nthreads = static_cast<int64_t>(gridDim.x) * blockDim.x;
      for (int64_t k = t; k < init.l2_ngroups; k += nthreads) init.l2_cnt[k] = 0u;
    }
  }
  if (b >= count) return;
  const float *x = rows + ((first + b) % n_rows) * d;
  double acc = 0.0;
  for (int i = lane; i < d; i += WAVE) { double v = x[i]; acc += v * v; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane == 0) {
    // |s~ + ||x||^2 - d| <= err_prod ||x|| ||c|| + err_sq (||x|| + ||c||)^2 =: delta3 for the GEMM in use (host:
    // prefilter_err3); tau = 2 delta3
    const double u = 5.9604644775390625e-08;                         // 2^-24
    const double cmax = sqrt(static_cast<double>(__uint_as_float(*cn_max_bits)) * (1.0 + 4.0 * d * u));
    const double a = sqrt(acc), s = a + cmax;
    const double t = 2.0 * (err_prod * a * cmax + err_sq * s * s) * 1.001;
    float tf = static_cast<float>(t);
    if (static_cast<double>(tf) < t) tf = __uint_as_float(__float_as_uint(tf) + 1);   // round up
    tau[b] = tf;
    if (tau1) {                                          // the level-1 (one-product) window of the two-level pre-filter:
      // |s~1 - s| <= l1_prod ||x|| ||c|| + l1_sq (||x|| + ||c||)^2 =: delta1 (host: prefilter_err_l1).  The window W above
      // the smallest level-1 group minimum has to (i) hold the exact winner's group: W >= 2 delta1, and (ii) leave
      // every group outside it with a level-1 minimum beyond what the re-rank looks at, min3 + tau <= s_min + 3 delta3,
      // i.e. W >= delta1 + 3 delta3 (a group outside has wmin1 > min1 + W >= s_min - delta1 + W).  delta1 >= 3 delta3 in
      // every ordinary case (the dropped lo parts dwarf the accumulation error), but not for a codebook of tiny norm.
      const double d1 = (l1_prod * a * cmax + l1_sq * s * s) * 1.001, d3 = 0.5 * static_cast<double>(tf);
      const double t1 = d1 + (d1 > 3.0 * d3 ? d1 : 3.0 * d3);
      float t1f = static_cast<float>(t1);
      if (static_cast<double>(t1f) < t1) t1f = __uint_as_float(__float_as_uint(t1f) + 1);
      tau1[b] = t1f;
      if (xw) {
        auto up = [](double v) { float f = static_cast<float>(v); if (static_cast<double>(f) < v) f = __uint_as_float(__float_as_uint(f) + 1); return f; };
        xw[b] = up(d1);
        xw[init.bpad + b] = up(d1 > 3.0 * d3 ? d1 : 3.0 * d3);
        xw[2 * init.bpad + b] = up(d3);
      }
    }
  }
}

// Epilogue shared by the fp32 and the split-bf16 distance GEMMs (same C/D register layout):
// s~ = cn - 2 dot, group minimum per sample, mask of rows within tau of it.
// Accumulator register r of block i is code row 32 i + (r&3) + 8 (r>>2) + 4 half; the lane's
// column is sample (lane & 31) of tile j.
__device__ __forceinline__ void prefilter_epilogue(const CbView &cb, f32x16 (&acc)[2][2], int64_t g,
                                                   int64_t st_first, int64_t nst, int lane,
                                                   const float *__restrict__ cn,
                                                   const float *__restrict__ tau, int64_t count,
                                                   int64_t bpad, float *__restrict__ wmin,
                                                   uint64_t *__restrict__ wmask) {
  if (g >= cb.ngroups) return;
  const int half = lane >> 5, l31 = lane & 31;
  float4 cnv[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int k = 0; k < 4; k++)
      cnv[i][k] = *reinterpret_cast<const float4 *>(cn + g * 64 + 32 * i + 8 * k + 4 * half);
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int64_t st = st_first + j;
    if (st >= nst) continue;
    const int64_t b = st * 32 + l31;
    float sv[2][16];
    float m = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float4 c4 = cnv[i][r >> 2];
        const float cnr = (r & 3) == 0 ? c4.x : (r & 3) == 1 ? c4.y : (r & 3) == 2 ? c4.z : c4.w;
        const float v = cnr - 2.0f * acc[i][j][r];
        sv[i][r] = v;
        m = fminf(m, v);
      }
    m = fminf(m, __shfl_xor(m, 32, WAVE));
    const float thr = m + (b < count ? tau[b] : 0.0f);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (sv[i][r] <= thr) bits |= 1u << (16 * i + r);
    const uint32_t other = __shfl_xor(bits, 32, WAVE);
    if (half == 0 && b < bpad) {
      wmin[g * bpad + b] = m;
      wmask[g * bpad + b] = static_cast<uint64_t>(bits) | (static_cast<uint64_t>(other) << 32);
    }
  }
}

__global__ __launch_bounds__(256, 2) void k_dist_mfma(CbView cb, const float4 *__restrict__ xt,
                                                      const float *__restrict__ cn,
                                                      const float *__restrict__ tau, int64_t count,
                                                      int64_t bpad, float *__restrict__ wmin,
                                                      uint64_t *__restrict__ wmask) {
  // [stage][ 2 groups x QB x 64 | 4 tiles x QB x 32 ] float4
  __shared__ float4 lds[2][2 * MF_QB * 64 + 4 * MF_QB * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: piece maps live in SGPRs
  const int wr = wave >> 1, wc = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 2;          // first row group of the WG
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 4;         // first sample tile of the WG
  const int64_t nst = bpad / 32;

  // ---- staging map: 32 pieces of 1 KiB per stage, 8 per wave, one float4 per lane each.
  // piece p < 16: codes, group p/8, chunk p%8; p >= 16: samples, tile (p-16)/4, chunk pair (p-16)%4
  const float4 *src[8];
  int dst[8];
  int stride[8];                     // float4 stride between stages in global memory
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = wave * 8 + i;
    if (p < 16) {
      const int gi = p >> 3, q = p & 7;
      int64_t g = g0 + gi < cb.ngroups ? g0 + gi : cb.ngroups - 1;
      src[i] = reinterpret_cast<const float4 *>(cb.tiles) + (g * cb.d4 + q) * 64 + lane;
      dst[i] = (gi * MF_QB + q) * 64 + lane;
      stride[i] = MF_QB * 64;
    } else {
      const int ti = (p - 16) >> 2, qp = (p - 16) & 3;
      int64_t st = st0 + ti < nst ? st0 + ti : nst - 1;
      src[i] = xt + (st * cb.d4 + qp * 2) * 32 + lane;              // chunks 2qp, 2qp+1
      dst[i] = 2 * MF_QB * 64 + (ti * MF_QB + qp * 2) * 32 + lane;
      stride[i] = MF_QB * 32;
    }
  }
  const int nstage = (cb.d4 + MF_QB - 1) / MF_QB;
  // chunks beyond d4 (d4 not a multiple of QB) must contribute zeros
  auto stage_load = [&](float4 (&r)[8], int s) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int p = wave * 8 + i;
      int q = s * MF_QB + (p < 16 ? (p & 7) : ((p - 16) & 3) * 2 + (lane >> 5));
      r[i] = q < cb.d4 ? src[i][static_cast<int64_t>(s) * stride[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  float4 regs[8];
  stage_load(regs, 0);
#pragma unroll
  for (int i = 0; i < 8; i++) lds[0][dst[i]] = regs[i];
  __syncthreads();

  for (int s = 0; s < nstage; s++) {
    const int cur = s & 1;
    if (s + 1 < nstage) stage_load(regs, s + 1);
    const float4 *lc = &lds[cur][(wr * MF_QB) * 64];
    const float4 *lx = &lds[cur][2 * MF_QB * 64 + (wc * 2 * MF_QB) * 32];
#pragma unroll
    for (int q = 0; q < MF_QB; q++) {
      const float4 a0 = lc[q * 64 + l31];
      const float4 a1 = lc[q * 64 + 32 + l31];
      const float4 b0 = lx[q * 32 + l31];
      const float4 b1 = lx[(MF_QB + q) * 32 + l31];
      // MFMA 32x32x2: lanes 0-31 carry k, lanes 32-63 carry k+1 (same rule for A and B)
      const float a0k = half ? a0.y : a0.x, a0m = half ? a0.w : a0.z;
      const float a1k = half ? a1.y : a1.x, a1m = half ? a1.w : a1.z;
      const float b0k = half ? b0.y : b0.x, b0m = half ? b0.w : b0.z;
      const float b1k = half ? b1.y : b1.x, b1m = half ? b1.w : b1.z;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0k, b0k, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0k, b1k, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1k, b0k, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1k, b1k, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0m, b0m, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0m, b1m, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1m, b0m, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1m, b1m, acc[1][1], 0, 0, 0);
    }
    if (s + 1 < nstage) {
#pragma unroll
      for (int i = 0; i < 8; i++) lds[cur ^ 1][dst[i]] = regs[i];
    }
    __syncthreads();
  }

  prefilter_epilogue(cb, acc, g0 + wr, st0 + wc * 2, nst, lane, cn, tau, count, bpad, wmin, wmask);
}

// =====================================================================================
// K2b: the same pre-filter on the bf16 matrix pipe (16x the fp32 MFMA rate) by operand
// splitting: v = hi + lo + r, hi = bf16(v), lo = bf16(v - hi), |r| <= 2^-16 |v|, and
//     <c, x>  ~  <c_hi, x_hi> + <c_hi, x_lo> + <c_lo, x_hi>          (3 MFMAs per K-step)
// Products of two bf16 are exact in fp32; what is lost is the dropped lo*lo / r terms
// (<= 3.1 * 2^-16 ||x|| ||c||) and the fp32 accumulation of 3d terms, both added to the
// error coefficients tau is built from (host_scan.inc prefilter_err3), so the exact
// re-rank downstream still returns the reference's bits.  Codes and samples are kept as
// bf16 tiles [group|tile][kb = dim/8][row][8] (16 B per row and k-block = one MFMA operand).
// =====================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_KB = 8;                  // k-blocks (of 8 dims) per stage: 64 dims

__device__ __forceinline__ uint32_t f2bf_rn(float v) {          // finite inputs
  uint32_t u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split_bf16(float v, uint32_t &hi, uint32_t &lo) {
  hi = f2bf_rn(v);
  const float r = v - __uint_as_float(hi << 16);                // exact
  lo = f2bf_rn(r);
}
__device__ __forceinline__ void split8(const float4 a, const float4 b, uint4 &hi, uint4 &lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; j++) split_bf16(v[j], h[j], l[j]);
  hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// squared norms (fp32) + bf16 hi/lo tiles of the codebook, one pass.  One workgroup per row group; its
// blockDim/64 waves (up to 16) take the k-blocks round-robin and their partial norms are added in wave order
// (the bound tau is built from holds for any summation order, and this one is fixed).
// rowmajor (may be null): a row-major fp32 copy of the rows, [ngroups * 64][d] -- every lane stores the 32 bytes of its row
// it holds anyway; the 64 lanes of a store go to 64 rows, but the lines come together in L2 (a row group's 64 rows are
// written by the waves of one workgroup) and leave it whole.  The exact re-rank of single rows reads from it.
__global__ void k_prep_codes_bf16(CbView cb, int d8, float *__restrict__ cn,
                                  unsigned int *__restrict__ cn_max_bits, uint4 *__restrict__ chi,
                                  uint4 *__restrict__ clo, float *__restrict__ rowmajor = nullptr) {
  __shared__ float s_part[16][WAVE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int64_t g = blockIdx.x;
  float acc = 0.0f;
  for (int kb = wave; kb < d8; kb += nw) {
    const float4 a = *tile_ptr(cb, g, 2 * kb, lane);
    const float4 b = (2 * kb + 1 < cb.d4) ? *tile_ptr(cb, g, 2 * kb + 1, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc += a.x * a.x; acc += a.y * a.y; acc += a.z * a.z; acc += a.w * a.w;
    acc += b.x * b.x; acc += b.y * b.y; acc += b.z * b.z; acc += b.w * b.w;
    uint4 hi, lo;
    split8(a, b, hi, lo);
    chi[(g * d8 + kb) * WAVE + lane] = hi;
    clo[(g * d8 + kb) * WAVE + lane] = lo;
    if (rowmajor) {                                       // (d % 4 == 0: the host asks for the copy only then)
      float4 *dst = reinterpret_cast<float4 *>(rowmajor + (g * WAVE + lane) * cb.d) + 2 * kb;
      dst[0] = a;
      if (2 * kb + 1 < cb.d4) dst[1] = b;
    }
  }
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  acc = s_part[0][lane];
  for (int w = 1; w < nw; w++) acc += s_part[w][lane];
  const int64_t row = g * WAVE + lane;
  cn[row] = row < cb.n ? acc : 3.0e38f;
  float m = row < cb.n ? acc : 0.0f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, WAVE));
  if (lane == 0) atomicMax(cn_max_bits, __float_as_uint(m));
}

// the same for a short list of rows (the rows one batch of the LVQ engine corrected): one wave per
// row, lanes over the k-blocks; the norm is a wave sum (any summation order satisfies the bound tau
// is built from)
__global__ __launch_bounds__(256) void k_prep_rows_bf16(CbView cb, int d8, const int32_t *__restrict__ list,
                                                        int nlist, const int32_t *__restrict__ nlist_dev,
                                                        float *__restrict__ cn,
                                                        uint4 *__restrict__ chi, uint4 *__restrict__ clo,
                                                        const int32_t *__restrict__ skip = nullptr) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (skip && *skip) return;                                    // (the LVQ loop's poison flag: the batch was not committed)
  if (w >= nlist || (nlist_dev && w >= *nlist_dev)) return;     // nlist: launch bound; *nlist_dev: the list's length on the device
  const int64_t row = list[w];
  const int64_t g = row >> 6;
  const int rl = static_cast<int>(row & 63);
  float acc = 0.0f;
  for (int kb = lane; kb < d8; kb += WAVE) {
    const float4 a = *tile_ptr(cb, g, 2 * kb, rl);
    const float4 b = (2 * kb + 1 < cb.d4) ? *tile_ptr(cb, g, 2 * kb + 1, rl) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc += a.x * a.x; acc += a.y * a.y; acc += a.z * a.z; acc += a.w * a.w;
    acc += b.x * b.x; acc += b.y * b.y; acc += b.z * b.z; acc += b.w * b.w;
    uint4 hi, lo;
    split8(a, b, hi, lo);
    chi[(g * d8 + kb) * WAVE + rl] = hi;
    clo[(g * d8 + kb) * WAVE + rl] = lo;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane == 0) cn[row] = acc;
}
// max over the live rows of cn (bits; the host zeroes *cn_max_bits first)
__global__ __launch_bounds__(256) void k_max_norm(CbView cb, const float *__restrict__ cn,
                                                  unsigned int *__restrict__ cn_max_bits) {
  float m = 0.0f;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r < cb.n;
       r += static_cast<int64_t>(gridDim.x) * blockDim.x) m = fmaxf(m, cn[r]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, WAVE));
  if ((threadIdx.x & 63) == 0) atomicMax(cn_max_bits, __float_as_uint(m));
}

// a run of samples -> bf16 hi/lo sample tiles xt[sb][kb][32][8].  grid = (32-sample tiles, slices of the k-blocks)
__global__ void k_pack_samples_bf16(const float *__restrict__ rows, int64_t n_rows, int d, int d8,
                                    int64_t first, int64_t count, uint4 *__restrict__ xhi,
                                    uint4 *__restrict__ xlo, unsigned int *__restrict__ zero_word,
                                    uint4 *__restrict__ xrow = nullptr) {
  // xrow (level 2 of the two-level pre-filter gathers single samples): the same pieces sample by sample,
  // xrow[(sample * d8 + kb) * 2 + {hi, lo}] -- a sample's operands of consecutive K-steps are then consecutive bytes
  const int64_t sb = blockIdx.x;
  if (zero_word && sb == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_word = 0u;   // max ||c||^2 accumulator of the next kernel
  const int per = (d8 + gridDim.y - 1) / gridDim.y;
  const int kb_lo = blockIdx.y * per, kb_hi = kb_lo + per < d8 ? kb_lo + per : d8;
  const bool vec = (d & 7) == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0;   // two aligned float4 per k-block
  for (int e = threadIdx.x; e < (kb_hi - kb_lo) * 32; e += blockDim.x) {
    const int kb = kb_lo + e / 32, sidx = e % 32;
    const int64_t smp = sb * 32 + sidx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (smp < count) {
      const float *x = rows + ((first + smp) % n_rows) * d;
      if (vec) {
        a = reinterpret_cast<const float4 *>(x)[2 * kb];
        b = reinterpret_cast<const float4 *>(x)[2 * kb + 1];
      } else {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j++) if (kb * 8 + j < d) v[j] = x[kb * 8 + j];
        a = make_float4(v[0], v[1], v[2], v[3]);
        b = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    uint4 hi, lo;
    split8(a, b, hi, lo);
    xhi[(sb * d8 + kb) * 32 + sidx] = hi;
    xlo[(sb * d8 + kb) * 32 + sidx] = lo;
    if (xrow) { xrow[(smp * d8 + kb) * 2] = hi; xrow[(smp * d8 + kb) * 2 + 1] = lo; }
  }
}

__global__ __launch_bounds__(256, 2) void k_dist_mfma_bf16(CbView cb, int d8,
                                                           const uint4 *__restrict__ chi,
                                                           const uint4 *__restrict__ clo,
                                                           const uint4 *__restrict__ xhi,
                                                           const uint4 *__restrict__ xlo,
                                                           const float *__restrict__ cn,
                                                           const float *__restrict__ tau, int64_t count,
                                                           int64_t bpad, float *__restrict__ wmin,
                                                           uint64_t *__restrict__ wmask) {
  // one stage: codes hi [2][KB][64] | codes lo | samples hi [4][KB][32] | samples lo   (uint4 each)
  constexpr int CH = 0, CL = 2 * BF_KB * 64, XH = 2 * CL, XL = XH + 4 * BF_KB * 32, TOT = XL + 4 * BF_KB * 32;
  __shared__ uint4 lds[TOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: piece maps live in SGPRs
  const int wr = wave >> 1, wc = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 2;
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 4;
  const int64_t nst = bpad / 32;

  // 64 pieces of 1 KiB per stage, 16 per wave, laid out so that everything but four base
  // pointers is a compile-time constant:
  //   wave w stages array (w & 1 ? lo : hi) of code group (w >> 1)      : 8 k-blocks
  //                 and of sample tiles 2(w >> 1), 2(w >> 1) + 1         : 4 k-block pairs each
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const int64_t t0s = st0 + 2 * sel < nst ? st0 + 2 * sel : nst - 1;
  const int64_t t1s = st0 + 2 * sel + 1 < nst ? st0 + 2 * sel + 1 : nst - 1;
  const uint4 *pc = (arr ? clo : chi) + (gsrc * d8) * 64 + lane;          // + kb * 64
  const uint4 *px0 = (arr ? xlo : xhi) + (t0s * d8) * 32 + lane;         // + kp * 64 (two k-blocks)
  const uint4 *px1 = (arr ? xlo : xhi) + (t1s * d8) * 32 + lane;
  const int dc = (arr ? CL : CH) + (sel * BF_KB) * 64 + lane;             // + kb * 64
  const int dx0 = (arr ? XL : XH) + ((2 * sel) * BF_KB) * 32 + lane;      // + kp * 64
  const int dx1 = (arr ? XL : XH) + ((2 * sel + 1) * BF_KB) * 32 + lane;
  const int nstage = (d8 + BF_KB - 1) / BF_KB;
  auto stage_load = [&](uint4 (&r)[16], int s) {
    const int kb0 = s * BF_KB;
    if (kb0 + BF_KB <= d8) {                      // full stage (wave-uniform): no per-piece predicates
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = pc[(kb0 + k) * 64];
#pragma unroll
      for (int k = 0; k < 4; k++) { r[8 + k] = px0[(kb0 + 2 * k) * 32]; r[12 + k] = px1[(kb0 + 2 * k) * 32]; }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = kb0 + k < d8 ? pc[(kb0 + k) * 64] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool ok = kb0 + 2 * k + half < d8;
        r[8 + k] = ok ? px0[(kb0 + 2 * k) * 32] : make_uint4(0u, 0u, 0u, 0u);
        r[12 + k] = ok ? px1[(kb0 + 2 * k) * 32] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto stage_store = [&](const uint4 (&r)[16]) {
#pragma unroll
    for (int k = 0; k < 8; k++) lds[dc + k * 64] = r[k];
#pragma unroll
    for (int k = 0; k < 4; k++) { lds[dx0 + k * 64] = r[8 + k]; lds[dx1 + k * 64] = r[12 + k]; }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  uint4 regs[16];
  stage_load(regs, 0);
  for (int s = 0; s < nstage; s++) {
    stage_store(regs);
    __syncthreads();
    if (s + 1 < nstage) stage_load(regs, s + 1);
#pragma unroll
    for (int m = 0; m < BF_KB / 2; m++) {
      const int kb = 2 * m + half;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ah[i] = __builtin_bit_cast(bf16x8, lds[CH + (wr * BF_KB + kb) * 64 + 32 * i + l31]);
        al[i] = __builtin_bit_cast(bf16x8, lds[CL + (wr * BF_KB + kb) * 64 + 32 * i + l31]);
        bh[i] = __builtin_bit_cast(bf16x8, lds[XH + ((wc * 2 + i) * BF_KB + kb) * 32 + l31]);
        bl[i] = __builtin_bit_cast(bf16x8, lds[XL + ((wc * 2 + i) * BF_KB + kb) * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  prefilter_epilogue(cb, acc, g0 + wr, st0 + wc * 2, nst, lane, cn, tau, count, bpad, wmin, wmask);
}

// The same GEMM with the operands brought in by LDS-DMA (global_load_lds_dwordx4: global -> LDS with
// no VGPR in between, one 1 KiB piece per wave instruction -- the operand tiles are laid out in
// exactly such pieces) into TWO 32 KiB stage buffers of 32 dims: the loads of stage s+1 are in
// flight while stage s is multiplied, one barrier per stage, no ds_write and 64 staging VGPRs fewer.
// Needs dim % 32 == 0 (no zero fill in a DMA); other shapes use the register-staged kernel above.
template <int BD_KB, int MINB>
__global__ __launch_bounds__(256, MINB) void k_dist_mfma_bf16_dma(CbView cb, int d8,
                                                               const uint4 *__restrict__ chi,
                                                               const uint4 *__restrict__ clo,
                                                               const uint4 *__restrict__ xhi,
                                                               const uint4 *__restrict__ xlo,
                                                               const float *__restrict__ cn,
                                                               const float *__restrict__ tau, int64_t count,
                                                               int64_t bpad, float *__restrict__ wmin,
                                                               uint64_t *__restrict__ wmask) {
  constexpr int CH = 0, CL = 2 * BD_KB * 64, XH = 2 * CL, XL = XH + 4 * BD_KB * 32, TOT = XL + 4 * BD_KB * 32;
  __shared__ uint4 lds[2 * TOT];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 2;
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 4;
  const int64_t nst = bpad / 32;
  // wave w brings array (w & 1 ? lo : hi) of code group (w >> 1) and of sample tiles 2(w >> 1), 2(w >> 1) + 1
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const int64_t t0s = st0 + 2 * sel < nst ? st0 + 2 * sel : nst - 1;
  const int64_t t1s = st0 + 2 * sel + 1 < nst ? st0 + 2 * sel + 1 : nst - 1;
  const uint4 *pc = (arr ? clo : chi) + (gsrc * d8) * 64 + lane;
  const uint4 *px0 = (arr ? xlo : xhi) + (t0s * d8) * 32 + lane;
  const uint4 *px1 = (arr ? xlo : xhi) + (t1s * d8) * 32 + lane;
  const int dc = (arr ? CL : CH) + (sel * BD_KB) * 64;              // piece bases (wave-uniform)
  const int dx0 = (arr ? XL : XH) + ((2 * sel) * BD_KB) * 32;
  const int dx1 = (arr ? XL : XH) + ((2 * sel + 1) * BD_KB) * 32;
  const int nstage = d8 / BD_KB;
  auto issue = [&](int s) {
    uint4 *buf = lds + (s & 1) * TOT;
    const int kb0 = s * BD_KB;
#pragma unroll
    for (int k = 0; k < BD_KB; k++)
      __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + k) * 64), (lds_void *)(buf + dc + k * 64), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < BD_KB / 2; k++) {
      __builtin_amdgcn_global_load_lds((glb_void *)(px0 + (kb0 + 2 * k) * 32), (lds_void *)(buf + dx0 + k * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void *)(px1 + (kb0 + 2 * k) * 32), (lds_void *)(buf + dx1 + k * 64), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  // stage s+1 is in flight while stage s is multiplied; the barrier at the top of an iteration also
  // tells that every wave is done with the buffer the next loads go to.  (Three buffers with two
  // stages in flight, counted vmcnt and a raw s_barrier, measured the same: 0.703 vs 0.710 ms.)
  issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nstage) issue(s + 1);
    const uint4 *buf = lds + (s & 1) * TOT;
#pragma unroll
    for (int m = 0; m < BD_KB / 2; m++) {
      const int kb = 2 * m + half;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ah[i] = __builtin_bit_cast(bf16x8, buf[CH + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
        al[i] = __builtin_bit_cast(bf16x8, buf[CL + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
        bh[i] = __builtin_bit_cast(bf16x8, buf[XH + ((wc * 2 + i) * BD_KB + kb) * 32 + l31]);
        bl[i] = __builtin_bit_cast(bf16x8, buf[XL + ((wc * 2 + i) * BD_KB + kb) * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  prefilter_epilogue(cb, acc, g0 + wr, st0 + wc * 2, nst, lane, cn, tau, count, bpad, wmin, wmask);
}

// Wide variant: 128 codes x 256 samples per workgroup, each wave 64 x 128 (2 x 4 MFMA tiles, 24 MFMAs
// per 16-dim k-step on 4 A + 8 B fragment reads instead of 12 on 8): a quarter less operand traffic
// into LDS and a quarter fewer LDS reads per MFMA.  Stages of BD_KB = 2 k-blocks (one k-step), two
// buffers of 24 KiB.
template <int BD_KB>
__global__ __launch_bounds__(256, 2) void k_dist_mfma_bf16_wide(CbView cb, int d8,
                                                                const uint4 *__restrict__ chi,
                                                                const uint4 *__restrict__ clo,
                                                                const uint4 *__restrict__ xhi,
                                                                const uint4 *__restrict__ xlo,
                                                                const float *__restrict__ cn,
                                                                const float *__restrict__ tau, int64_t count,
                                                                int64_t bpad, float *__restrict__ wmin,
                                                                uint64_t *__restrict__ wmask) {
  constexpr int CH = 0, CL = 2 * BD_KB * 64, XH = 2 * CL, XL = XH + 8 * BD_KB * 32, TOT = XL + 8 * BD_KB * 32;
  __shared__ uint4 lds[2 * TOT];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 2;
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 8;
  const int64_t nst = bpad / 32;
  // wave w brings array (w & 1 ? lo : hi) of code group (w >> 1) and of sample tiles 4(w >> 1) .. 4(w >> 1) + 3
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const uint4 *pc = (arr ? clo : chi) + (gsrc * d8) * 64 + lane;
  const uint4 *px[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int64_t ts = st0 + 4 * sel + t < nst ? st0 + 4 * sel + t : nst - 1;
    px[t] = (arr ? xlo : xhi) + (ts * d8) * 32 + lane;
  }
  const int dc = (arr ? CL : CH) + (sel * BD_KB) * 64;
  const int dx = (arr ? XL : XH) + ((4 * sel) * BD_KB) * 32;          // + t * BD_KB * 32
  const int nstage = d8 / BD_KB;
  auto issue = [&](int s) {
    uint4 *buf = lds + (s & 1) * TOT;
    const int kb0 = s * BD_KB;
#pragma unroll
    for (int k = 0; k < BD_KB; k++)
      __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + k) * 64), (lds_void *)(buf + dc + k * 64), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int k = 0; k < BD_KB / 2; k++)
        __builtin_amdgcn_global_load_lds((glb_void *)(px[t] + (kb0 + 2 * k) * 32),
                                         (lds_void *)(buf + dx + t * BD_KB * 32 + k * 64), 16, 0, 0);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

  issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nstage) issue(s + 1);
    const uint4 *buf = lds + (s & 1) * TOT;
#pragma unroll
    for (int m = 0; m < BD_KB / 2; m++) {
      const int kb = 2 * m + half;
      bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ah[i] = __builtin_bit_cast(bf16x8, buf[CH + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
        al[i] = __builtin_bit_cast(bf16x8, buf[CL + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bh[j] = __builtin_bit_cast(bf16x8, buf[XH + ((wc * 4 + j) * BD_KB + kb) * 32 + l31]);
        bl[j] = __builtin_bit_cast(bf16x8, buf[XL + ((wc * 4 + j) * BD_KB + kb) * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++) {
    f32x16 sub[2][2] = {{acc[0][2 * h2], acc[0][2 * h2 + 1]}, {acc[1][2 * h2], acc[1][2 * h2 + 1]}};
    prefilter_epilogue(cb, sub, g0 + wr, st0 + wc * 4 + 2 * h2, nst, lane, cn, tau, count, bpad, wmin, wmask);
  }
}

// =====================================================================================
// K2c: TWO-LEVEL pre-filter (round 2).  Level 1 is the same GEMM with ONE bf16 product per K-step, hi x hi: a third of
// the matrix work and half the operand bytes.  Dropping both lo terms costs |<c,x> - <c_hi,x_hi>| <= 2^-8 (1 + 2^-8)
// ||c|| ||x||, so s~1 is within delta1 ~ 2^-7 ||c|| ||x|| of s and the group of the exact winner has
// wmin1 <= min wmin1 + tau1, tau1 = delta1 + max(delta1, 3 delta3) (k_sample_tau; host: prefilter_err_l1, prefilter_err3).  Over a whole configs[3] run that keeps
// 2 ... 17 of the 1024 row groups per sample (tools/survivor_study.py).  Level 2 (k_dist_l2) runs the three-product
// GEMM on exactly those (group, sample) pairs -- gathered by group, the samples' bf16 pieces loaded per lane -- and
// writes the wmin / wmask the exact re-rank already consumes.  Entries of wmin that level 2 did not touch keep their
// level-1 value, which exceeds the sample's true minimum by more than delta1 >= 3 delta3: the re-rank's test
// wmin <= gmin + tau3 never selects them.
// =====================================================================================
__device__ __forceinline__ void prefilter_epilogue_min(const CbView &cb, f32x16 (&acc)[2][2], int64_t g, int64_t st_first,
                                                       int64_t nst, int lane, const float *__restrict__ cn, int64_t bpad,
                                                       float *__restrict__ wmin) {
  if (g >= cb.ngroups) return;
  const int half = lane >> 5, l31 = lane & 31;
  float4 cnv[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int k = 0; k < 4; k++)
      cnv[i][k] = *reinterpret_cast<const float4 *>(cn + g * 64 + 32 * i + 8 * k + 4 * half);
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int64_t st = st_first + j;
    if (st >= nst) continue;
    const int64_t b = st * 32 + l31;
    float m = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float4 c4 = cnv[i][r >> 2];
        const float cnr = (r & 3) == 0 ? c4.x : (r & 3) == 1 ? c4.y : (r & 3) == 2 ? c4.z : c4.w;
        m = fminf(m, cnr - 2.0f * acc[i][j][r]);
      }
    m = fminf(m, __shfl_xor(m, 32, WAVE));
    if (half == 0 && b < bpad) wmin[g * bpad + b] = m;
  }
}

// level 1: 128 codes x 256 samples per workgroup as the wide kernel; a stage is BD_KB = 4 k-blocks (two k-steps) of
// the hi arrays only: wave (arr, sel) brings k-blocks [2 arr, 2 arr + 2) of code group sel and of sample tiles 4 sel ..
template <int BD_KB>
__global__ __launch_bounds__(256, 2) void k_dist_mfma_bf16_l1(CbView cb, int d8, const uint4 *__restrict__ chi,
                                                              const uint4 *__restrict__ xhi, const float *__restrict__ cn,
                                                              int64_t bpad, float *__restrict__ wmin) {
  static_assert(BD_KB == 4, "two k-steps per stage, one per staging half");
  constexpr int CH = 0, XH = 2 * BD_KB * 64, TOT = XH + 8 * BD_KB * 32;
  __shared__ uint4 lds[2 * TOT];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 2;
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 8;
  const int64_t nst = bpad / 32;
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const uint4 *pc = chi + (gsrc * d8 + 2 * arr) * 64 + lane;
  const uint4 *px[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int64_t ts = st0 + 4 * sel + t < nst ? st0 + 4 * sel + t : nst - 1;
    px[t] = xhi + (ts * d8 + 2 * arr) * 32 + lane;
  }
  const int dc = CH + (sel * BD_KB + 2 * arr) * 64;
  const int dx = XH + ((4 * sel) * BD_KB + 2 * arr) * 32;             // + t * BD_KB * 32
  const int nstage = d8 / BD_KB;
  auto issue = [&](int s) {
    uint4 *buf = lds + (s & 1) * TOT;
    const int kb0 = s * BD_KB;
#pragma unroll
    for (int k = 0; k < 2; k++)
      __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + k) * 64), (lds_void *)(buf + dc + k * 64), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; t++)      // one instruction moves k-blocks kb0 + 2 arr and + 1 of a tile (lanes 0-31 / 32-63)
      __builtin_amdgcn_global_load_lds((glb_void *)(px[t] + kb0 * 32), (lds_void *)(buf + dx + t * BD_KB * 32), 16, 0, 0);
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nstage) issue(s + 1);
    const uint4 *buf = lds + (s & 1) * TOT;
#pragma unroll
    for (int m = 0; m < BD_KB / 2; m++) {
      const int kb = 2 * m + half;
      bf16x8 ah[2], bh[4];
#pragma unroll
      for (int i = 0; i < 2; i++) ah[i] = __builtin_bit_cast(bf16x8, buf[CH + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
#pragma unroll
      for (int j = 0; j < 4; j++) bh[j] = __builtin_bit_cast(bf16x8, buf[XH + ((wc * 4 + j) * BD_KB + kb) * 32 + l31]);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++) {
    f32x16 sub[2][2] = {{acc[0][2 * h2], acc[0][2 * h2 + 1]}, {acc[1][2 * h2], acc[1][2 * h2 + 1]}};
    prefilter_epilogue_min(cb, sub, g0 + wr, st0 + wc * 4 + 2 * h2, nst, lane, cn, bpad, wmin);
  }
}

// The same level-1 GEMM on a 256 codes x 256 samples workgroup tile (8 waves, each 64 x 128 as before): with one
// product per K-step the kernel is bound by the L2 -> LDS operand traffic, and the square tile moves a third less of it
// per MFMA ((256 + 256) / (256 * 256) against (128 + 256) / (128 * 256)).  Wave w brings k-blocks [2 (w & 1), + 2) of
// code group w >> 1 and of sample tiles 2 (w >> 1), 2 (w >> 1) + 1.
template <int BD_KB>
__global__ __launch_bounds__(512, 2) void k_dist_mfma_bf16_l1w(CbView cb, int d8, const uint4 *__restrict__ chi,
                                                               const uint4 *__restrict__ xhi, const float *__restrict__ cn,
                                                               int64_t bpad, float *__restrict__ wmin) {
  static_assert(BD_KB == 4, "two k-steps per stage, one per staging half");
  constexpr int CH = 0, XH = 4 * BD_KB * 64, TOT = XH + 8 * BD_KB * 32;
  __shared__ uint4 lds[2 * TOT];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                // this wave multiplies code group wr x sample tiles 4 wc .. 4 wc + 3
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t g0 = static_cast<int64_t>(blockIdx.y) * 4;
  const int64_t st0 = static_cast<int64_t>(blockIdx.x) * 8;
  const int64_t nst = bpad / 32;
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const uint4 *pc = chi + (gsrc * d8 + 2 * arr) * 64 + lane;
  const uint4 *px[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int64_t ts = st0 + 2 * sel + t < nst ? st0 + 2 * sel + t : nst - 1;
    px[t] = xhi + (ts * d8 + 2 * arr) * 32 + lane;
  }
  const int dc = CH + (sel * BD_KB + 2 * arr) * 64;
  const int dx = XH + ((2 * sel) * BD_KB + 2 * arr) * 32;             // + t * BD_KB * 32
  const int nstage = d8 / BD_KB;
  auto issue = [&](int s) {
    uint4 *buf = lds + (s & 1) * TOT;
    const int kb0 = s * BD_KB;
#pragma unroll
    for (int k = 0; k < 2; k++)
      __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + k) * 64), (lds_void *)(buf + dc + k * 64), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++)
      __builtin_amdgcn_global_load_lds((glb_void *)(px[t] + kb0 * 32), (lds_void *)(buf + dx + t * BD_KB * 32), 16, 0, 0);
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
  issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nstage) issue(s + 1);
    const uint4 *buf = lds + (s & 1) * TOT;
#pragma unroll
    for (int m = 0; m < BD_KB / 2; m++) {
      const int kb = 2 * m + half;
      bf16x8 ah[2], bh[4];
#pragma unroll
      for (int i = 0; i < 2; i++) ah[i] = __builtin_bit_cast(bf16x8, buf[CH + (wr * BD_KB + kb) * 64 + 32 * i + l31]);
#pragma unroll
      for (int j = 0; j < 4; j++) bh[j] = __builtin_bit_cast(bf16x8, buf[XH + ((wc * 4 + j) * BD_KB + kb) * 32 + l31]);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int h2 = 0; h2 < 2; h2++) {
    f32x16 sub[2][2] = {{acc[0][2 * h2], acc[0][2 * h2 + 1]}, {acc[1][2 * h2], acc[1][2 * h2 + 1]}};
    prefilter_epilogue_min(cb, sub, g0 + wr, st0 + wc * 4 + 2 * h2, nst, lane, cn, bpad, wmin);
  }
}

// The same workgroup tile, staging and LDS layout with v_mfma_f32_16x16x32_bf16: a stage of 4 k-blocks is ONE K-step of
// 32 dims; lane (r = lane & 15, kg = lane >> 4) supplies k-block kg of row / sample r, which is one 16-byte piece of the
// staged tiles as they are.  A wave's 64 x 128 tile is 4 x 8 MFMAs per stage on the same 12 fragment reads; register v
// of an output tile is row 4 (lane >> 4) + v, column lane & 15.  (The 16x16x32 form sustains a higher rate than
// 32x32x16 on this part: MI355X_MICROARCH.md, bare-loop measurements.)
template <int BD_KB>
__global__ __launch_bounds__(512, 2) void k_dist_mfma_bf16_l1w16(CbView cb, int d8, const uint4 *__restrict__ chi,
                                                                 const uint4 *__restrict__ xhi, const float *__restrict__ cn,
                                                                 int64_t bpad, float *__restrict__ wmin, int ntx, int nty) {
  static_assert(BD_KB == 4, "one K-step of 32 dims per stage");
  constexpr int CH = 0, XH = 4 * BD_KB * 64, TOT = XH + 8 * BD_KB * 32;
  __shared__ uint4 lds[2 * TOT];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;                // this wave multiplies code group wr x sample tiles 4 wc .. 4 wc + 3
  const int kg = lane >> 4, l15 = lane & 15;
  // 1-D grid over the ntx (256-sample columns) x nty (256-code blocks) tiles, in SUPER-COLUMNS of L1_SC columns: all code
  // blocks of 64 columns (16384 samples) before the next 64.  Workgroups go to the 8 XCDs round-robin, so an XCD multiplies
  // every 8th column; with the columns of a super-column only, the sample tiles it keeps re-reading are 8 x 256 KiB --
  // they stay in its 4 MiB L2 next to the code block in hand.  (Plain x-fastest order over 128 columns of a 32768-vector
  // batch: 16 columns = 4 MiB per XCD, and 3.3 GB per launch came from beyond L2 instead of 0.5.)
  constexpr int L1_SC = 64;
  int tx, ty;
  {
    const int lin = blockIdx.x, per_sc = L1_SC * nty;
    const int sc = lin / per_sc, rem = lin - sc * per_sc;
    const int cols = ntx - sc * L1_SC < L1_SC ? ntx - sc * L1_SC : L1_SC;   // the last super-column may be narrower
    ty = rem / cols;
    tx = sc * L1_SC + rem - ty * cols;
  }
  const int64_t g0 = static_cast<int64_t>(ty) * 4;
  const int64_t st0 = static_cast<int64_t>(tx) * 8;
  const int64_t nst = bpad / 32;
  const int arr = wave & 1, sel = wave >> 1;
  const int64_t gsrc = g0 + sel < cb.ngroups ? g0 + sel : cb.ngroups - 1;
  const uint4 *pc = chi + (gsrc * d8 + 2 * arr) * 64 + lane;
  const uint4 *px[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int64_t ts = st0 + 2 * sel + t < nst ? st0 + 2 * sel + t : nst - 1;
    px[t] = xhi + (ts * d8 + 2 * arr) * 32 + lane;
  }
  const int dc = CH + (sel * BD_KB + 2 * arr) * 64;
  const int dx = XH + ((2 * sel) * BD_KB + 2 * arr) * 32;             // + t * BD_KB * 32
  const int nstage = d8 / BD_KB;
  auto issue = [&](int s) {
    uint4 *buf = lds + (s & 1) * TOT;
    const int kb0 = s * BD_KB;
#pragma unroll
    for (int k = 0; k < 2; k++)
      __builtin_amdgcn_global_load_lds((glb_void *)(pc + (kb0 + k) * 64), (lds_void *)(buf + dc + k * 64), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++)
      __builtin_amdgcn_global_load_lds((glb_void *)(px[t] + kb0 * 32), (lds_void *)(buf + dx + t * BD_KB * 32), 16, 0, 0);
  };
  f32x4v acc[4][8];                                       // [16-row block of the group][16-sample block of the wave's 128]
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = 0.0f;
  issue(0);
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nstage) issue(s + 1);
    const uint4 *buf = lds + (s & 1) * TOT;
    bf16x8 ah[4], bh[8];
#pragma unroll
    for (int i = 0; i < 4; i++) ah[i] = __builtin_bit_cast(bf16x8, buf[CH + (wr * BD_KB + kg) * 64 + 16 * i + l15]);
#pragma unroll
    for (int j = 0; j < 8; j++) bh[j] = __builtin_bit_cast(bf16x8, buf[XH + ((wc * 4 + (j >> 1)) * BD_KB + kg) * 32 + 16 * (j & 1) + l15]);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  }
  // minimum over the group's 64 rows of ||c||^2 - 2 <c_hi, x_hi> per sample: rows 16 i + 4 kg + v in this lane, the other
  // three quarters of the rows in lanes l15 + 16, + 32, + 48
  const int64_t g = g0 + wr;
  if (g >= cb.ngroups) return;
  float4 cnv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) cnv[i] = *reinterpret_cast<const float4 *>(cn + g * 64 + 16 * i + 4 * kg);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int64_t st = st0 + wc * 4 + (j >> 1);
    float m = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      m = fminf(m, cnv[i].x - 2.0f * acc[i][j][0]);
      m = fminf(m, cnv[i].y - 2.0f * acc[i][j][1]);
      m = fminf(m, cnv[i].z - 2.0f * acc[i][j][2]);
      m = fminf(m, cnv[i].w - 2.0f * acc[i][j][3]);
    }
    m = fminf(m, __shfl_xor(m, 16, WAVE));
    m = fminf(m, __shfl_xor(m, 32, WAVE));
    const int64_t b = st * 32 + 16 * (j & 1) + l15;
    if (kg == 0 && st < nst && b < bpad) wmin[g * bpad + b] = m;
  }
}

// survivors of level 1, gathered by row group: list[g][0 .. cnt[g]) = the samples b with wmin1[g][b] <= gmin1[b] + tau1[b].
// Grid: (256-sample columns, chunks of groups); thread = sample, a wave = 64 consecutive samples of one group per trip
// (256 contiguous bytes of wmin; four groups' loads in flight), one list reservation per wave and group.
__global__ __launch_bounds__(256) void k_l2_select(int64_t ngroups, int64_t count, int64_t bpad, int64_t chunk,
                                                   const float *wmin, const uint32_t *__restrict__ gmin1,
                                                   const float *__restrict__ tau1, uint32_t *__restrict__ cnt,
                                                   uint16_t *__restrict__ list, float *mark = nullptr,
                                                   const float *__restrict__ xub = nullptr) {
  // xub (shard exchange): the bound agreed between the shards takes the place of this shard's own minimum
  // mark (= wmin; shard exchange): a (group, sample) pair that is left out gets the value 3.4e38 in place of its level-1
  // minimum -- with a bound from another shard a shard may keep no group at all for a sample, and what is left out must
  // never look like a candidate to the re-rank
  const int lane = threadIdx.x & 63;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t g_lo = static_cast<int64_t>(blockIdx.y) * chunk;
  const int64_t g_hi = g_lo + chunk < ngroups ? g_lo + chunk : ngroups;
  float thr = -3.4e38f;
  if (b < count) {
    const uint32_t o = gmin1[b];                         // order-preserving image of the float minimum (float_to_ordered)
    thr = (xub ? xub[b] : __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o)) + tau1[b];
  }
  const bool live = b < count;                           // (count <= bpad: the loads stay inside the rows of wmin)
  // sixteen groups per trip: their loads in flight together, and ONE vector atomic for the sixteen list reservations (lane k
  // reserves for group g0 + k) -- a reservation per group and wave, each waited for in turn, was this kernel's time
  // (64 round trips per wave: 90 us per 32768 samples whatever the size of the shard)
  const unsigned long long below = (1ull << lane) - 1ull;
  constexpr int GT = 16;
  for (int64_t g0 = g_lo; g0 < g_hi; g0 += GT) {
    float v[GT];
#pragma unroll
    for (int k = 0; k < GT; k++) v[k] = (live && g0 + k < g_hi) ? wmin[(g0 + k) * bpad + b] : 3.4e38f;
    unsigned long long bal[GT];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < GT; k++) {
      const bool valid = live && g0 + k < g_hi;
      const bool in = valid && v[k] <= thr;
      bal[k] = __ballot(in);
      if (mark && valid && !in) mark[(g0 + k) * bpad + b] = 3.4e38f;
      if (lane == k) mine = static_cast<uint32_t>(__popcll(bal[k]));
    }
    uint32_t base = 0;
    if (mine) base = atomicAdd(&cnt[g0 + lane], mine);   // (mine != 0 only in lanes 0..15 and only for groups below g_hi)
#pragma unroll
    for (int k = 0; k < GT; k++) {
      if (bal[k] == 0ull) continue;                      // wave-uniform
      const uint32_t at = __shfl(base, k, WAVE);
      if ((bal[k] >> lane) & 1ull) list[(g0 + k) * bpad + at + __popcll(bal[k] & below)] = static_cast<uint16_t>(b);
    }
  }
}

// level 2: the three-product GEMM of one row group against tiles of 32 of ITS surviving samples; one wave per tile,
// operands straight from global memory (each lane loads the 16-byte pieces the MFMA wants from it: for B the piece
// of its own gathered sample).  Writes wmin / wmask of the (group, sample) pairs it covers.
template <int DEPTH>                   // k-steps whose operands are requested together
__global__ __launch_bounds__(256) void k_dist_l2(CbView cb, int d8, const uint4 *__restrict__ chi, const uint4 *__restrict__ clo,
                                                 const uint4 *__restrict__ xhi, const uint4 *__restrict__ xlo,
                                                 const float *__restrict__ cn, const float *__restrict__ tau, int64_t bpad,
                                                 const uint32_t *__restrict__ cnt, const uint16_t *__restrict__ list,
                                                 float *__restrict__ wmin, uint64_t *__restrict__ wmask,
                                                 unsigned long long *__restrict__ stats, const uint4 *__restrict__ xrow = nullptr,
                                                 uint32_t *__restrict__ gmin = nullptr) {
  const int64_t g = blockIdx.x;
  const int n = static_cast<int>(cnt[g]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int ntiles = (n + 31) >> 5;
  if (stats && threadIdx.x == 0 && blockIdx.y == 0 && n) atomicAdd(stats, static_cast<unsigned long long>(n));
  // the tiles of a group are dealt to the 4 gridDim.y waves that serve it (a group may hold one tile or a hundred)
  for (int tile = blockIdx.y * 4 + wave; tile < ntiles; tile += 4 * gridDim.y) {
    const int slot = tile * 32 + l31;
    const bool valid = slot < n;
    const int64_t b = list[g * bpad + (valid ? slot : 0)];
    const uint4 *pa = chi + (g * d8 + half) * 64 + l31, *pl = clo + (g * d8 + half) * 64 + l31;
    // operand B: this lane's sample, k-block o + half -- from the sample-major copy when there is one (consecutive bytes
    // per sample: whole cache lines are used), else from the 32-sample tiles (16 bytes out of every 512)
    const uint4 *pxh = xrow ? xrow + (b * d8 + half) * 2 : xhi + ((b >> 5) * d8 + half) * 32 + (b & 31);
    const uint4 *pxl = xrow ? pxh + 1 : xlo + ((b >> 5) * d8 + half) * 32 + (b & 31);
    const int xs = xrow ? 2 : 32;                          // uint4 between consecutive k-blocks of one sample
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    // k-step ks uses k-blocks 2 ks + half.  The loop is a chain of global-memory round trips (nothing else hides them:
    // a group often has a single tile), so the operands of DEPTH k-steps are requested together: 1 / DEPTH of the trips.
    const int nks = d8 / 2;                              // a multiple of 2 (the host takes this path for d8 % 4 == 0)
    for (int ks0 = 0; ks0 < nks; ks0 += DEPTH) {
      uint4 rA0[DEPTH], rA1[DEPTH], rL0[DEPTH], rL1[DEPTH], rBH[DEPTH], rBL[DEPTH];
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
        const int o = 2 * (ks0 + u < nks ? ks0 + u : nks - 1);
        rA0[u] = pa[o * 64]; rA1[u] = pa[o * 64 + 32]; rL0[u] = pl[o * 64]; rL1[u] = pl[o * 64 + 32];
        rBH[u] = pxh[o * xs]; rBL[u] = pxl[o * xs];
      }
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
        if (ks0 + u < nks) {
          const bf16x8 a0 = __builtin_bit_cast(bf16x8, rA0[u]), a1 = __builtin_bit_cast(bf16x8, rA1[u]);
          const bf16x8 l0 = __builtin_bit_cast(bf16x8, rL0[u]), l1 = __builtin_bit_cast(bf16x8, rL1[u]);
          const bf16x8 bh = __builtin_bit_cast(bf16x8, rBH[u]), bl = __builtin_bit_cast(bf16x8, rBL[u]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bl, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, bh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, bh, acc[1], 0, 0, 0);
        }
      }
    }
    // same bit <-> row rule as prefilter_epilogue: register r of block i is row 32 i + (r & 3) + 8 (r >> 2) + 4 half
    float sv[2][16];
    float m = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = cn[g * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half] - 2.0f * acc[i][r];
        sv[i][r] = v;
        m = fminf(m, v);
      }
    m = fminf(m, __shfl_xor(m, 32, WAVE));
    const float thr = m + tau[b];
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (sv[i][r] <= thr) bits |= 1u << (16 * i + r);
    const uint32_t other = __shfl_xor(bits, 32, WAVE);
    if (half == 0 && valid) {
      wmin[g * bpad + b] = m;
      wmask[g * bpad + b] = static_cast<uint64_t>(bits) | (static_cast<uint64_t>(other) << 32);
      // the sample's smallest three-product minimum (what k_group_min would find: a group that level 1 left out keeps a
      // level-1 value above it, see the window's condition (ii)); ordered image of the float, no value returned
      if (gmin) { const uint32_t u = __float_as_uint(m); atomicMin(gmin + b, (u & 0x80000000u) ? ~u : (u | 0x80000000u)); }
    }
  }
}

// level 2 with the group's tiles resident in LDS (dims up to 512: 64 rows x 512 dims x (hi + lo) = 128 KiB): one workgroup
// per row group brings its 2 d8 one-KiB pieces in by LDS-DMA once, then its four waves take the group's sample tiles in
// turn -- operand A from LDS, operand B (each lane the pieces of its own gathered sample) from global memory, two blocks
// of eight K-steps in flight.  With 16 384-vector batches a group has four or five tiles: the operand-A traffic of
// k_dist_l2 (each tile re-reading the group's 128 KiB through L2) was most of that kernel's time.  Same products in the
// same order as k_dist_l2.
#ifndef L2_WAVES_V
#define L2_WAVES_V 8
#endif
#ifndef L2_DEPTH_V
#define L2_DEPTH_V 2       // K-steps of operand B per register block (two blocks in rotation).  Round 3, us per 32768-vector
#endif                   // launch along the configs[3] schedule: depth 8 (round 2) 546 ... 91, 4: 498 ... 80, 2: 444 ... 73, 1: 453 ... 79
constexpr int L2_WAVES = L2_WAVES_V;   // waves of a k_dist_l2_lds workgroup (one workgroup per CU: its LDS holds a group's tiles)
__global__ __launch_bounds__(64 * L2_WAVES, 1) void k_dist_l2_lds(CbView cb, int d8, const uint4 *__restrict__ chi, const uint4 *__restrict__ clo,
                                                        const uint4 *__restrict__ xhi, const uint4 *__restrict__ xlo,
                                                        const float *__restrict__ cn, const float *__restrict__ tau, int64_t bpad,
                                                        const uint32_t *__restrict__ cnt, const uint16_t *__restrict__ list,
                                                        float *__restrict__ wmin, uint64_t *__restrict__ wmask,
                                                        unsigned long long *__restrict__ stats, const uint4 *__restrict__ xrow = nullptr,
                                                        uint32_t *__restrict__ gmin = nullptr) {
  extern __shared__ uint4 s_l2a[];                         // [hi | lo][d8][64]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int64_t g = blockIdx.x;
  const int n = static_cast<int>(cnt[g]);
  const int ntiles = (n + 31) >> 5;
  // a crowded group's tiles are dealt to the gridDim.y workgroups of its row, a wave each; the others leave at once
  if (static_cast<int>(blockIdx.y) * L2_WAVES >= ntiles) return;
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  if (stats && threadIdx.x == 0 && blockIdx.y == 0) atomicAdd(stats, static_cast<unsigned long long>(n));
  for (int p = wave; p < 2 * d8; p += L2_WAVES) {
    const uint4 *src = (p < d8 ? chi + (g * d8 + p) * 64 : clo + (g * d8 + (p - d8)) * 64) + lane;
    __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(s_l2a + p * 64), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint4 *sah = s_l2a, *sal = s_l2a + d8 * 64;
  const int nks = d8 / 2;                                  // K-step ks uses k-blocks 2 ks + half
  constexpr int D = L2_DEPTH_V;
  for (int tile = blockIdx.y * L2_WAVES + wave; tile < ntiles; tile += L2_WAVES * gridDim.y) {
    const int slot = tile * 32 + l31;
    const bool valid = slot < n;
    const int64_t b = list[g * bpad + (valid ? slot : 0)];
    const uint4 *pxh = xrow ? xrow + (b * d8 + half) * 2 : xhi + ((b >> 5) * d8 + half) * 32 + (b & 31);   // (see k_dist_l2)
    const uint4 *pxl = xrow ? pxh + 1 : xlo + ((b >> 5) * d8 + half) * 32 + (b & 31);
    const int xs = xrow ? 2 : 32;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
    uint4 bh[2][D], bl[2][D];
    auto fetch = [&](int ks0, int buf) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int o = 2 * (ks0 + u < nks ? ks0 + u : nks - 1);
        bh[buf][u] = pxh[o * xs]; bl[buf][u] = pxl[o * xs];
      }
    };
    auto mult = [&](int ks0, int buf) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        if (ks0 + u < nks) {
          const int kb = 2 * (ks0 + u) + half;
          const bf16x8 a0 = __builtin_bit_cast(bf16x8, sah[kb * 64 + l31]), a1 = __builtin_bit_cast(bf16x8, sah[kb * 64 + 32 + l31]);
          const bf16x8 l0 = __builtin_bit_cast(bf16x8, sal[kb * 64 + l31]), l1 = __builtin_bit_cast(bf16x8, sal[kb * 64 + 32 + l31]);
          const bf16x8 xh = __builtin_bit_cast(bf16x8, bh[buf][u]), xl = __builtin_bit_cast(bf16x8, bl[buf][u]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xh, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xl, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xl, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, xh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, xh, acc[1], 0, 0, 0);
        }
      }
    };
    fetch(0, 0);
    for (int ks0 = 0; ks0 < nks; ks0 += 2 * D) {           // two register blocks in rotation (the loop is unrolled over both)
      if (ks0 + D < nks) fetch(ks0 + D, 1);
      mult(ks0, 0);
      if (ks0 + D < nks) {
        if (ks0 + 2 * D < nks) fetch(ks0 + 2 * D, 0);
        mult(ks0 + D, 1);
      }
    }
    // same bit <-> row rule as prefilter_epilogue: register r of block i is row 32 i + (r & 3) + 8 (r >> 2) + 4 half
    float sv[2][16];
    float m = 3.4e38f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = cn[g * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half] - 2.0f * acc[i][r];
        sv[i][r] = v;
        m = fminf(m, v);
      }
    m = fminf(m, __shfl_xor(m, 32, WAVE));
    const float thr = m + tau[b];
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (sv[i][r] <= thr) bits |= 1u << (16 * i + r);
    const uint32_t other = __shfl_xor(bits, 32, WAVE);
    if (half == 0 && valid) {
      wmin[g * bpad + b] = m;
      wmask[g * bpad + b] = static_cast<uint64_t>(bits) | (static_cast<uint64_t>(other) << 32);
      // the sample's smallest three-product minimum (what k_group_min would find: a group that level 1 left out keeps a
      // level-1 value above it, see the window's condition (ii)); ordered image of the float, no value returned
      if (gmin) { const uint32_t u = __float_as_uint(m); atomicMin(gmin + b, (u & 0x80000000u) ? ~u : (u | 0x80000000u)); }
    }
  }
}

}  // namespace somhip
