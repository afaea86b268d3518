// kernels.hpp -- gfx950 (MI355X, CDNA4) device code of the SOM/LVQ engine.
//
// Numerics contract (bit-exact with the reference's CPU code): distances are the
// left-to-right fp32 sum of (c_i - x_i)^2 with a separate rounding after the
// subtract, the multiply and the add (reference lvq_pak.c:70-71); updates are
// c = c + alpha*(x - c) with three roundings (lvq_pak.c:348-349).  This file is
// compiled with -ffp-contract=off and must contain no fma/mac in those chains
// (checked on the ISA by tests/test_build.py).
//
// Data layout in HBM ("row-group tiles"): the codebook is stored as
//     tile[g][q][lane][4]    g = row / 64, lane = row % 64, q = dim / 4
// i.e. for every group of 64 consecutive code rows and every chunk of 4 dims, the
// 64 float4 of that chunk are contiguous (1 KiB).  One lane of a wavefront owns one
// code row, walks its dims in order (which the sequential-sum contract needs) and
// every wave-level load/store is one fully coalesced 1 KiB access.  Rows >= n and
// dims >= d are zero padding (adding +0.0f to a sum of squares is exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace somhip {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr uint64_t KEY_NONE = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t FLT_MAX_BITS = 0x7F7FFFFFu;

struct CbView {
  float *tiles;         // [ngroups][d4][64][4]
  int64_t n;            // local rows
  int64_t ngroups;      // ceil(n / 64)
  int d, d4;
  int64_t row_offset;   // global unit index of the shard's first unit
  int xdim;             // map width (global)
  int topol, neigh;
  int patch_w;          // 0: storage row s holds unit row_offset + s (the reference's order);
                        // > 0 (= xdim/8): "8x8 patch" order -- every 64-row group is an 8x8 block of
                        // map units, so a round neighbourhood fills whole wavefronts instead of
                        // slivers of 64x1 strips.  Maps only, sides multiple of 8, shards on 8-row
                        // boundaries; indices seen outside the engine are always unit indices.
};

// global unit index (= the reference's row index, datafile.c:781,836) of local storage row `row`
__device__ __forceinline__ uint32_t unit_of_row(const CbView &cb, int64_t row) {
  if (cb.patch_w == 0) return static_cast<uint32_t>(row + cb.row_offset);
  const uint32_t p = static_cast<uint32_t>(row >> 6), i = static_cast<uint32_t>(row) & 63u;
  const uint32_t px = p % static_cast<uint32_t>(cb.patch_w), py = p / static_cast<uint32_t>(cb.patch_w);
  return static_cast<uint32_t>(cb.row_offset) + (py * 8 + (i >> 3)) * static_cast<uint32_t>(cb.xdim) + px * 8 + (i & 7);
}
// lattice coordinates of local storage row `row` (som_rout.c:493-494: x = unit % xdim, y = unit / xdim)
__device__ __forceinline__ void txty_of_row(const CbView &cb, int64_t row, int &tx, int &ty) {
  const uint32_t xd = static_cast<uint32_t>(cb.xdim);
  if (cb.patch_w == 0) {
    const uint32_t u = static_cast<uint32_t>(row + cb.row_offset);
    tx = static_cast<int>(u % xd); ty = static_cast<int>(u / xd);
    return;
  }
  const uint32_t p = static_cast<uint32_t>(row >> 6), i = static_cast<uint32_t>(row) & 63u;
  const uint32_t px = p % static_cast<uint32_t>(cb.patch_w), py = p / static_cast<uint32_t>(cb.patch_w);
  tx = static_cast<int>(px * 8 + (i & 7));
  ty = static_cast<int>(static_cast<uint32_t>(cb.row_offset) / xd + py * 8 + (i >> 3));
}

// local storage row of global unit index `unit` (inverse of unit_of_row)
__device__ __forceinline__ int64_t row_of_unit(const CbView &cb, uint32_t unit) {
  const uint32_t u = unit - static_cast<uint32_t>(cb.row_offset);
  if (cb.patch_w == 0) return static_cast<int64_t>(u);
  const uint32_t xd = static_cast<uint32_t>(cb.xdim);
  const uint32_t y = u / xd, x = u % xd;
  const uint32_t p = (y >> 3) * static_cast<uint32_t>(cb.patch_w) + (x >> 3);
  return static_cast<int64_t>(p) * 64 + ((y & 7) << 3) + (x & 7);
}

__device__ __forceinline__ const float4 *tile_ptr(const CbView &cb, int64_t g, int q, int lane) {
  return reinterpret_cast<const float4 *>(cb.tiles) + ((g * cb.d4 + q) * WAVE + lane);
}
__device__ __forceinline__ float4 *tile_ptr_w(const CbView &cb, int64_t g, int q, int lane) {
  return reinterpret_cast<float4 *>(cb.tiles) + ((g * cb.d4 + q) * WAVE + lane);
}

// ---- exact arithmetic helpers (no contraction: see file header) ----
__device__ __forceinline__ float sq_acc(float acc, float c, float x) {
  float t = c - x;
  float p = t * t;
  return acc + p;
}
__device__ __forceinline__ float adapt1(float c, float x, float a) {
  float t = x - c;
  float s = a * t;
  return c + s;
}
__device__ __forceinline__ float4 adapt4(float4 c, float4 x, float a) {
  return make_float4(adapt1(c.x, x.x, a), adapt1(c.y, x.y, a), adapt1(c.z, x.z, a),
                     adapt1(c.w, x.w, a));
}

// key = (distance bits << 32) | tag ; distances are >= 0 so unsigned order = value order
__device__ __forceinline__ uint64_t make_key(float dist, uint32_t tag) {
  return (static_cast<uint64_t>(__float_as_uint(dist)) << 32) | tag;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    uint64_t o = __shfl_xor(v, off, WAVE);
    v = o < v ? o : v;
  }
  return v;
}

// 64-bit unsigned minimum over the wave, result in every lane, without the LDS crossbar: four DPP
// butterfly steps inside each row of 16 lanes (ALU latency instead of a ds_bpermute round trip
// per step), then the four row results through SGPRs.  Used where the reduction sits on a serial
// critical path (K6); wave_min_u64 above is fine where many waves overlap.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_min_step(uint64_t v) {
  const uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
  const uint32_t olo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(lo), static_cast<int>(lo), CTRL, 0xF, 0xF, false));
  const uint32_t ohi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(hi), static_cast<int>(hi), CTRL, 0xF, 0xF, false));
  const uint64_t o = (static_cast<uint64_t>(ohi) << 32) | olo;
  return o < v ? o : v;
}
__device__ __forceinline__ uint64_t wave_min_u64_dpp(uint64_t v) {
  v = dpp_min_step<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_min_step<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_min_step<0x141>(v);    // row_half_mirror
  v = dpp_min_step<0x140>(v);    // row_mirror: all 16 lanes of a row now hold the row minimum
  uint64_t r[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v)), 16 * k));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(v >> 32)), 16 * k));
    r[k] = (static_cast<uint64_t>(hi) << 32) | lo;
  }
  const uint64_t a = r[0] < r[1] ? r[0] : r[1], b = r[2] < r[3] ? r[2] : r[3];
  return a < b ? a : b;
}

// ---- lattice distance, squared, exactly as the reference forms it before its sqrt
// hexa_dist som_rout.c:438-451, rect_dist :461-464.  The sqrt itself is folded into
// a host-computed threshold (bubble) or taken in double (gaussian).
__device__ __forceinline__ float lattice_sq(int topol, int bx, int by, int tx, int ty) {
  float dx = static_cast<float>(bx - tx);
  float dy = static_cast<float>(by - ty);
  if (topol == 4 /*rect*/) {
    float r = dx * dx;
    float r2 = dy * dy;
    return r + r2;
  }
  if (((by - ty) % 2) != 0) {
    dx = ((by % 2) == 0) ? static_cast<float>(static_cast<double>(dx) - 0.5)
                         : static_cast<float>(static_cast<double>(dx) + 0.5);
  }
  float r = dx * dx;
  double t = 0.75 * static_cast<double>(dy);
  t = t * static_cast<double>(dy);
  return static_cast<float>(static_cast<double>(r) + t);
}

// The same value without fp64, valid when both map sides are <= 1024: every intermediate
// (dx +- 0.5, dx^2, 0.75 dy^2, their sum) is then a multiple of 0.25 below 2^22 and exactly
// representable in fp32, so the reference's mixed float/double expression and this one
// round nowhere and agree bit for bit.
__device__ __forceinline__ float lattice_sq_small(int topol, int bx, int by, int tx, int ty) {
  float dx = static_cast<float>(bx - tx);
  const float dy = static_cast<float>(by - ty);
  if (topol == 4 /*rect*/) return dx * dx + dy * dy;
  if ((by - ty) & 1) dx += (by & 1) ? 0.5f : -0.5f;
  return dx * dx + 0.75f * (dy * dy);
}

// gaussian_adapt's factor, som_rout.c:539-542
__device__ __forceinline__ float gaussian_alpha(float lat_sq, float radius, float alpha) {
  float dd = static_cast<float>(sqrt(static_cast<double>(lat_sq)));
  float neg = -dd * dd;
  double den = 2.0 * static_cast<double>(radius);
  den = den * static_cast<double>(radius);
  float h = static_cast<float>(exp(static_cast<double>(neg) / den));
  return alpha * h;
}

// per-iteration scalars, computed on the host with the reference's own expressions
struct StepScalars {
  float alpha;     // talp after schedule (+ weights), som_rout.c:617-624
  float thresh;    // bubble: largest lattice_sq value still inside the radius; gaussian: trad
  int32_t fixed;   // >= 0: unit index from the sample's fixed point (som_rout.c:628-632)
  int32_t reach;   // >= 0: how many lattice rows the neighbourhood can span (conservative);
                   // -1: every component masked -> no search, no update (som_rout.c:635-640)
};

// =====================================================================================
// K-layout: row-major host rows <-> row-group tiles
// =====================================================================================
__global__ void k_rows_to_tiles(const float *__restrict__ rows, CbView cb) {
  int64_t g = blockIdx.x;
  int lane = threadIdx.x & 63;
  int64_t row = g * WAVE + lane;
  for (int q = threadIdx.x >> 6; q < cb.d4; q += blockDim.x >> 6) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int i = q * 4 + j;
      v[j] = (row < cb.n && i < cb.d) ? rows[(static_cast<int64_t>(unit_of_row(cb, row)) - cb.row_offset) * cb.d + i] : 0.0f;
    }
    *tile_ptr_w(cb, g, q, lane) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
__global__ void k_tiles_to_rows(float *__restrict__ rows, CbView cb) {
  int64_t g = blockIdx.x;
  int lane = threadIdx.x & 63;
  int64_t row = g * WAVE + lane;
  if (row >= cb.n) return;
  for (int q = threadIdx.x >> 6; q < cb.d4; q += blockDim.x >> 6) {
    float4 v = *tile_ptr(cb, g, q, lane);
    float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int i = q * 4 + j;
      if (i < cb.d) rows[(static_cast<int64_t>(unit_of_row(cb, row)) - cb.row_offset) * cb.d + i] = a[j];
    }
  }
}

// =====================================================================================
// K-pack: a run of samples, row-major [count][d] (wrapping inside the data set) ->
// sample tiles xt[sb][q][S][4]: for every block of S samples and every chunk of 4
// dims the S float4 are contiguous, so the scan kernel reads a sample tile with
// wave-uniform (scalar) loads.  Samples >= count and dims >= d are zero.
// =====================================================================================
template <int S>
__global__ void k_pack_samples(const float *__restrict__ rows, int64_t n_rows, int d, int d4,
                               int64_t first, int64_t count, float4 *__restrict__ xt) {
  int64_t sb = blockIdx.x;
  for (int e = threadIdx.x; e < d4 * S; e += blockDim.x) {
    int q = e / S, s = e % S;
    int64_t smp = sb * S + s;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (smp < count) {
      int64_t r = (first + smp) % n_rows;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int i = q * 4 + j;
        if (i < d) v[j] = rows[r * d + i];
      }
    }
    xt[(sb * d4 + q) * S + s] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

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
};
__global__ void k_sample_tau(const float *__restrict__ rows, int64_t n_rows, int d, int64_t first,
                             int64_t count, const unsigned int *__restrict__ cn_max_bits,
                             double err_coeff, float *__restrict__ tau, RerankInit init) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (init.gmin) {                                       // 64 * count threads >= bpad + 4 * ncols
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < count && init.keys) init.keys[t] = KEY_NONE;
    if (t < init.bpad) { init.gmin[t] = 0xFFFFFFFFu; init.gmin[init.bpad + t] = 0u; }
    if (t < 4 * static_cast<int64_t>(init.ncols)) init.gmin[2 * init.bpad + t] = 0u;
    if (t == 0) *init.pair_count = 0u;
  }
  if (b >= count) return;
  const float *x = rows + ((first + b) % n_rows) * d;
  double acc = 0.0;
  for (int i = lane; i < d; i += WAVE) { double v = x[i]; acc += v * v; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane == 0) {
    // err_coeff = the host's bound on |s~ + ||x||^2 - d| / (||x|| + ||c||)^2 for the GEMM in use
    const double u = 5.9604644775390625e-08;                         // 2^-24
    const double cmax = sqrt(static_cast<double>(__uint_as_float(*cn_max_bits)) * (1.0 + 4.0 * d * u));
    const double s = sqrt(acc) + cmax;
    const double t = 2.0 * err_coeff * s * s * 1.001;
    float tf = static_cast<float>(t);
    if (static_cast<double>(tf) < t) tf = __uint_as_float(__float_as_uint(tf) + 1);   // round up
    tau[b] = tf;
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
// error coefficient tau is built from (somhip.hip prefilter_err_coeff), so the exact
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

// squared norms (fp32, as k_row_norms) + bf16 hi/lo tiles of the codebook, one pass
__global__ void k_prep_codes_bf16(CbView cb, int d8, float *__restrict__ cn,
                                  unsigned int *__restrict__ cn_max_bits, uint4 *__restrict__ chi,
                                  uint4 *__restrict__ clo) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (g >= cb.ngroups) return;
  float acc = 0.0f;
  for (int kb = 0; kb < d8; kb++) {
    const float4 a = *tile_ptr(cb, g, 2 * kb, lane);
    const float4 b = (2 * kb + 1 < cb.d4) ? *tile_ptr(cb, g, 2 * kb + 1, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc += a.x * a.x; acc += a.y * a.y; acc += a.z * a.z; acc += a.w * a.w;
    acc += b.x * b.x; acc += b.y * b.y; acc += b.z * b.z; acc += b.w * b.w;
    uint4 hi, lo;
    split8(a, b, hi, lo);
    chi[(g * d8 + kb) * WAVE + lane] = hi;
    clo[(g * d8 + kb) * WAVE + lane] = lo;
  }
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
                                                        int nlist, float *__restrict__ cn,
                                                        uint4 *__restrict__ chi, uint4 *__restrict__ clo) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= nlist) return;
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

// a run of samples -> bf16 hi/lo sample tiles xt[sb][kb][32][8]
__global__ void k_pack_samples_bf16(const float *__restrict__ rows, int64_t n_rows, int d, int d8,
                                    int64_t first, int64_t count, uint4 *__restrict__ xhi,
                                    uint4 *__restrict__ xlo, unsigned int *__restrict__ zero_word) {
  const int64_t sb = blockIdx.x;
  if (zero_word && sb == 0 && threadIdx.x == 0) *zero_word = 0u;   // max ||c||^2 accumulator of the next kernel
  for (int e = threadIdx.x; e < d8 * 32; e += blockDim.x) {
    const int kb = e / 32, sidx = e % 32;
    const int64_t smp = sb * 32 + sidx;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (smp < count) {
      const float *x = rows + ((first + smp) % n_rows) * d;
#pragma unroll
      for (int j = 0; j < 8; j++) if (kb * 8 + j < d) v[j] = x[kb * 8 + j];
    }
    uint4 hi, lo;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
    xhi[(sb * d8 + kb) * 32 + sidx] = hi;
    xlo[(sb * d8 + kb) * 32 + sidx] = lo;
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
// K1m: masked variant, one sample per launch column (rare path: data with 'x'
// components, lvq_pak.c:65-69).  mask is wave-uniform per component.
// =====================================================================================
__global__ __launch_bounds__(256) void k_scan_masked(CbView cb, const float *__restrict__ rows,
                                                     const uint8_t *__restrict__ mask,
                                                     int64_t n_rows, int64_t first, int64_t count,
                                                     int tie_knn, uint64_t *__restrict__ keys) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t smp = blockIdx.y;
  const int64_t r = (first + smp) % n_rows;
  const float *x = rows + r * cb.d;
  const uint8_t *m = mask + r * cb.d;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (g >= cb.ngroups) return;
  float acc = 0.0f;
  for (int q = 0; q < cb.d4; q++) {
    float4 c = *tile_ptr(cb, g, q, lane);
    float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int i = q * 4 + j;
      if (i < cb.d && m[i] == 0) acc = sq_acc(acc, cc[j], x[i]);
    }
  }
  int64_t row = g * WAVE + lane;
  uint32_t grow = unit_of_row(cb, row);
  uint64_t k = row < cb.n ? make_key(acc, tie_knn ? ~grow : grow) : KEY_NONE;
  k = wave_min_u64(k);
  if (lane == 0)
    atomicMin(reinterpret_cast<unsigned long long *>(keys + smp), static_cast<unsigned long long>(k));
}

template <bool VEC>
__device__ __forceinline__ float4 load_x4(const float *__restrict__ xr, int q, int d) {
  if (VEC) return reinterpret_cast<const float4 *>(xr)[q];     // wave-uniform
  float4 x;
  x.x = q * 4 + 0 < d ? xr[q * 4 + 0] : 0.f;
  x.y = q * 4 + 1 < d ? xr[q * 4 + 1] : 0.f;
  x.z = q * 4 + 2 < d ? xr[q * 4 + 2] : 0.f;
  x.w = q * 4 + 3 < d ? xr[q * 4 + 3] : 0.f;
  return x;
}

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
    uint32_t widx = 0xFFFFFFFFu;
    if (s.fixed >= 0) widx = static_cast<uint32_t>(s.fixed);
    else {
      uint64_t k = keys[b];
      if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) widx = static_cast<uint32_t>(k);
    }
    if (widx != 0xFFFFFFFFu) o = make_int2(static_cast<int>(widx % static_cast<uint32_t>(xdim)),
                                           static_cast<int>(widx / static_cast<uint32_t>(xdim)));
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

template <bool GAUSS>
__global__ __launch_bounds__(256) void k_som_members(CbView cb, int64_t count,
                                                     const int2 *__restrict__ bxy,
                                                     const uint64_t *__restrict__ keys,
                                                     const StepScalars *__restrict__ sc,
                                                     uint32_t *__restrict__ cnt,
                                                     MemberEntry *__restrict__ ent,
                                                     unsigned long long *__restrict__ stats) {
  __shared__ uint32_t s_wcount[4];
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
  MemberEntry *out = ent + g * count;
  uint32_t base = 0;
  unsigned long long rows_total = 0, pairs_total = 0;

  constexpr int RR = 4;                 // samples per thread and trip: their loads are issued together
  for (int64_t b0 = 0; b0 < count; b0 += 256 * RR) {
    unsigned long long mm[RR];
    float al[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
      const int64_t b = b0 + 256 * r + tid;
      unsigned long long m = 0;
      float alpha_b = 0.f;
      if (b < count) {
      const StepScalars s = sc[b];
      int2 w;
      if (keys) {                                        // winners decoded here (K4a's rule), no extra launch
        w = make_int2(-1, -1);
        if (s.reach >= 0) {
          uint32_t widx = 0xFFFFFFFFu;
          if (s.fixed >= 0) widx = static_cast<uint32_t>(s.fixed);
          else { const uint64_t k = keys[b]; if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) widx = static_cast<uint32_t>(k); }
          if (widx != 0xFFFFFFFFu) w = make_int2(static_cast<int>(widx % xdim), static_cast<int>(widx / xdim));
        }
      } else {
        w = bxy[b];
      }
      alpha_b = s.alpha;
      // reach (rows) >= radius/0.866 + 1 also bounds the x extent (unit spacing 1, half-unit shifts)
      if (w.x >= 0 && w.y + s.reach >= g_ty0 && w.y - s.reach <= g_ty1 &&
          (GAUSS || (w.x + s.reach >= g_txa && w.x - s.reach <= g_tx1))) {
        if (GAUSS) m = live_mask;
        else if (cb.patch_w && small_map) {
          // 8x8 patch, exact integer form: with every lattice quantity a multiple of 1/4,
          //   lattice_sq <= thresh  <=>  (2dx)^2 + 3 dy^2 <= floor(4 thresh)   (hexa)
          //                              dx^2 + dy^2     <= floor(thresh)     (rect)
          // and in one lattice row the members are a contiguous run of tx.
          const bool rect = cb.topol == 4;
          const int K = static_cast<int>(floor(static_cast<double>(s.thresh) * (rect ? 1.0 : 4.0)));
          if (K >= 0) {
#pragma unroll
            for (int iy = 0; iy < 8; iy++) {
              const int ty = g_ty0 + iy, dy = w.y - ty;
              const int rem = K - (rect ? dy * dy : 3 * dy * dy);
              if (rem < 0) continue;
              int W = static_cast<int>(sqrtf(static_cast<float>(rem)));      // integer sqrt, corrected
              while ((W + 1) * (W + 1) <= rem) W++;
              while (W * W > rem) W--;
              // rect: |bx - tx| <= W.  hexa: |2(bx - tx) + o| <= W, o = 0 on same-parity rows,
              // -1 when by is even, +1 when by is odd (som_rout.c:440-447)
              int lo, hi;
              if (rect) { lo = w.x - W; hi = w.x + W; }
              else {
                const int o = (dy & 1) ? ((w.y & 1) ? 1 : -1) : 0;
                // 2 bx + o - W <= 2 tx <= 2 bx + o + W
                const int a = 2 * w.x + o - W, b = 2 * w.x + o + W;
                lo = (a + (a >= 0 ? 1 : 0)) / 2; if (2 * lo < a) lo++;      // ceil(a / 2)
                hi = b >= 0 ? b / 2 : -((-b + 1) / 2);                       // floor(b / 2)
              }
              lo = lo < g_tx0 ? g_tx0 : lo;
              hi = hi > g_tx0 + 7 ? g_tx0 + 7 : hi;
              if (lo <= hi) {
                const unsigned long long run = ((1ull << (hi - lo + 1)) - 1) << (lo - g_tx0);
                m |= run << (8 * iy);
              }
            }
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
#pragma unroll
    for (int r = 0; r < RR; r++) {
      if (b0 + 256 * r >= count) break;                 // uniform
      const int64_t b = b0 + 256 * r + tid;
      const unsigned long long m = mm[r];
      const bool on = m != 0;
      const unsigned long long bal = __ballot(on);
      if (lane == 0) s_wcount[wave] = __popcll(bal);
      __syncthreads();
      uint32_t off = base;
      for (int w2 = 0; w2 < wave; w2++) off += s_wcount[w2];
      if (on) {
        MemberEntry e;
        e.sample = static_cast<uint32_t>(b); e.alpha = al[r]; e.mask = m;
        out[off + __popcll(bal & ((1ull << lane) - 1))] = e;
        rows_total += __popcll(m);
        pairs_total += 1;
      }
      base += s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
      __syncthreads();
    }
  }
  if (tid == 0) cnt[g] = base;
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
template <int QW, int TB, bool GAUSS, bool MASKED>
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
  __shared__ float4 xs[TB][BQ];
  __shared__ uint32_t ms[MASKED ? TB : 1][MASKED ? BQ : 1];   // 4 mask bits per chunk
  __shared__ float s_ga[GAUSS ? TB : 1][GAUSS ? WAVE : 1];    // gaussian: per-lane alpha
  __shared__ unsigned long long s_mask[TB];
  __shared__ long long s_xoff[TB];
  __shared__ float s_alpha[TB], s_thr[TB];
  __shared__ int s_bx[TB], s_by[TB];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t g = order ? order[blockIdx.x] : blockIdx.x;     // heaviest groups first
  const uint32_t n_ent = cnt[g];
  if (n_ent == 0) return;                             // nothing in this run touches the group
  const int qblk = blockIdx.y * BQ;
  const int q0 = qblk + wave * QW;
  const bool vec = (cb.d & 3) == 0;
  const MemberEntry *list = ent + g * count;
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
    if (q0 < cb.d4) {
      for (int i = 0; i < tb; i++) {
        if ((s_mask[i] >> lane) & 1ull) {
          const float a = GAUSS ? s_ga[i][lane] : s_alpha[i];
#pragma unroll
          for (int j = 0; j < QW; j++) {
            const float4 n = adapt4(c[j], xs[i][wave * QW + j], a);
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
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < QW; j++)
    if (q0 + j < cb.d4) *tile_ptr_w(cb, g, q0 + j, lane) = c[j];
}

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
    uint32_t widx = 0xFFFFFFFFu;
    if (s.reach >= 0) {
      if (s.fixed >= 0) widx = static_cast<uint32_t>(s.fixed);
      else {
        uint64_t k = *prev_slot;
        if (static_cast<uint32_t>(k >> 32) < FLT_MAX_BITS) widx = static_cast<uint32_t>(k);
      }
    }
    if (widx != 0xFFFFFFFFu) {
      int tx, ty;
      txty_of_row(cb, row, tx, ty);
      const int bx = static_cast<int>(widx % xdim), by = static_cast<int>(widx / xdim);
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
    const bool q = gl < cb.ngroups && wmin[gl * bpad + b] <= thr;
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
                                                     uint64_t *__restrict__ keys_out) {
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

template <int K>
__global__ __launch_bounds__(256) void k_topk_select(CbView cb, int64_t count, int64_t bpad,
                                                     const float *__restrict__ wmin,
                                                     const float *__restrict__ tau, uint32_t cap,
                                                     uint2 *__restrict__ pairs, TopkSpan *__restrict__ span,
                                                     uint32_t *__restrict__ counter /* [0] fill, [1] overflow */) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= count) return;
  float mine[K];
#pragma unroll
  for (int t = 0; t < K; t++) mine[t] = 3.4e38f;
  for (int64_t g = lane; g < cb.ngroups; g += WAVE) {
    float v = wmin[g * bpad + b];
#pragma unroll
    for (int t = 0; t < K; t++) {
      const float lo = fminf(mine[t], v);
      v = fmaxf(mine[t], v);
      mine[t] = lo;
    }
  }
  float mk = 3.4e38f;
  for (int t = 0; t < K; t++) {
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
  const float thr = (mk >= 3.0e38f) ? 3.4e38f : mk + tau[b];
  uint32_t total = 0;
  for (int64_t gb = 0; gb < cb.ngroups; gb += WAVE) {
    const int64_t gl = gb + lane;
    total += __popcll(__ballot(gl < cb.ngroups && wmin[gl * bpad + b] <= thr));
  }
  uint32_t start = 0;
  if (lane == 0) {
    start = atomicAdd(counter, total);
    if (start + total > cap) atomicMax(counter + 1, 1u);
    span[b].start = start; span[b].n = total;
  }
  start = __shfl(start, 0, WAVE);
  if (start + total > cap) return;
  uint32_t at = start;
  for (int64_t gb = 0; gb < cb.ngroups; gb += WAVE) {
    const int64_t gl = gb + lane;
    const bool q = gl < cb.ngroups && wmin[gl * bpad + b] <= thr;
    const unsigned long long ball = __ballot(q);
    if (q) pairs[at + __popcll(ball & ((1ull << lane) - 1))] = make_uint2(static_cast<uint32_t>(b), static_cast<uint32_t>(gl));
    at += __popcll(ball);
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
                                                    uint64_t *__restrict__ keys_out) {
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
                                                       unsigned long long *__restrict__ stats) {
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
  const float thr = live ? ordered_to_float(gmin[b]) + tau[b] : -3.4e38f;
  unsigned ngr = 0, nrow = 0;
  for (int64_t g0 = g_lo; g0 < g_hi; g0 += 8) {
    const int64_t g = g0 + gy;
    unsigned long long mask = 0;
    if (live && g < g_hi && wmin[g * bpad + b] <= thr) {
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
    const unsigned n = __popcll(mask);
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
    base = __shfl(base, 0, WAVE) + pre - n;
    if (n) {
      ngr++; nrow += n;
      unsigned at = base;
      unsigned long long mm = mask;
      uint2 *seg = pairs + static_cast<size_t>(blockIdx.x) * cap_col;
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
                                                      unsigned long long *__restrict__ stats) {
  // A fixed, small grid (workgroup launches cost ~50 ns each: a grid sized for the worst case was
  // the whole cost of this kernel).  Every workgroup builds the same table of 256-entry chunks per
  // column (prefix sums of the segment fills) and takes chunks round-robin.
  __shared__ uint32_t s_pref[PAIR_MAX_COLS + 1];
  __shared__ uint32_t s_scan[256];
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
  for (int c = tid * per; c < (tid + 1) * per && c < ncols; c++) mine += (col_count[c] + 255u) >> 8;
  s_scan[tid] = mine;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t v = tid >= off ? s_scan[tid - off] : 0u;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  {
    uint32_t run = s_scan[tid] - mine;                   // exclusive prefix of this thread's columns
    for (int c = tid * per; c < (tid + 1) * per && c < ncols; c++) { s_pref[c] = run; run += (col_count[c] + 255u) >> 8; }
    if (tid == 255) s_pref[ncols] = s_scan[255];
  }
  __syncthreads();
  const uint32_t total = s_pref[ncols];
  const bool vec = (cb.d & 3) == 0;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    int lo = 0, hi = ncols;                              // column with s_pref[col] <= id < s_pref[col + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[mid] <= id) lo = mid; else hi = mid; }
    const uint32_t slot = (id - s_pref[lo]) * 256u + tid;
    if (slot >= col_count[lo]) continue;
    const uint2 pr = pairs[static_cast<size_t>(lo) * cap_col + slot];
    const int64_t row = pr.y;
    const float4 *crow = reinterpret_cast<const float4 *>(cb.tiles) + ((row >> 6) * cb.d4) * WAVE + (row & 63);
    const float *x = rows + ((first + pr.x) % n_rows) * cb.d;
    float acc = 0.0f;
    // 8 chunks of the row and of the sample in flight per lane (independent loads first)
    constexpr int UP = 8;
    int q = 0;
    for (; q + UP <= cb.d4; q += UP) {
      float4 cc[UP], xx[UP];
#pragma unroll
      for (int u = 0; u < UP; u++) {
        cc[u] = crow[static_cast<int64_t>(q + u) * WAVE];
        xx[u] = vec ? reinterpret_cast<const float4 *>(x)[q + u] : load_x4<false>(x, q + u, cb.d);
      }
#pragma unroll
      for (int u = 0; u < UP; u++) {
        acc = sq_acc(acc, cc[u].x, xx[u].x);
        acc = sq_acc(acc, cc[u].y, xx[u].y);
        acc = sq_acc(acc, cc[u].z, xx[u].z);
        acc = sq_acc(acc, cc[u].w, xx[u].w);
      }
    }
    for (; q < cb.d4; q++) {
      const float4 c = crow[static_cast<int64_t>(q) * WAVE];
      const float4 xv = vec ? reinterpret_cast<const float4 *>(x)[q] : load_x4<false>(x, q, cb.d);
      acc = sq_acc(acc, c.x, xv.x);
      acc = sq_acc(acc, c.y, xv.y);
      acc = sq_acc(acc, c.z, xv.z);
      acc = sq_acc(acc, c.w, xv.w);
    }
    const uint64_t k = make_key(acc, unit_of_row(cb, row));
    atomicMin(reinterpret_cast<unsigned long long *>(keys + pr.x), static_cast<unsigned long long>(k));
  }
}

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

// =====================================================================================
// K7: find_qerror2 (som_rout.c:823-885): per sample, the neighbourhood-weighted sum
//   bubble_qerror   :734-772   q = sum over units u with mapdist(bmu, u) <= radius of d_u * d_u
//   gaussian_qerror :775-818   q = sum over all units of (exp(-dd^2 / 2 radius^2) * d_u) * d_u
// with d_u = vector_dist_euc(code_u, sample) (lvq_pak.c:291-316: fp32 sum in dim order, masked
// components skipped, (float)sqrt((double)sum)), accumulated in fp32 IN UNIT ORDER.  One
// workgroup per sample; a chunk of 256 consecutive units is evaluated in parallel (thread =
// unit), then one thread adds the chunk's terms in order.  The host adds the per-sample sums in
// data order (the reference's outer float accumulator).
// =====================================================================================
template <bool GAUSS>
__global__ __launch_bounds__(256) void k_qerror2(CbView cb, int ydim, const float *__restrict__ rows,
                                                 const uint8_t *__restrict__ mask, int64_t n_rows,
                                                 int64_t first, const uint64_t *__restrict__ keys,
                                                 float radius, float thresh, int reach,
                                                 float *__restrict__ out) {
  extern __shared__ float q2_dyn[];
  float *s_x = q2_dyn;                                   // [d]
  uint8_t *s_mk = reinterpret_cast<uint8_t *>(q2_dyn + cb.d);   // [d]
  __shared__ float s_term[256];
  __shared__ uint8_t s_on[256];
  const int tid = threadIdx.x;
  const int64_t smp = blockIdx.x;
  const uint64_t key = keys[smp];
  if (static_cast<uint32_t>(key >> 32) >= FLT_MAX_BITS) { if (tid == 0) out[smp] = 0.0f; return; }
  const int64_t r = (first + smp) % n_rows;
  for (int i = tid; i < cb.d; i += 256) {
    s_x[i] = rows[r * cb.d + i];
    s_mk[i] = mask ? mask[r * cb.d + i] : 0;
  }
  const uint32_t widx = static_cast<uint32_t>(key);
  const int xdim = cb.xdim;
  const int bx = static_cast<int>(widx % static_cast<uint32_t>(xdim)), by = static_cast<int>(widx / static_cast<uint32_t>(xdim));
  int64_t u_lo = cb.row_offset, u_hi = cb.row_offset + cb.n;
  if (!GAUSS) {
    const int y0 = by - reach < 0 ? 0 : by - reach, y1 = by + reach + 1 > ydim ? ydim : by + reach + 1;
    const int64_t lo = static_cast<int64_t>(y0) * xdim, hi = static_cast<int64_t>(y1) * xdim;
    u_lo = lo > u_lo ? lo : u_lo;
    u_hi = hi < u_hi ? hi : u_hi;
  }
  __syncthreads();
  float q = 0.0f;
  for (int64_t base = u_lo; base < u_hi; base += 256) {
    const int64_t u = base + tid;
    bool on = false;
    float term = 0.0f;
    if (u < u_hi) {
      const int tx = static_cast<int>(u % xdim), ty = static_cast<int>(u / xdim);
      const float lsq = lattice_sq(cb.topol, bx, by, tx, ty);
      on = GAUSS || lsq <= thresh;
      if (on) {
        const int64_t row = row_of_unit(cb, static_cast<uint32_t>(u));
        const int64_t g = row >> 6;
        const int lane = static_cast<int>(row & 63);
        float acc = 0.0f;
        for (int qd = 0; qd < cb.d4; qd++) {
          const float4 c = *tile_ptr(cb, g, qd, lane);
          const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int i = qd * 4 + k;
            if (i < cb.d && s_mk[i] == 0) acc = sq_acc(acc, cc[k], s_x[i]);
          }
        }
        const float dv = static_cast<float>(sqrt(static_cast<double>(acc)));
        if (GAUSS) {
          const float h = gaussian_alpha(lsq, radius, 1.0f);     // 1.0f * h == h
          const float t = h * dv;
          term = t * dv;
        } else {
          term = dv * dv;
        }
      }
    }
    s_term[tid] = term;
    s_on[tid] = on ? 1 : 0;
    __syncthreads();
    if (tid == 0) {
      const int lim = static_cast<int>(u_hi - base < 256 ? u_hi - base : 256);
      for (int i = 0; i < lim; i++)
        if (s_on[i]) q = q + s_term[i];
    }
    __syncthreads();
  }
  if (tid == 0) out[smp] = q;
}

// =====================================================================================
// K8: the two data passes of lininit's find_eigenvectors (som_rout.c:211-289), exactly:
//   column sums    m[i]   += x[r][i]                       over unmasked components, rows in order
//   centred sums   R[i][j] += (x[r][i] - m[i]) * (x[r][j] - m[j])   for j >= i, rows in order
// Every output element is its own fp32 chain over the rows, so elements are the parallel axis
// (131 328 chains at dim 512) and nothing is re-associated.  K8b: a workgroup owns a 16x16
// block of (i, j); 64 rows at a time are centred once (x - m, one rounding, as the reference
// forms it) into LDS, then each thread runs mul + add down the 64 rows of its pair.
// =====================================================================================
__global__ __launch_bounds__(256) void k_column_sums(const float *__restrict__ rows, const uint8_t *__restrict__ mask,
                                                     int64_t n, int d, float *__restrict__ sum,
                                                     unsigned long long *__restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  float acc = 0.0f;
  unsigned long long k = 0;
  for (int64_t r = 0; r < n; r++) {
    if (!mask || mask[r * d + i] == 0) { acc = acc + rows[r * d + i]; k++; }
  }
  sum[i] = acc;
  cnt[i] = k;
}

__global__ __launch_bounds__(256) void k_centered_products(const float *__restrict__ rows,
                                                           const uint8_t *__restrict__ mask, int64_t n, int d,
                                                           const float *__restrict__ mean, float *__restrict__ R) {
  constexpr int RB = 64;
  __shared__ float s_i[RB][16], s_j[RB][16];
  __shared__ uint8_t s_mi[RB][16], s_mj[RB][16];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;                                   // only j >= i is ever read (som_rout.c:287-289)
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int i = bi * 16 + ti, j = bj * 16 + tj;
  float acc = 0.0f;
  for (int64_t r0 = 0; r0 < n; r0 += RB) {
    // stage 64 rows x (16 i-columns + 16 j-columns), centred
    for (int e = tid; e < RB * 32; e += 256) {
      const int rr = e >> 5, cc = e & 31;
      const int col = cc < 16 ? bi * 16 + cc : bj * 16 + (cc - 16);
      const int64_t r = r0 + rr;
      float v = 0.0f;
      uint8_t mk = 1;
      if (r < n && col < d) {
        mk = mask ? mask[r * d + col] : 0;
        v = rows[r * d + col] - mean[col];
      }
      if (cc < 16) { s_i[rr][cc] = v; s_mi[rr][cc] = mk; } else { s_j[rr][cc - 16] = v; s_mj[rr][cc - 16] = mk; }
    }
    __syncthreads();
    const int lim = static_cast<int>(n - r0 < RB ? n - r0 : RB);
    for (int rr = 0; rr < lim; rr++) {
      if (s_mi[rr][ti] == 0 && s_mj[rr][tj] == 0) {
        const float p = s_i[rr][ti] * s_j[rr][tj];
        acc = acc + p;
      }
    }
    __syncthreads();
  }
  if (i < d && j < d && j >= i) R[static_cast<int64_t>(i) * d + j] = acc;
}

// keys handed to a host-side collective: signed 64-bit MIN must order them like unsigned MIN, so
// the all-ones "no winner" key becomes INT64_MAX (still >= FLT_MAX in its distance half)
__global__ void k_clamp_keys(uint64_t *__restrict__ keys, int64_t n) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && (keys[i] >> 63)) keys[i] = 0x7FFFFFFFFFFFFFFFull;
}

__global__ void k_fill_u64(uint64_t *p, int64_t n, uint64_t v) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace somhip
