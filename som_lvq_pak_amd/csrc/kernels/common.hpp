// kernels/common.hpp -- shared types and exact-arithmetic helpers (CbView, keys, lattice distances, step scalars)
// (part of kernels.hpp; see the notes at the top of that file)
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
  int patch_stride;     // interleaved shards (patch order only): local patch p is map patch
  int patch_phase;      //   p * patch_stride + patch_phase; a contiguous shard / whole map has (1, 0)
};

// global unit index (= the reference's row index, datafile.c:781,836) of local storage row `row`
__host__ __device__ __forceinline__ uint32_t unit_of_row(const CbView &cb, int64_t row) {
  if (cb.patch_w == 0) return static_cast<uint32_t>(row + cb.row_offset);
  const uint32_t p = static_cast<uint32_t>(row >> 6) * static_cast<uint32_t>(cb.patch_stride) + static_cast<uint32_t>(cb.patch_phase);
  const uint32_t i = static_cast<uint32_t>(row) & 63u;
  const uint32_t px = p % static_cast<uint32_t>(cb.patch_w), py = p / static_cast<uint32_t>(cb.patch_w);
  return static_cast<uint32_t>(cb.row_offset) + (py * 8 + (i >> 3)) * static_cast<uint32_t>(cb.xdim) + px * 8 + (i & 7);
}
// index, in the host's row array of this shard, of local storage row `row`: contiguous shards hand their
// units over in unit order, interleaved shards in storage order (somhip_shard_units lists the units)
__device__ __forceinline__ int64_t host_row_of_row(const CbView &cb, int64_t row) {
  return cb.patch_stride > 1 ? row : static_cast<int64_t>(unit_of_row(cb, row)) - cb.row_offset;
}
// lattice coordinates of local storage row `row` (som_rout.c:493-494: x = unit % xdim, y = unit / xdim)
__device__ __forceinline__ void txty_of_row(const CbView &cb, int64_t row, int &tx, int &ty) {
  const uint32_t xd = static_cast<uint32_t>(cb.xdim);
  if (cb.patch_w == 0) {
    const uint32_t u = static_cast<uint32_t>(row + cb.row_offset);
    tx = static_cast<int>(u % xd); ty = static_cast<int>(u / xd);
    return;
  }
  const uint32_t p = static_cast<uint32_t>(row >> 6) * static_cast<uint32_t>(cb.patch_stride) + static_cast<uint32_t>(cb.patch_phase);
  const uint32_t i = static_cast<uint32_t>(row) & 63u;
  const uint32_t px = p % static_cast<uint32_t>(cb.patch_w), py = p / static_cast<uint32_t>(cb.patch_w);
  tx = static_cast<int>(px * 8 + (i & 7));
  ty = static_cast<int>(static_cast<uint32_t>(cb.row_offset) / xd + py * 8 + (i >> 3));
}

// local storage row of global unit index `unit` (inverse of unit_of_row; the unit must belong to the shard)
__device__ __forceinline__ int64_t row_of_unit(const CbView &cb, uint32_t unit) {
  const uint32_t u = unit - static_cast<uint32_t>(cb.row_offset);
  if (cb.patch_w == 0) return static_cast<int64_t>(u);
  const uint32_t xd = static_cast<uint32_t>(cb.xdim);
  const uint32_t y = u / xd, x = u % xd;
  const uint32_t p = ((y >> 3) * static_cast<uint32_t>(cb.patch_w) + (x >> 3)) / static_cast<uint32_t>(cb.patch_stride);
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
  int32_t fixed;   // >= 0: the sample's fixed point (som_rout.c:628-632) as (yfix << 15) | xfix -- lattice
                   // coordinates, not a unit: the reference hands them to the neighbourhood function as they are,
                   // so a point beyond the map's edge teaches the units within the radius of it
  int32_t reach;   // >= 0: how many lattice rows the neighbourhood can span (conservative);
                   // -1: every component masked -> no search, no update (som_rout.c:635-640)
};

__host__ __device__ __forceinline__ int32_t fixed_pack(int fx, int fy) { return (fy << 15) | fx; }   // 0 <= fx, fy < 32768
__host__ __device__ __forceinline__ int fixed_x(int32_t f) { return f & 32767; }
__host__ __device__ __forceinline__ int fixed_y(int32_t f) { return f >> 15; }

}  // namespace somhip
