// kernels/scan_masked.hpp -- K1m: masked-sample scan, padded sample loads
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "prefilter_mfma.hpp"

namespace somhip {

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

}  // namespace somhip
