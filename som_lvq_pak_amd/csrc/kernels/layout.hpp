// kernels/layout.hpp -- row-major rows <-> row-group tiles, sample tiles
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "common.hpp"

namespace somhip {

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
      v[j] = (row < cb.n && i < cb.d) ? rows[host_row_of_row(cb, row) * cb.d + i] : 0.0f;
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
      if (i < cb.d) rows[host_row_of_row(cb, row) * cb.d + i] = a[j];
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

}  // namespace somhip
