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

// ---- seeded Gaussian-mixture stream on the device (the host form: som_lvq_pak_amd/host/paklib.c pak_gen_row;
// SURVEY 8(d)).  Counter-based and integer up to one exact division, so both sides give the same bits.
__host__ __device__ __forceinline__ uint64_t gen_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ float gen_z(uint64_t seed, uint64_t counter) {
  int32_t sum = 0;
  for (int w = 0; w < 3; w++) {
    const uint64_t v = gen_splitmix64(seed ^ (3 * counter + w));
    sum += static_cast<int32_t>(v & 0xFFFF) + static_cast<int32_t>((v >> 16) & 0xFFFF) +
           static_cast<int32_t>((v >> 32) & 0xFFFF) + static_cast<int32_t>(v >> 48);
  }
  return static_cast<float>(sum - 6 * 65535) / 65536.0f;
}
// rows [row0, row0 + n) of the stream -> out[n][dim]; centre[n] (may be null) = mixture id of each row
__global__ __launch_bounds__(256) void k_gen_mixture(uint64_t seed, int k_centres, int dim, int64_t row0, int64_t n,
                                                     float *__restrict__ out, int32_t *__restrict__ centre) {
  const uint64_t seed_c = seed ^ 0xC3A5C85C97CB3127ull, seed_a = seed ^ 0xB492B66FBE98F273ull;
  const int64_t total = n * dim;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
       t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = t / dim, row = row0 + r;
    const int i = static_cast<int>(t - r * dim);
    const int k = static_cast<int>(gen_splitmix64(seed_a ^ static_cast<uint64_t>(row)) % static_cast<uint64_t>(k_centres));
    const float mu = 4.0f * gen_z(seed_c, static_cast<uint64_t>(k) * dim + i);
    out[t] = mu + gen_z(seed, static_cast<uint64_t>(row) * dim + i);
    if (centre && i == 0) centre[r] = k;
  }
}

}  // namespace somhip
