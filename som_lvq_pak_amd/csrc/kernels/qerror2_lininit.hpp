// kernels/qerror2_lininit.hpp -- K7/K8: find_qerror2, lininit data passes, small utilities
// (part of kernels.hpp; see the notes at the top of that file)
#pragma once
#include "lvq_batch.hpp"

namespace somhip {

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

// K8c: bounding box of the data (randinit_codes, som_rout.c:98-131): per component the smallest and largest
// unmasked value and how many there are.  min/max are order-independent, so rows are split over the grid and
// folded with integer atomics on an order-preserving image of the float (sign-magnitude -> biased unsigned).
__device__ __forceinline__ uint32_t f32_ordered(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(256) void k_column_minmax(const float *__restrict__ rows, const uint8_t *__restrict__ mask,
                                                       int64_t n, int d, int64_t rows_per_block,
                                                       uint32_t *__restrict__ omin, uint32_t *__restrict__ omax,
                                                       unsigned long long *__restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  unsigned long long k = 0;
  for (int64_t r = r0; r < r1; r++) {
    if (mask && mask[r * d + i]) continue;
    const uint32_t o = f32_ordered(rows[r * d + i]);
    lo = o < lo ? o : lo;
    hi = o > hi ? o : hi;
    k++;
  }
  if (k) {
    atomicMin(&omin[i], lo);
    atomicMax(&omax[i], hi);
    atomicAdd(&cnt[i], k);
  }
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
