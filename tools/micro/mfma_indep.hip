// Do INDEPENDENT VALU / LDS-read / readlane instructions issued between MFMAs of the same wave run in the shadow of the
// matrix pipe?  First table: fp32 MFMAs (v_mfma_f32_32x32x2_f32); second: bf16 (v_mfma_f32_32x32x16_bf16, -DBF16 build).  Per loop iteration: 8 MFMAs on 8 accumulators and NV other
// instructions, interleaved by sched_group_barrier.  KIND 0: v_mul on registers nobody else reads; 1: v_readlane +
// v_mul; 2: ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NM, int NV, int KIND>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float na) {
  __shared__ float4 lds[1024];
  f32x16 acc[8];
  for (int a = 0; a < 8; a++) for (int i = 0; i < 16; i++) acc[a][i] = threadIdx.x * 0.001f + i;
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  lds[threadIdx.x] = make_float4(v[0], v[1], v[2], v[3]);
  __syncthreads();
  const float a0 = threadIdx.x * 0.01f, b0 = threadIdx.x * 0.02f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(a0 + i); bb[i] = (__bf16)(b0 - i); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int m = 0; m < NM; m++) {
#ifdef BF16
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[m & 7], 0, 0, 0);
#else
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m & 7], 0, 0, 0);
#endif
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      if (KIND == 0) v[j & 7] = v[j & 7] * na;
      if (KIND == 1) v[j & 7] = v[j & 7] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[(j + 1) & 7]), j & 15));
      if (KIND == 2) { const float4 t = lds[(threadIdx.x + 16 * j + it) & 1023]; v[j & 7] += t.x + t.w; }
    }
    if constexpr (NM > 0 && NV > 0) {
      constexpr int PER = (NV + NM - 1) / NM;
#pragma unroll
      for (int m = 0; m < NM; m++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(KIND == 2 ? 0x100 : 0x002, PER * (KIND == 1 ? 2 : 1), 0);
        if (KIND == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2 * PER, 0);
      }
    }
  }
  float s = 0;
  for (int a = 0; a < 8; a++) for (int i = 0; i < 16; i++) s += acc[a][i];
  for (int i = 0; i < 8; i++) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV, int KIND> float run(float *out, int wg, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k<NM, NV, KIND><<<wg, 256>>>(out, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
#define ROW(NM, NV, KIND) printf("  MFMA %d + %2d x kind %d : %7.1f ns / iteration\n", NM, NV, KIND, run<NM, NV, KIND>(out, wg, iters) * 1e6 / iters)
int main() {
  float *out; hipMalloc(&out, 4 * 256 * 4096);
  const int iters = 4000;
  for (int wg = 256; wg <= 512; wg *= 2) {       // 1 or 2 workgroups of 4 waves per CU = 1 or 2 waves per SIMD
    printf("workgroups %d (%d wave(s) per SIMD)\n", wg, wg / 256);
    ROW(8, 0, 0); ROW(0, 32, 0); ROW(8, 16, 0); ROW(8, 32, 0); ROW(8, 64, 0);
    ROW(0, 16, 1); ROW(8, 16, 1); ROW(8, 32, 1);
    ROW(0, 8, 2); ROW(8, 8, 2); ROW(8, 16, 2);
  }
  return 0;
}
