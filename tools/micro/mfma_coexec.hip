// Does a fp32 MFMA (v_mfma_f32_32x32x1_2b_f32) overlap with fp32 VALU work of the same / other waves?
// mode 0: 64 VALU (32 mul + 32 add) per iteration; mode 1: 1 MFMA per iteration; mode 2: both (VALU consumes the MFMA result)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float na) {
  f32x32 c, d;
  for (int i = 0; i < 32; i++) { c[i] = threadIdx.x * 0.001f + i; d[i] = 0.5f * i; }
  float b = threadIdx.x * 0.01f;
  for (int it = 0; it < iters; it++) {
    if (MODE >= 1) d = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, b, c, 0, 0, 0);
    if (MODE != 1) {
#pragma unroll
      for (int v = 0; v < 32; v++) { float s = na * d[v]; c[v] = c[v] + s; }
    } else {
      c = d;
    }
  }
  float s = 0;
  for (int i = 0; i < 32; i++) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wg = 256; wg <= 2048; wg *= 2)          // 1, 2, 4, 8 workgroups per CU -> 1, 2, 2(+), ... waves per SIMD
    for (int mode = 0; mode < 3; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<wg, 256>>>(out, iters, -0.05f);
        else if (mode == 1) k<1><<<wg, 256>>>(out, iters, -0.05f);
        else k<2><<<wg, 256>>>(out, iters, -0.05f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      // cycles per iteration per wave-slot at 2.4 GHz if the chip were exactly filled
      printf("workgroups %4d mode %d: %.3f ms  (%.0f ns per iteration)\n", wg, mode, best, best * 1e6 / iters);
    }
  return 0;
}
