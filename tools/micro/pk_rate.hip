// Microbenchmark: element throughput of v_add_f32/v_mul_f32 vs v_pk_add_f32/v_pk_mul_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a) {
  float c[16], x[16];
  for (int i = 0; i < 16; i++) { c[i] = threadIdx.x * 0.001f + i; x[i] = i * 0.5f; }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        float t;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c[i]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(t));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(c[i]) : "v"(c[i]), "v"(t));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 cc = {c[i], c[i + 1]}, xx = {x[i], x[i + 1]}, aa = {a, a}, t;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(xx), "v"(cc));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(aa), "v"(t));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(cc) : "v"(cc), "v"(t));
        c[i] = cc.x; c[i + 1] = cc.y;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    if (mode == 0) k<0><<<4096, 256>>>(out, iters, 0.05f); else k<1><<<4096, 256>>>(out, iters, 0.05f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 4096.0 * 256 * iters * 16 * 3;
    printf("mode %d: %.3f ms  %.1f TFLOP/s (elementwise sub,mul,add)\n", mode, ms, flop / ms * 1e-9);
  }
  float h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); printf("%g\n", h[1]);
  return 0;
}
