// Prints the C/D layout of v_mfma_f32_32x32x1_2b_f32 on gfx950: which (block, row, col) each
// (lane, register) holds.  A[lane] = lane, B = 1  -> D = 32*block + row;  A = 1, B[lane] = lane -> D = 32*block + col.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void k(float *out) {
  const int lane = threadIdx.x;
  f32x32 z;
  for (int i = 0; i < 32; i++) z[i] = 0.f;
  f32x32 r = __builtin_amdgcn_mfma_f32_32x32x1f32((float)lane, 1.0f, z, 0, 0, 0);
  f32x32 c = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, (float)lane, z, 0, 0, 0);
  for (int v = 0; v < 32; v++) { out[(v * 64 + lane) * 2] = r[v]; out[(v * 64 + lane) * 2 + 1] = c[v]; }
}
int main() {
  float *d; hipMalloc(&d, 32 * 64 * 2 * 4);
  k<<<1, 64>>>(d);
  static float h[32 * 64 * 2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int v = 0; v < 32; v++) for (int l = 0; l < 64; l++) {
    int row = (int)h[(v * 64 + l) * 2], col = (int)h[(v * 64 + l) * 2 + 1];
    int b = v / 16, vv = v % 16;
    int erow = 32 * b + 8 * (vv / 4) + 4 * (l / 32) + (vv % 4), ecol = 32 * b + (l % 32);
    if (row != erow || col != ecol) { if (bad < 10) printf("v=%d lane=%d: row %d (exp %d) col %d (exp %d)\n", v, l, row, erow, col, ecol); bad++; }
  }
  printf("layout check: %d mismatches\n", bad);
  for (int v = 0; v < 32; v += 5) printf("v=%2d lane0 (%g,%g) lane31 (%g,%g) lane32 (%g,%g) lane63 (%g,%g)\n", v,
    h[(v*64+0)*2], h[(v*64+0)*2+1], h[(v*64+31)*2], h[(v*64+31)*2+1], h[(v*64+32)*2], h[(v*64+32)*2+1], h[(v*64+63)*2], h[(v*64+63)*2+1]);
  return 0;
}
