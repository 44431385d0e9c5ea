// Diagnostic: semantics (cbsz/abid A-broadcast) and rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
// Y[q][n] = sum_k X[q][k] W[n][k], q < 8, n < 64, K = 16.  rows 0-3 in lanes 0-3, rows 4-7 in lanes 4-7
__global__ void sem(const float* X, const float* W, float* Y) {
  const int lane = threadIdx.x;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  for (int k = 0; k < 16; ++k) {
    const float a = X[(lane & 7) * 16 + k];
    const float b = W[lane * 16 + k];
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a0, 4, 0, 0);   // cbsz=4: A of block 0 to all 16 blocks
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a1, 4, 1, 0);   // A of block 1 (lanes 4..7)
  }
  for (int i = 0; i < 4; ++i) { Y[i * 64 + lane] = a0[i]; Y[(4 + i) * 64 + lane] = a1[i]; }
}
template <int NACC>
__global__ void rate(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 0, 0);
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> int runrate(const char* tag) {
  float* d; CK(hipMalloc(&d, 256 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate<NACC><<<256, 256>>>(d, 10, 1.f, 2.f); CK(hipDeviceSynchronize());
  const int iters = 4000;
  CK(hipEventRecord(e0)); rate<NACC><<<256, 256>>>(d, iters, 1.f, 2.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double n = (double)iters * 16 * NACC;
  printf("%s acc=%d: %.2f ns per MFMA per SIMD (%.1f cyc @2.4GHz), chip %.1f TF\n", tag, NACC, ms * 1e6 / n, ms * 1e6 / n * 2.4,
         256.0 * 4 * n * 512 / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  float hX[8 * 16], hW[64 * 16], hY[8 * 64], *X, *W, *Y;
  for (int i = 0; i < 8 * 16; ++i) hX[i] = (float)((i * 7) % 13) - 6;
  for (int i = 0; i < 64 * 16; ++i) hW[i] = (float)((i * 5) % 11) - 5;
  CK(hipMalloc(&X, sizeof(hX))); CK(hipMalloc(&W, sizeof(hW))); CK(hipMalloc(&Y, sizeof(hY)));
  CK(hipMemcpy(X, hX, sizeof(hX), hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW, sizeof(hW), hipMemcpyHostToDevice));
  sem<<<1, 64>>>(X, W, Y); CK(hipMemcpy(hY, Y, sizeof(hY), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int q = 0; q < 8; ++q) for (int n = 0; n < 64; ++n) {
    float r = 0; for (int k = 0; k < 16; ++k) r += hX[q * 16 + k] * hW[n * 16 + k];
    if (r != hY[q * 64 + n]) { if (bad < 5) printf("mismatch q=%d n=%d want %g got %g\n", q, n, r, hY[q * 64 + n]); ++bad; }
  }
  printf("semantics: %d mismatches of 512\n", bad);
  runrate<1>("4x4x1"); runrate<2>("4x4x1"); runrate<4>("4x4x1"); runrate<8>("4x4x1");
  return 0;
}
