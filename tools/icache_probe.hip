// Diagnostic: cost of executing straight-line code larger than the instruction cache.
// Body = UNROLL x (4 MFMA 4x4x1, 8 bytes each); loop repeats the body.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
template <int UNROLL>
__global__ __launch_bounds__(256) void body(float* out, int iters, float a, float b) {
  f32x4 c0 = {0,0,0,0}, c1 = {0,0,0,0}, c2 = {0,0,0,0}, c3 = {0,0,0,0};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 4, 1, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 4, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 4, 1, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];
}
template <int UNROLL> int run(int blocks) {
  float* d; CK(hipMalloc(&d, 256 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long total_mfma = 1 << 20;                 // per wave
  const int iters = (int)(total_mfma / (4 * UNROLL));
  body<UNROLL><<<blocks, 256>>>(d, 2, 1.f, 2.f); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); body<UNROLL><<<blocks, 256>>>(d, iters, 1.f, 2.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("blocks=%3d body=%4d KB iters=%6d: %.2f ns per MFMA (%.1f cyc @2.4GHz)\n", blocks, UNROLL * 32 / 1024, iters,
         ms * 1e6 / ((double)iters * 4 * UNROLL), ms * 1e6 / ((double)iters * 4 * UNROLL) * 2.4);
  CK(hipFree(d));
  return 0;
}
int main() {
  for (int blocks : {57, 256}) {
    run<128>(blocks); run<512>(blocks); run<1024>(blocks); run<1536>(blocks); run<2048>(blocks); run<3072>(blocks); run<4096>(blocks);
  }
  return 0;
}
