// Diagnostic: sustained v_mfma_f32_16x16x4_f32 rate per SIMD on MI355X, by
// waves per SIMD and number of independent accumulators.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void probe(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int threads, const char* tag) {
  float* d; hipMalloc(&d, blocks * threads * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NACC><<<blocks, threads>>>(d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NACC><<<blocks, threads>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_wave = (double)iters * 16 * NACC;
  double waves_per_simd = threads / 64.0 / 4.0;
  double ns_per_mfma_simd = ms * 1e6 / (mfma_per_wave * waves_per_simd);
  printf("%-28s blocks=%3d thr=%4d acc=%d: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cyc @2.4GHz), chip %.1f TF\n", tag,
         blocks, threads, NACC, ms, ns_per_mfma_simd, ns_per_mfma_simd * 2.4,
         blocks * (threads / 64.0) * mfma_per_wave * 2048 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  run<1>(256, 256, "1 wave/SIMD 1 acc");
  run<2>(256, 256, "1 wave/SIMD 2 acc");
  run<4>(256, 256, "1 wave/SIMD 4 acc");
  run<2>(256, 512, "2 waves/SIMD 2 acc");
  run<2>(57, 512, "2 waves/SIMD 2 acc, 57 CUs");
  run<2>(57, 256, "1 wave/SIMD 2 acc, 57 CUs");
  run<4>(256, 1024, "4 waves/SIMD 4 acc");
  return 0;
}
