// Do MFMA and VALU instructions overlap on a gfx950 SIMD?  Every wave runs a loop of 8 independent
// MFMAs (8 accumulators) with NV independent v_fma_f32 behind each; W waves per SIMD (one workgroup of
// 256*W threads per CU).  Prints SIMD cycles per MFMA: the MFMA's own issue time when the VALU work
// hides behind it, the sum of both when it does not.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV> __device__ __forceinline__ void valu(float (&v)[8], float m) {
#pragma unroll
  for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(m));
}

template <int KIND, int NV>       // KIND 0: v_mfma_f32_4x4x1 (2 passes), 1: v_mfma_f32_16x16x4 (8 passes), 2: no MFMA
__global__ void probe(int iters, float* out) {
  f32x4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; }
  const float a = threadIdx.x * 0.5f, b = 1.0f / (1 + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
      if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      valu<NV>(v, b);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND, int NV> int run(int W, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<KIND, NV><<<256, 256 * W>>>(100, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<KIND, NV><<<256, 256 * W>>>(iters, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms * 1e-3 * 2.4e9;                       // at 2.4 GHz
  const double slots = (double)iters * 8 * W;                 // (MFMA + NV VALU) groups per SIMD
  printf("%-22s VALU per MFMA %2d  waves/SIMD %d : %7.2f cycles per group\n",
         KIND == 0 ? "v_mfma_f32_4x4x1" : KIND == 1 ? "v_mfma_f32_16x16x4" : "(no MFMA)", NV, W, cyc / slots);
  return 0;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 4));
  for (int W = 1; W <= 4; W *= 2) {
    run<2, 4>(W, out);
    run<0, 0>(W, out); run<0, 1>(W, out); run<0, 2>(W, out); run<0, 4>(W, out);
    run<1, 0>(W, out); run<1, 2>(W, out); run<1, 4>(W, out); run<1, 8>(W, out); run<1, 12>(W, out);
  }
  return 0;
}
