// What makes a launch expensive inside a replayed graph?  Trivial kernels that return
// at once, varied in dynamic LDS size, register count and grid size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Arg { int v[320]; };
__global__ __launch_bounds__(256) void k_small(Arg a, int* out) {
  if (a.v[5] == 123456) out[0] = 1;
}
// forces ~250 VGPRs + AGPRs live
__global__ __launch_bounds__(256) void k_regs(Arg a, int* out) {
  if (a.v[5] != 123456) return;
  float x[240];
#pragma unroll
  for (int i = 0; i < 240; ++i) x[i] = out[i + threadIdx.x];
  float s = 0;
#pragma unroll
  for (int i = 0; i < 240; ++i) s += x[i] * x[(i * 7) % 240];
  out[threadIdx.x] = (int)s;
}
extern __shared__ float dyn[];
__global__ __launch_bounds__(256) void k_lds(Arg a, int* out) {
  if (a.v[5] == 123456) { dyn[threadIdx.x] = 1.f; out[0] = (int)dyn[0]; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// AGPR accumulators + a big body behind a never-taken branch
template <int REP>
__global__ __launch_bounds__(256) void k_big(Arg a, int* out) {
  if (a.v[5] != 123456) return;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float x = out[threadIdx.x];
#pragma unroll
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x + r, x * i, acc[i], 4, 0, 0);
    x += acc[r & 15][0];
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = (int)s;
}
__constant__ int ctab[64] = {1, 2, 3};
__global__ __launch_bounds__(256) void k_const(Arg a, int* out) {
  if (a.v[5] == 123456) out[0] = ctab[threadIdx.x & 63];
}
template <typename K> int run(K kern, const char* name, int blocks, size_t lds, int* out) {
  Arg a; for (int i = 0; i < 320; ++i) a.v[i] = i;
  hipStream_t s; CK(hipStreamCreate(&s));
  if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a, out);
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a, out);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-10s blocks=%4d dyn LDS=%6zu B: %.2f us per launch\n", name, blocks, lds, best * 1e3 / 50);
  return 0;
}
int main() {
  int* out; CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(out, 0, 1 << 20));
  run(k_small, "small", 225, 0, out);
  run(k_small, "small", 1, 0, out);
  run(k_lds, "lds", 225, 32768, out);
  run(k_lds, "lds", 225, 65536, out);
  run(k_lds, "lds", 225, 140000, out);
  run(k_regs, "regs", 225, 0, out);
  run(k_regs, "regs", 1, 0, out);
  run(k_big<64>, "big64", 225, 0, out);
  run(k_big<1500>, "big1500", 225, 0, out);
  run(k_const, "const", 225, 0, out);
  return 0;
}
