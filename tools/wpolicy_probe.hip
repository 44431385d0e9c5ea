// Round 6: does the cache policy of the WEIGHT loads change what a CU gets out of its XCD's L2?
// 254 workgroups of 8 waves (one per CU, as a nine-frame chain launch) stream the same 3.2 MB "layer" (every workgroup reads all
// of it, a wave 1 KiB contiguous per load instruction, 16 loads = one 16 KiB item in flight per wave, as chain.hip wload), REP
// layers of different addresses back to back (25 MB in all: the L2s hold one layer, the Infinity Cache all of them).
// Variants of the load: plain | nt | sc0 | sc1 | sc0 sc1.   hipcc --offload-arch=gfx950 -O3 tools/wpolicy_probe.hip -o tools/wpolicy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int LAYER_FLOATS = 800 * 1024;       // 3.2 MB
constexpr int NLAYERS = 8;
template <int POL>
__device__ __forceinline__ float4 ldw(const float* p) {
  float4 v;
  if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int POL>
__global__ __launch_bounds__(512) void stream(const float* W, int rep, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = 0; r < rep; ++r) {
    const float* L = W + (size_t)(r % NLAYERS) * LAYER_FLOATS;
    // items of 16 KiB per wave: wave w takes items w, w + 8, ...
    for (int it = wave; it < LAYER_FLOATS / 4096; it += 8) {
      float4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = ldw<POL>(L + (size_t)it * 4096 + i * 256 + lane * 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
// the same stream with DEPTH items (16 KiB each) in flight per wave: is 16 KiB per wave enough to saturate the L2?
template <int DEPTH>
__global__ __launch_bounds__(512) void stream_deep(const float* W, int rep, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = 0; r < rep; ++r) {
    const float* L = W + (size_t)(r % NLAYERS) * LAYER_FLOATS;
    for (int it = wave * DEPTH; it + DEPTH <= LAYER_FLOATS / 4096; it += 8 * DEPTH) {
      float4 v[DEPTH][16];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[d][i] = ldw<0>(L + (size_t)(it + d) * 4096 + i * 256 + lane * 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.x += v[d][i].x; acc.y += v[d][i].y; acc.z += v[d][i].z; acc.w += v[d][i].w; }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int DEPTH> int run_deep(const float* W, float* out, int waves_note) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rep = 48;
  stream_deep<DEPTH><<<254, 512>>>(W, 8, out); CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int k = 0; k < 3; ++k) {
    CK(hipEventRecord(e0)); stream_deep<DEPTH><<<254, 512>>>(W, rep, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  const int items = LAYER_FLOATS / 4096 / (8 * DEPTH) * (8 * DEPTH);
  const double bytes = (double)rep * items * 16384.0;
  printf("plain, %d item(s) = %d KiB in flight per wave: %7.1f us per layer, %6.1f GB/s per CU = %5.1f B/clk\n", DEPTH, 16 * DEPTH, best * 1e3 / rep,
         bytes / best / 1e6, bytes / best / 1e6 / 2.4);
  return 0;
}
template <int POL> int run(const float* W, float* out, const char* tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rep = 48;
  stream<POL><<<254, 512>>>(W, 8, out); CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int k = 0; k < 3; ++k) {
    CK(hipEventRecord(e0)); stream<POL><<<254, 512>>>(W, rep, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  const double bytes = (double)rep * LAYER_FLOATS * 4;
  printf("%-8s %7.1f us per layer, %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz (one load stream deep: no prefetch of the next item)\n", tag, best * 1e3 / rep, bytes / best / 1e6, bytes / best / 1e6 / 2.4);
  return 0;
}
int main() {
  float *W, *out; CK(hipMalloc(&W, (size_t)NLAYERS * LAYER_FLOATS * 4)); CK(hipMalloc(&out, 254 * 512 * 4));
  CK(hipMemset(W, 0, (size_t)NLAYERS * LAYER_FLOATS * 4));
  if (run<0>(W, out, "plain")) return 1;
  if (run<1>(W, out, "nt")) return 1;
  if (run<2>(W, out, "sc0")) return 1;
  if (run<3>(W, out, "sc1")) return 1;
  if (run<4>(W, out, "sc0 sc1")) return 1;
  if (run<0>(W, out, "plain")) return 1;
  if (run_deep<1>(W, out, 8)) return 1;
  if (run_deep<2>(W, out, 8)) return 1;
  if (run_deep<3>(W, out, 8)) return 1;
  return 0;
}
