// Do the 225 workgroups of a fused chain throttle each other by reading the SAME packed weight
// lines at the same time (every workgroup walks the layer's 3.18 MB in the same order)?  Each
// wave streams its quarter of nmat [256 x 256] matrices (16 x 1 KiB wave-instructions per item,
// two register buffers, a few VALU ops per item) in three orders:
//   same    : every workgroup the same order (the chain kernels today)
//   rot-k   : workgroup b starts each matrix at k block (b % 4) (rotates the 64-deep k blocks)
//   rot-mat : workgroup b starts at matrix (b % nmat) (each workgroup elsewhere in the stream)
//   hipcc --offload-arch=gfx950 -O3 tools/hotspot_probe.hip -o tools/hotspot_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256, 2) void stream(const float* __restrict__ W, int nmat, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  float4 buf[2][16];
  // item (m, kb): packed tile `wave` of matrix m = 64 KiB; k block kb = 16 KiB = 16 x 1 KiB
  auto addr = [&](int it) {
    int m = it >> 2, kb = it & 3;
    if (MODE == 1) kb = (kb + b) & 3;
    if (MODE == 2) m = (m + b) % nmat;
    return W + (size_t)m * 65536 + (size_t)wave * 16384 + (size_t)kb * 4096 + lane * 4;
  };
  const int nit = nmat * 4;
  const float* p = addr(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) buf[0][j] = *(const float4*)(p + j * 256);
  for (int it = 0; it < nit; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* np = addr(min(it + h + 1, nit - 1));
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        buf[h ^ 1][j] = *(const float4*)(np + j * 256);
        acc.x += buf[h][j].x; acc.y += buf[h][j].y; acc.z += buf[h][j].z; acc.w += buf[h][j].w;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE>
int run(const float* W, float* out, int blocks, int nmat, const char* tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  stream<MODE><<<blocks, 256>>>(W, nmat, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(e0));
    stream<MODE><<<blocks, 256>>>(W, nmat, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-8s blocks=%3d: %6.1f us for %d matrices (%.2f MB per workgroup) = %5.1f GB/s per workgroup, %.0f cycles per 64 KiB item round at 2.4 GHz\n",
         tag, blocks, best * 1e3, nmat, nmat * 0.262144, 262144.0 * nmat / (best * 1e-3) / 1e9, best * 1e-3 * 2.4e9 / (nmat * 4));
  return 0;
}
int main() {
  const int nmat = 12;   // 3.1 MB: one decoder layer
  float *W, *out; CK(hipMalloc(&W, (size_t)nmat * 262144)); CK(hipMalloc(&out, 1024 * 256 * 4));
  CK(hipMemset(W, 0, (size_t)nmat * 262144));
  for (int blocks : {1, 8, 57, 113, 225, 450}) {
    run<0>(W, out, blocks, nmat, "same");
    run<1>(W, out, blocks, nmat, "rot-k");
    run<2>(W, out, blocks, nmat, "rot-mat");
  }
  return 0;
}
