// Diagnostic: how fast can a workgroup stream a [256 x 256] fp32 weight matrix
// from L2/HBM into VGPRs with the chain kernel's access pattern?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
// mode 0: chain pattern (lane (r,g): row r of tile, 16B at k = 32c+16h+4g), wave w: tiles w, w+8
// mode 1: fully coalesced (wave reads 1 KB contiguous per instruction)
template <int MODE>
__global__ __launch_bounds__(512) void stream(const float* W, int nmat, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int m = 0; m < nmat; ++m) {
    const float* Wm = W + (size_t)m * 65536;
    for (int kb = 0; kb < 2; ++kb) {
      float4 v[16];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (MODE == 0) {
            v[c * 4 + h * 2 + 0] = *(const float4*)(Wm + (size_t)(wave * 16 + r) * 256 + kb * 128 + 32 * c + 16 * h + 4 * g);
            v[c * 4 + h * 2 + 1] = *(const float4*)(Wm + (size_t)((wave + 8) * 16 + r) * 256 + kb * 128 + 32 * c + 16 * h + 4 * g);
          } else {
            const int idx = ((kb * 8 + c * 2 + h) * 2) * 8 + wave;   // 1 KB chunks
            v[c * 4 + h * 2 + 0] = *(const float4*)(Wm + (size_t)idx * 256 + lane * 4);
            v[c * 4 + h * 2 + 1] = *(const float4*)(Wm + (size_t)(idx + 8) * 256 + lane * 4);
          }
        }
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE>
int run(const float* W, float* out, int blocks, int nmat, const char* tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  stream<MODE><<<blocks, 512>>>(W, nmat, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    stream<MODE><<<blocks, 512>>>(W, nmat, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-34s blocks=%3d nmat=%3d: %.1f us total, %.2f us per 256KB matrix, %.1f GB/s per CU\n", tag, blocks, nmat,
         best * 1e3, best * 1e3 / nmat, 262144.0 * nmat / (best * 1e-3) / 1e9);
  return 0;
}
int main() {
  const int nmat = 96;   // 24 MB: larger than one XCD's L2
  float *W, *out; CK(hipMalloc(&W, (size_t)nmat * 262144)); CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMemset(W, 0, (size_t)nmat * 262144));
  run<0>(W, out, 57, 12, "chain pattern, 12 mats (3 MB)");
  run<0>(W, out, 57, 96, "chain pattern, 96 mats (24 MB)");
  run<1>(W, out, 57, 12, "coalesced, 12 mats");
  run<1>(W, out, 57, 96, "coalesced, 96 mats");
  run<0>(W, out, 1, 12, "chain pattern, 1 block");
  run<0>(W, out, 228, 12, "chain pattern, 228 blocks");
  run<1>(W, out, 228, 12, "coalesced, 228 blocks");
  return 0;
}
