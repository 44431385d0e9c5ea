// One-time weight re-layout for the fused row chains (chain.hip).
//
// nn.Linear stores W as [N][K].  The chain kernel's 4x4x1 MFMA wants, per lane
// n of a 64-column output tile, consecutive k of W[64t + n][.]: read from the
// checkpoint layout a wave-instruction touches 64 rows x 16 B and a CU sustains
// ~36 GB/s (tools/wstream_probe.hip).  Packed as
//     P[t][k/4][lane n][4]
// the same instruction reads 1 KiB contiguous (124 GB/s per CU), and a 64-row
// tile still spans 64 * Kpad floats, so row offsets (multiples of 64) keep their
// meaning.  N and K are zero padded to multiples of 64 (no guards in the loop).
// Done once per checkpoint load (tc_head_pack_weights), not per frame.
#include "kernels.hpp"

namespace tc {

__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ W, int N, int K,
                                                          float* __restrict__ P, int ntile, int nkq) {
  const size_t total = (size_t)ntile * nkq * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int j = i & 3;
    const int lane = (i >> 2) & 63;
    const size_t tk = i >> 8;
    const int kq = tk % nkq;
    const int t = tk / nkq;
    const int n = 64 * t + lane, k = 4 * kq + j;
    P[i] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.0f;
  }
}

size_t packed_floats(int N, int K) {
  return (size_t)((N + 63) / 64) * 64 * ((K + 63) / 64) * 64;
}

int launch_pack_linear(const float* W, int N, int K, float* P, hipStream_t s) {
  TC_REQUIRE(W != nullptr && P != nullptr && N > 0 && K > 0, "pack_linear: bad arguments");
  const int ntile = (N + 63) / 64, nkq = ((K + 63) / 64) * 16;
  const size_t total = (size_t)ntile * nkq * 256;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, s, W, N, K, P, ntile, nkq);
  return check_launch("pack_linear");
}

}  // namespace tc
