// FPN -> head handoff: NCHW [n_img, C, H, W] -> channels-last [n_img, H, W, C].
// The reference hands the head NCHW maps (DET:62-66); the sampling kernel wants
// a tap's 256 channels contiguous.  Pure HBM streaming (read + write the maps
// once), charged separately from the decoder in DESIGN.md; skipped entirely when
// the neck already emits channels_last tensors.
#include "kernels.hpp"

namespace tc {

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const size_t img = blockIdx.z;
  const float* s = src + img * (size_t)C * HW;
  float* d = dst + img * (size_t)C * HW;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = i * 4 + ty, px = p0 + tx;
    if (px < HW && c0 + c < C) tile[c][tx] = s[(size_t)(c0 + c) * HW + px];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int px = p0 + i * 4 + ty;
    if (px < HW && c0 + tx < C) d[(size_t)px * C + c0 + tx] = tile[tx][i * 4 + ty];
  }
}

int launch_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, hipStream_t s) {
  TC_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: empty input");
  const int HW = H * W;
  dim3 grid((HW + 63) / 64, (C + 63) / 64, n_img);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, s, src, dst, C, HW);
  return check_launch("nchw_to_nhwc");
}

}  // namespace tc
