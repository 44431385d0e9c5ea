// Camera cross-attention sampling: the fused replacement of feature_sampling
// (XFMR:381-422) plus the NaN rule, sigmoid weighting and (cam, level)
// reduction of Detr3DCrossAtten.forward (XFMR:367-373).
//
// The reference materialises [B,C,Q,6,1,4] with 24 F.grid_sample calls per
// decoder layer over NCHW maps (a channel of one tap is H*W floats apart) and
// touches all 6 cameras.  Here one wavefront owns one query: the 64 lanes hold
// the 256 channels as float4, so every bilinear tap of a channels-last (NHWC)
// map is ONE coalesced 1 KiB row; cameras that fail the visibility test
// contribute exactly 0 in the reference (weight = sigmoid * mask) and are
// skipped, which is what makes the algorithmic byte count "visibility-aware"
// (SURVEY.md section 8(d)).  The 16 taps of a visible camera are issued as 16
// independent 16-byte loads per lane before any is consumed (out-of-range taps
// read a clamped address with weight 0 = zeros padding), so a wave keeps 16 KiB
// in flight; 900 waves over 256 CUs.  Bound: HBM/L2 gather latency.
// The per-query routine lives in rowdev.hpp (shared with the fused decoder
// chain, chain.hip); this file is the stand-alone operator.
#include "kernels.hpp"
#include "rowdev.hpp"

namespace tc {

template <int L>
__global__ __launch_bounds__(256) void cam_sample_kernel(CamK p) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= p.B * p.Q) return;
  const int b = row / p.Q;
  int nvis = 0;
  const float4 acc = cam_sample_row<L>(p, row, b, p.logits + (size_t)row * p.num_cams * L, lane, nvis);
  st4(p.out + (size_t)row * 256 + 4 * lane, acc);
  if (p.pair_counter != nullptr && lane == 0 && nvis > 0)
    atomicAdd(p.pair_counter, (unsigned long long)nvis);
}

void fill_camk(const CamSampleArgs& a, CamK& p) {
  for (int l = 0; l < TC_MAX_LEVELS; ++l) {
    p.data[l] = a.feats.data[l]; p.H[l] = a.feats.H[l]; p.W[l] = a.feats.W[l];
  }
  p.num_levels = a.feats.num_levels; p.B = a.B; p.Q = a.Q; p.num_cams = a.num_cams;
  p.l2i = a.lidar2img; p.ref = a.ref; p.logits = a.logits;
  for (int i = 0; i < 6; ++i) p.pc[i] = a.pc[i];
  p.img_h = a.img_h; p.img_w = a.img_w;
  p.out = a.out; p.vis = a.vis; p.pair_counter = a.pair_counter;
}

int launch_cam_sample(const CamSampleArgs& a, hipStream_t s) {
  TC_REQUIRE(a.C == 256, "cam_sample: C=%d (256 supported)", a.C);
  TC_REQUIRE(a.feats.num_levels == 4, "cam_sample: num_levels=%d (4 supported)", a.feats.num_levels);
  for (int l = 0; l < a.feats.num_levels; ++l)      // pixel indices are 32-bit in the kernels
    TC_REQUIRE((long long)a.B * a.num_cams * a.feats.H[l] * a.feats.W[l] < (1ll << 31),
               "cam_sample: level %d has too many pixels for one call", l);
  CamK p;
  fill_camk(a, p);
  const int rows = a.B * a.Q;
  hipLaunchKernelGGL(cam_sample_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  return check_launch("cam_sample");
}

}  // namespace tc
