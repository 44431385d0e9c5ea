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
#include "kernels.hpp"

namespace tc {

struct CamK {
  const float* data[TC_MAX_LEVELS];
  int H[TC_MAX_LEVELS], W[TC_MAX_LEVELS];
  int num_levels, B, Q, num_cams;
  const float* l2i; const float* ref; const float* logits;
  float pc[6]; float img_h, img_w;
  float* out; unsigned char* vis; unsigned long long* pair_counter;
};

template <int L>
__global__ __launch_bounds__(256) void cam_sample_kernel(CamK p) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= p.B * p.Q) return;
  const int b = row / p.Q;
  const int N = p.num_cams;
  // XFMR:389-391
  const float rx = p.ref[(size_t)row * 3 + 0] * (p.pc[3] - p.pc[0]) + p.pc[0];
  const float ry = p.ref[(size_t)row * 3 + 1] * (p.pc[4] - p.pc[1]) + p.pc[1];
  const float rz = p.ref[(size_t)row * 3 + 2] * (p.pc[5] - p.pc[2]) + p.pc[2];
  const float* lg = p.logits + (size_t)row * N * L;

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int nvis = 0;
  for (int cam = 0; cam < N; ++cam) {
    const float* m = p.l2i + ((size_t)b * N + cam) * 16;
    // XFMR:398-409
    const float cx = m[0] * rx + m[1] * ry + m[2] * rz + m[3];
    const float cy = m[4] * rx + m[5] * ry + m[6] * rz + m[7];
    const float cz = m[8] * rx + m[9] * ry + m[10] * rz + m[11];
    const float eps = 1e-5f;
    const float zc = fmaxf(cz, eps);
    float u = (cx / zc) / p.img_w;
    float v = (cy / zc) / p.img_h;
    u = (u - 0.5f) * 2.0f;
    v = (v - 0.5f) * 2.0f;
    const bool visible = (cz > eps) && (u > -1.0f) && (u < 1.0f) && (v > -1.0f) && (v < 1.0f);
    if (p.vis != nullptr && lane == 0) p.vis[(size_t)row * N + cam] = visible ? 1 : 0;
    if (!visible) continue;
    ++nvis;

    float4 tap[L][4];
    float wgt[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = p.H[l], W = p.W[l];
      // F.grid_sample, bilinear, zeros padding, align_corners=False
      const float ix = ((u + 1.0f) * (float)W - 1.0f) * 0.5f;
      const float iy = ((v + 1.0f) * (float)H - 1.0f) * 0.5f;
      const float xw = floorf(ix), yn = floorf(iy);
      const float w_ = ix - xw, e_ = 1.0f - w_, n_ = iy - yn, s_ = 1.0f - n_;
      const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
      const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x1 >= 0) && (x1 < W);
      const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y1 >= 0) && (y1 < H);
      wgt[l][0] = (vx0 && vy0) ? s_ * e_ : 0.0f;   // nw
      wgt[l][1] = (vx1 && vy0) ? s_ * w_ : 0.0f;   // ne
      wgt[l][2] = (vx0 && vy1) ? n_ * e_ : 0.0f;   // sw
      wgt[l][3] = (vx1 && vy1) ? n_ * w_ : 0.0f;   // se
      const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
      const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y1, 0), H - 1);
      const float* base = p.data[l] + ((size_t)(b * N + cam) * H * W) * 256 + 4 * lane;
      tap[l][0] = ld4(base + ((size_t)yc0 * W + xc0) * 256);
      tap[l][1] = ld4(base + ((size_t)yc0 * W + xc1) * 256);
      tap[l][2] = ld4(base + ((size_t)yc1 * W + xc0) * 256);
      tap[l][3] = ld4(base + ((size_t)yc1 * W + xc1) * 256);
    }
    float4 camacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float4 s;
      s.x = tap[l][0].x * wgt[l][0] + tap[l][1].x * wgt[l][1] + tap[l][2].x * wgt[l][2] + tap[l][3].x * wgt[l][3];
      s.y = tap[l][0].y * wgt[l][0] + tap[l][1].y * wgt[l][1] + tap[l][2].y * wgt[l][2] + tap[l][3].y * wgt[l][3];
      s.z = tap[l][0].z * wgt[l][0] + tap[l][1].z * wgt[l][1] + tap[l][2].z * wgt[l][2] + tap[l][3].z * wgt[l][3];
      s.w = tap[l][0].w * wgt[l][0] + tap[l][1].w * wgt[l][1] + tap[l][2].w * wgt[l][2] + tap[l][3].w * wgt[l][3];
      // XFMR:367 -- NaN -> 0 on the sampled value
      if (s.x != s.x) s.x = 0.f;
      if (s.y != s.y) s.y = 0.f;
      if (s.z != s.z) s.z = 0.f;
      if (s.w != s.w) s.w = 0.f;
      const float a = sigmoidf_(lg[cam * L + l]);   // XFMR:370, mask == 1 here
      camacc.x += s.x * a; camacc.y += s.y * a; camacc.z += s.z * a; camacc.w += s.w * a;
    }
    acc.x += camacc.x; acc.y += camacc.y; acc.z += camacc.z; acc.w += camacc.w;
  }
  st4(p.out + (size_t)row * 256 + 4 * lane, acc);
  if (p.pair_counter != nullptr && lane == 0 && nvis > 0)
    atomicAdd(p.pair_counter, (unsigned long long)nvis);
}

int launch_cam_sample(const CamSampleArgs& a, hipStream_t s) {
  TC_REQUIRE(a.C == 256, "cam_sample: C=%d (256 supported)", a.C);
  TC_REQUIRE(a.feats.num_levels == 4, "cam_sample: num_levels=%d (4 supported)", a.feats.num_levels);
  CamK p;
  for (int l = 0; l < TC_MAX_LEVELS; ++l) {
    p.data[l] = a.feats.data[l]; p.H[l] = a.feats.H[l]; p.W[l] = a.feats.W[l];
  }
  p.num_levels = a.feats.num_levels; p.B = a.B; p.Q = a.Q; p.num_cams = a.num_cams;
  p.l2i = a.lidar2img; p.ref = a.ref; p.logits = a.logits;
  for (int i = 0; i < 6; ++i) p.pc[i] = a.pc[i];
  p.img_h = a.img_h; p.img_w = a.img_w;
  p.out = a.out; p.vis = a.vis; p.pair_counter = a.pair_counter;
  const int rows = a.B * a.Q;
  hipLaunchKernelGGL(cam_sample_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  return check_launch("cam_sample");
}

}  // namespace tc
