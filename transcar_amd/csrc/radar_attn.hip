// Distance-gated radar cross-attention (HEAD:549-579 and
// the _2/_3 copies).  The reference builds three 900x1500 cdist matrices, a
// boolean mask, selects the rows with a hit and runs nn.MultiheadAttention on
// them with the mask turned into -inf.  None of that is materialised here:
//
//   one wavefront = one query.  The 64 lanes test 64 radar tokens at a time
//   against the three circles (centre / front / rear, HEAD:556-571), a ballot
//   gives the hit set, and only the hit tokens (typically 0..10 of 1500) go
//   through the 8-head softmax(q k / sqrt(32)) v: lane l holds channels
//   4l..4l+3, i.e. 8 lanes per head, so a K or V row is one coalesced 1 KiB
//   load, a head's score is a 3-step xor-shuffle, and the softmax over hits is
//   an online (running max / sum) update -- identical to softmax over the
//   -inf-masked row.  Queries with no hit get a zero row and hit_count 0 (the
//   out_proj GEMM gates its residual add on that count, which reproduces the
//   reference's row-subset update).
//
// Distances follow torch.cdist's matmul expansion for >25 points
// (|x|^2 + |y|^2 - 2 x.y as a K=4 FMA chain, clamp 1e-30, sqrt): its rounding
// noise (up to 7e-4 m at these coordinates) is part of the reference's gate,
// and an "exact" distance would flip more gate decisions, not fewer.
#include "kernels.hpp"

namespace tc {

struct RadK {
  const float* qproj; const float* kv; const float* cxy; const float* box; const float* rxy;
  int ldq, ldkv, code, ld_xy, ld_c, B, Q, T, pad_mult;
  float rmin, rmax;
  float* attn_out; int* hits;
};

// torch.cdist(p=2) via _euclidean_dist: [-2x, |x|^2, 1] . [y, 1, |y|^2]
__device__ __forceinline__ float cdist_mm(float x0, float x1, float xn, float y0, float y1, float yn) {
  float t = __fmul_rn(__fmul_rn(-2.0f, x0), y0);
  t = fmaf(__fmul_rn(-2.0f, x1), y1, t);
  t = __fadd_rn(t, xn);
  t = __fadd_rn(t, yn);
  return sqrtf(fmaxf(t, 1e-30f));
}
__device__ __forceinline__ float sqnorm2(float a, float b) {
  return __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
}

__global__ __launch_bounds__(256) void radar_attn_kernel(RadK p) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= p.B * p.Q) return;
  const int b = row / p.Q;
  // gate geometry, HEAD:553-567
  const float cx = p.cxy[(size_t)row * p.ld_c + 0], cy = p.cxy[(size_t)row * p.ld_c + 1];
  const float* bx = p.box + (size_t)row * p.code;
  const float len = expf(bx[3]);
  const float rs = -bx[6], rc = -bx[7];
  const float ox = __fmul_rn(__fmul_rn(len, 0.25f), rs), oy = __fmul_rn(__fmul_rn(len, 0.25f), rc);
  const float fx = __fadd_rn(cx, ox), fy = __fadd_rn(cy, oy);
  const float bxx = __fsub_rn(cx, ox), byy = __fsub_rn(cy, oy);
  const float rad = fminf(fmaxf(len / 2.0f, p.rmin), p.rmax);
  const float cn = sqnorm2(cx, cy), fn = sqnorm2(fx, fy), bn = sqnorm2(bxx, byy);

  const float4 q4 = ld4(p.qproj + (size_t)row * p.ldq + 4 * lane);
  float m = -INFINITY, l = 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int count = 0;
  for (int t0 = 0; t0 < p.T; t0 += 64) {
    const int t = t0 + lane;
    bool hit = false;
    if (t < p.T) {
      const float* y = p.rxy + ((size_t)b * p.T + t) * p.ld_xy;
      const float y0 = y[0], y1 = y[1];
      const float yn = sqnorm2(y0, y1);
      hit = (cdist_mm(cx, cy, cn, y0, y1, yn) < rad) || (cdist_mm(fx, fy, fn, y0, y1, yn) < rad) ||
            (cdist_mm(bxx, byy, bn, y0, y1, yn) < rad);
    }
    unsigned long long mask = __ballot(hit);
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int tok = t0 + j;
      const int mult = (tok == p.T - 1) ? p.pad_mult : 1;
      count += mult;
      const float* kvr = p.kv + ((size_t)b * p.T + tok) * p.ldkv + 4 * lane;
      const float4 k4 = ld4(kvr);
      const float4 v4 = ld4(kvr + 256);
      float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const float mnew = fmaxf(m, s);
      const float alpha = expf(m - mnew);
      const float pw = (float)mult * expf(s - mnew);
      l = l * alpha + pw;
      acc.x = acc.x * alpha + pw * v4.x; acc.y = acc.y * alpha + pw * v4.y;
      acc.z = acc.z * alpha + pw * v4.z; acc.w = acc.w * alpha + pw * v4.w;
      m = mnew;
    }
  }
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (count > 0) {
    const float inv = 1.0f / l;
    o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
  st4(p.attn_out + (size_t)row * 256 + 4 * lane, o);
  if (lane == 0) p.hits[row] = count;
}

int launch_radar_attn(const RadarAttnArgs& a, hipStream_t s) {
  TC_REQUIRE(a.C == 256 && a.H == 8, "radar_attn: C=%d H=%d (256/8 supported)", a.C, a.H);
  TC_REQUIRE(a.T > 0 && a.pad_mult >= 1, "radar_attn: T=%d pad_mult=%d", a.T, a.pad_mult);
  RadK p;
  p.qproj = a.qproj; p.kv = a.kv; p.cxy = a.centre_xy; p.box = a.box; p.rxy = a.radar_xy;
  p.ldq = a.ldq; p.ldkv = a.ldkv; p.code = a.code; p.ld_xy = a.ld_xy; p.ld_c = a.ld_c;
  p.B = a.B; p.Q = a.Q; p.T = a.T; p.pad_mult = a.pad_mult; p.rmin = a.rmin; p.rmax = a.rmax;
  p.attn_out = a.attn_out; p.hits = a.hit_counts;
  const int rows = a.B * a.Q;
  hipLaunchKernelGGL(radar_attn_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  return check_launch("radar_attn");
}

}  // namespace tc
