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
// The per-query routine lives in rowdev.hpp (shared with chain.hip).
#include "kernels.hpp"
#include "rowdev.hpp"

namespace tc {

struct RadK {
  const float* qproj; const float* kv; const float* cxy; const float* box; const float* rxy;
  int ldq, ldkv, code, ld_xy, ld_c, B, Q, T, pad_mult;
  float rmin, rmax, qscale;
  float* attn_out; int* hits;
  DropK drop;
};

template <bool DROP>
__global__ __launch_bounds__(256) void radar_attn_kernel(RadK p) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= p.B * p.Q) return;
  const int b = row / p.Q;
  const float cx = p.cxy[(size_t)row * p.ld_c + 0], cy = p.cxy[(size_t)row * p.ld_c + 1];
  const float* bx = p.box + (size_t)row * p.code;
  float4 q4 = ld4(p.qproj + (size_t)row * p.ldq + 4 * lane);
  q4.x *= p.qscale; q4.y *= p.qscale; q4.z *= p.qscale; q4.w *= p.qscale;
  int count = 0;
  const float4 o = radar_attn_row<DROP>(cx, cy, bx[3], bx[6], bx[7], p.rmin, p.rmax, q4,
                                        p.rxy + (size_t)b * p.T * p.ld_xy, p.ld_xy,
                                        p.kv + (size_t)b * p.T * p.ldkv, p.ldkv, p.T, p.pad_mult, lane,
                                        count, p.drop, row);
  st4(p.attn_out + (size_t)row * 256 + 4 * lane, o);
  if (lane == 0) p.hits[row] = count;
}

int launch_radar_attn(const RadarAttnArgs& a, hipStream_t s) {
  TC_REQUIRE(a.C == 256 && a.H == 8, "radar_attn: C=%d H=%d (256/8 supported)", a.C, a.H);
  TC_REQUIRE(a.T > 0 && a.pad_mult >= 1, "radar_attn: T=%d pad_mult=%d", a.T, a.pad_mult);
  RadK p;
  p.qproj = a.qproj; p.kv = a.kv; p.cxy = a.centre_xy; p.box = a.box; p.rxy = a.radar_xy;
  p.ldq = a.ldq; p.ldkv = a.ldkv; p.code = a.code; p.ld_xy = a.ld_xy; p.ld_c = a.ld_c;
  p.B = a.B; p.Q = a.Q; p.T = a.T; p.pad_mult = a.pad_mult; p.rmin = a.rmin; p.rmax = a.rmax; p.qscale = a.qscale;
  p.attn_out = a.attn_out; p.hits = a.hit_counts; p.drop = a.drop;
  const int rows = a.B * a.Q;
  if (a.drop.thr != 0)
    hipLaunchKernelGGL(radar_attn_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(radar_attn_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, p);
  return check_launch("radar_attn");
}

}  // namespace tc
