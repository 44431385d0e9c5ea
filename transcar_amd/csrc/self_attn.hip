// Decoder self-attention core: softmax(Q K^T) V for 900 queries,
// 8 heads of 32, fp32, without materialising the 8 x 900 x 900 score tensor the
// reference's nn.MultiheadAttention builds (25.9 MB per layer).
//
// Flash-style on the f32 matrix core, laid out for wave64 / 16x16x4 MFMA:
//   * a workgroup = 4 waves = one (batch, head, 16-query tile); the 4 waves
//     split the 16-key tiles round-robin and merge their running (max, sum, O)
//     through LDS at the end -> 57 x 8 = 456 workgroups, 1824 waves for B = 1;
//   * scores are computed TRANSPOSED, S^T = K Q^T, so the accumulator of a
//     tile (lane = query column, registers = 4 keys) is already the B operand
//     of the second product O^T += V^T P^T: no LDS round trip, no shuffles for
//     the P matrix; the per-query max needs two xor-shuffles (lanes c, c+16,
//     c+32, c+48 hold the same query);
//   * V arrives transposed ([B, C, Qpad], written that way by the in_proj GEMM
//     epilogue) so a lane's 4 keys of one channel are one 16-byte load;
//   * Q is pre-scaled by 1/sqrt(32) in the in_proj epilogue, as torch does.
// K/V of one head are 115 KB each and stay in L2 across the 57 query tiles.
#include "kernels.hpp"

namespace tc {

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void self_attn_kernel(const float* __restrict__ q,
                                                        const float* __restrict__ k, int ld,
                                                        const float* __restrict__ vt, int ldt,
                                                        float* __restrict__ out, int ldo, int Q,
                                                        int C) {
  __shared__ float sm_m[4][16];
  __shared__ float sm_l[4][64];
  __shared__ float sm_o[4][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;

  const int qrow = min(q0 + r, Q - 1);
  const float* qp = q + ((size_t)b * Q + qrow) * ld + h * 32 + 8 * g;
  const float4 qa = ld4(qp), qb = ld4(qp + 4);
  const float* vbase = vt + ((size_t)b * C + h * 32 + r) * ldt;

  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lpart = 0.0f;
  const int ntiles = (Q + 15) / 16;
  for (int t = wave; t < ntiles; t += 4) {
    const int key0 = t * 16;
    const int krow = min(key0 + r, Q - 1);
    const float* kp = k + ((size_t)b * Q + krow) * ld + h * 32 + 8 * g;
    const float4 ka = ld4(kp), kb = ld4(kp + 4);
    float4 v0 = ld4(vbase + key0 + 4 * g);
    float4 v1 = ld4(vbase + (size_t)16 * ldt + key0 + 4 * g);

    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = MFMA4(ka.x, qa.x, s); s = MFMA4(ka.y, qa.y, s);
    s = MFMA4(ka.z, qa.z, s); s = MFMA4(ka.w, qa.w, s);
    s = MFMA4(kb.x, qb.x, s); s = MFMA4(kb.y, qb.y, s);
    s = MFMA4(kb.z, qb.z, s); s = MFMA4(kb.w, qb.w, s);
    // s[i] = S^T[key0 + 4g + i][q0 + r]
    const int kk = key0 + 4 * g;
    const bool ok0 = kk + 0 < Q, ok1 = kk + 1 < Q, ok2 = kk + 2 < Q, ok3 = kk + 3 < Q;
    float s0 = ok0 ? s[0] : -INFINITY, s1 = ok1 ? s[1] : -INFINITY;
    float s2 = ok2 ? s[2] : -INFINITY, s3 = ok3 ? s[3] : -INFINITY;
    if (!ok0) { v0.x = 0.f; v1.x = 0.f; }
    if (!ok1) { v0.y = 0.f; v1.y = 0.f; }
    if (!ok2) { v0.z = 0.f; v1.z = 0.f; }
    if (!ok3) { v0.w = 0.f; v1.w = 0.f; }
    float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(m, mx);           // finite: key0 + 0 < Q in every tile
    const float alpha = expf(m - mnew);
    const float p0 = expf(s0 - mnew), p1 = expf(s1 - mnew);
    const float p2 = expf(s2 - mnew), p3 = expf(s3 - mnew);
    lpart = lpart * alpha + ((p0 + p1) + (p2 + p3));
    m = mnew;
    o0 *= alpha; o1 *= alpha;
    // O^T[d][q] += V^T[d][key] P^T[key][q]
    o0 = MFMA4(v0.x, p0, o0); o1 = MFMA4(v1.x, p0, o1);
    o0 = MFMA4(v0.y, p1, o0); o1 = MFMA4(v1.y, p1, o1);
    o0 = MFMA4(v0.z, p2, o0); o1 = MFMA4(v1.z, p2, o1);
    o0 = MFMA4(v0.w, p3, o0); o1 = MFMA4(v1.w, p3, o1);
  }
  // merge the 4 key-slices
  if (g == 0) sm_m[wave][r] = m;
  sm_l[wave][lane] = lpart;
#pragma unroll
  for (int i = 0; i < 4; ++i) { sm_o[wave][lane][i] = o0[i]; sm_o[wave][lane][4 + i] = o1[i]; }
  __syncthreads();
  if (wave != 0) return;
  float mstar = fmaxf(fmaxf(sm_m[0][r], sm_m[1][r]), fmaxf(sm_m[2][r], sm_m[3][r]));
  float l = 0.0f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float sc = expf(sm_m[w][r] - mstar);
    l += sc * ((sm_l[w][r] + sm_l[w][r + 16]) + (sm_l[w][r + 32] + sm_l[w][r + 48]));
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += sc * sm_o[w][lane][i];
  }
  if (q0 + r < Q) {
    const float inv = 1.0f / l;
    float* op = out + ((size_t)b * Q + q0 + r) * ldo + h * 32 + 4 * g;
    st4(op, make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv));
    st4(op + 16, make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv));
  }
}

int launch_self_attn_core(const float* q, const float* k, int ld, const float* vt, int ldt,
                          float* out, int ldo, int B, int Q, int H, hipStream_t s) {
  TC_REQUIRE(Q > 0 && B > 0 && H > 0, "self_attn: empty problem");
  TC_REQUIRE((ldt & 3) == 0 && ldt >= ((Q + 15) / 16) * 16, "self_attn: ldt=%d too small for Q=%d", ldt, Q);
  dim3 grid((Q + 15) / 16, H, B);
  hipLaunchKernelGGL(self_attn_kernel, grid, dim3(256), 0, s, q, k, ld, vt, ldt, out, ldo, Q, H * 32);
  return check_launch("self_attn");
}

}  // namespace tc
