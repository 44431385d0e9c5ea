// Decoder self-attention core: softmax(Q K^T) V for 900 queries,
// 8 heads of 32, fp32, without materialising the 8 x 900 x 900 score tensor the
// reference's nn.MultiheadAttention builds (25.9 MB per layer).
//
// Flash-style on the f32 matrix core, laid out for wave64 / 16x16x4 MFMA:
//   * a workgroup = 8 waves = one (batch, head, 16-query tile); the waves
//     split the 16-key tiles round-robin and merge their running (max, sum, O)
//     through LDS at the end -> 57 x 8 = 456 workgroups, 3648 waves for B = 1
//     (3.6 per SIMD: the next tile's K/V loads of one wave hide behind the
//     MFMAs and exps of its neighbours);
//   * scores are computed TRANSPOSED, S^T = K Q^T, so the accumulator of a
//     tile (lane = query column, registers = 4 keys) is already the B operand
//     of the second product O^T += V^T P^T: no LDS round trip, no shuffles for
//     the P matrix; the per-query max is two permlane swaps (lanes c, c+16,
//     c+32, c+48 hold the same query);
//   * V arrives transposed ([B, C, Qpad], written that way by the in_proj GEMM
//     epilogue) so a lane's 4 keys of one channel are one 16-byte load;
//   * Q is pre-scaled by log2(e)/sqrt(32) in the in_proj epilogue, so the
//     softmax runs on v_exp_f32 (2^x) directly;
//   * the K/V fragments of tile t+NW are loaded before tile t is consumed.
// K/V of one head are 115 KB each and stay in L2 across the 57 query tiles.
// Algorithmic work: 4*Q*Q*32 flop per (batch, head); bound: f32 MFMA.
#include "kernels.hpp"

namespace tc {

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int SA_NW = 8;

// max over the four lanes c, c+16, c+32, c+48 (they hold the same query column) on
// the VALU: gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves /
// odd-even 16-lane rows between two registers, so swap(x, x) followed by one v_max is
// an xor-32 / xor-16 all-reduce.  (__shfl_xor goes through the LDS crossbar, and two
// of them sat in the dependency chain S -> max -> exp -> PV of every key tile.)
__device__ __forceinline__ float max_lanes_16_32(float x) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned xi = __builtin_bit_cast(unsigned, x);
  u2 r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  x = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
  xi = __builtin_bit_cast(unsigned, x);
  r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

struct KVFrag { float4 ka, kb, v0, v1; };

__device__ __forceinline__ KVFrag load_kv(const float* __restrict__ k, const float* vbase, int ld,
                                          int ldt, size_t brow, int h, int key0, int r, int g, int Q) {
  KVFrag f;
  const int krow = min(key0 + r, Q - 1);
  const float* kp = k + (brow + krow) * ld + h * 32 + 8 * g;
  f.ka = ld4(kp); f.kb = ld4(kp + 4);
  f.v0 = ld4(vbase + key0 + 4 * g);
  f.v1 = ld4(vbase + (size_t)16 * ldt + key0 + 4 * g);
  return f;
}

// QT = 16-query sub-tiles per workgroup: the K/V fragments of a key tile are loaded
// once and used for QT score / PV products (K/V re-reads from L2 are the kernel's
// main memory traffic: 57 query tiles x 8 heads x 230 KB at QT = 1).
//
// DROP (training statistics of the frozen decoder, tools/train.py:245-252 leaves its dropouts
// on): nn.MultiheadAttention's dropout on the attention probabilities -- the normalised
// probability of (batch b, head h, query i, key j) is multiplied by 0 or 1/(1-p), mask
// drop_keep(seed, site, ((b*H + h)*Q + i)*Q + j): only the PV product sees the mask, the
// normaliser l does not.  A separate instantiation: the eval kernel is unchanged.
template <int QT, bool DROP>
__global__ __launch_bounds__(SA_NW * 64) void self_attn_kernel(const float* __restrict__ q,
                                                               const float* __restrict__ k, int ld,
                                                               const float* __restrict__ vt, int ldt,
                                                               float* __restrict__ out, int ldo,
                                                               int Q, int C, DropK drop) {
  __shared__ float sm_m[SA_NW][QT][16];
  __shared__ float sm_l[SA_NW][QT][64];
  __shared__ float4 sm_o[SA_NW][QT][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 16 * QT, h = blockIdx.y, b = blockIdx.z;
  const size_t brow = (size_t)b * Q;

  float4 qa[QT], qb[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = min(q0 + 16 * u + r, Q - 1);
    const float* qp = q + (brow + qrow) * ld + h * 32 + 8 * g;
    qa[u] = ld4(qp); qb[u] = ld4(qp + 4);
  }
  const float* vbase = vt + ((size_t)b * C + h * 32 + r) * ldt;

  f32x4 o0[QT], o1[QT];
  float m[QT], lpart[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    o0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; o1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[u] = -INFINITY; lpart[u] = 0.0f;
  }
  const int ntiles = (Q + 15) / 16;
  int t = wave;
  KVFrag cur;
  if (t < ntiles) cur = load_kv(k, vbase, ld, ldt, brow, h, t * 16, r, g, Q);
  for (; t < ntiles; t += SA_NW) {
    const int key0 = t * 16;
    KVFrag nxt = cur;
    if (t + SA_NW < ntiles) nxt = load_kv(k, vbase, ld, ldt, brow, h, (t + SA_NW) * 16, r, g, Q);
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this tile's MFMAs
    float4 v0 = cur.v0, v1 = cur.v1;
    const bool ragged = key0 + 16 > Q;   // wave-uniform
    const int kk = key0 + 4 * g;
    if (ragged) {
      if (kk + 0 >= Q) { v0.x = 0.f; v1.x = 0.f; }
      if (kk + 1 >= Q) { v0.y = 0.f; v1.y = 0.f; }
      if (kk + 2 >= Q) { v0.z = 0.f; v1.z = 0.f; }
      if (kk + 3 >= Q) { v0.w = 0.f; v1.w = 0.f; }
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      s = MFMA4(cur.ka.x, qa[u].x, s); s = MFMA4(cur.ka.y, qa[u].y, s);
      s = MFMA4(cur.ka.z, qa[u].z, s); s = MFMA4(cur.ka.w, qa[u].w, s);
      s = MFMA4(cur.kb.x, qb[u].x, s); s = MFMA4(cur.kb.y, qb[u].y, s);
      s = MFMA4(cur.kb.z, qb[u].z, s); s = MFMA4(cur.kb.w, qb[u].w, s);
      // s[i] = log2(e) * S^T[key0 + 4g + i][q0 + 16u + r]
      float s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
      if (ragged) {
        if (kk + 0 >= Q) s0 = -INFINITY;
        if (kk + 1 >= Q) s1 = -INFINITY;
        if (kk + 2 >= Q) s2 = -INFINITY;
        if (kk + 3 >= Q) s3 = -INFINITY;
      }
      float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
      mx = max_lanes_16_32(mx);
      const float mnew = fmaxf(m[u], mx);           // finite: key0 + 0 < Q in every tile
      const float alpha = __builtin_amdgcn_exp2f(m[u] - mnew);
      const float p0 = __builtin_amdgcn_exp2f(s0 - mnew), p1 = __builtin_amdgcn_exp2f(s1 - mnew);
      const float p2 = __builtin_amdgcn_exp2f(s2 - mnew), p3 = __builtin_amdgcn_exp2f(s3 - mnew);
      lpart[u] = lpart[u] * alpha + ((p0 + p1) + (p2 + p3));
      m[u] = mnew;
      o0[u] *= alpha; o1[u] *= alpha;
      float d0 = p0, d1 = p1, d2 = p2, d3 = p3;
      if (DROP) {
        const unsigned qi = (unsigned)min(q0 + 16 * u + r, Q - 1);
        const unsigned base = (((unsigned)b * gridDim.y + h) * Q + qi) * Q + kk;
        d0 = drop_keep(drop.seed, drop.site, base + 0, drop.thr) ? p0 * drop.scale : 0.0f;
        d1 = drop_keep(drop.seed, drop.site, base + 1, drop.thr) ? p1 * drop.scale : 0.0f;
        d2 = drop_keep(drop.seed, drop.site, base + 2, drop.thr) ? p2 * drop.scale : 0.0f;
        d3 = drop_keep(drop.seed, drop.site, base + 3, drop.thr) ? p3 * drop.scale : 0.0f;
      }
      // O^T[d][q] += V^T[d][key] P^T[key][q]
      o0[u] = MFMA4(v0.x, d0, o0[u]); o1[u] = MFMA4(v1.x, d0, o1[u]);
      o0[u] = MFMA4(v0.y, d1, o0[u]); o1[u] = MFMA4(v1.y, d1, o1[u]);
      o0[u] = MFMA4(v0.z, d2, o0[u]); o1[u] = MFMA4(v1.z, d2, o1[u]);
      o0[u] = MFMA4(v0.w, d3, o0[u]); o1[u] = MFMA4(v1.w, d3, o1[u]);
    }
    cur = nxt;
  }
  // merge the key slices
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (g == 0) sm_m[wave][u][r] = m[u];
    sm_l[wave][u][lane] = lpart[u];
    sm_o[wave][u][0][lane] = make_float4(o0[u][0], o0[u][1], o0[u][2], o0[u][3]);
    sm_o[wave][u][1][lane] = make_float4(o1[u][0], o1[u][1], o1[u][2], o1[u][3]);
  }
  __syncthreads();
  // wave w finalises (sub-tile w/2, channel half w&1)
  if (wave >= 2 * QT) return;
  const int u = wave >> 1, half = wave & 1;
  float mstar = sm_m[0][u][r];
#pragma unroll
  for (int w = 1; w < SA_NW; ++w) mstar = fmaxf(mstar, sm_m[w][u][r]);
  float l = 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < SA_NW; ++w) {
    const float sc = __builtin_amdgcn_exp2f(sm_m[w][u][r] - mstar);
    l += sc * ((sm_l[w][u][r] + sm_l[w][u][r + 16]) + (sm_l[w][u][r + 32] + sm_l[w][u][r + 48]));
    const float4 v = sm_o[w][u][half][lane];
    acc.x += sc * v.x; acc.y += sc * v.y; acc.z += sc * v.z; acc.w += sc * v.w;
  }
  if (q0 + 16 * u + r < Q) {
    const float inv = 1.0f / l;
    float* op = out + (brow + q0 + 16 * u + r) * ldo + h * 32 + 16 * half + 4 * g;
    st4(op, make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv));
  }
}

int launch_self_attn_core(const float* q, const float* k, int ld, const float* vt, int ldt,
                          float* out, int ldo, int B, int Q, int H, hipStream_t s, const DropK* drop) {
  TC_REQUIRE(Q > 0 && B > 0 && H > 0, "self_attn: empty problem");
  TC_REQUIRE((ldt & 3) == 0 && ldt >= ((Q + 15) / 16) * 16, "self_attn: ldt=%d too small for Q=%d", ldt, Q);
  constexpr int QT = 2;
  dim3 grid((Q + 16 * QT - 1) / (16 * QT), H, B);
  if (drop != nullptr && drop->thr != 0) {
    TC_REQUIRE((unsigned long long)B * H * Q * Q < (1ull << 32), "self_attn: dropout index space (B*H*Q*Q) exceeds 32 bits");
    hipLaunchKernelGGL((self_attn_kernel<QT, true>), grid, dim3(SA_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q,
                       H * 32, *drop);
  } else {
    hipLaunchKernelGGL((self_attn_kernel<QT, false>), grid, dim3(SA_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q,
                       H * 32, DropK{0, 0, 1.0f, 0, 0});
  }
  return check_launch("self_attn");
}

}  // namespace tc
