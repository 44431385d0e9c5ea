// Token-wise linear layers of the fusion decoder on the gfx950 f32 matrix core.
//
// Every nn.Linear on the hot path is "M rows (900 queries or <=1500 radar tokens)
// x a small weight matrix" (k2 in/out proj, k8, k9, k11,
// k12, k14, k17-k19).  Parity is fp32 (boxes within 1e-3 of the reference), so
// the products run on v_mfma_f32_16x16x4_f32: exact f32 FMA chains at the f32
// matrix rate (157 TF/s dense on MI355X).
//
// Decomposition (B=1 makes every GEMM tiny, so the launch must spread over the
// chip): one workgroup = 4 waves = a 16-row x 64-column output tile, each wave
// one 16x16 tile over the full K.  900x256 -> 57 x 4 = 228 workgroups.
// Operands go global -> VGPR directly: lane (r = lane&15, g = lane>>4) reads
// 8 consecutive k of row r (two 16-byte loads, four lanes cover one 128-byte
// line) and feeds them to 8 MFMAs; the k -> MFMA-slot assignment is the same
// permutation on A and B, so the sum is unchanged.  A (16 x K, <=32 KB) is
// re-read by the 4 waves through L1; W panels stream from L2.
#include "kernels.hpp"

namespace tc {

struct GemmK {
  const float* X; const float* X2; const float* W; const float* bias;
  const float* R; const int* rowgate; float* Y; float* Yt;
  int ldx, x2_cols, ldw, ldr, ldy, t_col0, t_ld, t_rpb;
  int M, K, N, act, scale_cols;
  float scale;
};

__device__ __forceinline__ float4 ldk(const float* row, int k, int K) {
  if (k + 4 <= K) return ld4(row + k);
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool KFULL>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmK p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 16;
  const int n0 = blockIdx.x * 64 + wave * 16;
  if (n0 >= p.N) return;
  const int arow = min(m0 + r, p.M - 1);
  const int bcol = min(n0 + r, p.N - 1);
  const float* xa = p.X + (size_t)arow * p.ldx;
  const float* xb = (p.X2 != nullptr && n0 < p.x2_cols) ? p.X2 + (size_t)arow * p.ldx : nullptr;
  const float* wb = p.W + (size_t)bcol * p.ldw;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int K = p.K;
#pragma unroll 2
  for (int k0 = 0; k0 < K; k0 += 32) {
    const int k = k0 + 8 * g;
    float4 a0, a1, b0, b1;
    if (KFULL) {
      a0 = ld4(xa + k); a1 = ld4(xa + k + 4);
      b0 = ld4(wb + k); b1 = ld4(wb + k + 4);
    } else {
      a0 = ldk(xa, k, K); a1 = ldk(xa, k + 4, K);
      b0 = ldk(wb, k, K); b1 = ldk(wb, k + 4, K);
    }
    if (xb != nullptr) {
      float4 c0, c1;
      if (KFULL) { c0 = ld4(xb + k); c1 = ld4(xb + k + 4); }
      else { c0 = ldk(xb, k, K); c1 = ldk(xb, k + 4, K); }
      a0.x += c0.x; a0.y += c0.y; a0.z += c0.z; a0.w += c0.w;
      a1.x += c1.x; a1.y += c1.y; a1.z += c1.z; a1.w += c1.w;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
  }

  // C layout: col = lane&15, row = 4*(lane>>4) + reg
  const int col = n0 + r;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.0f;
  const float sc = (col < p.scale_cols) ? p.scale : 1.0f;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 4 * g + i;
    float t = (acc[i] + bv) * sc;
    if (p.act == 1) t = fmaxf(t, 0.0f);
    else if (p.act == 2) t = sigmoidf_(t);
    if (row < p.M) {
      if (p.rowgate != nullptr && p.rowgate[row] == 0) t = 0.0f;
      if (p.R != nullptr) t += p.R[(size_t)row * p.ldr + col];
    }
    v[i] = t;
  }
  if (p.Yt != nullptr && col >= p.t_col0) {
    const int c = col - p.t_col0;
    const int nct = p.N - p.t_col0;
    const int row0 = m0 + 4 * g;
    if (row0 + 3 < p.M && (p.t_rpb & 3) == 0) {
      const int b = row0 / p.t_rpb, q = row0 - b * p.t_rpb;
      st4(p.Yt + ((size_t)b * nct + c) * p.t_ld + q, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + i;
        if (row < p.M) {
          const int b = row / p.t_rpb, q = row - b * p.t_rpb;
          p.Yt[((size_t)b * nct + c) * p.t_ld + q] = v[i];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 4 * g + i;
      if (row < p.M) p.Y[(size_t)row * p.ldy + col] = v[i];
    }
  }
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  TC_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  TC_REQUIRE((a.K & 3) == 0 && (a.ldx & 3) == 0 && (a.ldw & 3) == 0,
             "gemm: K/ldx/ldw must be multiples of 4 (K=%d ldx=%d ldw=%d)", a.K, a.ldx, a.ldw);
  TC_REQUIRE((a.x2_cols & 15) == 0 || a.x2_cols >= a.N, "gemm: x2_cols must be 16-aligned");
  GemmK p;
  p.X = a.X; p.X2 = a.X2; p.W = a.W; p.bias = a.bias; p.R = a.R; p.rowgate = a.rowgate;
  p.Y = a.Y; p.Yt = a.Yt;
  p.ldx = a.ldx; p.x2_cols = a.x2_cols; p.ldw = a.ldw; p.ldr = a.ldr; p.ldy = a.ldy;
  p.t_col0 = a.t_col0; p.t_ld = a.t_ld; p.t_rpb = a.t_rows_per_batch > 0 ? a.t_rows_per_batch : 1;
  p.M = a.M; p.K = a.K; p.N = a.N; p.act = a.act; p.scale_cols = a.scale_cols; p.scale = a.scale;
  dim3 grid((a.N + 63) / 64, (a.M + 15) / 16);
  if ((a.K & 31) == 0)
    hipLaunchKernelGGL(gemm16_kernel<true>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(gemm16_kernel<false>, grid, dim3(256), 0, s, p);
  return check_launch("gemm16");
}

}  // namespace tc
