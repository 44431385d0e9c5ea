// Token-wise linear layers of the fusion decoder on the gfx950 f32 matrix core.
//
// Every nn.Linear on the hot path is "M rows (900 queries or <=1500 radar tokens)
// x a small weight matrix" (k2 in/out proj, k8, k9, k11,
// k12, k14, k17-k19).  Parity is fp32 (boxes within 1e-3 of the reference), so
// the products run on v_mfma_f32_16x16x4_f32: exact f32 FMA chains at the f32
// matrix rate (157 TF/s dense on MI355X).
//
// B = 1 makes every GEMM tiny (900x256x256 = 118 MFLOP), so a launch is bound
// by latency, not by the matrix pipe.  The decomposition is chosen for that:
//   * one workgroup = a 16-row x 64-column output tile -> 57 x 4 = 228
//     workgroups for 900x256, about one per CU;
//   * the NW waves of a workgroup SPLIT K: wave w owns 32-wide k-chunks
//     w*NCH .. w*NCH+NCH-1 for all four 16x16 column tiles, so each wave issues
//     ALL of its operand loads (<= 40 16-byte loads per lane) before the first
//     MFMA and the L2 latency is paid once per launch, not once per k-step;
//     A is read exactly once per workgroup (no redundancy between waves);
//   * operands go global -> VGPR directly: lane (r = lane&15, g = lane>>4)
//     reads 8 consecutive k of row r (two 16-byte loads; the four g-lanes cover
//     one 128-byte line) and feeds them to 8 MFMAs; the k -> MFMA-slot map is
//     the same permutation on A and B, so the dot product is unchanged;
//   * the NW partial tiles meet in LDS (16-32 KB), waves 0..3 each finalise
//     one column tile: bias, q-scaling, activation, row gate, residual, and
//     either a row-major store or a transposed float4 store (V^T for the
//     attention core).
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
__device__ __forceinline__ float comp(const float4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NW, int NCH>
__global__ __launch_bounds__(NW * 64) void gemm16_kernel(GemmK p) {
  __shared__ float4 red[NW][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 16;
  const int nb = blockIdx.x * 64;
  const int K = p.K;
  const int arow = min(m0 + r, p.M - 1);
  const float* xa = p.X + (size_t)arow * p.ldx;
  const bool use_x2 = (p.X2 != nullptr) && (nb < p.x2_cols);
  const float* xb = use_x2 ? p.X2 + (size_t)arow * p.ldx : xa;
  const int ntile = min(4, (p.N - nb + 15) / 16);      // valid 16-col tiles of this block

  float4 a[NCH][2];
  float4 b[4][NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = (wave * NCH + c) * 32 + 8 * g;
    a[c][0] = ldk(xa, k, K);
    a[c][1] = ldk(xa, k + 4, K);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float* wb = p.W + (size_t)min(nb + 16 * t + r, p.N - 1) * p.ldw;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = (wave * NCH + c) * 32 + 8 * g;
      if (t < ntile) {
        b[t][c][0] = ldk(wb, k, K);
        b[t][c][1] = ldk(wb, k + 4, K);
      } else {
        b[t][c][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        b[t][c][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (use_x2) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = (wave * NCH + c) * 32 + 8 * g;
      const float4 c0 = ldk(xb, k, K), c1 = ldk(xb, k + 4, K);
      a[c][0].x += c0.x; a[c][0].y += c0.y; a[c][0].z += c0.z; a[c][0].w += c0.w;
      a[c][1].x += c1.x; a[c][1].y += c1.y; a[c][1].z += c1.z; a[c][1].w += c1.w;
    }
  }

  // keep every load above issued before the first MFMA (hipcc otherwise sinks
  // each load next to its use and leaves 1-2 loads in flight per wave)
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av = comp(a[c][h], i);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MFMA4(av, comp(b[t][c][h], i), acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    red[wave][t][lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  __syncthreads();
  if (wave >= ntile) return;

  // wave t finalises column tile t.  C layout: col = lane&15, row = 4*(lane>>4) + reg
  const int t = wave;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float4 v = red[w][t][lane];
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
  const int n0 = nb + 16 * t;
  const int col = n0 + r;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.0f;
  const float sc = (col < p.scale_cols) ? p.scale : 1.0f;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 4 * g + i;
    float y = (s[i] + bv) * sc;
    if (p.act == 1) y = relu_(y);
    else if (p.act == 2) y = sigmoidf_(y);
    if (row < p.M) {
      if (p.rowgate != nullptr && p.rowgate[row] == 0) y = 0.0f;
      if (p.R != nullptr) y += p.R[(size_t)row * p.ldr + col];
    }
    v[i] = y;
  }
  if (p.Yt != nullptr && col >= p.t_col0) {
    const int c = col - p.t_col0;
    const int nct = p.N - p.t_col0;
    const int row0 = m0 + 4 * g;
    if (row0 + 3 < p.M && (p.t_rpb & 3) == 0) {
      const int bb = row0 / p.t_rpb, q = row0 - bb * p.t_rpb;
      st4(p.Yt + ((size_t)bb * nct + c) * p.t_ld + q, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + i;
        if (row < p.M) {
          const int bb = row / p.t_rpb, q = row - bb * p.t_rpb;
          p.Yt[((size_t)bb * nct + c) * p.t_ld + q] = v[i];
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
  TC_REQUIRE((a.x2_cols & 63) == 0 || a.x2_cols >= a.N, "gemm: x2_cols must be 64-aligned");
  TC_REQUIRE(a.K <= 1024, "gemm: K=%d > 1024 not supported", a.K);
  GemmK p;
  p.X = a.X; p.X2 = a.X2; p.W = a.W; p.bias = a.bias; p.R = a.R; p.rowgate = a.rowgate;
  p.Y = a.Y; p.Yt = a.Yt;
  p.ldx = a.ldx; p.x2_cols = a.x2_cols; p.ldw = a.ldw; p.ldr = a.ldr; p.ldy = a.ldy;
  p.t_col0 = a.t_col0; p.t_ld = a.t_ld; p.t_rpb = a.t_rows_per_batch > 0 ? a.t_rows_per_batch : 1;
  p.M = a.M; p.K = a.K; p.N = a.N; p.act = a.act; p.scale_cols = a.scale_cols; p.scale = a.scale;
  dim3 grid((a.N + 63) / 64, (a.M + 15) / 16);
  const int chunks = (a.K + 31) / 32;
  if (chunks <= 4)
    hipLaunchKernelGGL((gemm16_kernel<4, 1>), grid, dim3(256), 0, s, p);
  else if (chunks <= 8)
    hipLaunchKernelGGL((gemm16_kernel<8, 1>), grid, dim3(512), 0, s, p);
  else if (chunks <= 16)
    hipLaunchKernelGGL((gemm16_kernel<8, 2>), grid, dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL((gemm16_kernel<8, 4>), grid, dim3(512), 0, s, p);
  return check_launch("gemm16");
}

}  // namespace tc
