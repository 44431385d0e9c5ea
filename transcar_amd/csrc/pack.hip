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
//
// The 16-row tiles compute with v_mfma_f32_16x16x4 (one instruction = 16 rows x 16 columns x 4 k: a
// single wave issues the 4x4x1 at 12.3 cycles instead of 8, tools/issue_probe.hip).  Its B operand is
// one k per 16-lane group, so those tiles read a second copy with the same tile / item structure,
//     P16[t][k/64][kg][i][lane 16g + c][j] = W[64t + 16j + c][64 (k/64) + 16 kg + 4g + i]
// (kg = 16-wide k group of the item, i = MFMA of the group, j = 16-column sub-tile): again 1 KiB
// contiguous per wave-instruction, one float4 = the B operands of the four sub-tiles' MFMAs, and the
// A operand of a k group is one ds_read_b128 per lane (row c... 4 consecutive k at 16 kg + 4g).
//
// Round 4: a third copy for the 16-row tiles on the f16 matrix cores (chain.hip linear_step16h).  Every weight
// as TWO f16 planes, hi = f16(w) and lo = f16((w - hi) * 2^11) (round to nearest; the residual is exact in
// fp32 and the scale keeps it a normal f16: |w - (hi + 2^-11 lo)| <= 2^-24 |w|), in the operand order of
// v_mfma_f32_16x16x32_f16 -- lane 16g + c of a fragment holds 8 consecutive k of one output column:
//     PH[t][k/64][kk][j][p][lane 16g + c][e] = plane p of W[64t + 16j + c][64 (k/64) + 32 kk + 8g + e]
// (kk = 32-deep half of the item, j = 16-column sub-tile, p = hi / lo, e = 0..7): 16 fragments of 1 KiB per
// item, the same 4 bytes per weight as the fp32 copies, the same tile / item offsets.
#include "kernels.hpp"

namespace tc {

// 32-bit word `wq` (elements 2 wq, 2 wq + 1) of lane `lane` in fragment kq & 15 of item (t, kq >> 4)
__device__ __forceinline__ unsigned pack_h_word(const float* __restrict__ W, int N, int K, int transpose, int ldw,
                                                int t, int kq, int lane, int wq) {
  const int f = kq & 15, kb = kq >> 4;
  const int p = f & 1, j = (f >> 1) & 3, kk = f >> 3;
  const int g = lane >> 4, c = lane & 15;
  const int n = 64 * t + 16 * j + c;
  const int k0 = 64 * kb + 32 * kk + 8 * g + 2 * wq;
  unsigned short h[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int k = k0 + e;
    const float x = (n < N && k < K) ? (transpose ? W[(size_t)k * ldw + n] : W[(size_t)n * K + k]) : 0.0f;
    const _Float16 hi = (_Float16)x;
    const _Float16 v = p == 0 ? hi : (_Float16)((x - (float)hi) * 2048.0f);
    h[e] = __builtin_bit_cast(unsigned short, v);
  }
  return (unsigned)h[0] | ((unsigned)h[1] << 16);
}

__global__ __launch_bounds__(256) void pack_linear_kernel(const float* __restrict__ W, int N, int K,
                                                          float* __restrict__ P, float* __restrict__ P16,
                                                          float* __restrict__ PH, int ntile, int nkq) {
  const size_t total = (size_t)ntile * nkq * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int j = i & 3;
    const int lane = (i >> 2) & 63;
    const size_t tk = i >> 8;
    const int kq = tk % nkq;
    const int t = tk / nkq;
    const int n = 64 * t + lane, k = 4 * kq + j;
    P[i] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.0f;
    if (P16 != nullptr) {
      const int g = lane >> 4, c = lane & 15;
      const int n16 = 64 * t + 16 * j + c;
      const int k16 = 64 * (kq >> 4) + 16 * ((kq & 15) >> 2) + 4 * g + (kq & 3);
      P16[i] = (n16 < N && k16 < K) ? W[(size_t)n16 * K + k16] : 0.0f;
    }
    if (PH != nullptr) reinterpret_cast<unsigned*>(PH)[i] = pack_h_word(W, N, K, 0, 0, t, kq, lane, j);
  }
}

// Several weights in ONE launch (the trainable part after an optimizer step: 28 matrices, one launch instead of
// 28): the items travel in the kernel-argument segment, a block finds its item from the block offsets.
constexpr int PACK_GROUP_MAX = 60;        // 60 x 64 B of items + the count fit the 4 KiB kernel-argument segment
struct PackGroupItem { const float* W; float* P; float* P16; float* PH; int N, K, ntile, nkq, first_block, transpose, ldw; };
struct PackGroupK { PackGroupItem it[PACK_GROUP_MAX]; int n; };
static_assert(sizeof(PackGroupK) <= 4096, "the grouped pack's items travel as kernel arguments");
constexpr int PACK_EPT = 8;               // elements per thread
__global__ __launch_bounds__(256) void pack_group_kernel(PackGroupK g) {
  int i = 0;
#pragma unroll 1
  for (int j = 1; j < g.n; ++j)
    if ((int)blockIdx.x >= g.it[j].first_block) i = j;
  const PackGroupItem& it = g.it[i];
  const size_t total = (size_t)it.ntile * it.nkq * 256;
  const size_t base = ((size_t)(blockIdx.x - it.first_block) * 256 + threadIdx.x) * PACK_EPT;
#pragma unroll
  for (int e = 0; e < PACK_EPT; ++e) {
    const size_t idx = base + e;
    if (idx >= total) return;
    const int j = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t tk = idx >> 8;
    const int kq = tk % it.nkq;
    const int t = tk / it.nkq;
    const int n = 64 * t + lane, k = 4 * kq + j;
    it.P[idx] = (n < it.N && k < it.K) ? (it.transpose ? it.W[(size_t)k * it.ldw + n] : it.W[(size_t)n * it.K + k]) : 0.0f;
    if (it.P16 != nullptr) {
      const int gq = lane >> 4, c = lane & 15;
      const int n16 = 64 * t + 16 * j + c;
      const int k16 = 64 * (kq >> 4) + 16 * ((kq & 15) >> 2) + 4 * gq + (kq & 3);
      it.P16[idx] = (n16 < it.N && k16 < it.K) ? (it.transpose ? it.W[(size_t)k16 * it.ldw + n16] : it.W[(size_t)n16 * it.K + k16]) : 0.0f;
    }
    if (it.PH != nullptr)
      reinterpret_cast<unsigned*>(it.PH)[idx] = pack_h_word(it.W, it.N, it.K, it.transpose, it.ldw, t, kq, lane, j);
  }
}

int launch_pack_group(const PackJob* jobs, int n, hipStream_t s) {
  TC_REQUIRE(n >= 1 && n <= PACK_GROUP_MAX, "pack_group: %d weights (1..%d)", n, PACK_GROUP_MAX);
  PackGroupK g;
  g.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    TC_REQUIRE(jobs[i].W != nullptr && jobs[i].P != nullptr && jobs[i].N > 0 && jobs[i].K > 0, "pack_group: bad item %d", i);
    PackGroupItem& it = g.it[i];
    it.W = jobs[i].W; it.P = jobs[i].P; it.P16 = jobs[i].P16; it.PH = jobs[i].PH; it.N = jobs[i].N; it.K = jobs[i].K;
    it.transpose = jobs[i].transpose; it.ldw = jobs[i].ldw;
    it.ntile = (it.N + 63) / 64; it.nkq = ((it.K + 63) / 64) * 16;
    it.first_block = blocks;
    const size_t total = (size_t)it.ntile * it.nkq * 256;
    blocks += (int)((total + 256 * PACK_EPT - 1) / (256 * PACK_EPT));
  }
  hipLaunchKernelGGL(pack_group_kernel, dim3(blocks), dim3(256), 0, s, g);
  return check_launch("pack_group");
}

size_t packed_floats(int N, int K) {
  return (size_t)((N + 63) / 64) * 64 * ((K + 63) / 64) * 64;
}

int launch_pack_linear(const float* W, int N, int K, float* P, float* P16, float* PH, hipStream_t s) {
  TC_REQUIRE(W != nullptr && P != nullptr && N > 0 && K > 0, "pack_linear: bad arguments");
  const int ntile = (N + 63) / 64, nkq = ((K + 63) / 64) * 16;
  const size_t total = (size_t)ntile * nkq * 256;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, s, W, N, K, P, P16, PH, ntile, nkq);
  return check_launch("pack_linear");
}

}  // namespace tc
