// Row order for the fused radar chain (chain.hip, PROG_RADAR): queries with a radar return inside
// their gate first.
//
// About three of four queries have no radar token inside their three circles (HEAD:549-567; G5 rig:
// 214 / 219 / 69 of 900 have one), and for a row tile without a single hit the chain skips the q
// projection, the gated attention and the out_proj (x + 0 * (...) = x, F_IFHIT in chain.hip).  A tile
// of 16 consecutive queries is hit-free with probability 0.76^16; ordered by the gate of the first
// fusion layer three of four tiles are.  The order is a HINT: every layer's tiles decide from their own
// gates, rows are computed independently of their tile mates, outputs go back to the original row --
// bit-identical results, whatever the order.
//
//   radar_hit_flags_kernel   16 queries per workgroup: the first layer's gate (the chain's own predicate,
//                            rowdev.hpp GateGeom) against the sample's tokens, staged in LDS
//   radar_partition_kernel   one workgroup per sample: stable partition of its Q rows, hits first;
//                            perm[b*Q + i] = row (of the same sample: b = row / Q keeps its meaning)
#include "kernels.hpp"
#include "rowdev.hpp"

namespace tc {

struct CompactK {
  const float* ref_last; const float* box; const float* tokens;
  int code, cen_from_box, RI, B, Q, T;
  float pc[6]; float rmin, rmax;
  int* flags;
};

// A workgroup = 16 queries of ONE sample (4 per wave).  The tokens' xy sit 36 floats apart in the token
// matrix: fetched per wave (64 lanes x 144-byte stride, 8 instructions) the 7 200 queries of 8 frames
// pulled 236 MB of 64-byte sectors through the cache (11.5 us); staged once per workgroup in LDS, 256
// tokens at a time, it is 1/16 of that.
__global__ __launch_bounds__(256) void radar_hit_flags_kernel(CompactK p) {
  __shared__ float sx[256], sy[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = (p.Q + 15) / 16;
  const int b = blockIdx.x / tiles, q0 = (blockIdx.x - b * tiles) * 16;
  float cx[4], cy[4], b3[4], b6[4], b7[4];
  bool any[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = min(q0 + 4 * wave + i, p.Q - 1);
    const size_t row = (size_t)b * p.Q + q;
    const float* bx = p.box + row * p.code;
    if (p.cen_from_box) { cx[i] = bx[0]; cy[i] = bx[1]; }        // HEAD:615-617
    else {                                                       // HEAD:543-547
      cx[i] = __fadd_rn(__fmul_rn(p.ref_last[row * 3 + 0], p.pc[3] - p.pc[0]), p.pc[0]);
      cy[i] = __fadd_rn(__fmul_rn(p.ref_last[row * 3 + 1], p.pc[4] - p.pc[1]), p.pc[1]);
    }
    b3[i] = bx[3]; b6[i] = bx[6]; b7[i] = bx[7];
    any[i] = false;
  }
  const float* rxy = p.tokens + (size_t)b * p.T * p.RI;
  for (int t0 = 0; t0 < p.T; t0 += 256) {
    const int t = min(t0 + (int)threadIdx.x, p.T - 1);
    __syncthreads();
    sx[threadIdx.x] = rxy[(size_t)t * p.RI]; sy[threadIdx.x] = rxy[(size_t)t * p.RI + 1];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const GateGeom gg(cx[i], cy[i], b3[i], b6[i], b7[i], p.rmin, p.rmax);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int tt = t0 + 64 * w + lane;
        const float y0 = sx[64 * w + lane], y1 = sy[64 * w + lane];
        any[i] |= __ballot(tt < p.T && gg.hit_approx(y0, y1, sqnorm2(y0, y1))) != 0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q0 + 4 * wave + i;
    if (lane == 0 && q < p.Q) p.flags[(size_t)b * p.Q + q] = any[i] ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void radar_partition_kernel(const int* __restrict__ flags, int Q, int* __restrict__ perm) {
  __shared__ int wsum[2][4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* f = flags + (size_t)b * Q;
  // total number of hit rows of the sample, then two running offsets (hits from 0, the others behind them)
  int nh = 0;
  for (int i = threadIdx.x; i < Q; i += 256) nh += f[i];
  for (int o = 32; o > 0; o >>= 1) nh += __shfl_xor(nh, o, 64);
  if (lane == 0) wsum[0][wave] = nh;
  __syncthreads();
  const int total_hits = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
  __syncthreads();
  int hit_base = 0, miss_base = total_hits;
  for (int i0 = 0; i0 < Q; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const int v = i < Q ? f[i] : 0;
    const bool in = i < Q;
    const unsigned long long mh = __ballot(in && v != 0), mm = __ballot(in && v == 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { wsum[0][wave] = __popcll(mh); wsum[1][wave] = __popcll(mm); }
    __syncthreads();
    int ph = 0, pm = 0, th = 0, tm = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) { ph += wsum[0][w]; pm += wsum[1][w]; }
      th += wsum[0][w]; tm += wsum[1][w];
    }
    if (in) {
      const int pos = v != 0 ? hit_base + ph + __popcll(mh & below) : miss_base + pm + __popcll(mm & below);
      perm[(size_t)b * Q + pos] = b * Q + i;
    }
    hit_base += th; miss_base += tm;
    __syncthreads();
  }
}

// ---- self-check of GateGeom::hit's square-root-free comparison (tc_radar_gate_selfcheck) -------------
__global__ __launch_bounds__(256) void gate_selfcheck_kernel(int n_radii, unsigned long long seed,
                                                             unsigned long long* mismatches) {
  const int r = blockIdx.x;
  if (r >= n_radii) return;
  // radius r: the three clamp values first, then pseudo-random ones in [0.5, 2]
  unsigned long long h = (seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r + 1));
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
  float rad = 0.5f + 1.5f * (float)(h >> 40) * (1.0f / 16777216.0f);
  if (r == 0) rad = 0.5f;
  if (r == 1) rad = 1.0f;
  if (r == 2) rad = 2.0f;
  const float tstar = sqrt_threshold(rad);
  const unsigned centre = __builtin_bit_cast(unsigned, rad * rad);
  unsigned long long bad = 0;
  for (int i = threadIdx.x; i < 512 + 4096; i += 256) {
    float t;
    if (i < 512) t = __builtin_bit_cast(float, centre - 256u + (unsigned)i);
    else {
      unsigned long long g = h + 0xD1B54A32D192ED03ull * (unsigned long long)i;
      g ^= g >> 29; g *= 0xBF58476D1CE4E5B9ull; g ^= g >> 32;
      t = -1.0f + 10.0f * (float)(g >> 40) * (1.0f / 16777216.0f);      // [-1, 9): both sides, and below the clamp
    }
    const bool ref = sqrtf(fmaxf(t, 1e-30f)) < rad;
    const bool fast = fmaxf(t, 1e-30f) < tstar;
    bad += ref != fast;
  }
  if (bad) atomicAdd(mismatches, bad);
}

int launch_gate_selfcheck(int n_radii, unsigned long long seed, unsigned long long* mismatches, hipStream_t s) {
  TC_REQUIRE(n_radii > 0 && mismatches != nullptr, "radar_gate_selfcheck: bad arguments");
  hipLaunchKernelGGL(gate_selfcheck_kernel, dim3(n_radii), dim3(256), 0, s, n_radii, seed, mismatches);
  return check_launch("radar_gate_selfcheck");
}

int launch_radar_compact(const float* ref_last, const float* box, int code, int cen_from_box,
                         const float* pc6_host, const float* tokens, int RI, int B, int Q, int T,
                         float rmin, float rmax, int* flags, int* perm, hipStream_t s) {
  TC_REQUIRE(B > 0 && Q > 0 && T > 0 && flags != nullptr && perm != nullptr, "radar_compact: bad arguments");
  CompactK p;
  p.ref_last = ref_last; p.box = box; p.tokens = tokens; p.code = code; p.cen_from_box = cen_from_box;
  p.RI = RI; p.B = B; p.Q = Q; p.T = T; p.rmin = rmin; p.rmax = rmax; p.flags = flags;
  for (int i = 0; i < 6; ++i) p.pc[i] = pc6_host[i];
  hipLaunchKernelGGL(radar_hit_flags_kernel, dim3(B * ((Q + 15) / 16)), dim3(256), 0, s, p);
  hipLaunchKernelGGL(radar_partition_kernel, dim3(B), dim3(256), 0, s, flags, Q, perm);
  return check_launch("radar_compact");
}

}  // namespace tc
