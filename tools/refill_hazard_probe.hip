// Is refilling the A / B operand registers of a matrix-core MFMA IN PLACE (a global load into the registers an
// already issued v_mfma_f32_16x16x32_f16 has just read) safe on gfx950?  (VERDICT r4 item 3 / ADVICE r4: round 4
// found flaky wrong rows in chain.hip's first f16 item loop at two workgroups per CU and worked around it with a
// second register buffer; nothing outside chain.hip reproduced it and nobody had looked at the ISA.)
//
// This program is that item loop on its own: a workgroup of 4 waves owns 16 rows (activations in LDS as fp32, split
// into two f16 planes in registers), wave w streams the packed two-plane weights of its 64-column tile; per half item
// (64 columns x 32 k): 8 fragments of 1 KiB per wave, 12 MFMAs.  Y = X W^T (K = 256, N = 256) is recomputed `nrep`
// times from `nrep` copies of the weights laid end to end (a stream of 3 MB like a decoder layer's) and EVERY
// repetition's output is compared on the device with the first workgroup-independent reference (the same kernel's
// SAFE variant run alone, one workgroup per CU, checked against the host): mismatches are counted per workgroup.
// 512 workgroups = two per CU; the second workgroup of a CU (blocks b and b + 256 share one) starts late by a
// per-CU delay so that the pair runs out of phase -- the radar program's hit tiles did.
//
// Variants (template parameter V):
//   0  A   in place: the two fragments of sub-tile j are reloaded right behind their three MFMAs
//   1  J   two half-item buffers, the loop unrolled by two (chain.hip's production loop)
//   2  I   in place, one group late (three MFMAs of the next sub-tile between the last read and the load)
//   3  A + the accumulator chain kept out of the weight registers (an asm barrier pins the MFMA results)
//   4  A + s_nop 7 x 2 between the MFMAs and the loads
//   5  A + s_waitcnt vmcnt(0) in front of every half item (the stream drained: no overlap, the slow reference)
//
//   hipcc --offload-arch=gfx950 -O3 tools/refill_hazard_probe.hip -o tools/refill_hazard_probe
//   tools/refill_hazard_probe [launches per variant, default 20]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int LDA = 260;
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

__device__ __forceinline__ uint4 ldg16(const void* p) {
  const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split_f16(const float4& a, const float4& b, uint4& p1, uint4& p2) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t q1[4], q2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    q1[i] = pk_f16(x[2 * i], x[2 * i + 1]);
    const f16x2 h = __builtin_bit_cast(f16x2, q1[i]);
    q2[i] = pk_f16((x[2 * i] - (float)h[0]) * 2048.0f, (x[2 * i + 1] - (float)h[1]) * 2048.0f);
  }
  p1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
  p2 = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}

struct Acc { f32x4 hi[4]; f32x4 lo[4]; };

// one half item from w[base .. base + 7] ((sub-tile j, plane p) at base + 2 j + p); the refill goes to w[nb ..]
template <int V, int J>
struct Half {
  static __device__ __forceinline__ void run(Acc& acc, uint4 (&w)[16], const uint4& x1, const uint4& x2, const char* np,
                                             unsigned lo, const int base, const int nb) {
    acc.lo[J] = MFMA_H(w[base + 2 * J + 1], x1, acc.lo[J]);
    acc.lo[J] = MFMA_H(w[base + 2 * J], x2, acc.lo[J]);
    acc.hi[J] = MFMA_H(w[base + 2 * J], x1, acc.hi[J]);
    if constexpr (V == 3) {     // the weight fragments stay live past the three MFMAs: no result can be allocated into them
      const uint4 a0 = w[base + 2 * J], a1 = w[base + 2 * J + 1];
      const u32x4 k0 = {a0.x, a0.y, a0.z, a0.w}, k1 = {a1.x, a1.y, a1.z, a1.w};
      asm volatile("" ::"v"(k0), "v"(k1));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (V == 4) asm volatile("s_nop 7\n\ts_nop 7");
    if constexpr (V == 2) {
      // one group late: sub-tile J - 1's registers are reloaded behind sub-tile J's MFMAs (J = 0: nothing yet)
      if constexpr (J > 0) {
        w[nb + 2 * (J - 1)] = ldg16(np + lo + (2 * (J - 1)) * 1024);
        w[nb + 2 * (J - 1) + 1] = ldg16(np + lo + (2 * (J - 1) + 1) * 1024);
      }
      if constexpr (J == 3) {
        __builtin_amdgcn_sched_barrier(0);
        w[nb + 6] = ldg16(np + lo + 6 * 1024);
        w[nb + 7] = ldg16(np + lo + 7 * 1024);
      }
    } else {
      w[nb + 2 * J] = ldg16(np + lo + (2 * J) * 1024);
      w[nb + 2 * J + 1] = ldg16(np + lo + (2 * J + 1) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (J < 3) Half<V, J + 1>::run(acc, w, x1, x2, np, lo, base, nb);
  }
};

// W: [4 waves][nhalf_total half items][8 fragments][64 lanes][16 B]; a wave's stream: for rep, for tile-of-the-wave
// (one: N = 256 = 4 waves x 64), for kh in 0..7.  Y_ref: [M][256] (null: store Y instead of comparing).
template <int V>
__global__ __launch_bounds__(256, 2) void probe_kernel(const char* __restrict__ W, const float* __restrict__ X,
                                                       float* __restrict__ Y, const float* __restrict__ Yref,
                                                       int nrep, int nrep_packed, int M, int delay_mul,
                                                       unsigned* __restrict__ bad, long long* __restrict__ cyc) {
  extern __shared__ __align__(16) float lds[];          // [16][LDA]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = (blockIdx.x * 16) % M;
  for (int i = threadIdx.x; i < 16 * 64; i += 256) {
    const int row = i >> 6, c4 = i & 63;
    *reinterpret_cast<float4*>(&lds[row * LDA + 4 * c4]) = *reinterpret_cast<const float4*>(X + (size_t)(m0 + row) * 256 + 4 * c4);
  }
  __syncthreads();
  // the second workgroup of a CU starts late: a per-CU delay in shader cycles (0 .. 255 * delay_mul)
  if (blockIdx.x >= 256 && delay_mul > 0) {
    const long long until = __builtin_amdgcn_s_memtime() + (long long)(blockIdx.x & 255) * delay_mul;
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(1);
  }
  const unsigned lo = 16u * lane;
  const int nhalf = 8 * nrep;
  const char* wbase = W + (size_t)wave * (8 * nrep_packed) * 8192;
  uint4 w[16];
#pragma unroll
  for (int f = 0; f < 8; ++f) w[f] = ldg16(wbase + lo + f * 1024);
#pragma unroll
  for (int f = 8; f < 16; ++f) w[f] = make_uint4(0, 0, 0, 0);
  const float* arow = lds + (lane & 15) * LDA + 8 * (lane >> 4);
  Acc acc;
  unsigned nbad = 0;
  const int c = lane & 15, g = lane >> 4;
  // the expected values of this wave's tile (the same in every repetition): in registers, so that no load but the
  // weight stream's is issued inside the loop
  float4 want[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    want[j] = Yref != nullptr ? *reinterpret_cast<const float4*>(Yref + (size_t)(m0 + c) * 256 + 64 * wave + 16 * j + 4 * g)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  constexpr int STEP = V == 1 ? 2 : 1;
#pragma unroll 1
  for (int it = 0; it < nhalf; it += STEP) {
    const int kh = it & 7;
    if (kh == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc.hi[j] = f32x4{0, 0, 0, 0}; acc.lo[j] = f32x4{0, 0, 0, 0}; }
    }
    if constexpr (V == 1) {
      {
        uint4 x1, x2;
        split_f16(*reinterpret_cast<const float4*>(arow + kh * 32), *reinterpret_cast<const float4*>(arow + kh * 32 + 4), x1, x2);
        Half<V, 0>::run(acc, w, x1, x2, wbase + (size_t)(it + 1) * 8192, lo, 0, 8);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const int nx = it + 2 < nhalf ? it + 2 : 0;
        uint4 x1, x2;
        split_f16(*reinterpret_cast<const float4*>(arow + (kh + 1) * 32), *reinterpret_cast<const float4*>(arow + (kh + 1) * 32 + 4), x1, x2);
        Half<V, 0>::run(acc, w, x1, x2, wbase + (size_t)nx * 8192, lo, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const int nx = it + 1 < nhalf ? it + 1 : 0;
      if constexpr (V == 5) __builtin_amdgcn_s_waitcnt(0x0F70);
      uint4 x1, x2;
      split_f16(*reinterpret_cast<const float4*>(arow + kh * 32), *reinterpret_cast<const float4*>(arow + kh * 32 + 4), x1, x2);
      Half<V, 0>::run(acc, w, x1, x2, wbase + (size_t)nx * 8192, lo, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (((it + STEP - 1) & 7) == 7) {
      // a repetition's 64-column tile is complete: lane 16 g + c holds row c, columns 64 wave + 16 j + 4 g .. + 3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc.hi[j][i] + acc.lo[j][i] * (1.0f / 2048.0f);
        const size_t o = (size_t)(m0 + c) * 256 + 64 * wave + 16 * j + 4 * g;
        if (Yref != nullptr) {
          const float4 r = want[j];
          nbad += (v[0] != r.x) + (v[1] != r.y) + (v[2] != r.z) + (v[3] != r.w);
        } else if (it < 8) {
          *reinterpret_cast<float4*>(Y + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (nbad != 0 && bad != nullptr) atomicAdd(bad + blockIdx.x, nbad);
  if (threadIdx.x == 0 && cyc != nullptr) cyc[blockIdx.x] = t1 - t0;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// the 256 x 256 matrix as the waves' streams, `nrep` copies end to end
static std::vector<char> pack(const std::vector<float>& W, int nrep) {
  const int nhalf = 8 * nrep;
  std::vector<char> out((size_t)4 * nhalf * 8192);
  for (int w = 0; w < 4; ++w)
    for (int kh = 0; kh < 8; ++kh) {
      char* item = out.data() + ((size_t)w * nhalf + kh) * 8192;
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        for (int j = 0; j < 4; ++j) {
          uint16_t pl[2][8];
          for (int e = 0; e < 8; ++e) {
            const float x = W[(size_t)(64 * w + 16 * j + c) * 256 + 32 * kh + 8 * g + e];
            pl[0][e] = f2h(x); pl[1][e] = f2h((x - h2f(pl[0][e])) * 2048.0f);
          }
          for (int p = 0; p < 2; ++p) memcpy(item + (2 * j + p) * 1024 + lane * 16, pl[p], 16);
        }
      }
    }
  for (int w = 0; w < 4; ++w)
    for (int rep = 1; rep < nrep; ++rep)
      memcpy(out.data() + ((size_t)w * nhalf + 8 * rep) * 8192, out.data() + (size_t)w * nhalf * 8192, 8 * 8192);
  return out;
}

struct Dev { char* P; float *X, *Yref, *Y; unsigned* bad; long long* cyc; };

template <int V>
static void run_variant(const char* name, const Dev& d, int nrep, int M, int launches, const int* delays, int ndelays) {
  for (int di = 0; di < ndelays; ++di) {
    unsigned long long total_bad = 0;
    int bad_launches = 0, bad_wgs = 0, bad_late = 0;
    std::vector<long long> cyc(512);
    double us = 0;
    for (int l = 0; l < launches; ++l) {
      CK(hipMemset(d.bad, 0, 512 * 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((probe_kernel<V>), dim3(512), dim3(256), 16 * LDA * 4, 0, d.P, d.X, d.Y, d.Yref, nrep, nrep, M, delays[di], d.bad, d.cyc);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      us += ms * 1e3;
      std::vector<unsigned> bad(512);
      CK(hipMemcpy(bad.data(), d.bad, 512 * 4, hipMemcpyDeviceToHost));
      unsigned long long nb = 0;
      for (int b = 0; b < 512; ++b) { nb += bad[b]; if (bad[b]) { ++bad_wgs; if (b >= 256) ++bad_late; } }
      total_bad += nb;
      bad_launches += nb != 0;
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    CK(hipMemcpy(cyc.data(), d.cyc, 512 * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    printf("%-44s delay x%-3d : %2d of %2d launches wrong, %8llu wrong elements in %4d workgroups (%d of them late starters); "
           "%7.1f us per launch, %5.0f cycles per half item (median workgroup)\n",
           name, delays[di], bad_launches, launches, total_bad, bad_wgs, bad_late, us / launches, (double)cyc[256] / (8.0 * nrep));
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20;
  const int M = 4096, nrep = 48;            // 48 x 256 KiB per wave stream: the 3 MB of a decoder layer per workgroup
  std::vector<float> X((size_t)M * 256), W(256 * 256);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  auto gauss = [&]() { const double u = rnd() + 1e-300, v = rnd(); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
  for (auto& x : X) x = (float)(gauss() * 1.5);
  for (auto& w : W) w = (float)((rnd() * 2 - 1) * 0.108);
  Dev d;
  std::vector<char> P = pack(W, nrep);
  CK(hipMalloc(&d.P, P.size())); CK(hipMalloc(&d.X, X.size() * 4)); CK(hipMalloc(&d.Yref, X.size() * 4)); CK(hipMalloc(&d.Y, X.size() * 4));
  CK(hipMalloc(&d.bad, 512 * 4)); CK(hipMalloc(&d.cyc, 512 * 8));
  CK(hipMemcpy(d.P, P.data(), P.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d.X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  // reference: the drained variant, 256 workgroups (one per CU), stored; checked against the host in fp64
  CK(hipMemset(d.Yref, 0, X.size() * 4));
  hipLaunchKernelGGL((probe_kernel<5>), dim3(256), dim3(256), 16 * LDA * 4, 0, d.P, d.X, d.Yref, (const float*)nullptr, 1, nrep, M, 0, (unsigned*)nullptr, (long long*)nullptr);
  CK(hipDeviceSynchronize());
  {
    std::vector<float> Y((size_t)M * 256);
    CK(hipMemcpy(Y.data(), d.Yref, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxe = 0;
    for (int m = 0; m < M; m += 7)
      for (int n = 0; n < 256; ++n) {
        double a = 0;
        for (int k = 0; k < 256; ++k) a += (double)X[(size_t)m * 256 + k] * (double)W[(size_t)n * 256 + k];
        maxe = std::max(maxe, fabs(a - (double)Y[(size_t)m * 256 + n]));
      }
    printf("reference (drained loop, one workgroup per CU) vs host fp64: max |err| %.3e\n", maxe);
  }
  // and the reference is the same from every variant run ALONE, one workgroup per CU (no partner on the SIMDs)
  {
    CK(hipMemset(d.bad, 0, 512 * 4));
    hipLaunchKernelGGL((probe_kernel<0>), dim3(256), dim3(256), 16 * LDA * 4, 0, d.P, d.X, d.Y, d.Yref, nrep, nrep, M, 0, d.bad, d.cyc);
    hipLaunchKernelGGL((probe_kernel<1>), dim3(256), dim3(256), 16 * LDA * 4, 0, d.P, d.X, d.Y, d.Yref, nrep, nrep, M, 0, d.bad, d.cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> bad(512);
    CK(hipMemcpy(bad.data(), d.bad, 512 * 4, hipMemcpyDeviceToHost));
    unsigned long long nb = 0;
    for (unsigned b : bad) nb += b;
    printf("variants A and J alone (256 workgroups, one per CU): %llu wrong elements\n", nb);
    // ... and what ONE workgroup per CU streams at (no second workgroup to share the L1 with: every byte crosses the CU's
    // L2 port): 32 KiB per half item and workgroup
    std::vector<long long> cyc(256);
    CK(hipMemcpy(cyc.data(), d.cyc, 256 * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double per_half = (double)cyc[128] / (8.0 * nrep);
    printf("one workgroup per CU (J): %.0f cycles per half item = %.1f B/clk per CU through the L2 port\n", per_half, 32768.0 / per_half);
  }
  const int delays[] = {0, 1, 4, 16, 64};
  const int nd = (int)(sizeof(delays) / sizeof(int));
  run_variant<0>("A  in place, right behind the MFMAs", d, nrep, M, launches, delays, nd);
  run_variant<1>("J  two half-item buffers (production)", d, nrep, M, launches, delays, nd);
  run_variant<2>("I  in place, one group late", d, nrep, M, launches, delays, nd);
  run_variant<3>("A + MFMA results pinned out of the weights", d, nrep, M, launches, delays, nd);
  run_variant<4>("A + 16 wait states before the loads", d, nrep, M, launches, delays, nd);
  run_variant<5>("A + stream drained per half item", d, nrep, M, launches, delays, nd);
  return 0;
}
