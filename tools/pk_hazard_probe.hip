// Round 5: THE REPRODUCTION of round 4's "flaky rows at two workgroups per CU" outside the library
// (profiles/r5_refill_hazard.txt).
//
// Two roles on every CU (blocks b and b + 256 share one; 512 workgroups of 4 waves, two waves per SIMD):
//  * AGGRESSOR (blocks >= 256): the 16-row f16 item loop of chain.hip on its own (tools/refill_hazard_probe.hip) -- 8
//    fragments of 1 KiB per wave and half item, 12 v_mfma_f32_16x16x32_f16, the next half item's fragments loaded
//      A  in place, right behind the MFMAs that read the registers      J  into a second half-item buffer (production)
//      I  in place, one group late   3  A with the MFMA results kept out of the weight registers
//      4  A + 16 wait states in front of the loads                      5  A + the stream drained per half item
//    Its OWN outputs are verified every repetition: they are exact in every variant.
//  * VICTIM (blocks < 256): never touches the item loop.  It runs the packed-f32 sequence hipcc's SLP vectoriser formed
//    in the radar attention's inner loop (rowdev.hpp radar_attn_row_g: pv * (v.x, v.y), pv * (v.z, v.w)), on fixed
//    registers through inline asm, and verifies every product:
//        v_pk_mul_f32 v[48:49], v[42:43], v[44:45]            ; v49 = pv
//        s_nop 0
//        v_add_f32    v50, v48, v49                           ; (the soft-max normaliser)
//        v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel:[0,1]   ; LOW result = v46 * v49 (the HIGH half of source 1)
//        v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel:[0,1]
//    victim modes: 0 as above; 1 s_nop 3 behind the producer; 2 the consumers broadcast the LOW half (op_sel_hi:[1,0]);
//    4 no v_add between; 5 the producer is two plain v_mul_f32; 6 the consumers are plain v_mul_f32; 7 old v49 = 3.0
//
// Measured on MI355X (ROCm 7.2; profiles/r5_refill_hazard.txt): beside aggressor A the victim's FIRST op_sel consumer
// returns exactly 0 in lanes 48..63 of its LOW result (~0.2 % of the wave-rounds; never the high result, never the
// v_add, never another lane group; 0 also when the register's old content was 3.0: it is not a stale read), with wait
// states or not; beside aggressor J never (2.5e9 wave-rounds); with the low-half broadcast or plain v_mul_f32
// consumers never.  I / 4 fail like A, 3 a hundred times less often, 5 fifteen times less often.
//
//   hipcc --offload-arch=gfx950 -O3 tools/pk_hazard_probe.hip -o tools/pk_hazard_probe
//   tools/pk_hazard_probe [launches, default 20] [J: only the combinations that must stay exact, for long runs]
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

__device__ __forceinline__ float urand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return 0.25f + (float)(s >> 9) * (1.0f / 8388608.0f);        // [0.25, 1.25)
}

// MODE 0: the sequence as hipcc emitted it (s_nop 0 behind the producer, the consumers read the HIGH half through op_sel)
// MODE 1: four wait states behind the producer
// MODE 2: the consumers broadcast the LOW half (op_sel_hi:[1,0]) of a pair whose low half carries pv
// MODE 3: as 0 with a vector-memory wait between producer and consumers (a load in flight, as in the library)
template <int MODE>
__device__ __forceinline__ void victim_round(unsigned& seed, const float* __restrict__ gsrc, unsigned& nbad, unsigned* detail, int lane) {
  const float a0 = urand(seed), a1 = urand(seed), b0 = urand(seed), b1 = urand(seed);
  const float c0 = urand(seed), c1 = urand(seed), c2 = urand(seed), c3 = urand(seed);
  float r0, r1, r2, r3, lsum;
  if constexpr (MODE == 2) {
    asm volatile(
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "v_pk_mul_f32 v[48:49], v[42:43], v[44:45]\n\t"              // v48 = a0 b0 (pv here), v49 = a1 b1
        "s_nop 0\n\t"
        "v_add_f32 v50, v48, v49\n\t"
        "v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel_hi:[1,0]\n\t"
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");
    const float pv = __fmul_rn(a0, b0);
    nbad += (r0 != __fmul_rn(c0, pv)) + (r1 != __fmul_rn(c1, pv)) + (r2 != __fmul_rn(c2, pv)) + (r3 != __fmul_rn(c3, pv));
    return;
  }
#define SEQ(BEHIND, WAIT)                                                                                       \
    asm volatile(                                                                                               \
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"               \
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"                  \
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"                                                              \
        "v_pk_mul_f32 v[48:49], v[42:43], v[44:45]\n\t"              /* v48 = a0 b0, v49 = a1 b1 (= pv) */      \
        BEHIND                                                                                                  \
        "v_add_f32 v50, v48, v49\n\t"                                                                           \
        WAIT                                                                                                    \
        "v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel:[0,1]\n\t"                                            \
        "v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel:[0,1]\n\t"                                            \
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50" \
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)                                               \
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                \
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53")
  if constexpr (MODE == 4) {          // no v_add between: producer, s_nop 0, consumer xy, consumer zw, (l last)
    asm volatile(
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "v_pk_mul_f32 v[48:49], v[42:43], v[44:45]\n\t"
        "s_nop 0\n\t"
        "v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel:[0,1]\n\t"
        "v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel:[0,1]\n\t"
        "v_add_f32 v50, v48, v49\n\t"
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");
  } else if constexpr (MODE == 5) {   // the producer is a plain v_mul_f32 into v49 (and one into v48)
    asm volatile(
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "v_mul_f32 v48, v42, v44\n\tv_mul_f32 v49, v43, v45\n\t"
        "s_nop 0\n\t"
        "v_add_f32 v50, v48, v49\n\t"
        "v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel:[0,1]\n\t"
        "v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel:[0,1]\n\t"
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");
  } else if constexpr (MODE == 6) {   // the consumers are plain v_mul_f32 reading v49
    asm volatile(
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "v_pk_mul_f32 v[48:49], v[42:43], v[44:45]\n\t"
        "s_nop 0\n\t"
        "v_add_f32 v50, v48, v49\n\t"
        "v_mul_f32 v46, v46, v49\n\tv_mul_f32 v47, v47, v49\n\tv_mul_f32 v52, v52, v49\n\tv_mul_f32 v53, v53, v49\n\t"
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");
  } else if constexpr (MODE == 7) {   // the stale candidate is NOT zero: v49 preset to 3.0 -> a stale read gives 3 c0
    asm volatile(
        "v_mov_b32 v42, %9\n\tv_mov_b32 v43, %10\n\tv_mov_b32 v44, %11\n\tv_mov_b32 v45, %12\n\t"
        "v_mov_b32 v46, %5\n\tv_mov_b32 v47, %6\n\tv_mov_b32 v52, %7\n\tv_mov_b32 v53, %8\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0x40400000\n\t"
        "v_pk_mul_f32 v[48:49], v[42:43], v[44:45]\n\t"
        "s_nop 0\n\t"
        "v_add_f32 v50, v48, v49\n\t"
        "v_pk_mul_f32 v[46:47], v[46:47], v[48:49] op_sel:[0,1]\n\t"
        "v_pk_mul_f32 v[52:53], v[52:53], v[48:49] op_sel:[0,1]\n\t"
        "v_mov_b32 %0, v46\n\tv_mov_b32 %1, v47\n\tv_mov_b32 %2, v52\n\tv_mov_b32 %3, v53\n\tv_mov_b32 %4, v50"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(lsum)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53");
    if (r0 == __fmul_rn(c0, 3.0f)) atomicAdd(detail + 21, 1u);
  } else if constexpr (MODE == 0) SEQ("s_nop 0\n\t", "");
  else if constexpr (MODE == 1) SEQ("s_nop 3\n\t", "");
  else {
    // a load in flight whose arrival the sequence waits for between producer and consumers
    float4 dummy = *reinterpret_cast<const float4*>(gsrc + ((seed >> 8) & 0xffff) * 4);
    SEQ("s_nop 0\n\t", "s_waitcnt vmcnt(0)\n\t");
    if (dummy.x == 1234.5f) nbad += 1000000;
  }
#undef SEQ
  const float pv = __fmul_rn(a1, b1);
  const bool e0 = r0 != __fmul_rn(c0, pv), e1 = r1 != __fmul_rn(c1, pv), e2 = r2 != __fmul_rn(c2, pv), e3 = r3 != __fmul_rn(c3, pv);
  const bool el = lsum != __fadd_rn(__fmul_rn(a0, b0), pv);
  nbad += e0 + e1 + e2 + e3 + el;
  if (e0) atomicAdd(detail + 0 * 4 + (lane >> 4), 1u);
  if (e1) atomicAdd(detail + 1 * 4 + (lane >> 4), 1u);
  if (e2) atomicAdd(detail + 2 * 4 + (lane >> 4), 1u);
  if (e3) atomicAdd(detail + 3 * 4 + (lane >> 4), 1u);
  if (el) atomicAdd(detail + 16 + (lane >> 4), 1u);
  if (e0 && r0 == 0.0f) atomicAdd(detail + 20, 1u);             // ... and how many of them are exactly 0 (the stale v49)
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
template <int V, int VM = 0>
__global__ __launch_bounds__(256, 2) void probe_kernel(const char* __restrict__ W, const float* __restrict__ X,
                                                       float* __restrict__ Y, const float* __restrict__ Yref,
                                                       int nrep, int nrep_packed, int M, int delay_mul,
                                                       unsigned* __restrict__ bad, long long* __restrict__ cyc) {
  extern __shared__ __align__(16) float lds[];          // [16][LDA]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x < 256 && delay_mul < 0) {
    // victim role: the packed op_sel sequence, verified, for about as long as the partner's stream lasts
    unsigned seed = blockIdx.x * 9781u + threadIdx.x * 6271u + 17u, nb = 0;
    for (int it = 0; it < -delay_mul; ++it) victim_round<VM>(seed, X, nb, bad + 512, lane);
    if (nb) atomicAdd(bad + blockIdx.x, nb);
    return;
  }
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
  CK(hipMalloc(&d.bad, 600 * 4)); CK(hipMalloc(&d.cyc, 512 * 8));
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
  }
  auto combo = [&](auto kern, const char* pname, const char* vname) {
    unsigned long long tot = 0, totp = 0;
    std::vector<unsigned> bad(600);
    for (int l = 0; l < launches; ++l) {
      CK(hipMemset(d.bad, 0, 600 * 4));
      hipLaunchKernelGGL(kern, dim3(512), dim3(256), 16 * LDA * 4, 0, d.P, d.X, d.Y, d.Yref, nrep, nrep, M, -60000, d.bad, d.cyc);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(bad.data(), d.bad, 600 * 4, hipMemcpyDeviceToHost));
      for (int b = 0; b < 256; ++b) tot += bad[b];
      for (int b = 256; b < 512; ++b) totp += bad[b];
    }
    printf("partner %-44s victim %-40s: wrong packed products %9llu (partners' wrong outputs %llu)  last launch [x y z w][lane group]:", pname, vname, tot, totp);
    for (int i = 0; i < 22; ++i) printf(" %u", bad[512 + i]);
    printf("\n");
  };
  const char* PN[6] = {"A  in place", "J  two half-item buffers", "I  in place, one group late", "A + results pinned out of the weights",
                       "A + 16 wait states before the loads", "A + stream drained per half item"};
  const char* VN[3] = {"s_nop 0, op_sel:[0,1] (hipcc)", "s_nop 3 behind the producer", "consumers broadcast the LOW half"};
  if (argc > 2) { combo(probe_kernel<1, 0>, PN[1], VN[0]); combo(probe_kernel<1, 4>, PN[1], "no v_add between"); combo(probe_kernel<0, 2>, PN[0], VN[2]); combo(probe_kernel<0, 6>, PN[0], "consumers plain v_mul_f32"); return 0; }
  combo(probe_kernel<0, 0>, PN[0], VN[0]);
  combo(probe_kernel<0, 4>, PN[0], "no v_add between producer and consumers");
  combo(probe_kernel<0, 5>, PN[0], "producer = two plain v_mul_f32");
  combo(probe_kernel<0, 6>, PN[0], "consumers = plain v_mul_f32 reading v49");
  combo(probe_kernel<0, 7>, PN[0], "old v49 = 3.0 (x == 3 c0 counted last)");
  combo(probe_kernel<1, 0>, PN[1], VN[0]);
  combo(probe_kernel<1, 4>, PN[1], "no v_add between producer and consumers");
  return 0;
}
