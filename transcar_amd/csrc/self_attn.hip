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
#include <string.h>

#include "kernels.hpp"
#include "rowdev.hpp"

namespace tc {

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int SA_NW = 8;
constexpr float SA_TAU = 8.0f;       // re-centre when a score exceeds the running reference by 2^8

// max over the four lanes c, c+16, c+32, c+48 (they hold the same query column) on
// the VALU: gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves /
// odd-even 16-lane rows between two registers, so swap(x, x) followed by one v_max is
// an xor-32 / xor-16 all-reduce.  (__shfl_xor goes through the LDS crossbar, and two
// of them sat in the dependency chain S -> max -> exp -> PV of every key tile.)
// ROUND 4 FIX.  Rounds 2-3 wrote this with the builtins: r = __builtin_amdgcn_permlane32_swap(x, x); x =
// fmaxf(r[0], r[1]); ...  hipcc (ROCm 7.2) folds maxnum(extractvalue 0, extractvalue 1) of a swap to element 0 --
// the emitted code has NO v_max between the two swaps, and the "maximum" every lane got was lane group 0's own value
// (tools/permlane_probe: wrong in all 64 lanes; a 10-line repro, the operands made opaque or not).  The fp32 kernel
// never showed it: softmax is shift invariant and 2^(score - any of the tile's scores) fits fp32, so every parity
// test passed with the wrong reference.  The f16x2 kernel turns probabilities into f16 planes (<= 65 504):
// test_sdpa_lazy_recentring_extreme_scores[huge_negative_start-f16x2] came out NaN.  Inline asm, with the wait
// states of "VALU write -> v_permlane read" (2) inside the string: hipcc pads nothing in there.
__device__ __forceinline__ float max_lanes_16_32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1"
               : "+v"(a), "+v"(b));
  return a;
}

struct KVFrag { float4 ka, kb, v0, v1; };

// per-lane pointers of a wave's current key tile: K rows (lane r = key, 8 channels of group g) and
// the two V^T channel rows (r, r + 16; 4 keys of group g); advanced by whole strides, no index math
struct KVPtr { const float* k; const float* v0; const float* v1; };

__device__ __forceinline__ KVFrag load_kv(const KVPtr& p) {
  KVFrag f;
  f.ka = ld4(p.k); f.kb = ld4(p.k + 4);
  f.v0 = ld4(p.v0); f.v1 = ld4(p.v1);
  return f;
}

// "does any of the four scores exceed tau (> 0)?" on the integer unit: for non-NaN floats x > tau is the
// signed comparison of the bit patterns (negative floats are negative integers), and v_max3_i32 needs no
// canonicalising v_max per MFMA-produced operand as an fmaxf chain does.  (NOT inline asm: hipcc's hazard
// recogniser does not see inside it and the MFMA -> VALU read needs its wait states.)
__device__ __forceinline__ bool any_above(float a, float b, float c, float d, float tau) {
  const int m = max(max(max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)), __builtin_bit_cast(int, c)),
                    __builtin_bit_cast(int, d));
  return m > __builtin_bit_cast(int, tau);
}

// Running softmax state of one 16-query sub-tile: O^T (2 x 16 channels x 16 queries), the partial
// normaliser and the NEGATED running reference as the four equal entries of an MFMA C operand.
struct SAState { f32x4 o0, o1, negm; float l; };

// One 16-key tile against one 16-query sub-tile.
//
// Lazy re-centring: on gfx950 every VALU instruction costs the f32 MFMA pipe its issue cycles (the
// f32 matrix instructions run on the same FMA lanes: tools/issue_probe.hip), so the softmax
// bookkeeping is kept off the common path.  The scores leave the MFMA chain already relative to the
// running reference (-negm is the chain's C operand), and as long as none exceeds it by more than
// SA_TAU (p <= 2^SA_TAU: harmless in fp32) a tile costs max + compare + 4 exp + 3 add.  Only a tile
// that does (always the wave's first) pays for the cross-lane max, alpha and the rescale of O.
// softmax is shift invariant: the result is the running-max formulation's up to rounding.
// nvalid < 16 (the ragged last tile): keys >= nvalid are masked.
template <bool DROP>
__device__ __forceinline__ void sa_tile(const KVFrag& f, const float4& qa, const float4& qb, SAState& st,
                                        bool first, int nvalid, int g, const DropK& drop, unsigned drop_base) {
  f32x4 s = MFMA4(f.ka.x, qa.x, st.negm);
  s = MFMA4(f.ka.y, qa.y, s); s = MFMA4(f.ka.z, qa.z, s); s = MFMA4(f.ka.w, qa.w, s);
  s = MFMA4(f.kb.x, qb.x, s); s = MFMA4(f.kb.y, qb.y, s);
  s = MFMA4(f.kb.z, qb.z, s); s = MFMA4(f.kb.w, qb.w, s);
  // s[i] = log2(e) * S^T[key0 + 4g + i][query r] + negm
  float s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
  float4 v0 = f.v0, v1 = f.v1;
  if (nvalid < 16) {                                   // wave-uniform
    const int kk = 4 * g;
    if (kk + 0 >= nvalid) { s0 = -INFINITY; v0.x = 0.f; v1.x = 0.f; }
    if (kk + 1 >= nvalid) { s1 = -INFINITY; v0.y = 0.f; v1.y = 0.f; }
    if (kk + 2 >= nvalid) { s2 = -INFINITY; v0.z = 0.f; v1.z = 0.f; }
    if (kk + 3 >= nvalid) { s3 = -INFINITY; v0.w = 0.f; v1.w = 0.f; }
  }
  if (first || __builtin_amdgcn_ballot_w64(any_above(s0, s1, s2, s3, SA_TAU)) != 0) {
    const float mx = max_lanes_16_32(fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)));   // finite: key 0 of every tile is valid
    const float delta = first ? mx : fmaxf(mx, 0.0f);
    const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // first tile: l = O = 0
    s0 -= delta; s1 -= delta; s2 -= delta; s3 -= delta;
    st.l *= alpha;
    st.o0 *= alpha; st.o1 *= alpha;      // (scalar multiplies: the library is built without packed-f32 ops, Makefile)
    st.negm -= delta;
  }
  const float p0 = __builtin_amdgcn_exp2f(s0), p1 = __builtin_amdgcn_exp2f(s1);
  const float p2 = __builtin_amdgcn_exp2f(s2), p3 = __builtin_amdgcn_exp2f(s3);
  st.l += (p0 + p1) + (p2 + p3);
  float d0 = p0, d1 = p1, d2 = p2, d3 = p3;
  if (DROP) {
    const unsigned dm = drop_keep4(drop.seed, drop.site, drop_base, drop.thr);     // the lane's four consecutive keys
    d0 = (dm & 1u) ? p0 * drop.scale : 0.0f;
    d1 = (dm & 2u) ? p1 * drop.scale : 0.0f;
    d2 = (dm & 4u) ? p2 * drop.scale : 0.0f;
    d3 = (dm & 8u) ? p3 * drop.scale : 0.0f;
  }
  // O^T[d][q] += V^T[d][key] P^T[key][q]
  st.o0 = MFMA4(v0.x, d0, st.o0); st.o1 = MFMA4(v1.x, d0, st.o1);
  st.o0 = MFMA4(v0.y, d1, st.o0); st.o1 = MFMA4(v1.y, d1, st.o1);
  st.o0 = MFMA4(v0.z, d2, st.o0); st.o1 = MFMA4(v1.z, d2, st.o1);
  st.o0 = MFMA4(v0.w, d3, st.o0); st.o1 = MFMA4(v1.w, d3, st.o1);
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
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 16 * QT, h = blockIdx.y, b = blockIdx.z;
  const size_t brow = (size_t)b * Q;

  float4 qa[QT], qb[QT];
  SAState st[QT];
  unsigned dbase[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = min(q0 + 16 * u + r, Q - 1);
    const float* qp = q + (brow + qrow) * ld + h * 32 + 8 * g;
    qa[u] = ld4(qp); qb[u] = ld4(qp + 4);
    st[u].o0 = f32x4{0.f, 0.f, 0.f, 0.f}; st[u].o1 = f32x4{0.f, 0.f, 0.f, 0.f};
    st[u].negm = f32x4{0.f, 0.f, 0.f, 0.f}; st[u].l = 0.0f;
    // (a batch of frames with per-sample seeds: the index is the one sample b has when it is launched alone)
    dbase[u] = DROP ? (((drop.rows_per_sample ? 0u : (unsigned)b) * gridDim.y + h) * Q + (unsigned)qrow) * Q + 4 * g : 0u;
  }
  if (DROP) drop.seed += (unsigned long long)b * drop.seed_stride;
  // the wave's key tiles: wave, wave + NW, ... among the nfull whole tiles (two fragment buffers,
  // each loaded one tile ahead), then the ragged tile if it is this wave's turn
  const int nfull = Q >> 4;
  const int n = wave < nfull ? (nfull - wave + SA_NW - 1) / SA_NW : 0;
  const size_t kstep = (size_t)SA_NW * 16 * ld;
  KVPtr p;
  p.k = k + (brow + wave * 16 + r) * ld + h * 32 + 8 * g;
  p.v0 = vt + ((size_t)b * C + h * 32 + r) * ldt + wave * 16 + 4 * g;
  p.v1 = p.v0 + (size_t)16 * ldt;
  auto advance = [&]() { p.k += kstep; p.v0 += SA_NW * 16; p.v1 += SA_NW * 16; };
  auto tile = [&](const KVFrag& f, int i, int nvalid) {
#pragma unroll
    for (int u = 0; u < QT; ++u)
      sa_tile<DROP>(f, qa[u], qb[u], st[u], i == 0, nvalid, g, drop, dbase[u] + (unsigned)(wave + i * SA_NW) * 16u);
  };
  KVFrag fa, fb;
  if (n > 0) fa = load_kv(p);
  int i = 0;
  for (; i + 2 <= n; i += 2) {
    advance();
    fb = load_kv(p);                       // i + 1 < n
    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this tile's MFMAs
    tile(fa, i, 16);
    advance();
    if (i + 2 < n) fa = load_kv(p);
    __builtin_amdgcn_sched_barrier(0);
    tile(fb, i + 1, 16);
  }
  if (i < n) { tile(fa, i, 16); ++i; }
  const int rag = Q & 15;
  if (rag != 0 && wave == (nfull % SA_NW)) {           // i == n: this wave's next tile is the ragged one
    KVPtr pr;
    pr.k = k + (brow + min(nfull * 16 + r, Q - 1)) * ld + h * 32 + 8 * g;
    pr.v0 = vt + ((size_t)b * C + h * 32 + r) * ldt + nfull * 16 + 4 * g;    // ldt >= 16 * (nfull + 1)
    pr.v1 = pr.v0 + (size_t)16 * ldt;
    const KVFrag fr = load_kv(pr);
    tile(fr, i, rag);
    ++i;
  }
  const bool idle = i == 0;                             // a wave without a tile: weight 0 in the merge
  // merge the key slices
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (g == 0) sm_m[wave][u][r] = idle ? -INFINITY : -st[u].negm[0];
    sm_l[wave][u][lane] = st[u].l;
    sm_o[wave][u][0][lane] = make_float4(st[u].o0[0], st[u].o0[1], st[u].o0[2], st[u].o0[3]);
    sm_o[wave][u][1][lane] = make_float4(st[u].o1[0], st[u].o1[1], st[u].o1[2], st[u].o1[3]);
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

// ---- the same core on the f16 MATRIX CORES, fp32-accurate (round 4) ------------------------------------------
// After the chains moved to the matrix cores this kernel was the largest share of a frame (5 x 84 us of 1 290 us per
// nine frames), 58 % of its SIMD cycles f32 MFMAs on the vector pipe.  Operands as two f16 planes (hi = f16(x),
// lo = f16(x - hi)), three v_mfma_f32_16x16x32_f16 per product (hi hi + lo hi + hi lo):
//   * d = 32 is ONE MFMA's k: S^T of a 16-key tile is three MFMAs into one accumulator;
//   * the PV product sums over KEYS, 32 per MFMA: key tiles go in pairs, and the rows of the two K tiles are chosen so
//     that a lane's eight scores (4 + 4) are the eight CONSECUTIVE keys 8g .. 8g + 7 of the pair (tile A holds keys
//     8g' + i, tile B keys 8g' + 4 + i at row 4g' + i): the probabilities are the PV product's B operand as they
//     stand -- split into planes in registers (8 values per lane) -- and V^T's operand is one 16-byte read per plane.
// FIRST FORM (measured, not kept: tools/experiments/README.md): the fp32 kernel's structure -- 8 waves split the keys,
// planes built by a conversion pass -- 78 us + 10 us against 84 us per nine frames: 203 VGPRs, one workgroup per CU, and
// a wave's 3.6 dependent key pairs + the merge are latency, not matrix time.  SECOND FORM (below, in tc_head_forward):
// every wave walks all keys, K / V^T staged through LDS: 43 us.
typedef _Float16 sa_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sa_f16x2 __attribute__((ext_vector_type(2)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sa_f16x8, (a)), __builtin_bit_cast(sa_f16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned sa_pk(float a, float b) {
  const sa_f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// ---- round 4, second form: every wave walks ALL keys; K / V^T staged through LDS as MFMA fragments ------------
// The split-key form above is bound by the latency of a wave's 3.6 dependent key pairs and by its merge.  Here a wave
// owns QT 16-query sub-tiles for the whole key range (29 pairs of 32 keys: a real loop to pipeline, no merge, the
// normaliser is two cross-lane adds at the end), and the NW = 4 waves of a workgroup share every K / V^T fragment:
//   * the workgroup reads the chain's fp32 q | k rows and V^T ONCE per pair (one 32-byte task per thread: waves 0-1
//     the 32 keys x 4 channel groups of K, waves 2-3 the 32 channels x 4 key groups of V^T), splits them into the two
//     f16 planes in registers and writes them to LDS IN FRAGMENT ORDER ([pair][ka_h ka_l kb_h kb_l v0_h v0_l v1_h
//     v1_l][lane] x 16 B): a wave's operand read is one conflict-free ds_read_b128 per fragment; no conversion pass,
//     no plane buffers in HBM -- the kernel takes exactly what the fp32 core takes;
//   * chunks of two pairs, double buffered (32 KB of LDS): the loads of chunk c + 1 are issued before chunk c is
//     consumed, converted and stored after it, ONE barrier per chunk;
//   * the block index is mapped so that the workgroups of one (batch, head) -- which read the same 230 KB of K / V^T
//     -- run on ONE XCD (block i lands on XCD i % 8), and the workgroups that hold a sample's last, nearly empty
//     query tile (Q = 900: 7 x 128 queries + 4) come last in the grid.
// Per pair and sub-tile a SIMD issues ~90 VALU instructions (8 v_exp at quarter rate, the fp32 -> plane split of the
// probabilities) beside 12 matrix-core MFMAs of 16 cycles: VALU bound, the matrix cores a third busy.
#ifndef SX_NW_VALUE
#define SX_NW_VALUE 4
#endif
#ifndef SX_QT_VALUE
#define SX_QT_VALUE 2
#endif
constexpr int SX_NW = SX_NW_VALUE, SX_CP = 2;
constexpr int SX_PG_ROWS = 32;      // rows of a pre-gather workgroup (8 per wave: two projection rounds of four rows)
static_assert(SX_PG_ROWS % (4 * SX_NW) == 0, "a pre-gather wave projects four rows at a time");
#ifndef SX_NPC
#define SX_NPC 1
#endif
#ifndef SX_OCC
#define SX_OCC 3
#endif

// The staged form keeps the LOW planes UNSCALED (lo = f16(x - hi)): the matrix cores honour f16 subnormals
// (measured: tools/r4_attn_time.py, error unchanged at 3e-7), so a lo plane's error is at most 2^-25 ABSOLUTE per element
// -- below fp32's own rounding for operands of magnitude >= 0.5, and the scores / probabilities of this kernel are
// O(0.1 .. 256) -- and all three products of a split product (hi hi + lo hi + hi lo) chain into ONE accumulator: no
// 2^-11 recombination, no second accumulator set, no scale multiplies in the split.
// running state of one 16-query sub-tile: O^T, the normaliser (an MFMA too with SX_L_MFMA: an all-ones A operand sums a
// pair's probabilities over the keys) and the negated reference as the score products' C operand
#ifndef SX_L_MFMA
#define SX_L_MFMA 0
#endif
struct SXState { f32x4 o0, o1, lsum, negm; float l; };

// x (8 fp32) -> hi plane (round to nearest f16) and lo plane f16(x - hi): one v_cvt_pk per two elements + one
// v_fma_mix{lo,hi}_f16 per element (the packed hi half is an operand as it stands)
__device__ __forceinline__ void sx_split1(const float* x, float4& hi, float4& lo) {
  unsigned hh[4], ll[4];
  float m1 = -1.0f;
  asm("" : "+s"(m1));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = sa_pk(x[2 * i], x[2 * i + 1]);
    asm volatile("" : "+v"(hh[i]));      // opaque: the halves are read out of the PACKED register, not re-converted
    const sa_f16x2 v = __builtin_bit_cast(sa_f16x2, hh[i]);
    // (round 6: the multiplier opaque -- with a literal -1 hipcc turns fma(h, -1, x) into v_cvt_f32_f16 + v_sub_f32, two
    // instructions per value; as it stands it is ONE v_fma_mix_f32 reading the half out of the packed register.  x - h is
    // exact either way: same bits.)
    ll[i] = sa_pk(fmaf((float)v[0], m1, x[2 * i]), fmaf((float)v[1], m1, x[2 * i + 1]));
  }
  hi = make_float4(__uint_as_float(hh[0]), __uint_as_float(hh[1]), __uint_as_float(hh[2]), __uint_as_float(hh[3]));
  lo = make_float4(__uint_as_float(ll[0]), __uint_as_float(ll[1]), __uint_as_float(ll[2]), __uint_as_float(ll[3]));
}
__device__ __forceinline__ void sx_split(const float4& x0, const float4& x1, float4& hi, float4& lo) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  sx_split1(x, hi, lo);
}

// One chunk of NP key pairs (32 keys each) against the wave's QT sub-tiles, as ONE straight-line block: all score
// products first, ONE lazy re-centring test for the whole chunk (a chunk is re-centred as a unit: softmax is shift
// invariant), then exponentials, plane split and the PV products.
// DROP (train-mode statistics of the frozen decoder: the trainer's batched look-ahead): the mask of self_attn_kernel --
// drop_keep(seed, site, ((b * H + h) * Q + query) * Q + key) on the normalised probability, the normaliser does not see
// it -- on the lane's eight consecutive keys: two hashes (drop_keep4).  dbase[u]: the index of (query r of sub-tile u,
// key 8 g) at pair 0; key0 = the first key of the chunk's first pair.
template <int QT, int NP, bool DROP = false>
__device__ __forceinline__ void sx_chunk(const float4 (*fb)[8][64], int lane, int g, const float4* q_h, const float4* q_l,
                                         SXState* st, bool first, int nvalid_last, const DropK* drop = nullptr,
                                         const unsigned* dbase = nullptr, unsigned key0 = 0u) {
  float sc[NP][QT][8];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const float4 ka_h = fb[pp][0][lane], ka_l = fb[pp][1][lane], kb_h = fb[pp][2][lane], kb_l = fb[pp][3][lane];
    f32x4 sa[QT], sb[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) { sa[u] = MFMAH(ka_h, q_h[u], st[u].negm); sb[u] = MFMAH(kb_h, q_h[u], st[u].negm); }
#pragma unroll
    for (int u = 0; u < QT; ++u) { sa[u] = MFMAH(ka_l, q_h[u], sa[u]); sb[u] = MFMAH(kb_l, q_h[u], sb[u]); }
#pragma unroll
    for (int u = 0; u < QT; ++u) { sa[u] = MFMAH(ka_h, q_l[u], sa[u]); sb[u] = MFMAH(kb_h, q_l[u], sb[u]); }
    // sc[i] = log2(e) * S^T[key0 + 8g + i][query r] + negm
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) { sc[pp][u][i] = sa[u][i]; sc[pp][u][4 + i] = sb[u][i]; }
  }
  if (nvalid_last < 32) {                                // wave-uniform: the ragged last pair (its V planes are 0 there)
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) if (8 * g + i >= nvalid_last) sc[NP - 1][u][i] = -INFINITY;
  }
  // "does any score exceed tau?" on the integer unit, branch-free (any_above's trick: one max3 tree over all of them)
  int top = __builtin_bit_cast(int, sc[0][0][0]);
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
      for (int i = 0; i < 8; i += 2) top = max(max(top, __builtin_bit_cast(int, sc[pp][u][i])), __builtin_bit_cast(int, sc[pp][u][i + 1]));
  const bool above = top > __builtin_bit_cast(int, SA_TAU);
  if (first || __builtin_amdgcn_ballot_w64(above) != 0) {
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float mx = sc[0][u][0];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, sc[pp][u][i]);
      mx = max_lanes_16_32(mx);                           // finite: key 0 of every pair is a real key
      const float delta = first ? mx : fmaxf(mx, 0.0f);
      const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // first chunk: l = O = 0
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int i = 0; i < 8; ++i) sc[pp][u][i] -= delta;
      st[u].o0 *= alpha; st[u].o1 *= alpha; st[u].lsum *= alpha; st[u].l *= alpha;
      st[u].negm -= delta;
    }
  }
  const unsigned one2 = 0x3C003C00u;                       // (1.0h, 1.0h)
  const float4 ones = make_float4(__uint_as_float(one2), __uint_as_float(one2), __uint_as_float(one2), __uint_as_float(one2));
  (void)ones;
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const float4 v0h = fb[pp][4][lane], v0l = fb[pp][5][lane], v1h = fb[pp][6][lane], v1l = fb[pp][7][lane];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(sc[pp][u][i]);
      if (!SX_L_MFMA) st[u].l += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if constexpr (DROP) {
        const unsigned idx = dbase[u] + key0 + 32u * (unsigned)pp;
        const unsigned m0 = drop_keep4(drop->seed, drop->site, idx, drop->thr), m1 = drop_keep4(drop->seed, drop->site, idx + 4u, drop->thr);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          p[i] = ((m0 >> i) & 1u) ? p[i] * drop->scale : 0.0f;
          p[4 + i] = ((m1 >> i) & 1u) ? p[4 + i] * drop->scale : 0.0f;
        }
      }
      // P^T as two f16 planes: the lane's eight keys are the MFMA's eight k slots
      float4 p_h, p_l;
#ifdef SX_NO_PLO      // experiment (upper bound of VERDICT r4 item 8's "hi plane only for small probabilities"): no lo plane of P at all
      {
        unsigned hh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hh[i] = sa_pk(p[2 * i], p[2 * i + 1]);
        p_h = make_float4(__uint_as_float(hh[0]), __uint_as_float(hh[1]), __uint_as_float(hh[2]), __uint_as_float(hh[3]));
        p_l = p_h;
      }
#else
      sx_split1(p, p_h, p_l);
#endif
      // O^T[d][q] += V^T[d][key] P^T[key][q];  l[q] += sum over the keys
      st[u].o0 = MFMAH(v0h, p_h, st[u].o0); st[u].o1 = MFMAH(v1h, p_h, st[u].o1);
      if (SX_L_MFMA) st[u].lsum = MFMAH(ones, p_h, st[u].lsum);
      st[u].o0 = MFMAH(v0l, p_h, st[u].o0); st[u].o1 = MFMAH(v1l, p_h, st[u].o1);
#ifndef SX_NO_PLO
      st[u].o0 = MFMAH(v0h, p_l, st[u].o0); st[u].o1 = MFMAH(v1h, p_l, st[u].o1);
      if (SX_L_MFMA) st[u].lsum = MFMAH(ones, p_l, st[u].lsum);
#endif
    }
  }
}

template <int QT, bool DROP = false>
__global__ __launch_bounds__(SX_NW * 64) __attribute__((amdgpu_waves_per_eu(SX_OCC, SX_OCC))) void self_attn_x_kernel(
    const float* __restrict__ q, const float* __restrict__ k, int ld, const float* __restrict__ vt, int ldt,
    float* __restrict__ out, int ldo, int Q, int C, int H, int BH, DropK drop, PreGatherK pg) {
  constexpr int QW = 16 * QT * SX_NW;                       // queries per workgroup
  __shared__ float4 frag[2][SX_CP][8][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (!DROP) {
    // round 6: the LAST pg.nblocks workgroups of the launch gather the camera taps of the decoder chain that follows
    // (rowdev.hpp cam_pregather_rows: SX_PG_ROWS rows per workgroup).  576 attention workgroups of a nine-frame launch
    // leave one of three slots free on 192 CUs: the gather workgroups start there at once, beside the VALU-bound cores.
    const int nattn = (int)gridDim.x - pg.nblocks;
    if ((int)blockIdx.x >= nattn) {
      const int r0 = ((int)blockIdx.x - nattn) * SX_PG_ROWS + wave * (SX_PG_ROWS / SX_NW);
      cam_pregather_rows<4>(pg, r0, SX_PG_ROWS / SX_NW, lane);
      return;
    }
  }
  const int r = lane & 15, g = lane >> 4;
  // work item: (batch * head, query group)
  const int Gf = Q / QW, nfull = BH * Gf;
  int bh, qg;
  {
    const int idx = blockIdx.x;
    if (idx >= nfull) { bh = idx - nfull; qg = Gf; }                      // the ragged query group of a (batch, head)
    else if ((BH & 7) == 0) { const int slot = idx >> 3; bh = (slot / Gf) * 8 + (idx & 7); qg = slot % Gf; }
    else { bh = idx / Gf; qg = idx % Gf; }
  }
  const int b = bh / H, h = bh - b * H;
  const size_t brow = (size_t)b * Q;
  const int q0 = qg * QW + wave * 16 * QT;
  const bool active = q0 < Q;                                             // a wave without a real query only stages
  // the thread's staging tasks of a pair: 256 tasks of 32 bytes (tasks 0..127: key t >> 2 of K, channel group t & 3;
  // 128..255: channel (t - 128) >> 2 of V^T, key group t & 3), task t + i * threads for thread t
  constexpr int NT = SX_NW * 64, TPT = NT >= 256 ? 1 : 256 / NT;
  const bool stager = tid < 256;
  bool is_k[TPT];
  int gg[TPT], rowi[TPT], dfrag[TPT], dlane[TPT];
  const float* src[TPT];
#pragma unroll
  for (int tk = 0; tk < TPT; ++tk) {
    const int t = tid + tk * NT;
    is_k[tk] = __builtin_amdgcn_readfirstlane(t) < 128;                 // (wave-uniform: a wave's 64 tasks lie on one side)
    const int tt = t & 127;
    gg[tk] = tt & 3; rowi[tk] = tt >> 2;                                // K: key `rowi` of the pair; V^T: channel `rowi`
    dfrag[tk] = is_k[tk] ? 2 * ((rowi[tk] >> 2) & 1) : 4 + 2 * (rowi[tk] >> 4);
    dlane[tk] = is_k[tk] ? 16 * gg[tk] + 4 * (rowi[tk] >> 3) + (rowi[tk] & 3) : 16 * gg[tk] + (rowi[tk] & 15);
    src[tk] = is_k[tk] ? k + brow * ld + h * 32 + 8 * gg[tk] : vt + ((size_t)b * C + h * 32 + rowi[tk]) * ldt + 8 * gg[tk];
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int j, int tk, float4& x0, float4& x1) {
    const int key0 = 32 * j;
    if (is_k[tk]) {
      const float* p = src[tk] + (size_t)min(key0 + rowi[tk], Q - 1) * ld;   // keys >= Q: any finite row (their scores are masked)
      x0 = ld4(p); x1 = ld4(p + 4);
    } else {
      const int kk = key0 + 8 * gg[tk];
      const float* p = src[tk] + key0;
      x0 = kk + 3 < ldt ? ld4(p) : zero4;
      x1 = kk + 7 < ldt ? ld4(p + 4) : zero4;
      if (kk + 7 >= Q) {                                                  // keys >= Q weigh nothing
        if (kk + 0 >= Q) x0.x = 0.f;
        if (kk + 1 >= Q) x0.y = 0.f;
        if (kk + 2 >= Q) x0.z = 0.f;
        if (kk + 3 >= Q) x0.w = 0.f;
        if (kk + 4 >= Q) x1.x = 0.f;
        if (kk + 5 >= Q) x1.y = 0.f;
        if (kk + 6 >= Q) x1.z = 0.f;
        if (kk + 7 >= Q) x1.w = 0.f;
      }
    }
  };
  auto put = [&](int buf, int pp, int tk, const float4& x0, const float4& x1) {
    float4 hi, lo;
    sx_split(x0, x1, hi, lo);
    frag[buf][pp][dfrag[tk]][dlane[tk]] = hi;
    frag[buf][pp][dfrag[tk] + 1][dlane[tk]] = lo;
  };
  const int np = (Q + 31) >> 5, nchunk = (np + SX_CP - 1) / SX_CP;
  float4 x[SX_CP][TPT][2];
  if (stager) {
#pragma unroll
    for (int pp = 0; pp < SX_CP; ++pp)
#pragma unroll
      for (int tk = 0; tk < TPT; ++tk)
        if (pp < np) fetch(pp, tk, x[pp][tk][0], x[pp][tk][1]);
  }
  float4 q_h[QT], q_l[QT];
  SXState st[QT];
  unsigned dbase[QT];
  if (DROP) drop.seed += (unsigned long long)b * drop.seed_stride;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = min(q0 + 16 * u + r, Q - 1);                         // (rows past Q: a copy of the last query, not stored)
    // (a batch of frames with per-sample seeds: the index is the one sample b has when it is launched alone)
    dbase[u] = DROP ? (((drop.rows_per_sample ? 0u : (unsigned)b) * (unsigned)H + (unsigned)h) * (unsigned)Q + (unsigned)qrow) * (unsigned)Q + 8u * (unsigned)g : 0u;
    const float* qp = q + (brow + qrow) * ld + h * 32 + 8 * g;
#ifdef SX_Q_NT
    sx_split(ldg4_stream(qp), ldg4_stream(qp + 4), q_h[u], q_l[u]);      // (round-6 experiment: a query row is read by one workgroup only)
#else
    sx_split(ld4(qp), ld4(qp + 4), q_h[u], q_l[u]);
#endif
    st[u].o0 = st[u].o1 = st[u].lsum = st[u].negm = f32x4{0.f, 0.f, 0.f, 0.f};
    st[u].l = 0.0f;
  }
  if (stager) {
#pragma unroll
    for (int pp = 0; pp < SX_CP; ++pp)
#pragma unroll
      for (int tk = 0; tk < TPT; ++tk)
        if (pp < np) put(0, pp, tk, x[pp][tk][0], x[pp][tk][1]);
  }
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
#ifdef SX_NO_STAGE
    const bool more = false;
#else
    const bool more = c + 1 < nchunk;
#endif
    if (more && stager) {
#pragma unroll
      for (int pp = 0; pp < SX_CP; ++pp)
#pragma unroll
        for (int tk = 0; tk < TPT; ++tk)
          if ((c + 1) * SX_CP + pp < np) fetch((c + 1) * SX_CP + pp, tk, x[pp][tk][0], x[pp][tk][1]);
    }
    __builtin_amdgcn_sched_barrier(0);                 // the next chunk's loads stay in front of this chunk's products
#ifndef SX_NO_COMPUTE
    if (active) {
      const int left = np - c * SX_CP;                 // pairs in this chunk: SX_CP, or fewer in the last one
      const int nvalid_last = min(32, Q - 32 * (c * SX_CP + min(left, SX_CP) - 1));
      static_assert(SX_CP == 2, "the chunk dispatch below is written for two pairs");
      const unsigned key0 = 32u * (unsigned)(c * SX_CP);
      if (SX_NPC == 2 && left >= 2) sx_chunk<QT, 2, DROP>(frag[buf], lane, g, q_h, q_l, st, c == 0, nvalid_last, &drop, dbase, key0);
      else {
        sx_chunk<QT, 1, DROP>(frag[buf], lane, g, q_h, q_l, st, c == 0, left >= 2 ? 32 : nvalid_last, &drop, dbase, key0);
        if (left >= 2) sx_chunk<QT, 1, DROP>(frag[buf] + 1, lane, g, q_h, q_l, st, false, nvalid_last, &drop, dbase, key0 + 32u);
      }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (more && stager) {
#pragma unroll
      for (int pp = 0; pp < SX_CP; ++pp)
#pragma unroll
        for (int tk = 0; tk < TPT; ++tk)
          if ((c + 1) * SX_CP + pp < np) put(buf ^ 1, pp, tk, x[pp][tk][0], x[pp][tk][1]);
    }
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = q0 + 16 * u + r;
    if (qrow < Q) {
      float l = st[u].lsum[0];                         // SX_L_MFMA: every row of the ones-product holds the sum
      if (!SX_L_MFMA) { l = st[u].l; l += __shfl_xor(l, 16); l += __shfl_xor(l, 32); }   // the lane's keys 8g .. 8g + 7 of every pair
      const float inv = 1.0f / l;
      float* op = out + (brow + qrow) * ldo + h * 32 + 4 * g;
      st4(op, make_float4(st[u].o0[0] * inv, st[u].o0[1] * inv, st[u].o0[2] * inv, st[u].o0[3] * inv));
      st4(op + 16, make_float4(st[u].o1[0] * inv, st[u].o1[1] * inv, st[u].o1[2] * inv, st[u].o1[3] * inv));
    }
  }
}

// q, k rows [B*Q, ld] (q pre-scaled), vt [B, C, ldt] fp32 -> out [B*Q, C]: the operands of launch_self_attn_core
void fill_camk(const CamSampleArgs& a, CamK& p);   // cam_sample.hip

int launch_self_attn_core_x(const float* q, const float* k, int ld, const float* vt, int ldt, float* out, int ldo,
                            int B, int Q, int H, hipStream_t s, const DropK* drop, const PreGatherArgs* pregather) {
  TC_REQUIRE(Q > 0 && B > 0 && H > 0, "self_attn(f16x2): empty problem");
  TC_REQUIRE((ldt & 3) == 0 && ldt >= ((Q + 15) / 16) * 16, "self_attn(f16x2): ldt=%d too small for Q=%d", ldt, Q);
  constexpr int QT = SX_QT_VALUE, QW = 16 * QT * SX_NW;
  const int BH = B * H, Gf = Q / QW, G = (Q + QW - 1) / QW;
  const int nattn = BH * Gf + (G > Gf ? BH : 0);
  PreGatherK pg;
  memset(&pg, 0, sizeof(pg));
  if (drop != nullptr && drop->thr != 0) {
    TC_REQUIRE(pregather == nullptr, "self_attn(f16x2): the pre-gather rides in eval launches only");
    TC_REQUIRE((unsigned long long)(drop->rows_per_sample ? 1 : B) * H * Q * Q < (1ull << 32),
               "self_attn(f16x2): dropout index space (B*H*Q*Q) exceeds 32 bits");
    hipLaunchKernelGGL((self_attn_x_kernel<QT, true>), dim3(nattn), dim3(SX_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q, H * 32, H, BH, *drop, pg);
  } else {
    if (pregather != nullptr) {
      const PreGatherArgs& a = *pregather;
      TC_REQUIRE(a.out != nullptr && a.mask != nullptr && a.M > 0, "self_attn(f16x2): pre-gather buffers");
      TC_REQUIRE(a.cam.C == 256 && a.cam.feats.num_levels == 4 && a.cam.num_cams <= 8, "self_attn(f16x2): pre-gather shape");
      for (int l = 0; l < a.cam.feats.num_levels; ++l)      // pixel indices are 32-bit in the kernels
        TC_REQUIRE((long long)a.cam.B * a.cam.num_cams * a.cam.feats.H[l] * a.cam.feats.W[l] < (1ll << 31),
                   "self_attn(f16x2): pre-gather level %d has too many pixels for one call", l);
      fill_camk(a.cam, pg.cam);
      pg.cam.vis = nullptr; pg.cam.out = nullptr; pg.cam.pair_counter = nullptr; pg.cam.logits = nullptr;
      pg.M = a.M; pg.ref_mod = a.ref_mod; pg.out = a.out; pg.mask = a.mask;
      pg.nblocks = (a.M + SX_PG_ROWS - 1) / SX_PG_ROWS;
    }
    hipLaunchKernelGGL((self_attn_x_kernel<QT, false>), dim3(nattn + pg.nblocks), dim3(SX_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q, H * 32, H, BH,
                       DropK{0, 0, 1.0f, 0, 0, 0, 0, 0}, pg);
  }
  return check_launch("self_attn(f16x2, staged)");
}

int launch_self_attn_core(const float* q, const float* k, int ld, const float* vt, int ldt,
                          float* out, int ldo, int B, int Q, int H, hipStream_t s, const DropK* drop) {
  TC_REQUIRE(Q > 0 && B > 0 && H > 0, "self_attn: empty problem");
  TC_REQUIRE((ldt & 3) == 0 && ldt >= ((Q + 15) / 16) * 16, "self_attn: ldt=%d too small for Q=%d", ldt, Q);
  constexpr int QT = 2;
  dim3 grid((Q + 16 * QT - 1) / (16 * QT), H, B);
  if (drop != nullptr && drop->thr != 0) {
    TC_REQUIRE((unsigned long long)(drop->rows_per_sample ? 1 : B) * H * Q * Q < (1ull << 32),
               "self_attn: dropout index space (B*H*Q*Q) exceeds 32 bits");
    hipLaunchKernelGGL((self_attn_kernel<QT, true>), grid, dim3(SA_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q,
                       H * 32, *drop);
  } else {
    hipLaunchKernelGGL((self_attn_kernel<QT, false>), grid, dim3(SA_NW * 64), 0, s, q, k, ld, vt, ldt, out, ldo, Q,
                       H * 32, DropK{0, 0, 1.0f, 0, 0, 0, 0, 0});
  }
  return check_launch("self_attn");
}

}  // namespace tc
