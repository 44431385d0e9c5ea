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
    st.o0 *= alpha; st.o1 *= alpha;
    st.negm -= delta;
  }
  const float p0 = __builtin_amdgcn_exp2f(s0), p1 = __builtin_amdgcn_exp2f(s1);
  const float p2 = __builtin_amdgcn_exp2f(s2), p3 = __builtin_amdgcn_exp2f(s3);
  st.l += (p0 + p1) + (p2 + p3);
  float d0 = p0, d1 = p1, d2 = p2, d3 = p3;
  if (DROP) {
    d0 = drop_keep(drop.seed, drop.site, drop_base + 0, drop.thr) ? p0 * drop.scale : 0.0f;
    d1 = drop_keep(drop.seed, drop.site, drop_base + 1, drop.thr) ? p1 * drop.scale : 0.0f;
    d2 = drop_keep(drop.seed, drop.site, drop_base + 2, drop.thr) ? p2 * drop.scale : 0.0f;
    d3 = drop_keep(drop.seed, drop.site, drop_base + 3, drop.thr) ? p3 * drop.scale : 0.0f;
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
// nine frames), 58 % of its SIMD cycles f32 MFMAs on the vector pipe.  MEASURED RESULT (round 4): exact (every
// attention test passes at the fp32 kernel's tolerances, 1e-6 from it) but NOT faster -- 78 us + a 10 us conversion
// pass against 84 us per nine frames -- so tc_head_forward keeps the fp32 core; this one is the operator
// tc_sdpa_fwd_f16x2.  Why: 203 VGPRs (two fragment buffers of 32 registers, 42 of running state for two query
// sub-tiles, the P planes) = one workgroup per CU instead of two, and with the MFMA time gone (1.3 of 10 us per
// workgroup) what is left is the latency of the dependent chain load -> QK -> exp -> split -> PV of a wave's 3.6 key
// pairs; forcing 128 registers spills 140-720 bytes (190-380 us), one query sub-tile per workgroup 121 us, four 94 us.
// Operands as two f16 planes (hi, 2^11-scaled lo:
// chain.hip "16-row tiles on the f16 MATRIX CORES"), three v_mfma_f32_16x16x32_f16 per product:
//   * Q | K and V^T arrive ALREADY SPLIT (attn_planes_kernel: one pass over the chain's fp32 outputs per layer; split
//     inside this kernel every one of the 29 query-tile workgroups of a head would redo the same K / V conversion --
//     ~140 VALU instructions per 32 keys, as much issue time as the f32 MFMAs they replace);
//   * d = 32 is ONE MFMA's k: S^T of a 16-key tile = K_hi Q_hi + 2^-11 (K_lo Q_hi + K_hi Q_lo);
//   * the PV product sums over KEYS, 32 per MFMA: key tiles go in pairs, and the rows of the two K tiles are chosen so
//     that a lane's eight scores (4 + 4) are the eight CONSECUTIVE keys 8g .. 8g + 7 of the pair (tile A holds keys
//     8g' + i, tile B keys 8g' + 4 + i at row 4g' + i): the probabilities are the PV product's B operand as they
//     stand -- split into planes in registers (8 values per lane) -- and V^T's operand is one 16-byte load per plane;
//   * everything else (lazy re-centring, key pairs round-robin over the 8 waves, merge through LDS) as above.
typedef _Float16 sa_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sa_f16x2 __attribute__((ext_vector_type(2)));
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sa_f16x8, (a)), __builtin_bit_cast(sa_f16x8, (b)), (c), 0, 0, 0)
constexpr float SA_LO = 2048.0f, SA_ILO = 1.0f / 2048.0f;

__device__ __forceinline__ unsigned sa_pk(float a, float b) {
  const sa_f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// fp32 -> (hi, lo) planes: Q | K rows [M, ncol] -> [M, ncol] halves each; V^T [B*C, ldt] -> [B*C, ldt2] (zero padded)
__global__ __launch_bounds__(256) void attn_planes_kernel(const float* __restrict__ qk, size_t n_qk4,
                                                          unsigned short* __restrict__ qk_h, unsigned short* __restrict__ qk_l,
                                                          const float* __restrict__ vt, int rows_vt, int ldt, int Q, int ldt2,
                                                          unsigned short* __restrict__ vt_h, unsigned short* __restrict__ vt_l) {
  const size_t n_vt4 = (size_t)rows_vt * (ldt2 / 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_qk4 + n_vt4; i += (size_t)gridDim.x * 256) {
    float4 x;
    unsigned short *dh, *dl;
    if (i < n_qk4) {
      x = ld4(qk + 4 * i); dh = qk_h + 4 * i; dl = qk_l + 4 * i;
    } else {
      const size_t j = i - n_qk4;
      const size_t row = j / (ldt2 / 4);
      const int c = 4 * (int)(j - row * (ldt2 / 4));
      x = c + 3 < ldt ? ld4(vt + row * ldt + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (c + 0 >= Q) x.x = 0.f;
      if (c + 1 >= Q) x.y = 0.f;
      if (c + 2 >= Q) x.z = 0.f;
      if (c + 3 >= Q) x.w = 0.f;
      dh = vt_h + row * ldt2 + c; dl = vt_l + row * ldt2 + c;
    }
    const unsigned h0 = sa_pk(x.x, x.y), h1 = sa_pk(x.z, x.w);
    const sa_f16x2 a = __builtin_bit_cast(sa_f16x2, h0), b = __builtin_bit_cast(sa_f16x2, h1);
    const unsigned l0 = sa_pk((x.x - (float)a[0]) * SA_LO, (x.y - (float)a[1]) * SA_LO);
    const unsigned l1 = sa_pk((x.z - (float)b[0]) * SA_LO, (x.w - (float)b[1]) * SA_LO);
    typedef unsigned sa_u2 __attribute__((ext_vector_type(2)));
    *(TC_GLOBAL sa_u2*)(dh) = sa_u2{h0, h1};
    *(TC_GLOBAL sa_u2*)(dl) = sa_u2{l0, l1};
  }
}

struct KVFragH { float4 ka_h, ka_l, kb_h, kb_l, v0_h, v0_l, v1_h, v1_l; };     // (8 halves each)
struct KVPtrH { const unsigned short* ka; const unsigned short* kb; const unsigned short* v0; size_t klo, vlo, v1off; };
__device__ __forceinline__ float4 ldh8(const unsigned short* p) { return ld4(reinterpret_cast<const float*>(p)); }
__device__ __forceinline__ KVFragH load_kv_h(const KVPtrH& p) {
  KVFragH f;
  f.ka_h = ldh8(p.ka); f.ka_l = ldh8(p.ka + p.klo); f.kb_h = ldh8(p.kb); f.kb_l = ldh8(p.kb + p.klo);
  f.v0_h = ldh8(p.v0); f.v0_l = ldh8(p.v0 + p.vlo); f.v1_h = ldh8(p.v0 + p.v1off); f.v1_l = ldh8(p.v0 + p.v1off + p.vlo);
  return f;
}
struct SAStateH { f32x4 o0h, o0l, o1h, o1l, negm; float l; };

// One PAIR of 16-key tiles (32 keys) against one 16-query sub-tile.  nvalid < 32: the ragged last pair.
__device__ __forceinline__ void sa_pair_h(const KVFragH& f, const float4& q_h, const float4& q_l, SAStateH& st, bool first,
                                          int nvalid, int g) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 al = MFMAH(f.ka_l, q_h, zero); al = MFMAH(f.ka_h, q_l, al);
  f32x4 bl = MFMAH(f.kb_l, q_h, zero); bl = MFMAH(f.kb_h, q_l, bl);
  const f32x4 ah = MFMAH(f.ka_h, q_h, st.negm), bh = MFMAH(f.kb_h, q_h, st.negm);
  // s[i] = log2(e) * S^T[key0 + 8g + i][query r] + negm, i = 0..7
  float s[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { s[i] = fmaf(al[i], SA_ILO, ah[i]); s[4 + i] = fmaf(bl[i], SA_ILO, bh[i]); }
  float4 v0h = f.v0_h, v0l = f.v0_l, v1h = f.v1_h, v1l = f.v1_l;
  if (nvalid < 32) {                                   // wave-uniform: keys >= nvalid are masked (their V planes are 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) if (8 * g + i >= nvalid) s[i] = -INFINITY;
  }
  const bool above = any_above(s[0], s[1], s[2], s[3], SA_TAU) || any_above(s[4], s[5], s[6], s[7], SA_TAU);
  if (first || __builtin_amdgcn_ballot_w64(above) != 0) {
    const float mx = max_lanes_16_32(fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]))));
    const float delta = first ? mx : fmaxf(mx, 0.0f);
    const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] -= delta;
    st.l *= alpha;
    st.o0h *= alpha; st.o0l *= alpha; st.o1h *= alpha; st.o1l *= alpha;
    st.negm -= delta;
  }
  float p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(s[i]);
  st.l += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  // P^T as two f16 planes: the lane's eight keys are the MFMA's eight k slots
  unsigned ph[4], pl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ph[i] = sa_pk(p[2 * i], p[2 * i + 1]);
    const sa_f16x2 hh = __builtin_bit_cast(sa_f16x2, ph[i]);
    pl[i] = sa_pk((p[2 * i] - (float)hh[0]) * SA_LO, (p[2 * i + 1] - (float)hh[1]) * SA_LO);
  }
  const float4 p_h = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
  const float4 p_l = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
  // O^T[d][q] += V^T[d][key] P^T[key][q]
  st.o0l = MFMAH(v0l, p_h, st.o0l); st.o0l = MFMAH(v0h, p_l, st.o0l); st.o0h = MFMAH(v0h, p_h, st.o0h);
  st.o1l = MFMAH(v1l, p_h, st.o1l); st.o1l = MFMAH(v1h, p_l, st.o1l); st.o1h = MFMAH(v1h, p_h, st.o1h);
}

template <int QT>
__global__ __launch_bounds__(SA_NW * 64) void self_attn_h_kernel(const unsigned short* __restrict__ qk_h, size_t qk_lo, int ld,
                                                                 const unsigned short* __restrict__ vt_h, size_t vt_lo, int ldt2,
                                                                 float* __restrict__ out, int ldo, int Q, int C) {
  __shared__ float sm_m[SA_NW][QT][16];
  __shared__ float sm_l[SA_NW][QT][64];
  __shared__ float4 sm_o[SA_NW][QT][2][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 16 * QT, h = blockIdx.y, b = blockIdx.z;
  const size_t brow = (size_t)b * Q;
  float4 q_h[QT], q_l[QT];
  SAStateH st[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int qrow = min(q0 + 16 * u + r, Q - 1);
    const unsigned short* qp = qk_h + (brow + qrow) * ld + h * 32 + 8 * g;
    q_h[u] = ldh8(qp); q_l[u] = ldh8(qp + qk_lo);
    st[u].o0h = st[u].o0l = st[u].o1h = st[u].o1l = f32x4{0.f, 0.f, 0.f, 0.f};
    st[u].negm = f32x4{0.f, 0.f, 0.f, 0.f}; st[u].l = 0.0f;
  }
  // the wave's key PAIRS: wave, wave + NW, ... among the nfull whole pairs, then the ragged pair if it is this wave's turn
  const int nfull = Q >> 5;
  const int n = wave < nfull ? (nfull - wave + SA_NW - 1) / SA_NW : 0;
  const int ra = 8 * (r >> 2) + (r & 3);                // row r of tile A holds key ra of the pair, tile B key ra + 4
  const unsigned short* kbase = qk_h + C + h * 32 + 8 * g;     // K columns of the (q | k) rows
  KVPtrH p;
  p.klo = qk_lo; p.vlo = vt_lo; p.v1off = (size_t)16 * ldt2;
  p.ka = kbase + (brow + wave * 32 + ra) * ld;
  p.kb = p.ka + (size_t)4 * ld;
  p.v0 = vt_h + ((size_t)b * C + h * 32 + r) * ldt2 + wave * 32 + 8 * g;
  const size_t kstep = (size_t)SA_NW * 32 * ld;
  auto advance = [&]() { p.ka += kstep; p.kb += kstep; p.v0 += SA_NW * 32; };
  auto pair = [&](const KVFragH& f, int i, int nvalid) {
#pragma unroll
    for (int u = 0; u < QT; ++u) sa_pair_h(f, q_h[u], q_l[u], st[u], i == 0, nvalid, g);
  };
  KVFragH fa, fb;
  if (n > 0) fa = load_kv_h(p);
  int i = 0;
  for (; i + 2 <= n; i += 2) {
    advance();
    fb = load_kv_h(p);
    __builtin_amdgcn_sched_barrier(0);
    pair(fa, i, 32);
    advance();
    if (i + 2 < n) fa = load_kv_h(p);
    __builtin_amdgcn_sched_barrier(0);
    pair(fb, i + 1, 32);
  }
  if (i < n) { pair(fa, i, 32); ++i; }
  const int rag = Q & 31;
  if (rag != 0 && wave == (nfull % SA_NW)) {            // this wave's next pair is the ragged one
    KVPtrH pr = p;
    const int k0 = nfull * 32;
    pr.ka = kbase + (brow + min(k0 + ra, Q - 1)) * ld;
    pr.kb = kbase + (brow + min(k0 + ra + 4, Q - 1)) * ld;
    pr.v0 = vt_h + ((size_t)b * C + h * 32 + r) * ldt2 + k0 + 8 * g;          // ldt2 >= 32 * (nfull + 1), zero padded
    const KVFragH fr = load_kv_h(pr);
    pair(fr, i, rag);
    ++i;
  }
  const bool idle = i == 0;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (g == 0) sm_m[wave][u][r] = idle ? -INFINITY : -st[u].negm[0];
    sm_l[wave][u][lane] = st[u].l;
    sm_o[wave][u][0][lane] = make_float4(fmaf(st[u].o0l[0], SA_ILO, st[u].o0h[0]), fmaf(st[u].o0l[1], SA_ILO, st[u].o0h[1]),
                                         fmaf(st[u].o0l[2], SA_ILO, st[u].o0h[2]), fmaf(st[u].o0l[3], SA_ILO, st[u].o0h[3]));
    sm_o[wave][u][1][lane] = make_float4(fmaf(st[u].o1l[0], SA_ILO, st[u].o1h[0]), fmaf(st[u].o1l[1], SA_ILO, st[u].o1h[1]),
                                         fmaf(st[u].o1l[2], SA_ILO, st[u].o1h[2]), fmaf(st[u].o1l[3], SA_ILO, st[u].o1h[3]));
  }
  __syncthreads();
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

size_t self_attn_h_ws_bytes(int B, int Q, int H) {
  const size_t M = (size_t)B * Q, C = (size_t)H * 32, ldt2 = ((size_t)Q + 31) / 32 * 32;
  return 2 * arena_slice(M * 2 * C, 2) + 2 * arena_slice((size_t)B * C * ldt2, 2);
}

// q | k rows [B*Q, 2C] (q pre-scaled), vt [B, C, ldt] fp32 -> planes in `ws` -> out [B*Q, C]
int launch_self_attn_core_h(const float* qk, const float* vt, int ldt, float* out, int ldo, int B, int Q, int H,
                            void* ws, size_t ws_bytes, hipStream_t s) {
  TC_REQUIRE(Q > 0 && B > 0 && H > 0, "self_attn(f16x2): empty problem");
  TC_REQUIRE((ldt & 3) == 0 && ldt >= ((Q + 15) / 16) * 16, "self_attn(f16x2): ldt=%d too small for Q=%d", ldt, Q);
  TC_REQUIRE(ws != nullptr && ws_bytes >= self_attn_h_ws_bytes(B, Q, H), "self_attn(f16x2): workspace too small");
  const int C = H * 32, ld = 2 * C, ldt2 = (Q + 31) / 32 * 32;
  const size_t M = (size_t)B * Q;
  Arena a(ws, ws_bytes);
  unsigned short* qk_h = a.take<unsigned short>(M * ld);
  unsigned short* qk_l = a.take<unsigned short>(M * ld);
  unsigned short* vt_h = a.take<unsigned short>((size_t)B * C * ldt2);
  unsigned short* vt_l = a.take<unsigned short>((size_t)B * C * ldt2);
  const size_t n4 = M * ld / 4 + (size_t)B * C * (ldt2 / 4);
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(attn_planes_kernel, dim3(blocks), dim3(256), 0, s, qk, M * ld / 4, qk_h, qk_l, vt, B * C, ldt, Q, ldt2,
                     vt_h, vt_l);
  if (const int rc = check_launch("attn_planes"); rc != 0) return rc;
  constexpr int QT = 2;
  dim3 grid((Q + 16 * QT - 1) / (16 * QT), H, B);
  hipLaunchKernelGGL((self_attn_h_kernel<QT>), grid, dim3(SA_NW * 64), 0, s, qk_h, (size_t)(qk_l - qk_h), ld, vt_h,
                     (size_t)(vt_l - vt_h), ldt2, out, ldo, Q, C);
  return check_launch("self_attn(f16x2)");
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
