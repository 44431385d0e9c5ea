// Row-local device routines shared by the stand-alone operator kernels and the
// fused row-chain kernels (chain.hip).  "Row" = one query (or radar token) of
// C = 256 channels owned by ONE wavefront: lane l holds channels 4l..4l+3.
#pragma once
#include "common.hpp"
#include "../../include/transcar_hip.h"

namespace tc {

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(relu_(v.x), relu_(v.y), relu_(v.z), relu_(v.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// LayerNorm over the 256 channels of a row held as one float4 per lane; eps 1e-5.
__device__ __forceinline__ float4 ln_row(float4 v, const float* g, const float* b, int lane) {
  const float4 gg = ld4(g + 4 * lane), bb = ld4(b + 4 * lane);   // in flight under the reductions
  float s = wave_sum(v.x + v.y + v.z + v.w);
  const float mean = s * (1.0f / 256.0f);
  float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  float q = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
  const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
  return make_float4(d.x * rstd * gg.x + bb.x, d.y * rstd * gg.y + bb.y,
                     d.z * rstd * gg.z + bb.z, d.w * rstd * gg.w + bb.w);
}

// Wave64 sums of FOUR rows at once (round 6).  In: each lane's partial sum of rows a, b, c, d.  Out (in `a`): the wave
// total of row j in every lane whose bank (lane >> 2) & 3 is j.  Same reduction TREE as four wave_sum()s -- pairs at
// distance 1, 2, quad + quad, 8 + 8, row + row, half + half; IEEE addition commutes, so which of a pair's two lanes does the
// add does not matter -- hence the same bits; but from the third stage on ONE register carries two, then four rows
// (DPP bank masks merge them), and the two cross-row stages are gfx950's half-wave / row swaps instead of row_bcast (which
// would broadcast lane 15 / 31, i.e. one bank): 20 vector instructions instead of 4 x 12.  A LayerNorm step of the 16- and
// 32-row chains is VALU-bound -- two waves per SIMD, 4 cycles per wave64 instruction: ~460 instructions for a wave's
// four rows were the step's 5.3 K cycles (profiles/r6_ln_valu.txt: the same with its parameter loads removed, the same
// when repeated through warm code) -- and a quarter of those were the eight separate reductions and the four copies of
// 1 / sqrt(var + eps), all lanes computing one number.
// (hazards: VALU write -> DPP read of that register, and -> v_permlane read: 2 wait states; hipcc pads nothing inside an
// asm string)
__device__ __forceinline__ float wave_sum4_packed(float a, float b, float c, float d) {
  float t;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      // quad + quad: a | b share %0 (even | odd banks), c | d share %2
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      // 8 + 8 (row_ror:8 keeps a lane's bank parity; row_mirror would not): banks 0..3 of %0 = a, b, c, d
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      // row + row, half + half
      "v_mov_b32 %4, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %4, %0\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %4\n\t"
      "v_mov_b32 %4, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %4, %0\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %4\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t));
  return a;
}
__device__ __forceinline__ float packed_row(float v, int j) {        // row j's value of a wave_sum4_packed() register
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 4 * j));
}

// The same for NR rows at once (gamma / beta already in registers): the rows' reduction chains are
// independent and interleave.  Same operations per row as ln_row: bit-identical results.
template <int NR>
__device__ __forceinline__ void ln_rows(float4 (&v)[NR], const float4 gg, const float4 bb) {
#ifndef TC_LN_UNPACKED
  if constexpr (NR == 4) {
    // four rows: the packed reductions, and 1 / sqrt(var + eps) ONCE, on the register that carries the four variances
    const float sp = wave_sum4_packed(v[0].x + v[0].y + v[0].z + v[0].w, v[1].x + v[1].y + v[1].z + v[1].w,
                                      v[2].x + v[2].y + v[2].z + v[2].w, v[3].x + v[3].y + v[3].z + v[3].w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mean = packed_row(sp, i) * (1.0f / 256.0f);
      v[i] = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
    }
    float qq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qq[i] = v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    const float qp = wave_sum4_packed(qq[0], qq[1], qq[2], qq[3]);
    const float rp = 1.0f / sqrtf(qp * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float rstd = packed_row(rp, i);
      v[i] = make_float4(v[i].x * rstd * gg.x + bb.x, v[i].y * rstd * gg.y + bb.y,
                         v[i].z * rstd * gg.z + bb.z, v[i].w * rstd * gg.w + bb.w);
    }
    return;
  }
#endif
  float s[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] = wave_sum(v[i].x + v[i].y + v[i].z + v[i].w);
  float q[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float mean = s[i] * (1.0f / 256.0f);
    v[i] = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) q[i] = wave_sum(v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float rstd = 1.0f / sqrtf(q[i] * (1.0f / 256.0f) + 1e-5f);
    v[i] = make_float4(v[i].x * rstd * gg.x + bb.x, v[i].y * rstd * gg.y + bb.y,
                       v[i].z * rstd * gg.z + bb.z, v[i].w * rstd * gg.w + bb.w);
  }
}

// position-encoder layer 0: relu(LN(W0 p + b0)), W0 [256,3]
__device__ __forceinline__ float4 posenc_l0_row(float p0, float p1, float p2, const float* w0,
                                                const float* b0, const float* g, const float* beta,
                                                int lane, float4* pre = nullptr) {
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * lane + i;
    v[i] = ldg1(w0 + c * 3 + 0) * p0 + ldg1(w0 + c * 3 + 1) * p1 + ldg1(w0 + c * 3 + 2) * p2 + ldg1(b0 + c);
  }
  if (pre != nullptr) *pre = make_float4(v[0], v[1], v[2], v[3]);      // training tape: the LayerNorm's input
  return relu4(ln_row(make_float4(v[0], v[1], v[2], v[3]), g, beta, lane));
}

#ifndef CAM_STAMP
#define CAM_STAMP(slot) do {} while (0)      // s_memtime stamps in the STAMPS=1 build of chain.hip
#endif
// ---- camera sampling of one query (XFMR:365-373, 381-422) -------------------
struct CamK {
  const float* data[TC_MAX_LEVELS];
  int H[TC_MAX_LEVELS], W[TC_MAX_LEVELS];
  int num_levels, B, Q, num_cams;
  const float* l2i; const float* ref; const float* logits;
  float pc[6]; float img_h, img_w;
  float* out; unsigned char* vis; unsigned long long* pair_counter;
};

// Reference point `row` of sample b projected onto camera `cam` (XFMR:389-409), any
// (row, camera) per lane: u, v in [-1, 1] grid_sample coordinates; true if visible.
__device__ __forceinline__ bool cam_project_lane(const CamK& p, int row, int b, int cam, bool active,
                                                 float& u, float& v) {
  const int N = p.num_cams;
  // XFMR:389-391
  const float rx = p.ref[(size_t)row * 3 + 0] * (p.pc[3] - p.pc[0]) + p.pc[0];
  const float ry = p.ref[(size_t)row * 3 + 1] * (p.pc[4] - p.pc[1]) + p.pc[1];
  const float rz = p.ref[(size_t)row * 3 + 2] * (p.pc[5] - p.pc[2]) + p.pc[2];
  const float* m = p.l2i + ((size_t)b * N + cam) * 16;
  // XFMR:398-409
  const float cx = m[0] * rx + m[1] * ry + m[2] * rz + m[3];
  const float cy = m[4] * rx + m[5] * ry + m[6] * rz + m[7];
  const float cz = m[8] * rx + m[9] * ry + m[10] * rz + m[11];
  const float eps = 1e-5f;
  const float zc = fmaxf(cz, eps);
  u = (cx / zc) / p.img_w;
  v = (cy / zc) / p.img_h;
  u = (u - 0.5f) * 2.0f;
  v = (v - 0.5f) * 2.0f;
  const bool visible = active && (cz > eps) && (u > -1.0f) && (u < 1.0f) && (v > -1.0f) && (v < 1.0f);
  if (p.vis != nullptr && active) p.vis[(size_t)row * N + cam] = visible ? 1 : 0;
  return visible;
}
// Lane c < num_cams projects reference point `row` onto camera c; returns the mask of visible
// cameras.  The N matrix reads are one memory round trip (as a loop over the cameras each
// iteration's reads waited behind the previous camera's visibility branch).
__device__ __forceinline__ unsigned long long cam_project(const CamK& p, int row, int b, int lane,
                                                          float& u, float& v) {
  return __ballot(cam_project_lane(p, row, b, min(lane, p.num_cams - 1), lane < p.num_cams, u, v));
}

// Bilinear tap geometry, one tap per lane: lane j < 4 L owns tap t = j & 3 (nw, ne, sw, se)
// of level l = j >> 2 around (u_, v_) on camera `cam` -- its weight (zero outside the map:
// F.grid_sample, bilinear, zeros padding, align_corners=False) and the index of its
// (clamped) pixel in the level's [B * num_cams, H, W] grid.  (With every lane computing all
// 4 L taps the sampling step was instruction bound: ~700 VALU instructions per camera.)
// the lane's level l = lane >> 2: its map's height and width.  Depends on the lane only: callers that walk several rows
// fetch it ONCE (round 6: hipcc turns the select chain into an indexed load from the kernel-argument segment -- inside the
// row loop of the chain's sampling step that was one more dependent memory round trip per row, in front of the taps)
template <int L>
__device__ __forceinline__ void cam_level_dims(const CamK& p, int lane, int& H, int& W) {
  const int l = min(lane >> 2, L - 1);
  H = p.H[0]; W = p.W[0];
#pragma unroll
  for (int i = 1; i < L; ++i) {
    int hi = p.H[i], wi = p.W[i];
    asm("" : "+s"(hi), "+s"(wi));          // (opaque scalars: selects between registers, no table in memory)
    if (l == i) { H = hi; W = wi; }
  }
}
template <int L>
__device__ __forceinline__ void cam_tap_lane(const CamK& p, int b, int cam, float u_, float v_, int lane,
                                             float& wgt, int& pix, int H = -1, int W = -1) {
  if (H < 0) cam_level_dims<L>(p, lane, H, W);
  const float ix = ((u_ + 1.0f) * (float)W - 1.0f) * 0.5f;
  const float iy = ((v_ + 1.0f) * (float)H - 1.0f) * 0.5f;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w_ = ix - xw, e_ = 1.0f - w_, n_ = iy - yn, s_ = 1.0f - n_;
  const int xs = lane & 1, ys = (lane >> 1) & 1;
  const int x = (int)xw + xs, y = (int)yn + ys;
  const bool valid = (x >= 0) && (x < W) && (y >= 0) && (y < H);
  wgt = valid ? (ys ? n_ : s_) * (xs ? w_ : e_) : 0.0f;
  const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1);
  pix = ((b * p.num_cams + cam) * H + yc) * W + xc;
}
// this lane's 4 channels of pixel `pix` (uniform) of level l
__device__ __forceinline__ const float* cam_tap_ptr(const CamK& p, int l, int pix, int lane) {
  return p.data[l] + ((size_t)(unsigned)pix << 8) + 4 * lane;
}
// a tap row (and a pre-gathered level value) is read ONCE: the non-temporal hint keeps the gather out of the way of the
// lines that are reused -- the packed weights above all (round 6, chain.hip; +2.4 % frames/s, bit-identical)
__device__ __forceinline__ float4 cam_tap_ld(const float* p) { return ldg4_stream(p); }
__device__ __forceinline__ float lane_f(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// The bilinear value of one level of one (query, camera) pair: the four taps times their weights, NaN -> 0 on the
// sampled value (XFMR:367).  ONE definition for the in-chain sampling (cam_sample_core) and for the pre-gather
// workgroups of the attention-core launch (cam_pregather_rows, round 6): same expression, same contraction, same bits.
__device__ __forceinline__ float4 cam_level_value(const float4 (&tap)[4], const float (&wgt)[4]) {
  float4 s;
  s.x = tap[0].x * wgt[0] + tap[1].x * wgt[1] + tap[2].x * wgt[2] + tap[3].x * wgt[3];
  s.y = tap[0].y * wgt[0] + tap[1].y * wgt[1] + tap[2].y * wgt[2] + tap[3].y * wgt[3];
  s.z = tap[0].z * wgt[0] + tap[1].z * wgt[1] + tap[2].z * wgt[2] + tap[3].z * wgt[3];
  s.w = tap[0].w * wgt[0] + tap[1].w * wgt[1] + tap[2].w * wgt[2] + tap[3].w * wgt[3];
  if (s.x != s.x) s.x = 0.f;
  if (s.y != s.y) s.y = 0.f;
  if (s.z != s.z) s.z = 0.f;
  if (s.w != s.w) s.w = 0.f;
  return s;
}

// Weighted sum over the visible cameras (XFMR:365-373).  u, v: lane c holds camera c's
// coordinates; lg: the query's num_cams*L attention logits (any address space);
// fetch(c, l, t, ptr): this lane's 4 channels of tap t of level l of the c-th visible camera.
// lane0: lane lane0 + c holds camera c's coordinates (0 for a single projected row; the row chains project
// the four rows of a wave at once, 16 lanes apart)
template <int L, typename Fetch>
__device__ __forceinline__ float4 cam_sample_core(const CamK& p, int b, const float* lg, int lane,
                                                  unsigned long long vmask, float u, float v, Fetch fetch,
                                                  int lane0 = 0, int Hl = -1, int Wl = -1) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0;
  // XFMR:370: the num_cams * L attention weights of the query, one per lane, ONCE (round 3: every lane evaluated
  // sigmoid(lg[cam * L + l]) for itself inside the camera loop -- ~80 of its ~330 VALU instructions per visible
  // camera; same function of the same value: bit-identical)
  const float sg_lane = sigmoidf_(lg[min(lane, p.num_cams * L - 1)]);
  while (vmask) {
    const int cam = __ffsll((long long)vmask) - 1;
    vmask &= vmask - 1;
    const float u_ = lane_f(u, lane0 + cam), v_ = lane_f(v, lane0 + cam);
    float4 tap[L][4];
    float wgt[L][4];
    float w_lane;
    int pix_lane;
    cam_tap_lane<L>(p, b, cam, u_, v_, lane, w_lane, pix_lane, Hl, Wl);
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wgt[l][t] = lane_f(w_lane, 4 * l + t);
        tap[l][t] = fetch(c, l, t, cam_tap_ptr(p, l, __builtin_amdgcn_readlane(pix_lane, 4 * l + t), lane));
      }
    }
    CAM_STAMP(2);
    float4 camacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float4 s = cam_level_value(tap[l], wgt[l]);
      const float a = lane_f(sg_lane, cam * L + l);   // XFMR:370, mask == 1 here
      camacc.x += s.x * a; camacc.y += s.y * a; camacc.z += s.z * a; camacc.w += s.w * a;
    }
    acc.x += camacc.x; acc.y += camacc.y; acc.z += camacc.z; acc.w += camacc.w;
    CAM_STAMP(3);
    ++c;
  }
  return acc;
}

// ---- round 6: the camera gather off the decoder chain's critical path ------------------------------
// The reference points of decoder layer l are final when chain l - 1 ends, and the attention core of layer l runs in
// between: extra workgroups of THAT launch (self_attn.hip) project every row, fetch the 16 taps of every visible
// (row, camera) pair and store the four bilinear level values (cam_level_value: 4 x 1 KiB per pair) plus the row's
// visibility mask.  The chain's sampling step then reads 4 KiB contiguous per pair and only weighs and sums
// (sigmoid(attention_weights) . mask, XFMR:367-373) -- the same products in the same order: bit-identical outputs.
//   out  [M][num_cams][L][256] floats (only visible pairs are written / read), mask [M] ints (bit c: camera c visible)
struct PreGatherK {
  CamK cam;                   // data / H / W / l2i / ref (this layer's reference points) / pc / img size
  int M, ref_mod;             // rows; reference rows taken modulo ref_mod when > 0
  float* out; int* mask;
  int nblocks;                // workgroups of the launch that run this role (0: none)
};
// the rows r0 .. r0 + nrows - 1 (nrows a multiple of 4) of one WAVE
template <int L>
__device__ __forceinline__ void cam_pregather_rows(const PreGatherK& g, int r0, int nrows, int lane) {
  const CamK& p = g.cam;
  int Hl, Wl;
  cam_level_dims<L>(p, lane, Hl, Wl);
#pragma unroll 1
  for (int base = 0; base < nrows; base += 4) {
    // lane 16 i + c: row r0 + base + i, camera c -- the projections of four rows in one round trip
    float pu, pv;
    unsigned long long vm;
    {
      const int i = lane >> 4, c = lane & 15;
      const int row = r0 + base + i;
      const int grow = min(row, g.M - 1);
      const bool act = row < g.M && c < p.num_cams;
      vm = __ballot(cam_project_lane(p, g.ref_mod > 0 ? grow % g.ref_mod : grow, grow / p.Q, min(c, p.num_cams - 1), act, pu, pv));
    }
    if (lane < 4 && r0 + base + lane < g.M) g.mask[r0 + base + lane] = (int)((vm >> (16 * lane)) & 0xFFFFull);
    unsigned long long rest = vm;
#pragma unroll 1
    while (rest) {
      const int bit = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
      const int cam = bit & 15;
      const int grow = r0 + base + (bit >> 4);
      float w_lane;
      int pix_lane;
      cam_tap_lane<L>(p, grow / p.Q, cam, lane_f(pu, bit), lane_f(pv, bit), lane, w_lane, pix_lane, Hl, Wl);
      // two levels (eight taps: 8 KiB per wave in flight) at a time: 32 tap registers instead of 64 -- the role must fit
      // the attention core's 128-register budget (four waves per SIMD: at 156 registers the core itself lost a quarter of
      // its occupancy and 5 us per launch, measured)
      float* o = g.out + (((size_t)grow * p.num_cams + cam) * L) * 256 + 4 * lane;
#pragma unroll
      for (int l0 = 0; l0 < L; l0 += 2) {
        float4 tap[2][4];
        float wgt[2][4];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            wgt[l][t] = lane_f(w_lane, 4 * (l0 + l) + t);
            tap[l][t] = cam_tap_ld(cam_tap_ptr(p, l0 + l, __builtin_amdgcn_readlane(pix_lane, 4 * (l0 + l) + t), lane));
          }
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const float4 s = cam_level_value(tap[l], wgt[l]);
          st4(o + (l0 + l) * 256, s);
        }
        __builtin_amdgcn_sched_barrier(0);          // (the second half's taps are not hoisted above the first half's sums)
      }
    }
  }
}

template <int L>
__device__ __forceinline__ float4 cam_sample_row(const CamK& p, int row, int b, const float* lg,
                                                 int lane, int& nvis) {
  CAM_STAMP(0);
  float u, v;
  const unsigned long long vmask = cam_project(p, row, b, lane, u, v);
  nvis = __popcll(vmask);
  CAM_STAMP(1);
  const float4 acc = cam_sample_core<L>(p, b, lg, lane, vmask, u, v,
                                        [](int, int, int, const float* ptr) { return cam_tap_ld(ptr); });
  CAM_STAMP(4);
  return acc;
}

// ---- distance-gated radar attention of one query (HEAD:549-579) -------------
// torch.cdist(p=2) via _euclidean_dist: [-2x, |x|^2, 1] . [y, 1, |y|^2]
// the squared distance exactly as _euclidean_dist forms it (before clamp and square root)
__device__ __forceinline__ float cdist_sq(float x0, float x1, float xn, float y0, float y1, float yn) {
  float t = __fmul_rn(__fmul_rn(-2.0f, x0), y0);
  t = fmaf(__fmul_rn(-2.0f, x1), y1, t);
  t = __fadd_rn(t, xn);
  t = __fadd_rn(t, yn);
  return t;
}
__device__ __forceinline__ float cdist_mm(float x0, float x1, float xn, float y0, float y1, float yn) {
  return sqrtf(fmaxf(cdist_sq(x0, x1, xn, y0, y1, yn), 1e-30f));
}
// The smallest float t with sqrtf(t) >= rad: sqrtf is monotone, so  sqrtf(max(t, 1e-30)) < rad  <=>
// max(t, 1e-30) < t*  -- the gate's comparison without a correctly-rounded square root per (query, token,
// circle).  Found from rad * rad by stepping single floats (a few sqrtf per QUERY).
__device__ __forceinline__ float sqrt_threshold(float rad) {
  float c = rad * rad;
  for (int i = 0; i < 8 && sqrtf(c) >= rad; ++i) c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) - 1u);
  for (int i = 0; i < 8; ++i) {
    const float up = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) + 1u);
    if (sqrtf(up) >= rad) return up;
    c = up;
  }
  return c;        // not reached for rad in the clamp range [0.5, 2]
}
__device__ __forceinline__ float sqnorm2(float a, float b) {
  return __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
}

// cx,cy: gate centre (m); b3,b6,b7: log-length, sin, cos of the previous box;
// q4: this lane's 4 channels of the projected, scaled query;
// rxy: xy of token t at rxy[t*ld_xy + {0,1}]; kv: token t at kv[t*ldkv]: K | V (256 each).
// Returns the attention output (zero if no hit); count = gated tokens (with multiplicity).
// DROP (training forward): dropout on the attention probabilities (nn.MultiheadAttention
// dropout = 0.1, HEAD:129): the softmax denominator counts every hit token, the weighted sum only
// the kept ones, scaled by 1 / (1 - p); mask index ((row * 8 + head) * tokens_ref + token).
// gate geometry of one query, HEAD:553-567: three circles (centre, front, back) of one radius
struct GateGeom {
  float cx, cy, fx, fy, bx, by, cn, fn, bn, rad, tstar;
  __device__ __forceinline__ GateGeom(float cx_, float cy_, float b3, float b6, float b7, float rmin, float rmax) {
    const float len = expf(b3);
    const float rs = -b6, rc = -b7;
    const float ox = __fmul_rn(__fmul_rn(len, 0.25f), rs), oy = __fmul_rn(__fmul_rn(len, 0.25f), rc);
    cx = cx_; cy = cy_;
    fx = __fadd_rn(cx, ox); fy = __fadd_rn(cy, oy);
    bx = __fsub_rn(cx, ox); by = __fsub_rn(cy, oy);
    rad = fminf(fmaxf(len / 2.0f, rmin), rmax);
    cn = sqnorm2(cx, cy); fn = sqnorm2(fx, fy); bn = sqnorm2(bx, by);
    tstar = sqrt_threshold(rad);
  }
  // The expensive part of the constructor above (expf, the clamp, sqrt_threshold's square roots: ~300
  // instructions that every lane of every wave repeated per row) evaluated ONCE per row by one thread:
  // (ox, oy, tstar) = Pre; the remaining adds and norms are the same operations in the same order.
  struct Pre { float ox, oy, tstar; };
  static __device__ __forceinline__ Pre precompute(float b3, float b6, float b7, float rmin, float rmax) {
    const float len = expf(b3);
    const float rs = -b6, rc = -b7;
    Pre p;
    p.ox = __fmul_rn(__fmul_rn(len, 0.25f), rs); p.oy = __fmul_rn(__fmul_rn(len, 0.25f), rc);
    p.tstar = sqrt_threshold(fminf(fmaxf(len / 2.0f, rmin), rmax));
    return p;
  }
  __device__ __forceinline__ GateGeom(float cx_, float cy_, const Pre& p) {
    cx = cx_; cy = cy_;
    fx = __fadd_rn(cx, p.ox); fy = __fadd_rn(cy, p.oy);
    bx = __fsub_rn(cx, p.ox); by = __fsub_rn(cy, p.oy);
    rad = 0.0f;                    // hit() does not use it
    cn = sqnorm2(cx, cy); fn = sqnorm2(fx, fy); bn = sqnorm2(bx, by);
    tstar = p.tstar;
  }
  // (cdist < rad) for any of the three circles, HEAD:568-571 -- on the squared distances against tstar
  __device__ __forceinline__ bool hit(float y0, float y1, float yn) const {
    const float tc_ = fmaxf(cdist_sq(cx, cy, cn, y0, y1, yn), 1e-30f), tf = fmaxf(cdist_sq(fx, fy, fn, y0, y1, yn), 1e-30f),
                tb = fmaxf(cdist_sq(bx, by, bn, y0, y1, yn), 1e-30f);
    return fminf(tc_, fminf(tf, tb)) < tstar;
  }
  // the literal form (a square root per circle): the device check of the equivalence, tests only
  __device__ __forceinline__ bool hit_sqrt(float y0, float y1, float yn) const {
    return (cdist_mm(cx, cy, cn, y0, y1, yn) < rad) || (cdist_mm(fx, fy, fn, y0, y1, yn) < rad) ||
           (cdist_mm(bx, by, bn, y0, y1, yn) < rad);
  }
  // the same on squared distances (no correctly-rounded square roots): may differ from hit() for a token
  // within rounding of a circle -- only for the row-ORDER hint of radar_compact.hip, never for a result
  __device__ __forceinline__ bool hit_approx(float y0, float y1, float yn) const {
    const float r2 = rad * rad;
    const float dc = fmaf(-2.0f * cx, y0, fmaf(-2.0f * cy, y1, cn + yn));
    const float df = fmaf(-2.0f * fx, y0, fmaf(-2.0f * fy, y1, fn + yn));
    const float db = fmaf(-2.0f * bx, y0, fmaf(-2.0f * by, y1, bn + yn));
    return fminf(dc, fminf(df, db)) < r2;
  }
};

// The gate alone: number of radar tokens inside the three circles (the last token counts pad_mult
// times, as in radar_attn_row).  Same predicate, same arithmetic: the two always agree.
__device__ __forceinline__ int radar_gate_count(float cx, float cy, float b3, float b6, float b7, float rmin,
                                                float rmax, const float* rxy, int ld_xy, int T, int pad_mult,
                                                int lane, unsigned long long* masks_out = nullptr) {
  const GateGeom gg(cx, cy, b3, b6, b7, rmin, rmax);
  int count = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    bool hit = false;
    if (t < T) {
      const float* y = rxy + (size_t)t * ld_xy;
      const float y0 = y[0], y1 = y[1];
      hit = gg.hit(y0, y1, sqnorm2(y0, y1));
    }
    const unsigned long long mask = __ballot(hit);
    count += __popcll(mask);
    if (T - 1 >= t0 && T - 1 < t0 + 64 && ((mask >> (T - 1 - t0)) & 1ull)) count += pad_mult - 1;
    if (masks_out != nullptr && lane == 0) masks_out[t0 >> 6] = mask;
  }
  return count;
}

template <bool DROP = false>
__device__ __forceinline__ float4 radar_attn_row_g(const GateGeom& gg, float4 q4,
                                                   const float* rxy, int ld_xy, const float* kv,
                                                   int ldkv, int T, int pad_mult, int lane, int& count,
                                                   DropK drop = DropK(), int row = 0,
                                                   const unsigned long long* hit_masks = nullptr) {
  // hit_masks (chain.hip, K_RADAR_GATE): the gate of this row already evaluated, one 64-token word per
  // chunk -- the same predicate, so the same tokens
  float m = -INFINITY, l = 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  count = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    unsigned long long mask;
    if (hit_masks != nullptr) {
      mask = hit_masks[t0 >> 6];
    } else {
      const int t = t0 + lane;
      bool hit = false;
      if (t < T) {
        const float* y = rxy + (size_t)t * ld_xy;
        const float y0 = y[0], y1 = y[1];
        const float yn = sqnorm2(y0, y1);
        hit = gg.hit(y0, y1, yn);
      }
      mask = __ballot(hit);
    }
    while (mask) {
      const int j = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int tok = t0 + j;
      const int mult = (tok == T - 1) ? pad_mult : 1;
      count += mult;
      const float* kvr = kv + (size_t)tok * ldkv + 4 * lane;
      const float4 k4 = ld4(kvr);
      const float4 v4 = ld4(kvr + 256);
      float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const float mnew = fmaxf(m, s);
      const float alpha = expf(m - mnew);
      const float pw = (float)mult * expf(s - mnew);
      l = l * alpha + pw;
      float pv = pw;
      if (DROP)
        pv = drop_keep(drop.seed, drop.site, ((unsigned)row * 8u + (unsigned)(lane >> 3)) * drop.tokens_ref + (unsigned)tok,
                       drop.thr) ? pw * drop.scale : 0.0f;
      acc.x = acc.x * alpha + pv * v4.x; acc.y = acc.y * alpha + pv * v4.y;
      acc.z = acc.z * alpha + pv * v4.z; acc.w = acc.w * alpha + pv * v4.w;
      m = mnew;
    }
  }
  if (count > 0) {
    const float inv = 1.0f / l;
    return make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool DROP = false>
__device__ __forceinline__ float4 radar_attn_row(float cx, float cy, float b3, float b6, float b7,
                                                 float rmin, float rmax, float4 q4,
                                                 const float* rxy, int ld_xy, const float* kv,
                                                 int ldkv, int T, int pad_mult, int lane, int& count,
                                                 DropK drop = DropK(), int row = 0,
                                                 const unsigned long long* hit_masks = nullptr) {
  const GateGeom gg(cx, cy, b3, b6, b7, rmin, rmax);
  return radar_attn_row_g<DROP>(gg, q4, rxy, ld_xy, kv, ldkv, T, pad_mult, lane, count, drop, row, hit_masks);
}

// ---- backward of the gated attention core for ONE query row (one wavefront) ----------------------
// q4: this lane's 4 channels of the SCALED projected query; dO / o4: gradient and value of the attention
// output; the gate is re-evaluated exactly as the operator-level backward always did (cdist_mm < radius:
// the same decisions as the forward's squared-distance form, tc_radar_gate_selfcheck); softmax statistics
// are recomputed over the few hit tokens; dK | dV go to the token rows with atomics.  Returns dq w.r.t. the
// scaled query (the caller multiplies by the scale).  Used by train.hip (radar_attn_bwd_kernel) and by the
// backward row chain (chain.hip K_ATTN_BWD).
__device__ __forceinline__ float head_sum8(float s) {      // 8 lanes = one 32-channel head
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  return s;
}
__device__ __forceinline__ float4 radar_attn_bwd_row(float cx, float cy, float b3, float b6, float b7, float rmin,
                                                     float rmax, float4 q4, const float* rxy, int ld_xy,
                                                     const float* kv, float* dkv, int ldkv, int T, int pad_mult,
                                                     float4 dO, float4 o4, const DropK& drop, int row, int lane,
                                                     long long* sdkv = nullptr) {      // the shadow of dkv (DetAcc) or null
  // gate geometry: identical to radar_attn_row (HEAD:553-567)
  const float len = expf(b3);
  const float rs = -b6, rc = -b7;
  const float ox = __fmul_rn(__fmul_rn(len, 0.25f), rs), oy = __fmul_rn(__fmul_rn(len, 0.25f), rc);
  const float fx = __fadd_rn(cx, ox), fy = __fadd_rn(cy, oy);
  const float bxx = __fsub_rn(cx, ox), byy = __fsub_rn(cy, oy);
  const float rad = fminf(fmaxf(len / 2.0f, rmin), rmax);
  const float cn = sqnorm2(cx, cy), fn = sqnorm2(fx, fy), bn = sqnorm2(bxx, byy);
  const float D = head_sum8(dO.x * o4.x + dO.y * o4.y + dO.z * o4.z + dO.w * o4.w);
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  float m = -INFINITY, l = 0.f;
  // pass 0: softmax statistics over the hit tokens;  pass 1: gradients
  for (int pass = 0; pass < 2; ++pass) {
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      bool hit = false;
      if (t < T) {
        const float* y = rxy + (size_t)t * ld_xy;
        const float y0 = y[0], y1 = y[1];
        const float yn = sqnorm2(y0, y1);
        hit = (cdist_mm(cx, cy, cn, y0, y1, yn) < rad) || (cdist_mm(fx, fy, fn, y0, y1, yn) < rad) ||
              (cdist_mm(bxx, byy, bn, y0, y1, yn) < rad);
      }
      unsigned long long mask = __ballot(hit);
      while (mask) {
        const int j = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int tok = t0 + j;
        const float mult = (tok == T - 1) ? (float)pad_mult : 1.0f;
        const float* kvr = kv + (size_t)tok * ldkv + 4 * lane;
        const float4 k4 = ld4(kvr);
        const float sc = head_sum8(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w);
        if (pass == 0) {
          const float mnew = fmaxf(m, sc);
          l = l * expf(m - mnew) + mult * expf(sc - mnew);
          m = mnew;
        } else {
          const float4 v4 = ld4(kvr + 256);
          const float pj = mult * expf(sc - m) / l;
          // O = sum_j keep_j p_j v_j: dP_j = keep_j (dO . v_j), dV_j = keep_j p_j dO; D = dO . O as without
          float keep = 1.0f;
          if (drop.thr != 0)
            keep = drop_keep(drop.seed, drop.site,
                             ((unsigned)row * 8u + (unsigned)(lane >> 3)) * drop.tokens_ref + (unsigned)tok,
                             drop.thr) ? drop.scale : 0.0f;
          const float dp = keep * head_sum8(dO.x * v4.x + dO.y * v4.y + dO.z * v4.z + dO.w * v4.w);
          const float ds = pj * (dp - D);
          const float pk = pj * keep;
          dq.x += ds * k4.x; dq.y += ds * k4.y; dq.z += ds * k4.z; dq.w += ds * k4.w;
          float* dk = dkv + (size_t)tok * ldkv + 4 * lane;
          if (sdkv != nullptr) {       // deterministic mode (wave-uniform): integer atomics on the shadow of dkv
            long long* sk = sdkv + (size_t)tok * ldkv + 4 * lane;
            acc_add_at(sk + 0, dk + 0, ds * q4.x); acc_add_at(sk + 1, dk + 1, ds * q4.y);
            acc_add_at(sk + 2, dk + 2, ds * q4.z); acc_add_at(sk + 3, dk + 3, ds * q4.w);
            acc_add_at(sk + 256, dk + 256, pk * dO.x); acc_add_at(sk + 257, dk + 257, pk * dO.y);
            acc_add_at(sk + 258, dk + 258, pk * dO.z); acc_add_at(sk + 259, dk + 259, pk * dO.w);
          } else {
            unsafeAtomicAdd(dk + 0, ds * q4.x); unsafeAtomicAdd(dk + 1, ds * q4.y);
            unsafeAtomicAdd(dk + 2, ds * q4.z); unsafeAtomicAdd(dk + 3, ds * q4.w);
            unsafeAtomicAdd(dk + 256, pk * dO.x); unsafeAtomicAdd(dk + 257, pk * dO.y);
            unsafeAtomicAdd(dk + 258, pk * dO.z); unsafeAtomicAdd(dk + 259, pk * dO.w);
          }
        }
      }
    }
  }
  return dq;
}

}  // namespace tc
