// FPN -> head hand-off: NCHW [n_img, C, H, W] -> channels-last [n_img, H, W, C].
// The reference hands the head NCHW maps (DET:62-66); the sampling kernel wants
// a tap's 256 channels contiguous.  Pure HBM streaming (read + write the maps
// once), charged separately from the decoder in DESIGN.md; skipped entirely when
// the neck already emits channels_last tensors (ops.to_nhwc: zero-copy view).
//
// All levels of a frame in ONE launch (a launch per level left three tails and
// three ~1.5 us boundaries in a 0.1 ms hand-off).  A workgroup moves a 64-channel
// x 64-pixel tile through LDS: 16-byte global loads along the pixels (whenever a
// channel row is 16-byte aligned, i.e. H*W % 4 == 0 -- the two large levels of
// both configs, 94 % of the bytes), 16-byte global stores along the channels; the
// transposition itself is the scalar LDS traffic (row stride 65: at most 2-way
// bank conflicts on either side, LDS has 10x the bandwidth this needs).  The
// round-1 kernel moved 4 bytes per lane on both sides: 2.9 TB/s.
#include "kernels.hpp"

namespace tc {

namespace {

constexpr int TP = 64, TCH = 64, TLD = 65;

struct TrLevel { const float* src; float* dst; int HW, px_tiles, first_tile, vec, vst; };
struct TrArgs { TrLevel lv[TC_MAX_LEVELS]; int num_levels, C, c_tiles, n_img; };

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(TrArgs a) {
  __shared__ float tile[TCH][TLD];
  // which level: the tile ranges are ascending
  int l = 0;
#pragma unroll
  for (int i = 1; i < TC_MAX_LEVELS; ++i)
    if (i < a.num_levels && (int)blockIdx.x >= a.lv[i].first_tile) l = i;
  const TrLevel lv = a.lv[l];
  int t = (int)blockIdx.x - lv.first_tile;
  const int pt = t % lv.px_tiles; t /= lv.px_tiles;
  const int ct = t % a.c_tiles;
  const size_t img = (size_t)(t / a.c_tiles);
  const int p0 = pt * TP, c0 = ct * TCH, HW = lv.HW, C = a.C;
  const float* s = lv.src + img * (size_t)C * HW;
  float* d = lv.dst + img * (size_t)C * HW;
  const int tid = threadIdx.x;
  if (lv.vec) {
    // thread -> (channel c0 + (tid >> 4) + 16 i, pixels p0 + 4 (tid & 15) ..+3)
    const int px = p0 + 4 * (tid & 15);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + (tid >> 4) + 16 * i;
      v[i] = (px < HW && c < C) ? ldg4_stream(s + (size_t)c * HW + px) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* row = &tile[(tid >> 4) + 16 * i][4 * (tid & 15)];
      row[0] = v[i].x; row[1] = v[i].y; row[2] = v[i].z; row[3] = v[i].w;
    }
  } else {
    const int tx = tid & 63, ty = tid >> 6;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = c0 + i * 4 + ty, px = p0 + tx;
      v[i] = (px < HW && c < C) ? s[(size_t)c * HW + px] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[i * 4 + ty][tx] = v[i];
  }
  __syncthreads();
  if (lv.vst) {
    // thread -> (pixel p0 + (tid >> 4) + 16 i, channels c0 + 4 (tid & 15) ..+3)
    const int cc = 4 * (tid & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pl = (tid >> 4) + 16 * i, px = p0 + pl;
      const float4 o = make_float4(tile[cc][pl], tile[cc + 1][pl], tile[cc + 2][pl], tile[cc + 3][pl]);
      if (px < HW && c0 + cc < C) stg4_stream(d + (size_t)px * C + c0 + cc, o);
    }
  } else {
    const int tx = tid & 63, ty = tid >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int px = p0 + i * 4 + ty;
      if (px < HW && c0 + tx < C) d[(size_t)px * C + c0 + tx] = tile[tx][i * 4 + ty];
    }
  }
}

}  // namespace

int launch_nchw_to_nhwc_levels(const float* const* src, float* const* dst, int num_levels, int n_img,
                               int C, const int* H, const int* W, hipStream_t s) {
  TC_REQUIRE(num_levels >= 1 && num_levels <= TC_MAX_LEVELS, "nchw_to_nhwc: num_levels=%d", num_levels);
  TC_REQUIRE(n_img > 0 && C > 0, "nchw_to_nhwc: empty input");
  TrArgs a;
  a.num_levels = num_levels; a.C = C; a.c_tiles = (C + TCH - 1) / TCH; a.n_img = n_img;
  long long tiles = 0;
  for (int l = 0; l < TC_MAX_LEVELS; ++l) {
    TrLevel& lv = a.lv[l];
    if (l >= num_levels) { lv = TrLevel{nullptr, nullptr, 0, 1, 0x7fffffff, 0, 0}; continue; }
    TC_REQUIRE(src[l] != nullptr && dst[l] != nullptr && H[l] > 0 && W[l] > 0,
               "nchw_to_nhwc: level %d is empty", l);
    lv.src = src[l]; lv.dst = dst[l]; lv.HW = H[l] * W[l];
    lv.px_tiles = (lv.HW + TP - 1) / TP;
    lv.first_tile = (int)tiles;
    // 16-byte loads need every channel row 16-byte aligned
    lv.vec = ((lv.HW & 3) == 0 && (reinterpret_cast<size_t>(src[l]) & 15) == 0) ? 1 : 0;
    // 16-byte stores along the channels: C % 4 == 0 AND a 16-byte aligned destination (a caller's `out` view)
    lv.vst = ((C & 3) == 0 && (reinterpret_cast<size_t>(dst[l]) & 15) == 0) ? 1 : 0;
    tiles += (long long)lv.px_tiles * a.c_tiles * n_img;
    TC_REQUIRE(tiles < (1ll << 31), "nchw_to_nhwc: too many tiles");
  }
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
  return check_launch("nchw_to_nhwc");
}

int launch_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, hipStream_t s) {
  return launch_nchw_to_nhwc_levels(&src, &dst, 1, n_img, C, &H, &W, s);
}

}  // namespace tc
