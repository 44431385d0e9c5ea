// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace tc {

constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);   // hipGetLastError -> message; returns code

#define TC_REQUIRE(cond, ...)                       \
  do {                                              \
    if (!(cond)) {                                  \
      tc::set_error(__VA_ARGS__);                   \
      return -1;                                    \
    }                                               \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// "done once per DEVICE" (hipFuncSetAttribute is per device: a second GPU used by the same
// process needs its own call).  Racing first calls both set the attribute -- harmless.
struct DeviceOnce {
  unsigned long long mask[4] = {0, 0, 0, 0};     // up to 256 devices; the only shared state
  // returns the calling thread's device when its attribute call is still to be made, -1 when done
  // (the device travels through the caller: two host threads on different GPUs share this object)
  int need() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) return 256;     // unknown: always set
    return ((__atomic_load_n(&mask[dev >> 6], __ATOMIC_ACQUIRE) >> (dev & 63)) & 1ull) == 0 ? dev : -1;
  }
  void done(int dev) { if (dev >= 0 && dev < 256) __atomic_fetch_or(&mask[dev >> 6], 1ull << (dev & 63), __ATOMIC_RELEASE); }
};

// bump allocator over the caller's workspace (256-byte aligned slices)
struct Arena {
  char* base;
  size_t cap;
  size_t off;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), cap(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(base + off);
    off += bytes;
    return p;
  }
  bool ok() const { return off <= cap; }
};
inline size_t arena_slice(size_t count, size_t elt) { return (count * elt + 255) & ~size_t(255); }

// ---- device helpers --------------------------------------------------------
// Wave64 sum on the DPP data path (quad_perm, row_half_mirror, row_mirror,
// row_bcast:15/31 -> lane 63): ~6 VALU ops instead of six ds_bpermute round trips
// through the LDS crossbar (a LayerNorm row needs two of these back to back).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);    // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);    // row_mirror: every lane of a 16-row holds the row sum
  v = dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// torch.relu keeps a NaN (fmaxf would return 0 -- and hide, behind every ReLU, what an overflow of the f16 planes did)
__device__ __forceinline__ float relu_(float x) { return x < 0.0f ? 0.0f : x; }

// XFMR:17-32
__device__ __forceinline__ float inverse_sigmoidf_(float x) {
  const float eps = 1e-5f;
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  float x1 = fmaxf(x, eps);
  float x2 = fmaxf(1.0f - x, eps);
  return logf(x1 / x2);
}

// Global-memory accessors with an explicit address space: a pointer that went
// through LDS or an integer (the chain kernel's resolved step records) is "generic"
// to the compiler, which then emits flat_load/flat_store -- those tick lgkmcnt as
// well as vmcnt, so every LDS read afterwards waits for the weight stream.
#define TC_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 ld4(const float* p) {
  const f32x4 v = *(const TC_GLOBAL f32x4*)(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *(TC_GLOBAL f32x4*)(p) = f32x4{v.x, v.y, v.z, v.w};
}
// read-once / write-once streams (the hand-off transposes): non-temporal hint
__device__ __forceinline__ float4 ldg4_stream(const float* p) {
  const f32x4 v = __builtin_nontemporal_load((const TC_GLOBAL f32x4*)(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stg4_stream(float* p, float4 v) {
  *(TC_GLOBAL f32x4*)(p) = f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ float ldg1(const float* p) { return *(const TC_GLOBAL float*)(p); }
__device__ __forceinline__ void stg1(float* p, float v) { *(TC_GLOBAL float*)(p) = v; }


// Deterministic accumulation (round 5, tc_radar_train_bwd_fused_det).  A value that would be added with a FLOAT atomic --
// whose sum depends on the order the adds arrive in -- is rounded to a 64-bit fixed-point number (units of 2^-40; exact for
// |v| >= 2^-16, an absolute 9e-13 below) and added with an INTEGER atomic into a shadow of its target: integer sums are
// exact, so the result does not depend on the order.  det_flush adds the shadows back into the targets (and zeroes them).
// Two address ranges: the flat gradient bucket, and the backward workspace's dK | dV accumulators.  shadow[r] == nullptr:
// off (plain float atomics).  The fixed-point word holds |v| < 2^23 (8.4e6); a value it cannot hold -- NaN, inf, or
// larger -- goes to the FLOAT target with a float atomic instead (det_fits / acc_add_at; round 6): the flush adds
// the shadow onto it, so an exploding or non-finite gradient stays visible in the parameters' gradients exactly as in the
// default mode (float2ll of such a value is undefined and would have produced a finite wrong sum).
struct DetAcc {
  const float* lo[2]; const float* hi[2]; long long* shadow[2];
};
constexpr float DET_SCALE = 1099511627776.0f;        // 2^40
// the shadow word of *p, or nullptr (p outside both ranges / mode off): resolved ONCE per target tensor where the target
// is known at launch (the GEMMs' C and column sums, LayerNorm parameter gradients, a sample's dK | dV) -- a range test
// per atomic cost the float-atomic default 2 % of an iteration
__host__ __device__ __forceinline__ long long* det_shadow_of(const DetAcc& d, const float* p) {
  for (int r = 0; r < 2; ++r)
    if (d.shadow[r] != nullptr && p >= d.lo[r] && p < d.hi[r]) return d.shadow[r] + (p - d.lo[r]);
  return nullptr;
}
__device__ __forceinline__ bool det_fits(float v) { return fabsf(v) < 8388608.0f; }      // false for NaN / inf / |v| >= 2^23
// *p += v: through the shadow word sp (of p) when there is one (wave-uniform test) and it can hold v, a float atomic otherwise
__device__ __forceinline__ void acc_add_at(long long* sp, float* p, float v) {
  if (sp != nullptr && det_fits(v)) atomicAdd(reinterpret_cast<unsigned long long*>(sp), (unsigned long long)__float2ll_rn(v * DET_SCALE));
  else unsafeAtomicAdd(p, v);
}
__device__ __forceinline__ void acc_add(const DetAcc& d, float* p, float v) { acc_add_at(det_shadow_of(d, p), p, v); }

// ---- dropout (training): counter-based Bernoulli masks ----------------------
// keep(seed, site, idx): ONE splitmix64 of (seed, site, idx / 4) decides the four elements 4 (idx / 4) .. + 3 from its
// four 16-bit fields (field idx % 4 >= round(p * 2^16)): the forward and the backward regenerate the same mask from
// (seed, site, element index) -- no mask tensor is stored.  Round 4: until then one hash per ELEMENT (>= thr =
// round(p * 2^32)): three 64-bit multiplies per element were ~65 us of a 150 us train-mode attention core and of a
// 170 us train-mode decoder chain (58 M probabilities / 19 M activations per nine frames); a lane that owns four
// consecutive elements (the 16-row epilogues, the attention probabilities) now pays one (drop_keep4).
// site = 4 * radar layer + {0: attention probabilities, 1: rf_dropout2, 2: rf_dropout (FFN), 3: rf_dropout3}
// (HEAD:129-171); the frozen decoder's sites: 16 + 8 * layer + {0..4}.
struct DropK {
  unsigned long long seed;
  unsigned thr;                // 0: dropout off
  float scale;                 // 1 / (1 - p)
  unsigned site;
  unsigned tokens_ref;         // stride of the probability index (1500 reference tokens)
  // Frozen decoder, several frames in one launch (tc_head_options.dropout_seed_stride): sample b draws its masks from
  // seed + b * seed_stride with element indices relative to the sample -- the masks that frame would draw launched
  // alone with that seed.  rows_per_sample = 0: one index space over the whole batch (every other caller).
  unsigned long long seed_stride;
  unsigned rows_per_sample;
  unsigned pad_;
};
__host__ __device__ __forceinline__ unsigned long long drop_hash(unsigned long long seed, unsigned site, unsigned group) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((((unsigned long long)site) << 32) | group) + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__host__ __device__ __forceinline__ unsigned drop_thr16(unsigned thr) { return (unsigned)(((unsigned long long)thr + 0x8000ull) >> 16); }
__host__ __device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned site, unsigned idx,
                                                   unsigned thr) {
  const unsigned long long z = drop_hash(seed, site, idx >> 2);
  return (unsigned)((z >> (16u * (idx & 3u))) & 0xFFFFull) >= drop_thr16(thr);
}
// the decisions of elements idx .. idx + 3 as bits 0..3: one hash when idx is a multiple of 4 (the callers' layouts
// make it one), two otherwise
__host__ __device__ __forceinline__ unsigned drop_keep4(unsigned long long seed, unsigned site, unsigned idx, unsigned thr) {
  const unsigned t = drop_thr16(thr);
  const unsigned long long z = drop_hash(seed, site, idx >> 2);
  const unsigned a = idx & 3u;
  unsigned m = 0;
  if (a == 0u) {
#pragma unroll
    for (unsigned i = 0; i < 4u; ++i) m |= (unsigned)(((z >> (16u * i)) & 0xFFFFull) >= t) << i;
    return m;
  }
  const unsigned long long z1 = drop_hash(seed, site, (idx >> 2) + 1u);
#pragma unroll
  for (unsigned i = 0; i < 4u; ++i) {
    const unsigned e = a + i;
    const unsigned long long zz = e < 4u ? z : z1;
    m |= (unsigned)(((zz >> (16u * (e & 3u))) & 0xFFFFull) >= t) << i;
  }
  return m;
}
inline DropK make_drop(float p, unsigned long long seed, unsigned site, unsigned tokens_ref) {
  DropK d;
  d.seed = seed; d.site = site; d.tokens_ref = tokens_ref;
  d.seed_stride = 0; d.rows_per_sample = 0; d.pad_ = 0;
  d.thr = p > 0.0f ? (unsigned)((double)p * 4294967296.0 + 0.5) : 0u;
  d.scale = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}

}  // namespace tc
