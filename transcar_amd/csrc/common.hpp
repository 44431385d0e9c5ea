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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// XFMR:17-32
__device__ __forceinline__ float inverse_sigmoidf_(float x) {
  const float eps = 1e-5f;
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  float x1 = fmaxf(x, eps);
  float x2 = fmaxf(1.0f - x, eps);
  return logf(x1 / x2);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

}  // namespace tc
