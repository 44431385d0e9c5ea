// Feasibility of the "shared weight ring" row chain: a workgroup of 4 waves owns 16 rows, wave w the
// rows 4w..4w+3; every 16 KiB weight item (64 output columns x 64 k, packed [k/4][lane][4] as
// pack.hip writes it) is fetched ONCE per workgroup into an LDS ring by LDS-DMA (each wave issues a
// quarter of the item) and read by all four waves (ds_read_b128, lane-linear: conflict free), 64
// v_mfma_f32_4x4x1 per wave and item.  Per row the weight bytes through the CU's vector-memory path
// are 1/4 of the 4-row tiles' (which run AT that path's limit, tools/hotspot_probe.hip).
// Prints cycles per item (all workgroups stream the same nitems x 16 KiB).
//   hipcc --offload-arch=gfx950 -O3 tools/ring_probe.hip -o tools/ring_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA44(a, b, c, grp) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (grp), 0)

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NS, int MODE>      // MODE 0: LDS-DMA ring; 1: the same without MFMAs (transfer floor); 2: without LDS reads + MFMA
__global__ __launch_bounds__(256) void ring(const float* __restrict__ W, int nitems, float* out, long long* cyc) {
  extern __shared__ __align__(16) float lds[];           // NS * 4096 floats ring + 16 * 64 floats A
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* A = lds + NS * 4096;
  for (int i = threadIdx.x; i < 16 * 64; i += 256) A[i] = 0.001f * i;
  const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
  constexpr int D = NS - 1;
  auto issue = [&](int item) {
    const int slot = item % NS;
    const float* src = W + (size_t)item * 4096 + (wave * 4) * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      glds16(src + c * 256, ring_base + (unsigned)(slot * 4096 + (wave * 4 + c) * 256) * 4u);
  };
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < D && i < nitems; ++i) issue(i);
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int i = 0; i < nitems; ++i) {
    // my quarter of item i has landed when at most 4 * (D - 1) of my transfers are outstanding
    if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // everybody's quarter landed; everybody finished reading item i - 1
    if (i + D < nitems) issue(i + D);       // into the slot of item i - 1
    if (MODE == 2) continue;
    const float* slot = lds + (i % NS) * 4096 + lane * 4;
    float4 b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = *reinterpret_cast<const float4*>(slot + j * 256);
    const float4 ar = *reinterpret_cast<const float4*>(A + (wave * 4 + (lane & 3)) * 64 + 4 * (lane >> 2));
    if (MODE == 1) { acc0[0] += b[0].x + b[5].y + b[10].z + b[15].w + ar.x; continue; }
#define STEP(j)                                  \
    acc0 = MFMA44(ar.x, b[j].x, acc0, j);          \
    acc1 = MFMA44(ar.y, b[j].y, acc1, j);          \
    acc0 = MFMA44(ar.z, b[j].z, acc0, j);          \
    acc1 = MFMA44(ar.w, b[j].w, acc1, j);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NS, int MODE>
int run(const float* W, float* out, long long* cyc, int blocks, int nitems, const char* tag) {
  const size_t lds = (size_t)(NS * 4096 + 16 * 64) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ring<NS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  ring<NS, MODE><<<blocks, 256, lds>>>(W, nitems, out, cyc);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    ring<NS, MODE><<<blocks, 256, lds>>>(W, nitems, out, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  long long h[1024]; CK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
  long long mx = 0; for (int i = 0; i < blocks; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%-22s ring %d slots, %3d workgroups: %6.1f us per %d items = %5.0f ns per item (%4.0f cycles at 2.4 GHz); in-kernel s_memtime ticks per item %.0f\n",
         tag, NS, blocks, best * 1e3, nitems, best * 1e6 / nitems, best * 1e-3 * 2.4e9 / nitems, (double)mx / nitems);
  return 0;
}
int main() {
  const int nitems = 195;     // one decoder layer: 3.15 MB of packed weights
  float *W, *out; long long* cyc;
  CK(hipMalloc(&W, (size_t)nitems * 16384)); CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
  CK(hipMemset(W, 0, (size_t)nitems * 16384));
  for (int blocks : {1, 113, 225}) {
    run<2, 0>(W, out, cyc, blocks, nitems, "DMA + read + MFMA");
    run<3, 0>(W, out, cyc, blocks, nitems, "DMA + read + MFMA");
    run<4, 0>(W, out, cyc, blocks, nitems, "DMA + read + MFMA");
    run<5, 0>(W, out, cyc, blocks, nitems, "DMA + read + MFMA");
    run<4, 1>(W, out, cyc, blocks, nitems, "DMA + read");
    run<4, 2>(W, out, cyc, blocks, nitems, "DMA only");
  }
  return 0;
}
