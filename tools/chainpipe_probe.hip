// Diagnostic for chain.hip's weight pipeline: 225 workgroups x 4 waves, every wave
// streams NITEMS packed 16 KiB weight items (all workgroups read the SAME 3.2 MB,
// as in a decoder layer), 16 ds_read_b128 + 64 v_mfma_f32_4x4x1 per item (R = 4).
// Variants: prefetch depth D (register buffers in flight), with / without the MFMA
// work, item streams split into "steps" of S items that drain the pipeline.
//   hipcc -O3 --offload-arch=gfx950 tools/chainpipe_probe.hip -o tools/chainpipe_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA44(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, 0, 0)

struct WBuf { float4 b[16]; };
__device__ __forceinline__ void wload(WBuf& w, const float* p) {
#pragma unroll
  for (int i = 0; i < 16; ++i) w.b[i] = *(const float4*)(p + i * 256);
}
__device__ __forceinline__ void compute(f32x4& a0, f32x4& a1, const WBuf& w, const float* arow, int do_mfma) {
  float4 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = *(const float4*)(arow + 4 * i);
  if (do_mfma) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a0 = MFMA44(a[i].x, w.b[i].x, a0);
      a1 = MFMA44(a[i].y, w.b[i].y, a1);
      a0 = MFMA44(a[i].z, w.b[i].z, a0);
      a1 = MFMA44(a[i].w, w.b[i].w, a1);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[0] += w.b[i].x * a[i].x; a1[0] += w.b[i].y + w.b[i].z + w.b[i].w; }
  }
}

// D buffers, items of one "step" (S items) pipelined, pipeline drained between steps
template <int D>
__global__ __launch_bounds__(256) void pipe(const float* W, int nsteps, int S, int wrap, int do_mfma, float* out,
                                            long long* cyc) {
  __shared__ float A[4][516];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 516; i += 256) (&A[0][0])[i] = 0.001f * i;
  __syncthreads();
  const float* arow = &A[lane & 3][0];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  int item = wave * 7;
  for (int st = 0; st < nsteps; ++st) {
    WBuf w[D];
    auto src = [&](int i) { return W + (size_t)((item + i) % wrap) * 4096 + 4 * lane; };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) wload(w[d], src(d < S ? d : S - 1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int i = 0; i < S; i += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int nx = i + d + D - 1;
        wload(w[(d + D - 1) % D], src(nx < S ? nx : S - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (i + d < S) compute(a0, a1, w[d], arow + ((i + d) & 7) * 64, do_mfma);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    item += S;
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a0[1] + a1[0] + a1[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D>
int run(const float* W, float* out, long long* cyc, int blocks, int nsteps, int S, int wrap, int do_mfma) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pipe<D><<<blocks, 256>>>(W, nsteps, S, wrap, do_mfma, out, cyc);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    pipe<D><<<blocks, 256>>>(W, nsteps, S, wrap, do_mfma, out, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const int items = nsteps * S;
  printf("D=%d blocks=%3d steps=%3d x %2d items mfma=%d: %7.1f us, %6.0f ns/item = %5.0f cyc@2.4GHz, %.1f GB/s per CU (useful)\n",
         D, blocks, nsteps, S, do_mfma, best * 1e3, best * 1e6 / items, best * 1e6 / items * 2.4,
         4.0 * 16384.0 * items / (best * 1e-3) / 1e9);
  return 0;
}

// interleaved: the 16 loads of item i+1 are issued one per 4 MFMAs of item i
template <int GAP>
__device__ __forceinline__ void compute_il(f32x4& a0, f32x4& a1, const WBuf& w, WBuf& nx, const float* np,
                                           const float* arow) {
  float4 a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = *(const float4*)(arow + 4 * i);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    nx.b[i] = *(const float4*)(np + i * 256);
    if (GAP) __builtin_amdgcn_sched_barrier(0);
    a0 = MFMA44(a[i].x, w.b[i].x, a0);
    a1 = MFMA44(a[i].y, w.b[i].y, a1);
    a0 = MFMA44(a[i].z, w.b[i].z, a0);
    a1 = MFMA44(a[i].w, w.b[i].w, a1);
    if (GAP) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int GAP>
__global__ __launch_bounds__(256) void pipe_il(const float* W, int nsteps, int S, int wrap, float* out, int desync = 0) {
  __shared__ float A[4][516];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 516; i += 256) (&A[0][0])[i] = 0.001f * i;
  __syncthreads();
  const float* arow = &A[lane & 3][0];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  int item = wave * 7 + desync * blockIdx.x * 13;
  for (int st = 0; st < nsteps; ++st) {
    WBuf w0, w1;
    auto src = [&](int i) { return W + (size_t)((item + (i < S ? i : S - 1)) % wrap) * 4096 + 4 * lane; };
    wload(w0, src(0));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int i = 0; i < S; i += 2) {
      compute_il<GAP>(a0, a1, w0, w1, src(i + 1), arow + (i & 7) * 64);
      __builtin_amdgcn_sched_barrier(0);
      compute_il<GAP>(a0, a1, w1, w0, src(i + 2), arow + ((i + 1) & 7) * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
    item += S;
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a0[1] + a1[0] + a1[3];
}

template <int GAP>
int run_il(const float* W, float* out, int blocks, int nsteps, int S, int wrap, int desync = 0) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pipe_il<GAP><<<blocks, 256>>>(W, nsteps, S, wrap, out, desync);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    pipe_il<GAP><<<blocks, 256>>>(W, nsteps, S, wrap, out, desync);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const int items = nsteps * S;
  printf("interleaved gap=%d desync=%d blocks=%3d steps=%3d x %2d items: %7.1f us, %6.0f ns/item = %5.0f cyc@2.4GHz, %.1f GB/s per CU\n",
         GAP, desync, blocks, nsteps, S, best * 1e3, best * 1e6 / items, best * 1e6 / items * 2.4,
         4.0 * 16384.0 * items / (best * 1e-3) / 1e9);
  return 0;
}

// single buffer: b[j] is reloaded with the next item's chunk right after its 4 MFMAs
__global__ __launch_bounds__(256) void pipe_sb(const float* W, int nsteps, int S, int wrap, float* out) {
  __shared__ float A[4][516];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 516; i += 256) (&A[0][0])[i] = 0.001f * i;
  __syncthreads();
  const float* arow0 = &A[lane & 3][0];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  int item = wave * 7;
  WBuf w;
  wload(w, W + (size_t)(item % wrap) * 4096 + 4 * lane);
  for (int st = 0; st < nsteps; ++st) {
#pragma unroll 1
    for (int i = 0; i < S; ++i) {
      const float* np = W + (size_t)((item + i + 1) % wrap) * 4096 + 4 * lane;   // crosses step boundaries
      const float* arow = arow0 + (i & 7) * 64;
      float4 a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = *(const float4*)(arow + 4 * j);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        a0 = MFMA44(a[j].x, w.b[j].x, a0);
        a1 = MFMA44(a[j].y, w.b[j].y, a1);
        a0 = MFMA44(a[j].z, w.b[j].z, a0);
        a1 = MFMA44(a[j].w, w.b[j].w, a1);
        __builtin_amdgcn_sched_barrier(0);
        w.b[j] = *(const float4*)(np + j * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    item += S;
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a0[1] + a1[0] + a1[3] + w.b[3].x;
}

int run_sb(const float* W, float* out, int blocks, int nsteps, int S, int wrap) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pipe_sb<<<blocks, 256>>>(W, nsteps, S, wrap, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    pipe_sb<<<blocks, 256>>>(W, nsteps, S, wrap, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const int items = nsteps * S;
  printf("single-buffer cross-step blocks=%3d steps=%3d x %2d items: %7.1f us, %6.0f ns/item = %5.0f cyc@2.4GHz, %.1f GB/s per CU\n",
         blocks, nsteps, S, best * 1e3, best * 1e6 / items, best * 1e6 / items * 2.4,
         4.0 * 16384.0 * items / (best * 1e-3) / 1e9);
  return 0;
}

// interleaved, THREE buffers: item i computes while the loads of item i+2 are issued
__global__ __launch_bounds__(256) void pipe_il3(const float* W, int nsteps, int S, int wrap, float* out) {
  __shared__ float A[4][516];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 516; i += 256) (&A[0][0])[i] = 0.001f * i;
  __syncthreads();
  const float* arow = &A[lane & 3][0];
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  int item = wave * 7;
  for (int st = 0; st < nsteps; ++st) {
    WBuf w0, w1, w2;
    auto src = [&](int i) { return W + (size_t)((item + (i < S ? i : S - 1)) % wrap) * 4096 + 4 * lane; };
    wload(w0, src(0));
    wload(w1, src(1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int i = 0; i < S; i += 3) {
      compute_il<1>(a0, a1, w0, w2, src(i + 2), arow + (i & 7) * 64);
      __builtin_amdgcn_sched_barrier(0);
      compute_il<1>(a0, a1, w1, w0, src(i + 3), arow + ((i + 1) & 7) * 64);
      __builtin_amdgcn_sched_barrier(0);
      compute_il<1>(a0, a1, w2, w1, src(i + 4), arow + ((i + 2) & 7) * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
    item += S;
    __syncthreads();
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a0[1] + a1[0] + a1[3];
}
int run_il3(const float* W, float* out, int blocks, int nsteps, int S, int wrap) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  pipe_il3<<<blocks, 256>>>(W, nsteps, S, wrap, out);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    pipe_il3<<<blocks, 256>>>(W, nsteps, S, wrap, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const int items = nsteps * S;
  printf("interleaved 3 buffers blocks=%3d steps=%3d x %2d items: %7.1f us, %6.0f ns/item = %5.0f cyc@2.4GHz\n",
         blocks, nsteps, S, best * 1e3, best * 1e6 / items, best * 1e6 / items * 2.4);
  return 0;
}

int main() {
  const int wrap = 200;   // 200 items x 16 KiB = 3.2 MB, one decoder layer's weights
  float *W, *out; long long* cyc;
  CK(hipMalloc(&W, (size_t)(wrap + 64) * 16384)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMemset(W, 0, (size_t)(wrap + 64) * 16384));
  for (int mf = 1; mf >= 0; --mf) {
    run<2>(W, out, cyc, 225, 14, 4, wrap, mf);     // decoder-like: 14 steps of 4 items
    run<3>(W, out, cyc, 225, 14, 4, wrap, mf);
    run<4>(W, out, cyc, 225, 14, 4, wrap, mf);
    run<2>(W, out, cyc, 225, 7, 8, wrap, mf);
    run<3>(W, out, cyc, 225, 7, 8, wrap, mf);
    run<4>(W, out, cyc, 225, 7, 8, wrap, mf);
    run<2>(W, out, cyc, 225, 1, 56, wrap, mf);     // one long stream
    run<3>(W, out, cyc, 225, 1, 56, wrap, mf);
    run<4>(W, out, cyc, 225, 1, 56, wrap, mf);
  }
  for (int S : {4, 8, 56}) {
    run_il<0>(W, out, 225, 56 / S, S, wrap);
    run_il<1>(W, out, 225, 56 / S, S, wrap);
  }
  run_il<1>(W, out, 1, 1, 56, wrap);
  run_il3(W, out, 225, 1, 57, wrap);
  run_il3(W, out, 225, 7, 9, wrap);
  run_il3(W, out, 225, 19, 3, wrap);
  for (int S : {4, 8, 56}) run_il<1>(W, out, 225, 56 / S, S, wrap, 1);
  for (int S : {4, 8, 56}) run_sb(W, out, 225, 56 / S, S, wrap);
  run_sb(W, out, 1, 1, 56, wrap);
  run<2>(W, out, cyc, 113, 1, 56, wrap, 1);
  run<3>(W, out, cyc, 57, 1, 56, wrap, 1);
  run<3>(W, out, cyc, 1, 1, 56, wrap, 1);
  return 0;
}
