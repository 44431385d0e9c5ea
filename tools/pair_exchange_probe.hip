// Round 6: what would a per-step exchange between a PAIR of workgroups cost?
//
// At one frame per launch the 4-row decoder chain is 225 workgroups, each streaming the layer's 3.18 MB of packed weights
// through its CU (78 % of a 52 us layer, profiles/r5_chain_stamps_4row.txt).  Proposal: a pair of workgroups shares 4 rows,
// each computes HALF the columns of every linear step (half the weight bytes per workgroup) and the two exchange their
// halves of the step's output through L2 before the next step (4 rows x 128 columns x 4 B = 2 KiB each way, 10 linear
// steps per layer).  This probe measures the exchange alone: 2 N workgroups of 256 threads (one per CU), pairs (b, b + d),
// STEPS rounds of { write 2 KiB, release a flag, spin on the partner's flag, read the partner's 2 KiB }, with d = 8 (both
// on one XCD: workgroups are dealt round-robin over the eight XCDs) and d = 1 (neighbouring XCDs: through the fabric).
//
//   hipcc --offload-arch=gfx950 -O3 tools/pair_exchange_probe.hip -o tools/pair_exchange_probe && tools/pair_exchange_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int STEPS = 200;
constexpr int HALF = 512;              // floats a workgroup hands over per step: 4 rows x 128 columns

__global__ __launch_bounds__(256) void exchange_kernel(float* buf, unsigned* flags, int npairs, int dist, long long* cycles, float* sink) {
  // pair p = blocks (lo, lo + dist): with dist = 8, p -> lo = (p / 8) * 16 + p % 8
  const int b = blockIdx.x;
  const int grp = b / (2 * dist), within = b % (2 * dist);
  const int side = within / dist, p = grp * dist + within % dist;
  if (p >= npairs) return;
  float* mine = buf + ((size_t)p * 2 + side) * HALF;
  const float* theirs = buf + ((size_t)p * 2 + (1 - side)) * HALF;
  unsigned* myflag = flags + (p * 2 + side) * 16;                 // one flag per 64-byte line
  volatile unsigned* theirflag = flags + (p * 2 + (1 - side)) * 16;
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 1; s <= STEPS; ++s) {
    // "the step's output": 2 floats per thread
    mine[threadIdx.x] = acc + (float)s;
    mine[threadIdx.x + 256] = acc - (float)s;
    __threadfence();                                              // the data is visible device-wide ...
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(myflag, (unsigned)s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // ... before the flag
      while (__hip_atomic_load((unsigned*)theirflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    const float a = __builtin_nontemporal_load(theirs + threadIdx.x), c = __builtin_nontemporal_load(theirs + threadIdx.x + 256);
    acc = acc * 0.5f + a - c;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[b] = t1 - t0;
  sink[(size_t)b * 256 + threadIdx.x] = acc;
}

int main() {
  const int npairs = 112;                 // 224 workgroups: one per CU, as the 225 tiles of a one-frame launch
  float *buf, *sink;
  unsigned* flags;
  long long* cyc;
  CHECK(hipMalloc(&buf, (size_t)npairs * 2 * HALF * 4));
  CHECK(hipMalloc(&flags, (size_t)npairs * 2 * 16 * 4));
  CHECK(hipMalloc(&cyc, 256 * 8));
  CHECK(hipMalloc(&sink, 256 * 256 * 4));
  long long h[256];
  for (int dist : {8, 1}) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(flags, 0, (size_t)npairs * 2 * 16 * 4));
      CHECK(hipMemset(cyc, 0, 256 * 8));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(exchange_kernel, dim3(2 * npairs), dim3(256), 0, 0, buf, flags, npairs, dist, cyc, sink);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost));
    long long mn = 1ll << 60, mx = 0, sum = 0;
    int n = 0;
    for (int i = 0; i < 2 * npairs; ++i) if (h[i] > 0) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; sum += h[i]; ++n; }
    printf("pairs (b, b + %d) %s: %d workgroups, %d exchanges of 2 x 2 KiB: kernel %.1f us = %.2f us per exchange (HIP events); s_memtime ticks per\n"
           "  exchange min / mean / max %.0f / %.0f / %.0f\n", dist, dist == 8 ? "[same XCD]" : "[neighbouring XCDs]", n, STEPS, ms * 1e3,
           ms * 1e3 / STEPS, (double)mn / STEPS, (double)sum / n / STEPS, (double)mx / STEPS);
  }
  printf("a decoder layer has 10 linear steps: the exchanges alone cost 10 x the figure above; halving a 4-row workgroup's weight stream\n"
         "(3.18 MB at 35 B/clk per CU = 38 us of a 52 us layer) could save at most ~19 us per layer.\n");
  return 0;
}
