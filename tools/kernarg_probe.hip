// Can a kernel take 16 KB of by-value arguments on this stack, and what does a launch
// with such an argument cost inside a replayed hipGraph?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int* out) {
  int s = 0;
  for (int i = threadIdx.x; i < N; i += 64) s += b.v[i];
  if (s == 12345678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = b.v[N - 1];
}
template <int N> int run(int* out) {
  Big<N> b;
  for (int i = 0; i < N; ++i) b.v[i] = i;
  hipStream_t s; CK(hipStreamCreate(&s));
  k<N><<<225, 256, 0, s>>>(b, out);
  CK(hipStreamSynchronize(s));
  int h[2]; CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 50; ++i) k<N><<<225, 256, 0, s>>>(b, out);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("kernarg %6d bytes: last=%d (expect %d), %.2f us per launch in a 50-launch graph\n", N * 4, h[1], N - 1, ms * 1e3 / 50);
  return 0;
}
int main() {
  int* out; CK(hipMalloc(&out, 64));
  run<16>(out); run<320>(out); run<1000>(out); run<2048>(out); run<4000>(out);
  return 0;
}
