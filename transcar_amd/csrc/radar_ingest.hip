// Radar ingest on the device: the reference builds the 36-feature rows
// in numpy on the main thread inside Detr3DHead.forward (HEAD:301-536).  Here the raw devkit rows
// of a sample (18 float64 fields per point, five radars concatenated in RADAR_CHANNELS order) are
// uploaded as they are and ONE workgroup turns them into the [T, 36] float32 token matrix the
// fusion chains read: velocity rotation radar -> ego -> lidar frame (HEAD:317-327), time offsets
// relative to each radar's newest sweep (HEAD:453-455), one-hot state columns (HEAD:499-510),
// the point-range filter (HEAD:304, 512-521) as an order-preserving compaction, float32
// conversion and the 500.0 padding (HEAD:523-530).  All arithmetic in float64 like the reference,
// rounded to float32 once.  With a fixed T this makes the radar input of a frame pipeline lane a
// pure device-side refill (pad_mult = 1500 - T + 1 does not depend on the frame).
#include "kernels.hpp"

namespace tc {

namespace {

constexpr int RI_RAW = 18, RI_OUT = 36, MAX_CHAN = 8, NT = 256;
constexpr int REF_TOKENS = 1500;     // HEAD:526: the reference always attends over 1500 tokens

struct IngestK {
  const double* raw;        // [N, 18] point-major
  const double* times;      // [N]
  int chan_start[MAX_CHAN + 1];
  int num_chan, N, T;
  double rot_radar[MAX_CHAN][9];   // radar -> ego, row-major
  double rot_ref[9];               // lidar -> ego (applied transposed)
  double lo[3], hi[3];
  float* tokens;            // [T, 36]
  int* count;               // [1]
};

__device__ __forceinline__ unsigned long long ord(double x) {   // order-preserving map double -> u64
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unord(unsigned long long u) {
  const unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)b);
}

__device__ __forceinline__ void radar_ingest_body(const IngestK& k) {
  __shared__ unsigned long long tmax[MAX_CHAN];
  __shared__ int wave_cnt[NT / 64];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < MAX_CHAN) tmax[tid] = 0ull;                       // ord(x) > 0 for every double
  if (tid == 0) base = 0;
  __syncthreads();
  // newest sweep time of every radar (HEAD:453-455: times - max(times))
  for (int c = 0; c < k.num_chan; ++c) {
    unsigned long long m = 0ull;
    for (int i = k.chan_start[c] + tid; i < k.chan_start[c + 1]; i += NT) m = max(m, ord(k.times[i]));
    if (m) atomicMax(&tmax[c], m);
  }
  __syncthreads();
  // T < 1500: row T - 1 stands for the 1500 - T + 1 pad rows the head folds into it (pad_mult), so it must
  // STAY a 500.0 pad row -- a real return there would be weighted pad_mult times by the gate and the
  // attention.  At most T - 1 points are kept then; `count` (> T - 1) tells the caller the frame did not fit.
  const int limit = k.T < REF_TOKENS ? k.T - 1 : k.T;
  for (int i0 = 0; i0 < k.N; i0 += NT) {
    const int i = i0 + tid;
    bool keep = false;
    const double* r = k.raw + (size_t)min(i, k.N - 1) * RI_RAW;
    if (i < k.N)
      keep = r[0] > k.lo[0] && r[1] > k.lo[1] && r[2] > k.lo[2] && r[0] < k.hi[0] && r[1] < k.hi[1] && r[2] < k.hi[2];
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && off < limit) {
      int c = 0;
      while (c + 1 < k.num_chan && i >= k.chan_start[c + 1]) ++c;
      const double* R = k.rot_radar[c];
      const double t = k.times[i] - unord(tmax[c]);
      float* o = k.tokens + (size_t)off * RI_OUT;
      // x y z id rcs is_quality_valid invalid_state (HEAD:499)
      o[0] = (float)r[0]; o[1] = (float)r[1]; o[2] = (float)r[2]; o[3] = (float)r[4]; o[4] = (float)r[5];
      o[5] = (float)r[10]; o[6] = (float)r[14];
      o[7] = (float)t; o[8] = (float)t;
      double vxy[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {            // s = 0: compensated velocity (fields 8, 9); 1: raw (6, 7)
        const double vx = r[s == 0 ? 8 : 6], vy = r[s == 0 ? 9 : 7];
        double e[3];                            // radar -> ego, z component of the input is 0
#pragma unroll
        for (int a = 0; a < 3; ++a) e[a] = fma(R[3 * a + 2], 0.0, fma(R[3 * a + 1], vy, R[3 * a + 0] * vx));
#pragma unroll
        for (int a = 0; a < 2; ++a)             // ego -> lidar: rot_ref^T
          vxy[s][a] = fma(k.rot_ref[6 + a], e[2], fma(k.rot_ref[3 + a], e[1], k.rot_ref[a] * e[0]));
      }
      o[9] = (float)(vxy[0][0] * t); o[10] = (float)(vxy[0][1] * t);
      o[11] = (float)vxy[0][0]; o[12] = (float)vxy[0][1];
      o[13] = (float)vxy[1][0]; o[14] = (float)vxy[1][1];
      const int dyn = (int)r[3], amb = (int)r[11], pdh = (int)r[15];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[15 + j] = j == dyn ? 1.0f : 0.0f;
#pragma unroll
      for (int j = 0; j < 5; ++j) o[23 + j] = j == amb ? 1.0f : 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[28 + j] = j == pdh ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (tid == 0) { int s = base; for (int w = 0; w < NT / 64; ++w) s += wave_cnt[w]; base = s; }
    __syncthreads();
  }
  // pad tokens (HEAD:526-530)
  const int filled = min(base, limit);
  for (int j = filled * RI_OUT + tid; j < k.T * RI_OUT; j += NT) k.tokens[j] = 500.0f;
  if (tid == 0 && k.count != nullptr) k.count[0] = base;
}

// one sample, its parameters by value (kernel arguments)
__global__ __launch_bounds__(NT) void radar_ingest_kernel(IngestK k) { radar_ingest_body(k); }

// P samples in one launch, workgroup b = sample b, its parameters from a DEVICE-resident descriptor: the form a
// captured hipGraph can replay on new frames (a lane's producer refills raw rows + descriptors, nothing is baked
// into the graph's kernel arguments).  raw [P, cap, 18], times [P, cap], tokens [P, T, 36], count [P].
struct IngestBatchK { const double* raw; const double* times; const tc_radar_frame_desc* desc; int cap, T; float* tokens; int* count; };
__global__ __launch_bounds__(NT) void radar_ingest_batch_kernel(IngestBatchK a) {
  __shared__ IngestK sk;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    const tc_radar_frame_desc& d = a.desc[b];
    sk.raw = a.raw + (size_t)b * a.cap * RI_RAW;
    sk.times = a.times + (size_t)b * a.cap;
    int nc = d.num_chan < 1 ? 1 : d.num_chan > MAX_CHAN ? MAX_CHAN : d.num_chan;
    sk.num_chan = nc;
    int prev = 0;
    for (int c = 0; c <= MAX_CHAN; ++c) {             // ascending, inside [0, cap]: a bad descriptor cannot read out of bounds
      int v = d.chan_start[c <= nc ? c : nc];
      v = c == 0 ? 0 : (v < prev ? prev : v > a.cap ? a.cap : v);
      sk.chan_start[c] = v; prev = v;
    }
    sk.N = sk.chan_start[nc]; sk.T = a.T;
    for (int c = 0; c < MAX_CHAN; ++c)
      for (int j = 0; j < 9; ++j) sk.rot_radar[c][j] = c < nc ? d.radar_rot[c * 9 + j] : 0.0;
    for (int j = 0; j < 9; ++j) sk.rot_ref[j] = d.lidar_rot[j];
    for (int j = 0; j < 3; ++j) { sk.lo[j] = d.point_range[j]; sk.hi[j] = d.point_range[3 + j]; }
    sk.tokens = a.tokens + (size_t)b * a.T * RI_OUT;
    sk.count = a.count != nullptr ? a.count + b : nullptr;
  }
  __syncthreads();
  radar_ingest_body(sk);
}

}  // namespace

int launch_radar_ingest_batch(const double* raw, const double* times, const tc_radar_frame_desc* desc, int P, int cap,
                              float* tokens, int T, int* count, hipStream_t s) {
  static_assert(TC_MAX_RADAR_CHANNELS == MAX_CHAN, "descriptor layout");
  TC_REQUIRE(P >= 1 && cap >= 1 && T >= 1, "radar_ingest_batch: P=%d cap=%d T=%d", P, cap, T);
  TC_REQUIRE(raw != nullptr && times != nullptr && desc != nullptr && tokens != nullptr, "radar_ingest_batch: null argument");
  IngestBatchK a{raw, times, desc, cap, T, tokens, count};
  hipLaunchKernelGGL(radar_ingest_batch_kernel, dim3(P), dim3(NT), 0, s, a);
  return check_launch("radar_ingest_batch");
}

int launch_radar_ingest(const double* raw, const double* times, const int* chan_start_host, int num_chan,
                        const double* radar_rot_host, const double* lidar_rot_host,
                        const double* point_range_host, float* tokens, int T, int* count, hipStream_t s) {
  TC_REQUIRE(num_chan >= 1 && num_chan <= MAX_CHAN, "radar_ingest: num_chan=%d (1..%d)", num_chan, MAX_CHAN);
  TC_REQUIRE(tokens != nullptr && T >= 1 && chan_start_host != nullptr && radar_rot_host != nullptr &&
                 lidar_rot_host != nullptr && point_range_host != nullptr, "radar_ingest: null argument");
  IngestK k;
  k.raw = raw; k.times = times; k.num_chan = num_chan; k.T = T; k.tokens = tokens; k.count = count;
  for (int c = 0; c <= MAX_CHAN; ++c) k.chan_start[c] = chan_start_host[c <= num_chan ? c : num_chan];
  TC_REQUIRE(k.chan_start[0] == 0, "radar_ingest: chan_start[0] must be 0");
  for (int c = 0; c < num_chan; ++c)
    TC_REQUIRE(k.chan_start[c + 1] >= k.chan_start[c], "radar_ingest: chan_start must ascend");
  k.N = k.chan_start[num_chan];
  TC_REQUIRE(k.N == 0 || (raw != nullptr && times != nullptr), "radar_ingest: null points");
  for (int c = 0; c < MAX_CHAN; ++c)
    for (int j = 0; j < 9; ++j) k.rot_radar[c][j] = c < num_chan ? radar_rot_host[c * 9 + j] : 0.0;
  for (int j = 0; j < 9; ++j) k.rot_ref[j] = lidar_rot_host[j];
  for (int j = 0; j < 3; ++j) { k.lo[j] = point_range_host[j]; k.hi[j] = point_range_host[3 + j]; }
  hipLaunchKernelGGL(radar_ingest_kernel, dim3(1), dim3(NT), 0, s, k);
  return check_launch("radar_ingest");
}

}  // namespace tc
