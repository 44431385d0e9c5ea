// Training targets and losses on the device.
// The reference builds the Hungarian cost matrix and the focal / L1 losses from
// ~25 small PyTorch ops per decoder output, three times per iteration, with a
// device->host sync each (HEAD:742-917, ASSIGN:106-125, COST:15-26, mmdet FocalLoss /
// FocalLossCost / L1Loss).  Here:
//   normalize_gt_kernel   UTIL:4-24 on the ground-truth boxes
//   match_cost_kernel     cost[l,b,q,g] = w_cls * FocalLossCost + w_reg * |pred - gt|_1
//                         for all decoder outputs in ONE launch (one D2H copy, then
//                         scipy's linear_sum_assignment on the host as in the reference)
//   detr_loss_kernel      given the assignment: the six losses AND their gradients with
//                         respect to the logits / box codes in one pass (closed form,
//                         no autograd graph); the normalisers (mean number of positives
//                         over ranks, clamped to 1) are read from device memory so the
//                         all-reduce that produces them needs no host sync.
#include "kernels.hpp"

namespace tc {

__device__ __forceinline__ float softplusf_(float z) { return fmaxf(z, 0.0f) + log1pf(expf(-fabsf(z))); }

// (cx,cy,cz,w,l,h,rot,vx,vy) -> (cx,cy,log w,log l,cz,log h,sin,cos,vx,vy)
__global__ void normalize_gt_kernel(const float* gt, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = gt + (size_t)i * 9;
  float* o = out + (size_t)i * 10;
  o[0] = b[0]; o[1] = b[1]; o[2] = logf(b[3]); o[3] = logf(b[4]); o[4] = b[2]; o[5] = logf(b[5]);
  o[6] = sinf(b[6]); o[7] = cosf(b[6]); o[8] = b[7]; o[9] = b[8];
}

int launch_normalize_gt(const float* gt, int n, float* out, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(normalize_gt_kernel, dim3((n + 127) / 128), dim3(128), 0, s, gt, n, out);
  return check_launch("normalize_gt");
}

struct CostK {
  const float* cls; const float* box; const float* gtn; const int* gt_labels; const int* gt_counts;
  int Lyr, B, Q, ncls, code, Gmax;
  float wcls, wreg, alpha, gamma, eps;
  float* cost;
};

// one thread per (l, b, q, g)
__global__ void match_cost_kernel(CostK p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)p.Lyr * p.B * p.Q * p.Gmax;
  if (i >= total) return;
  const int g = (int)(i % p.Gmax);
  const size_t row = i / p.Gmax;                    // (l*B + b)*Q + q
  const int b = (int)((row / p.Q) % p.B);
  float c = 0.0f;
  if (g < p.gt_counts[b]) {
    const int label = p.gt_labels[(size_t)b * p.Gmax + g];
    const float x = p.cls[row * p.ncls + label];
    const float pr = 1.0f / (1.0f + expf(-x));
    // mmdet FocalLossCost: pos - neg at the ground-truth class
    const float neg = -logf(1.0f - pr + p.eps) * (1.0f - p.alpha) * powf(pr, p.gamma);
    const float pos = -logf(pr + p.eps) * p.alpha * powf(1.0f - pr, p.gamma);
    float l1 = 0.0f;
    const float* pb = p.box + row * p.code;
    const float* gb = p.gtn + ((size_t)b * p.Gmax + g) * 10;
#pragma unroll
    for (int j = 0; j < 10; ++j) l1 += fabsf(pb[j] - gb[j]);      // COST:15-26 (torch.cdist p = 1)
    c = (pos - neg) * p.wcls + l1 * p.wreg;
  }
  p.cost[i] = c;
}

int launch_match_cost(const CostK& p, hipStream_t s) {
  const size_t total = (size_t)p.Lyr * p.B * p.Q * p.Gmax;
  if (total == 0) return 0;
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  return check_launch("match_cost");
}

struct LossK {
  const float* cls; const float* box; const float* gtn; const int* gt_labels; const int* assigned;
  const float* avg; const float* code_w;
  int Lyr, B, Q, ncls, code, Gmax;
  float alpha, gamma, wcls, wbox;
  float* losses; float* d_cls; float* d_box;
  int avg_min1;                 // 1: `avg` holds raw counts (tc_lsa_assign's num_pos): the normaliser is max(count, 1)
};

// one thread per (l, b, q): focal loss over the classes + L1 over the box code of a
// matched query, and the gradients.  mmdet py_sigmoid_focal_loss:
//   L = BCEwithLogits(x, t) * (alpha t + (1-alpha)(1-t)) * pt^gamma,  pt = (1-p) t + p (1-t)
//   t = 1: L = alpha (1-p)^g softplus(-x),   dL/dx = alpha (1-p)^g (g p log p - (1 - p))
//   t = 0: L = (1-alpha) p^g softplus(x),    dL/dx = (1-alpha) p^g (p - g (1-p) log(1-p))
// 16 threads per row (round 4: one thread per row -- ten classes of expf / powf / log1p each, twelve workgroups for
// 3 x 900 rows -- was 30 us of pure latency on the iteration's critical path): thread (row, sub) takes classes sub,
// sub + 16, ... and box codes sub, sub + 16, ...
__global__ __launch_bounds__(256) void detr_loss_kernel(LossK p) {
  __shared__ float red[2][4];
  const int rows_per_layer = p.B * p.Q;
  const int l = blockIdx.y;
  const int sub = threadIdx.x & 15;
  const int r = blockIdx.x * 16 + (threadIdx.x >> 4);        // row within the layer
  float lc = 0.0f, lb = 0.0f;
  if (r < rows_per_layer) {
    const size_t row = (size_t)l * rows_per_layer + r;
    const int b = r / p.Q;
    const int a = p.assigned[row];
    const float inv = 1.0f / (p.avg_min1 ? fmaxf(p.avg[2 * l], 1.0f) : p.avg[2 * l]);
    const float invb = 1.0f / (p.avg_min1 ? fmaxf(p.avg[2 * l + 1], 1.0f) : p.avg[2 * l + 1]);
    const int label = a >= 0 ? p.gt_labels[(size_t)b * p.Gmax + a] : p.ncls;
    const float* x = p.cls + row * p.ncls;
    float* dx = p.d_cls + row * p.ncls;
    for (int c = sub; c < p.ncls; c += 16) {
      const float v = x[c];
      const float pr = 1.0f / (1.0f + expf(-v));
      float loss, grad;
      if (c == label) {
        const float mod = powf(1.0f - pr, p.gamma);
        const float logp = -softplusf_(-v);
        loss = -p.alpha * mod * logp;
        grad = p.alpha * mod * (p.gamma * pr * logp - (1.0f - pr));
      } else {
        const float mod = powf(pr, p.gamma);
        const float log1mp = -softplusf_(v);
        loss = -(1.0f - p.alpha) * mod * log1mp;
        grad = (1.0f - p.alpha) * mod * (pr - p.gamma * (1.0f - pr) * log1mp);
      }
      lc += loss;
      dx[c] = grad * p.wcls * inv;
    }
    const float* pb = p.box + row * p.code;
    float* db = p.d_box + row * p.code;
    bool ok = a >= 0;
    const float* gb = ok ? p.gtn + ((size_t)b * p.Gmax + a) * 10 : nullptr;
    if (ok) {
      for (int j = 0; j < 10; ++j) ok = ok && isfinite(gb[j]);      // HEAD:905-906 isfinite filter
    }
    for (int j = sub; j < p.code; j += 16) {
      float g = 0.0f;
      if (ok && j < 10) {
        const float d = pb[j] - gb[j];
        const float wj = p.code_w[j];
        lb += wj * fabsf(d);
        g = wj * (d > 0.0f ? 1.0f : d < 0.0f ? -1.0f : 0.0f) * p.wbox * invb;
      }
      db[j] = g;
    }
  }
  lc = wave_sum(lc); lb = wave_sum(lb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = lc; red[1][wave] = lb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float inv = 1.0f / (p.avg_min1 ? fmaxf(p.avg[2 * l], 1.0f) : p.avg[2 * l]);
    const float invb = 1.0f / (p.avg_min1 ? fmaxf(p.avg[2 * l + 1], 1.0f) : p.avg[2 * l + 1]);
    unsafeAtomicAdd(p.losses + 2 * l + 0, (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * p.wcls * inv);
    unsafeAtomicAdd(p.losses + 2 * l + 1, (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * p.wbox * invb);
  }
}

int launch_detr_loss(const LossK& p, hipStream_t s) {
  const int rows = p.B * p.Q;
  if (rows == 0 || p.Lyr == 0) return 0;
  hipLaunchKernelGGL(detr_loss_kernel, dim3((rows + 15) / 16, p.Lyr), dim3(256), 0, s, p);
  return check_launch("detr_loss");
}


// ---- the Hungarian assignment itself, on the device (round 4) -------------------------------------------------
// ASSIGN:117-125 moves the [Q, G] cost matrix to the host and calls scipy's linear_sum_assignment; with the
// decoder hidden behind a look-ahead that round trip (D2H, three solves of 900 x 24, H2D: ~0.30 ms) was the largest
// single piece of a 0.96 ms iteration's critical path.  Here: the same algorithm scipy implements (the shortest
// augmenting path form of Jonker-Volgenant in Crouse's rectangular variant, scipy/optimize/rectangular_lsap: the G
// ground-truth boxes are the rows, the Q queries the columns), in float64 like scipy, one workgroup of LSA_NT = 256
// threads per (decoder output, sample): thread t owns columns t, t + 256, ...; a step of an augmenting path is a
// register update of the thread's columns plus a workgroup-wide arg-min (wave DPP, then the four waves' candidates
// through LDS: two barriers per step).  First build: ONE wave, 16 columns per lane -- 143 us per iteration (a step =
// 16 x ~10 float64 instructions per lane + the transposing load of the costs by 64 threads), the largest single item
// on the iteration's critical path; four waves: see DESIGN.md section 5.
// The optimal assignment is unique unless two candidate sets have exactly equal cost; on an exact tie between
// columns an unassigned column wins, then the lowest (column mod 64), then the lowest column (scipy: the one met
// last in its work list) -- same total cost, possibly another of the tied queries.  The order is a property of the
// COLUMN, not of the thread that holds it: the one-wave build resolved ties the same way.
// A non-finite cost (scipy raises ValueError) leaves the sample unassigned and sets *status.
constexpr int LSA_NT = 256;                        // threads per problem
constexpr int LSA_COLS = 4;                        // columns per thread: Q <= LSA_NT * LSA_COLS = 1024
constexpr int LSA_QMAX = LSA_NT * LSA_COLS;
constexpr int LSA_GMAX = 128;                      // ground-truth boxes per sample
struct LsaK {
  const float* cost;                               // [P, Q, Gmax] (P = outputs x samples), 0 beyond a sample's count
  const int* gt_counts;                            // [B]
  int P, B, Q, Gmax;
  int* assigned;                                   // [P, Q]: gt index or -1
  float* num_pos;                                  // [outputs, 2] += matched boxes, both columns (zero first) or null
  int* status;                                     // += 1 per sample with a non-finite cost, or null
  float* poison;                                   // [outputs, 2] loss accumulators or null: NaN into the output's pair
                                                   // when a sample of it cannot be assigned (scipy would have raised)
};

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, 64));
  return v;
}
// a column's place in the tie order: unassigned first, then column mod 64, then column / 64 (unique per column)
__device__ __forceinline__ unsigned lsa_key(int j, bool is_free) { return (is_free ? 0u : 1u << 20) | ((unsigned)(j & 63) << 10) | (unsigned)(j >> 6); }
__device__ __forceinline__ int lsa_key_col(unsigned key) { return (int)((key >> 10) & 63u) + 64 * (int)(key & 1023u); }

// LDS = true: the sample's costs are transposed into LDS first (a step then reads consecutive words); false (the
// matrix does not fit: more than ~40 boxes at 900 queries): straight from global memory (L2), stride Gmax
template <bool LDS>
__global__ __launch_bounds__(LSA_NT) void lsa_kernel(LsaK p) {
  extern __shared__ __align__(16) float lsa_cost[];          // [G][Qpad] (transposed: a row = one ground-truth box)
  __shared__ double u[LSA_GMAX];
  __shared__ double spc_of_col4row[LSA_GMAX];
  __shared__ int col4row[LSA_GMAX];
  __shared__ int SR[LSA_GMAX];
  __shared__ int path_s[LSA_QMAX];                           // column -> the row it was reached from (this row's search)
  __shared__ int row4col_s[LSA_QMAX];                        // column -> its row, or -1
  __shared__ double red_val[LSA_NT / 64];
  __shared__ unsigned red_key[LSA_NT / 64];
  __shared__ int bad_s;
  const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = prob % p.B;
  const int G = min(p.gt_counts[b], p.Gmax), Q = p.Q;
  const int Qpad = (Q + 63) & ~63;
  int* out = p.assigned + (size_t)prob * Q;
  for (int j = tid; j < Q; j += LSA_NT) out[j] = -1;
  if (G <= 0) return;                                         // (workgroup-uniform)
  if (tid == 0) bad_s = 0;
  for (int j = tid; j < LSA_QMAX; j += LSA_NT) row4col_s[j] = -1;
  __syncthreads();
  // the sample's costs, transposed into LDS (coalesced global reads: g fastest; 8 loads in flight per thread)
  const float* cg = p.cost + (size_t)prob * Q * p.Gmax;
  const int total = Q * p.Gmax;
  int bad = 0;
#pragma unroll 1
  for (int base = 0; base < total; base += LSA_NT * 8) {
    float c[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = base + LSA_NT * t + tid;
      c[t] = i < total ? ldg1(cg + i) : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = base + LSA_NT * t + tid;
      const int q = i / p.Gmax, g = i - q * p.Gmax;
      if (i < total && g < G) {
        bad |= !(fabsf(c[t]) <= 3.0e38f);
        if (LDS) lsa_cost[g * Qpad + q] = c[t];
      }
    }
  }
  if (bad) bad_s = 1;
  for (int i = tid; i < G; i += LSA_NT) { u[i] = 0.0; col4row[i] = -1; }
  __syncthreads();
  if (bad_s) {                                      // scipy: "matrix contains invalid numeric entries"
    if (tid == 0 && p.status != nullptr) atomicAdd(p.status, 1);
    if (tid < 2 && p.poison != nullptr) p.poison[2 * (prob / p.B) + tid] = __int_as_float(0x7fc00000);
    return;
  }
  double v[LSA_COLS];
  int row4col[LSA_COLS];
#pragma unroll
  for (int k = 0; k < LSA_COLS; ++k) { v[k] = 0.0; row4col[k] = -1; }
  const double INF = __longlong_as_double(0x7FF0000000000000ll);
  for (int cur = 0; cur < G; ++cur) {
    double spc[LSA_COLS];
    unsigned sc = 0;                                // bit k: column tid + LSA_NT k has been scanned (left `remaining`)
#pragma unroll
    for (int k = 0; k < LSA_COLS; ++k) spc[k] = INF;
    for (int i = tid; i < G; i += LSA_NT) SR[i] = 0;
    __syncthreads();
    double minVal = 0.0;
    int i = cur, sink = -1;
    while (sink < 0) {
      if (tid == 0) SR[i] = 1;
      const double ui = u[i];
      const float* crow = lsa_cost + i * Qpad;
      double best = INF;
      unsigned bkey = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k) {
        const int j = tid + LSA_NT * k;
        if (j < Q && !((sc >> k) & 1u)) {
          const float cij = LDS ? crow[j] : ldg1(cg + (size_t)j * p.Gmax + i);
          const double r = minVal + (double)cij - ui - v[k];
          if (r < spc[k]) { spc[k] = r; path_s[j] = i; }
          const unsigned key = lsa_key(j, row4col[k] < 0);
          if (spc[k] < best || (spc[k] == best && key < bkey)) { best = spc[k]; bkey = key; }
        }
      }
      // the workgroup's lowest (value, key): wave, then the waves' candidates through LDS
      const double wlow = wave_min_f64(best);
      const unsigned wkey = wave_min_u32(best == wlow ? bkey : 0xFFFFFFFFu);
      if (lane == 0) { red_val[wave] = wlow; red_key[wave] = wkey; }
      __syncthreads();
      double lowest = red_val[0];
      unsigned key = red_key[0];
#pragma unroll
      for (int w = 1; w < LSA_NT / 64; ++w) {
        const double t = red_val[w];
        const unsigned tk = red_key[w];
        if (t < lowest || (t == lowest && tk < key)) { lowest = t; key = tk; }
      }
      if (!(lowest < INF)) {                        // infeasible (cannot happen with finite costs); uniform
        if (tid == 0 && p.status != nullptr) atomicAdd(p.status, 1);
        if (tid < 2 && p.poison != nullptr) p.poison[2 * (prob / p.B) + tid] = __int_as_float(0x7fc00000);
        return;
      }
      minVal = lowest;
      const int j = lsa_key_col(key);
      if ((j & (LSA_NT - 1)) == tid) sc |= 1u << (j / LSA_NT);          // its owner takes it out of `remaining`
      const int r4c = (key >> 20) ? row4col_s[j] : -1;                   // (the key says whether it is assigned)
      if (r4c < 0) sink = j; else i = r4c;
      __syncthreads();                              // red_* are free again; SR / path_s of this step are visible
    }
    // dual update (scipy: u[cur] += minVal; u[i] += minVal - spc[col4row[i]] for the other scanned rows;
    // v[j] -= minVal - spc[j] for the scanned columns)
#pragma unroll
    for (int k = 0; k < LSA_COLS; ++k) {
      if ((sc >> k) & 1u) {
        v[k] -= minVal - spc[k];
        if (row4col[k] >= 0) spc_of_col4row[row4col[k]] = spc[k];      // the row assigned to a scanned column
      }
    }
    __syncthreads();
    for (int r = tid; r < G; r += LSA_NT) {
      if (r == cur) u[r] += minVal;
      else if (SR[r]) u[r] += minVal - spc_of_col4row[r];
    }
    // augment along the path from the sink back to `cur`: one thread walks it (at most G steps, usually one or two)
    if (tid == 0) {
      int j = sink;
      for (;;) {
        const int pi = path_s[j];
        row4col_s[j] = pi;
        const int prev = col4row[pi];
        col4row[pi] = j;
        j = prev;
        if (pi == cur) break;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LSA_COLS; ++k) row4col[k] = row4col_s[tid + LSA_NT * k];
  }
  for (int r = tid; r < G; r += LSA_NT) out[col4row[r]] = r;
  if (tid < 2 && p.num_pos != nullptr) atomicAdd(p.num_pos + 2 * (prob / p.B) + tid, (float)G);
}

int launch_lsa(const float* cost, const int* gt_counts, int P, int B, int Q, int Gmax, int* assigned, float* num_pos,
               int* status, float* poison, hipStream_t s) {
  TC_REQUIRE(P >= 1 && B >= 1 && P % B == 0, "lsa: P=%d B=%d", P, B);
  TC_REQUIRE(Q >= 1 && Q <= LSA_QMAX && Gmax >= 1 && Gmax <= LSA_GMAX && Gmax <= Q,
             "lsa: Q=%d (<= %d) Gmax=%d (<= %d, <= Q)", Q, LSA_QMAX, Gmax, LSA_GMAX);
  const size_t lds = (size_t)Gmax * ((Q + 63) & ~63) * sizeof(float);
  const bool in_lds = lds <= 140 * 1024;        // (beside ~12 KB of static LDS)
  static DeviceOnce once;
  if (const int once_dev = once.need(); once_dev >= 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lsa_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    if (e != hipSuccess) { set_error("lsa: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    once.done(once_dev);
  }
  LsaK p;
  p.cost = cost; p.gt_counts = gt_counts; p.P = P; p.B = B; p.Q = Q; p.Gmax = Gmax; p.assigned = assigned;
  p.num_pos = num_pos; p.status = status; p.poison = poison;
  if (in_lds) hipLaunchKernelGGL(lsa_kernel<true>, dim3(P), dim3(LSA_NT), lds, s, p);
  else hipLaunchKernelGGL(lsa_kernel<false>, dim3(P), dim3(LSA_NT), 0, s, p);
  return check_launch("lsa");
}

}  // namespace tc

using namespace tc;

extern "C" {

int tc_normalize_bbox(const float* gt_boxes, int n, float* out, tc_stream_t stream) {
  return launch_normalize_gt(gt_boxes, n, out, as_stream(stream));
}

int tc_match_cost(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                  int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                  const int* gt_counts, int Gmax, float cls_weight, float reg_weight, float alpha,
                  float gamma, float eps, float* cost, tc_stream_t stream) {
  TC_REQUIRE(code_size >= 10, "match_cost: code_size=%d (>= 10)", code_size);
  CostK p;
  p.cls = all_cls; p.box = all_box; p.gtn = gt_norm; p.gt_labels = gt_labels; p.gt_counts = gt_counts;
  p.Lyr = num_outputs; p.B = B; p.Q = Q; p.ncls = num_classes; p.code = code_size; p.Gmax = Gmax;
  p.wcls = cls_weight; p.wreg = reg_weight; p.alpha = alpha; p.gamma = gamma; p.eps = eps;
  p.cost = cost;
  return launch_match_cost(p, as_stream(stream));
}

int tc_detr_loss_fwd_bwd(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                         int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                         int Gmax, const int* assigned, const float* avg_factors,
                         const float* code_weights, float alpha, float gamma, float cls_loss_weight,
                         float bbox_loss_weight, float* losses, float* d_all_cls, float* d_all_box,
                         tc_stream_t stream) {
  TC_REQUIRE(code_size >= 10, "detr_loss: code_size=%d (>= 10)", code_size);
  LossK p;
  p.cls = all_cls; p.box = all_box; p.gtn = gt_norm; p.gt_labels = gt_labels; p.assigned = assigned;
  p.avg = avg_factors; p.code_w = code_weights;
  p.Lyr = num_outputs; p.B = B; p.Q = Q; p.ncls = num_classes; p.code = code_size; p.Gmax = Gmax;
  p.alpha = alpha; p.gamma = gamma; p.wcls = cls_loss_weight; p.wbox = bbox_loss_weight;
  p.losses = losses; p.d_cls = d_all_cls; p.d_box = d_all_box; p.avg_min1 = 0;
  return launch_detr_loss(p, as_stream(stream));
}

// the same with `avg_factors` holding raw COUNTS of matched boxes (tc_lsa_assign's num_pos): the normalisers are
// max(count, 1) (HEAD:889-902 on one rank), no host in between
int tc_detr_loss_fwd_bwd_counts(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                         int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                         int Gmax, const int* assigned, const float* avg_factors,
                         const float* code_weights, float alpha, float gamma, float cls_loss_weight,
                         float bbox_loss_weight, float* losses, float* d_all_cls, float* d_all_box,
                         tc_stream_t stream) {
  TC_REQUIRE(code_size >= 10, "detr_loss: code_size=%d (>= 10)", code_size);
  LossK p;
  p.cls = all_cls; p.box = all_box; p.gtn = gt_norm; p.gt_labels = gt_labels; p.assigned = assigned;
  p.avg = avg_factors; p.code_w = code_weights;
  p.Lyr = num_outputs; p.B = B; p.Q = Q; p.ncls = num_classes; p.code = code_size; p.Gmax = Gmax;
  p.alpha = alpha; p.gamma = gamma; p.wcls = cls_loss_weight; p.wbox = bbox_loss_weight;
  p.losses = losses; p.d_cls = d_all_cls; p.d_box = d_all_box; p.avg_min1 = 1;
  return launch_detr_loss(p, as_stream(stream));
}

int tc_lsa_assign(const float* cost, const int* gt_counts, int num_outputs, int B, int Q, int Gmax, int* assigned,
                  float* num_pos, int* status, tc_stream_t stream) {
  TC_REQUIRE(cost != nullptr && gt_counts != nullptr && assigned != nullptr, "lsa_assign: null argument");
  return launch_lsa(cost, gt_counts, num_outputs * B, B, Q, Gmax, assigned, num_pos, status, nullptr, as_stream(stream));
}

int tc_lsa_assign_ex(const float* cost, const int* gt_counts, int num_outputs, int B, int Q, int Gmax, int* assigned,
                     float* num_pos, int* status, float* poison_losses, tc_stream_t stream) {
  TC_REQUIRE(cost != nullptr && gt_counts != nullptr && assigned != nullptr, "lsa_assign: null argument");
  return launch_lsa(cost, gt_counts, num_outputs * B, B, Q, Gmax, assigned, num_pos, status, poison_losses, as_stream(stream));
}

}  // extern "C"
