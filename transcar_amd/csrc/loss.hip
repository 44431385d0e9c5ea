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
};

// one thread per (l, b, q): focal loss over the classes + L1 over the box code of a
// matched query, and the gradients.  mmdet py_sigmoid_focal_loss:
//   L = BCEwithLogits(x, t) * (alpha t + (1-alpha)(1-t)) * pt^gamma,  pt = (1-p) t + p (1-t)
//   t = 1: L = alpha (1-p)^g softplus(-x),   dL/dx = alpha (1-p)^g (g p log p - (1 - p))
//   t = 0: L = (1-alpha) p^g softplus(x),    dL/dx = (1-alpha) p^g (p - g (1-p) log(1-p))
__global__ __launch_bounds__(256) void detr_loss_kernel(LossK p) {
  __shared__ float red[2][4];
  const int rows_per_layer = p.B * p.Q;
  const int l = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;       // row within the layer
  float lc = 0.0f, lb = 0.0f;
  if (r < rows_per_layer) {
    const size_t row = (size_t)l * rows_per_layer + r;
    const int b = r / p.Q;
    const int a = p.assigned[row];
    const float inv = 1.0f / p.avg[2 * l], invb = 1.0f / p.avg[2 * l + 1];
    const int label = a >= 0 ? p.gt_labels[(size_t)b * p.Gmax + a] : p.ncls;
    const float* x = p.cls + row * p.ncls;
    float* dx = p.d_cls + row * p.ncls;
    for (int c = 0; c < p.ncls; ++c) {
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
    for (int j = 0; j < p.code; ++j) {
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
    const float inv = 1.0f / p.avg[2 * l], invb = 1.0f / p.avg[2 * l + 1];
    unsafeAtomicAdd(p.losses + 2 * l + 0, (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * p.wcls * inv);
    unsafeAtomicAdd(p.losses + 2 * l + 1, (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * p.wbox * invb);
  }
}

int launch_detr_loss(const LossK& p, hipStream_t s) {
  const int rows = p.B * p.Q;
  if (rows == 0 || p.Lyr == 0) return 0;
  hipLaunchKernelGGL(detr_loss_kernel, dim3((rows + 255) / 256, p.Lyr), dim3(256), 0, s, p);
  return check_launch("detr_loss");
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
  p.losses = losses; p.d_cls = d_all_cls; p.d_box = d_all_box;
  return launch_detr_loss(p, as_stream(stream));
}

}  // extern "C"
