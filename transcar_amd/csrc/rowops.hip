// Row-local (per query / per radar token) pieces of the fusion decoder that are
// not GEMMs: LayerNorm chains, the first layer of the position encoders
// (in_features = 3), reference-point bookkeeping.  One wavefront owns one row
// of C = 256 channels: each lane holds a float4, a row is one coalesced 1 KiB
// access, and the LayerNorm statistics are two wave-wide shuffle reductions.
// All of it is HBM/L2-bound elementwise work.
#include "kernels.hpp"
#include "rowdev.hpp"

namespace tc {

struct Pc6 { float v[6]; };

struct LnK {
  const float *a, *b, *c, *g2, *b2, *gamma, *beta, *d;
  float* y;
  int M, relu, d_relu;
};

__global__ __launch_bounds__(256) void ln256_kernel(LnK p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const size_t off = (size_t)row * 256 + 4 * lane;
  float4 v = ld4(p.a + off);
  if (p.b) v = add4(v, ld4(p.b + off));
  if (p.c) v = add4(v, relu4(ln_row(ld4(p.c + off), p.g2, p.b2, lane)));
  if (p.gamma) v = ln_row(v, p.gamma, p.beta, lane);
  if (p.relu) v = relu4(v);
  if (p.d) {
    float4 dd = ld4(p.d + off);
    if (p.d_relu) dd = relu4(dd);
    v = add4(v, dd);
  }
  st4(p.y + off, v);
}

int launch_ln256(const LnArgs& a, hipStream_t s) {
  TC_REQUIRE(a.M > 0, "ln256: M=%d", a.M);
  LnK p{a.a, a.b, a.c, a.g2, a.b2, a.gamma, a.beta, a.d, a.y, a.M, a.relu, a.d_relu};
  hipLaunchKernelGGL(ln256_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, p);
  return check_launch("ln256");
}

// ---- position-encoder layer 0: Linear(3,256) + LN + ReLU --------------------
__global__ __launch_bounds__(256) void posenc_l1_kernel(const float* src, int ld, int inv_sig,
                                                        const float* w0, const float* b0,
                                                        const float* g, const float* beta,
                                                        float* y, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float p0 = src[(size_t)row * ld + 0], p1 = src[(size_t)row * ld + 1], p2 = src[(size_t)row * ld + 2];
  if (inv_sig) { p0 = inverse_sigmoidf_(p0); p1 = inverse_sigmoidf_(p1); p2 = inverse_sigmoidf_(p2); }
  const float4 o = posenc_l0_row(p0, p1, p2, w0, b0, g, beta, lane);
  st4(y + (size_t)row * 256 + 4 * lane, o);
}

int launch_posenc_l1(const float* src, int ld, int inv_sigmoid, const float* w0, const float* b0,
                     const float* g, const float* beta, float* y, int M, hipStream_t s) {
  hipLaunchKernelGGL(posenc_l1_kernel, dim3((M + 3) / 4), dim3(256), 0, s, src, ld, inv_sigmoid, w0,
                     b0, g, beta, y, M);
  return check_launch("posenc_l1");
}

// ---- XFMR:119-123: split the embedding, initial reference points ------------
__global__ __launch_bounds__(256) void split_embed_kernel(const float* qe, int Q, int B, float* pos,
                                                          float* x) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Q) return;
  const int q = row % Q;
  st4(pos + (size_t)row * 256 + 4 * lane, ld4(qe + (size_t)q * 512 + 4 * lane));
  st4(x + (size_t)row * 256 + 4 * lane, ld4(qe + (size_t)q * 512 + 256 + 4 * lane));
}
int launch_split_embed(const float* qe, int Q, int C, int B, float* pos, float* x, hipStream_t s) {
  TC_REQUIRE(C == 256, "split_embed: C=%d (256 supported)", C);
  hipLaunchKernelGGL(split_embed_kernel, dim3((B * Q + 3) / 4), dim3(256), 0, s, qe, Q, B, pos, x);
  return check_launch("split_embed");
}

__global__ __launch_bounds__(256) void init_ref_kernel(const float* qe, int Q, const float* w,
                                                       const float* b, float* ref, int B) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * Q) return;
  const int q = row % Q;
  const float4 p = ld4(qe + (size_t)q * 512 + 4 * lane);   // query_pos half
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 ww = ld4(w + j * 256 + 4 * lane);
    float t = wave_sum(p.x * ww.x + p.y * ww.y + p.z * ww.z + p.w * ww.w);
    if (lane == 0) ref[(size_t)row * 3 + j] = sigmoidf_(t + b[j]);
  }
}
int launch_init_ref(const float* qe, int Q, int C, const float* w, const float* b, float* ref, int B,
                    hipStream_t s) {
  TC_REQUIRE(C == 256, "init_ref: C=%d (256 supported)", C);
  hipLaunchKernelGGL(init_ref_kernel, dim3((B * Q + 3) / 4), dim3(256), 0, s, qe, Q, w, b, ref, B);
  return check_launch("init_ref");
}

// ---- XFMR:195-203 + HEAD:287-293 --------------------------------------------
__global__ void ref_update_kernel(const float* tmp, int code, const float* ref, float* new_ref,
                                  float* box_m, Pc6 pcs, int M) {
  const float* pc = pcs.v;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  const float* t = tmp + (size_t)row * code;
  const float* r = ref + (size_t)row * 3;
  const float nx = sigmoidf_(t[0] + inverse_sigmoidf_(r[0]));
  const float ny = sigmoidf_(t[1] + inverse_sigmoidf_(r[1]));
  const float nz = sigmoidf_(t[4] + inverse_sigmoidf_(r[2]));
  if (new_ref) {
    new_ref[(size_t)row * 3 + 0] = nx;
    new_ref[(size_t)row * 3 + 1] = ny;
    new_ref[(size_t)row * 3 + 2] = nz;
  }
  if (box_m) {
    float* o = box_m + (size_t)row * code;
    for (int j = 0; j < code; ++j) o[j] = t[j];
    o[0] = nx * (pc[3] - pc[0]) + pc[0];
    o[1] = ny * (pc[4] - pc[1]) + pc[1];
    o[4] = nz * (pc[5] - pc[2]) + pc[2];
  }
}
int launch_ref_update(const float* tmp, int code, const float* ref, float* new_ref, float* box_m,
                      const float* pc6_host, int M, hipStream_t s) {
  Pc6 pc_dev; for (int i = 0; i < 6; ++i) pc_dev.v[i] = pc6_host[i];
  hipLaunchKernelGGL(ref_update_kernel, dim3((M + 255) / 256), dim3(256), 0, s, tmp, code, ref,
                     new_ref, box_m, pc_dev, M);
  return check_launch("ref_update");
}

__global__ void box_add_ref_kernel(const float* reg, int code, const float* rxy, int ld_xy,
                                   const float* rz, int ld_z, float* box, float* next_ref3, int M) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  const float* t = reg + (size_t)row * code;
  float* o = box + (size_t)row * code;
  for (int j = 0; j < code; ++j) o[j] = t[j];
  const float x = t[0] + rxy[(size_t)row * ld_xy + 0];
  const float y = t[1] + rxy[(size_t)row * ld_xy + 1];
  const float z = t[4] + rz[(size_t)row * ld_z];
  o[0] = x; o[1] = y; o[4] = z;
  if (next_ref3) {
    next_ref3[(size_t)row * 3 + 0] = x;
    next_ref3[(size_t)row * 3 + 1] = y;
    next_ref3[(size_t)row * 3 + 2] = z;
  }
}
int launch_box_add_ref(const float* reg_out, int code, const float* ref_xy, int ld_xy,
                       const float* ref_z, int ld_z, float* box, float* next_ref3, int M,
                       hipStream_t s) {
  hipLaunchKernelGGL(box_add_ref_kernel, dim3((M + 255) / 256), dim3(256), 0, s, reg_out, code,
                     ref_xy, ld_xy, ref_z, ld_z, box, next_ref3, M);
  return check_launch("box_add_ref");
}

__global__ void radar_ref_l1_kernel(const float* ref, Pc6 pcs, float* cxy, float* addref, int M) {
  const float* pc = pcs.v;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  // separate roundings (torch: mul, then add) -- these feed the distance gate
  const float x = __fadd_rn(__fmul_rn(ref[(size_t)row * 3 + 0], pc[3] - pc[0]), pc[0]);
  const float y = __fadd_rn(__fmul_rn(ref[(size_t)row * 3 + 1], pc[4] - pc[1]), pc[1]);
  cxy[(size_t)row * 2 + 0] = x;
  cxy[(size_t)row * 2 + 1] = y;
  addref[(size_t)row * 3 + 0] = x;
  addref[(size_t)row * 3 + 1] = y;
  addref[(size_t)row * 3 + 2] = ref[(size_t)row * 3 + 2];   // z stays normalised: HEAD:598
}
int launch_radar_ref_l1(const float* ref, const float* pc6_host, float* centre_xy, float* addref3,
                        int M, hipStream_t s) {
  Pc6 pc_dev; for (int i = 0; i < 6; ++i) pc_dev.v[i] = pc6_host[i];
  hipLaunchKernelGGL(radar_ref_l1_kernel, dim3((M + 255) / 256), dim3(256), 0, s, ref, pc_dev,
                     centre_xy, addref3, M);
  return check_launch("radar_ref_l1");
}

}  // namespace tc
