// One training iteration of the trainable (radar) part of the head as two C calls:
// tc_radar_train_fwd keeps every activation the backward needs on a caller-provided
// tape, tc_radar_train_bwd walks it in reverse and ADDS the parameter gradients into a
// tc_head_weights-shaped table of gradient pointers (the flat gradient bucket of
// transcar_amd/trainer.py).  Same kernels as the operator-level entry points
// (gemm.hip, rowops.hip, radar_attn.hip, train.hip); what goes away is the host side:
// ~60 forward and ~100 backward autograd nodes, each a Python -> ctypes round trip.
// Reference: HEAD:531-729 (forward), tools/train.py:245-252 (what is trainable).
#include <math.h>
#include <string.h>

#include "kernels.hpp"

namespace tc {

#define TS_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)
#define TS_HIP(expr)                                          \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) {                                   \
      tc::set_error("%s: %s", #expr, hipGetErrorString(_e)); \
      return (int)_e;                                         \
    }                                                         \
  } while (0)

namespace {

struct LayerTape {
  float *qp, *kv, *ao, *x1, *x2, *h, *ff, *x3, *c0, *c1, *c2, *c3, *t0, *t1, *treg;
  int* hits;
};
struct Tape {
  float *xyz4, *w0p, *u0, *u1, *u2, *pos, *f0, *f1, *f2, *mem, *cxy, *addref;
  LayerTape L[TC_MAX_RADAR_LAYERS];
  // backward scratch
  float *dbox, *dA, *dB, *dC, *dD, *dh, *dqp, *dkv, *dmem, *dqin, *dt64, *dt128, *dw0p;
};

size_t tape_layout(const tc_head_weights* w, int B, int T, void* base, size_t cap, Tape* out) {
  const size_t rows = (size_t)B * w->num_query, rt = (size_t)B * T;
  const size_t C = w->embed_dims, F = w->ffn_dims, code = w->code_size;
  Arena a(base, cap);
  Tape t;
  t.xyz4 = a.take<float>(rt * 4); t.w0p = a.take<float>(C * 4);
  t.u0 = a.take<float>(rt * C); t.u1 = a.take<float>(rt * C); t.u2 = a.take<float>(rt * C);
  t.pos = a.take<float>(rt * C);
  t.f0 = a.take<float>(rt * 64); t.f1 = a.take<float>(rt * 128); t.f2 = a.take<float>(rt * C);
  t.mem = a.take<float>(rt * C);
  t.cxy = a.take<float>(rows * 2); t.addref = a.take<float>(rows * 3);
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    LayerTape& l = t.L[r];
    l.qp = a.take<float>(rows * C); l.kv = a.take<float>(rt * 2 * C); l.ao = a.take<float>(rows * C);
    l.x1 = a.take<float>(rows * C); l.x2 = a.take<float>(rows * C); l.h = a.take<float>(rows * F);
    l.ff = a.take<float>(rows * C); l.x3 = a.take<float>(rows * C);
    l.c0 = a.take<float>(rows * C); l.c1 = a.take<float>(rows * C); l.c2 = a.take<float>(rows * C);
    l.c3 = a.take<float>(rows * C);
    l.t0 = a.take<float>(rows * C); l.t1 = a.take<float>(rows * C); l.treg = a.take<float>(rows * code);
    l.hits = a.take<int>(rows);
  }
  t.dbox = a.take<float>(rows * code);
  t.dA = a.take<float>(rows * C); t.dB = a.take<float>(rows * C); t.dC = a.take<float>(rows * C);
  t.dD = a.take<float>(rows * C);
  t.dh = a.take<float>(rows * F); t.dqp = a.take<float>(rows * C);
  t.dkv = a.take<float>(rt * 2 * C); t.dmem = a.take<float>(rt * C); t.dqin = a.take<float>(rows * C);
  t.dt64 = a.take<float>(rt * 64); t.dt128 = a.take<float>(rt * 128); t.dw0p = a.take<float>(C * 4);
  if (out) *out = t;
  return a.off;
}

// dst[r*ld_dst + c] (=|+=) src[r*ld_src + c] for c < cols: pads / unpads the 3-column
// operands of radar_position_encoder.0 to the 4 columns the MFMA tiles want
__global__ void copy_cols_kernel(const float* src, int ld_src, float* dst, int ld_dst, int rows,
                                 int cols, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i - r * cols;
  const float v = src[(size_t)r * ld_src + c];
  float* d = dst + (size_t)r * ld_dst + c;
  if (accumulate) *d += v; else *d = v;
}
int copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols,
              int accumulate, hipStream_t s) {
  const int n = rows * cols;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, ld_src, dst, ld_dst,
                     rows, cols, accumulate);
  return check_launch("copy_cols");
}

// dst[i] (+=) src[i]
__global__ void add_kernel(const float* src, float* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
int add_into(const float* src, float* dst, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
  return check_launch("add_into");
}

int lin(const float* x, const tc_linear& w, int M, int K, int N, int act, float* y, hipStream_t s,
        const float* res = nullptr, const int* gate = nullptr) {
  GemmArgs g;
  g.X = x; g.ldx = K; g.W = w.w; g.ldw = K; g.bias = w.b; g.R = res; g.ldr = N; g.rowgate = gate;
  g.Y = y; g.ldy = N; g.M = M; g.K = K; g.N = N; g.act = act;
  return launch_gemm(g, s);
}
int lnorm(const float* a, const float* b, const tc_lnorm& n, float* y, int M, int relu, hipStream_t s) {
  LnArgs l;
  l.a = a; l.b = b; l.gamma = n.g; l.beta = n.b; l.y = y; l.M = M; l.relu = relu;
  return launch_ln256(l, s);
}
// backward of y = act(x W^T + b): dW += dY~^T x, db += colsum dY~, dx (=|+=) dY~ W
//   y_relu: this layer's saved ReLU output (mask on dY) or nullptr; x_relu: mask on dx
//   dx_scale: 1 / (1 - p) when x is a dropped-out ReLU output (zeros of x_relu = ReLU zeros + dropped)
int lin_bwd(const float* x, const float* dy, const float* y_relu, const int* gate, const tc_linear& w,
            const tc_linear& g, const float* x_relu, float* dx, int accumulate, int M, int K, int N,
            hipStream_t s, float dx_scale = 1.0f) {
  TS_TRY(launch_linear_bwd_weight(x, dy, y_relu, gate, const_cast<float*>(g.w), const_cast<float*>(g.b),
                                  M, K, N, 1.0f, s));
  if (dx != nullptr)
    TS_TRY(launch_linear_bwd_data(dy, y_relu, gate, w.w, x_relu, dx, M, K, N, dx_scale, accumulate, s));
  return 0;
}
int ln_bwd(const float* a, const float* b, const tc_lnorm& n, const tc_lnorm& g, const float* dy,
           const float* relu_out, float* dz, int M, hipStream_t s) {
  return launch_ln256_bwd(a, b, n.g, dy, relu_out, dz, const_cast<float*>(g.g), const_cast<float*>(g.b),
                          M, s);
}

RadarAttnArgs core_args(const tc_head_weights* w, int r, const Tape& t, const float* box_prev,
                        const float* tokens, int B, int T, int pad_mult, float drop_p = 0.0f,
                        unsigned long long seed = 0) {
  const int C = w->embed_dims, code = w->code_size;
  RadarAttnArgs ra;
  ra.qproj = t.L[r].qp; ra.ldq = C; ra.kv = t.L[r].kv; ra.ldkv = 2 * C;
  ra.centre_xy = r == 0 ? t.cxy : box_prev; ra.ld_c = r == 0 ? 2 : code;
  ra.box = box_prev; ra.code = code; ra.radar_xy = tokens; ra.ld_xy = w->radar_in_dims;
  ra.B = B; ra.Q = w->num_query; ra.T = T; ra.C = C; ra.H = w->num_heads; ra.pad_mult = pad_mult;
  ra.rmin = w->radar[r].radius_min; ra.rmax = w->radar[r].radius_max;
  ra.attn_out = t.L[r].ao; ra.hit_counts = t.L[r].hits;
  ra.qscale = 1.0f / sqrtf((float)(C / w->num_heads));
  ra.drop = make_drop(drop_p, seed, 4u * r + 0u, (unsigned)w->num_radar_tokens_ref);
  return ra;
}

int check(const tc_head_weights* w, int B, int T) {
  TC_REQUIRE(w != nullptr && w->abi_version == TC_ABI_VERSION, "radar_train: bad weights struct");
  TC_REQUIRE(w->embed_dims == 256 && w->num_heads == 8 && w->num_radar_layers == TC_MAX_RADAR_LAYERS,
             "radar_train: embed_dims=%d heads=%d radar layers=%d", w->embed_dims, w->num_heads,
             w->num_radar_layers);
  TC_REQUIRE(B >= 1 && T >= 1 && (w->radar_in_dims & 3) == 0, "radar_train: B=%d T=%d", B, T);
  return 0;
}

}  // namespace
}  // namespace tc

using namespace tc;

extern "C" {

size_t tc_radar_train_tape_bytes(const tc_head_weights* w, int B, int T) {
  if (check(w, B, T) != 0) return 0;
  return tape_layout(w, B, T, nullptr, ~size_t(0), nullptr);
}

int tc_radar_train_fwd(const tc_head_weights* w, const float* hs_last, const float* ref_last,
                       const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                       float* all_cls_scores, float* all_bbox_preds, void* tape, size_t tape_bytes,
                       float dropout_p, unsigned long long dropout_seed, tc_stream_t stream) {
  TS_TRY(check(w, B, T));
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_train_fwd: dropout_p=%g", (double)dropout_p);
  const bool drop = dropout_p > 0.0f;
  const unsigned tref = (unsigned)w->num_radar_tokens_ref;
  Tape t;
  TC_REQUIRE(tape_layout(w, B, T, tape, tape_bytes, &t) <= tape_bytes, "radar_train_fwd: tape too small");
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, code = w->code_size;
  const int ncls = w->num_classes, RI = w->radar_in_dims;
  const int rows = B * Q, rt = B * T;
  // radar encoders, HEAD:531-536 (Linear(3,C) as K = 4 with a zero column)
  const tc_pos_encoder& pe = w->radar_position_encoder;
  TS_HIP(hipMemsetAsync(t.xyz4, 0, (size_t)rt * 4 * 4, s));
  TS_HIP(hipMemsetAsync(t.w0p, 0, (size_t)C * 4 * 4, s));
  TS_TRY(copy_cols(radar_tokens, RI, t.xyz4, 4, rt, 3, 0, s));
  TS_TRY(copy_cols(pe.l0.w, 3, t.w0p, 4, C, 3, 0, s));
  TS_TRY(lin(t.xyz4, tc_linear{t.w0p, pe.l0.b}, rt, 4, C, 0, t.u0, s));
  TS_TRY(lnorm(t.u0, nullptr, pe.n1, t.u1, rt, 1, s));
  TS_TRY(lin(t.u1, pe.l3, rt, C, C, 0, t.u2, s));
  TS_TRY(lnorm(t.u2, nullptr, pe.n4, t.pos, rt, 1, s));
  TS_TRY(lin(radar_tokens, w->radar_feat0, rt, RI, 64, 1, t.f0, s));
  TS_TRY(lin(t.f0, w->radar_feat2, rt, 64, 128, 1, t.f1, s));
  TS_TRY(lin(t.f1, w->radar_feat4, rt, 128, C, 1, t.f2, s));
  {
    LnArgs l;      // mem = pos + f2 (plain sum)
    l.a = t.pos; l.b = t.f2; l.y = t.mem; l.M = rt;
    TS_TRY(launch_ln256(l, s));
  }
  TS_TRY(launch_radar_ref_l1(ref_last, w->pc_range, t.cxy, t.addref, rows, s));
  const float* qin = hs_last;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    const tc_radar_layer& rl = w->radar[r];
    LayerTape& l = t.L[r];
    float* cls_out = all_cls_scores + (size_t)r * rows * ncls;
    float* box_out = all_bbox_preds + (size_t)r * rows * code;
    const float* box_prev = r == 0 ? last_box : all_bbox_preds + (size_t)(r - 1) * rows * code;
    const tc_linear wq{rl.attn.in_proj.w, rl.attn.in_proj.b};
    const tc_linear wkv{rl.attn.in_proj.w + (size_t)C * C, rl.attn.in_proj.b + C};
    TS_TRY(lin(qin, wq, rows, C, C, 0, l.qp, s));
    TS_TRY(lin(t.mem, wkv, rt, C, 2 * C, 0, l.kv, s));
    TS_TRY(launch_radar_attn(core_args(w, r, t, box_prev, radar_tokens, B, T, pad_mult, dropout_p, dropout_seed), s));
    if (drop) {     // x1 = qin + gate * rf_dropout2(out_proj(ao)), HEAD:581 (l.ff is free until linear2)
      TS_TRY(lin(l.ao, rl.attn.out_proj, rows, C, C, 0, l.ff, s));
      TS_TRY(launch_dropout(l.ff, qin, l.hits, rows, C, make_drop(dropout_p, dropout_seed, 4u * r + 1u, tref), l.x1, s));
    } else {
      TS_TRY(lin(l.ao, rl.attn.out_proj, rows, C, C, 0, l.x1, s, qin, l.hits));
    }
    TS_TRY(lnorm(l.x1, nullptr, rl.norm2, l.x2, rows, 0, s));
    TS_TRY(lin(l.x2, rl.linear1, rows, C, F, 1, l.h, s));
    if (drop)       // rf_dropout(relu(linear1)), HEAD:584: the tape keeps the dropped activations
      TS_TRY(launch_dropout(l.h, nullptr, nullptr, rows, F, make_drop(dropout_p, dropout_seed, 4u * r + 2u, tref), l.h, s));
    TS_TRY(lin(l.h, rl.linear2, rows, F, C, 0, l.ff, s));
    if (drop)       // rf_dropout3(ffn_out), HEAD:585
      TS_TRY(launch_dropout(l.ff, nullptr, nullptr, rows, C, make_drop(dropout_p, dropout_seed, 4u * r + 3u, tref), l.ff, s));
    TS_TRY(lnorm(l.x2, l.ff, rl.norm3, l.x3, rows, 0, s));
    TS_TRY(lin(l.x3, rl.final_cls.l0, rows, C, C, 0, l.c0, s));
    TS_TRY(lnorm(l.c0, nullptr, rl.final_cls.n1, l.c1, rows, 1, s));
    TS_TRY(lin(l.c1, rl.final_cls.l3, rows, C, C, 0, l.c2, s));
    TS_TRY(lnorm(l.c2, nullptr, rl.final_cls.n4, l.c3, rows, 1, s));
    TS_TRY(lin(l.c3, rl.final_cls.l6, rows, C, ncls, 0, cls_out, s));
    TS_TRY(lin(l.x3, rl.final_reg.l0, rows, C, C, 1, l.t0, s));
    TS_TRY(lin(l.t0, rl.final_reg.l2, rows, C, C, 1, l.t1, s));
    TS_TRY(lin(l.t1, rl.final_reg.l4, rows, C, code, 0, l.treg, s));
    if (r == 0)
      TS_TRY(launch_box_add_ref(l.treg, code, t.addref, 3, t.addref + 2, 3, box_out, nullptr, rows, s));
    else
      TS_TRY(launch_box_add_ref(l.treg, code, box_prev, code, box_prev + 4, code, box_out, nullptr, rows, s));
    qin = l.x3;
  }
  return 0;
}

// The same forward as THREE launches of the fused row chains (chain.hip PROG_RADAR_ENC_TRAIN / PROG_RADAR_TRAIN)
// plus the reference set-up: the encoder program writes the token-side tape (u0 u1 u2 pos f0 f1 f2 mem and the
// three layers' K | V), the radar program walks the three fusion layers with the dropout sites in its epilogues
// and stores every query-side tape tensor as it is produced.  The tape is the one tc_radar_train_fwd writes:
// tc_radar_train_bwd takes either.  `packed_view`: the head's packed weights, re-packed for the CURRENT
// parameters (tc_head_repack_trainable_ex(w, view, 1, ...) after an optimizer step).
int tc_radar_train_fwd_fused(const tc_head_weights* packed_view, const float* hs_last, const float* ref_last,
                             const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                             float* all_cls_scores, float* all_bbox_preds, void* tape, size_t tape_bytes,
                             float dropout_p, unsigned long long dropout_seed, tc_stream_t stream) {
  const tc_head_weights* w = packed_view;
  TS_TRY(check(w, B, T));
  TC_REQUIRE(w->l0_attn_out != nullptr, "radar_train_fwd_fused: packed_view was not produced by tc_head_pack_weights");
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_train_fwd_fused: dropout_p=%g", (double)dropout_p);
  // (round 3 refused T > 512 here although the chains take any T <= 1500 -- beyond 512 tokens the gate's hit masks
  // are not kept in LDS and the attention step re-evaluates the gate, as the inference chain does; a frame with more
  // than 511 radar points then failed mid-training: ADVICE r3.  Covered at T = 576 and T = 1500 by
  // tests/test_gpu_training.py::test_fused_training_with_many_radar_tokens.)
  TC_REQUIRE(w->code_size <= 10, "radar_train_fwd_fused: code_size=%d", w->code_size);
  Tape t;
  TC_REQUIRE(tape_layout(w, B, T, tape, tape_bytes, &t) <= tape_bytes, "radar_train_fwd_fused: tape too small");
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, code = w->code_size, RI = w->radar_in_dims;
  const int rows = B * Q, rt = B * T;
  float* slots[TS_COUNT];
  const LayerTape& l0 = t.L[0];
  slots[TS_QP] = l0.qp; slots[TS_AO] = l0.ao; slots[TS_X1] = l0.x1; slots[TS_X2] = l0.x2; slots[TS_H] = l0.h;
  slots[TS_SUM] = l0.ff; slots[TS_X3] = l0.x3; slots[TS_C0] = l0.c0; slots[TS_C1] = l0.c1; slots[TS_C2] = l0.c2;
  slots[TS_C3] = l0.c3; slots[TS_T0] = l0.t0; slots[TS_T1] = l0.t1; slots[TS_TREG] = l0.treg;
  slots[TS_U0] = t.u0; slots[TS_U1] = t.u1; slots[TS_U2] = t.u2; slots[TS_POS] = t.pos;
  slots[TS_F0] = t.f0; slots[TS_F1] = t.f1; slots[TS_F2] = t.f2; slots[TS_MEM] = t.mem;
  // encoders + K | V of the three layers (HEAD:531-536, 573-575): one launch
  RadarEncodeArgs re;
  re.tokens = radar_tokens; re.RI = RI; re.M = rt;
  re.rpe = w->radar_position_encoder; re.f0 = w->radar_feat0; re.f2 = w->radar_feat2; re.f4 = w->radar_feat4;
  re.nlayers = w->num_radar_layers;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    const tc_mha& m = w->radar[r].attn;
    re.kvproj[r] = tc_linear{m.in_proj.w ? m.in_proj.w + (size_t)C * C : nullptr, m.in_proj.b ? m.in_proj.b + C : nullptr};
    re.kv[r] = t.L[r].kv;
  }
  re.radar_feat = nullptr; re.w16_delta = w->packed16_delta; re.tape = slots;
  TS_TRY(launch_radar_encode(re, s));
  TS_TRY(launch_radar_ref_l1(ref_last, w->pc_range, t.cxy, t.addref, rows, s));
  // the three fusion layers: one launch
  RadarChainArgs rc;
  rc.qf = hs_last; rc.ref_last = ref_last; rc.box_m = last_box; rc.tokens = radar_tokens; rc.RI = RI;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) { rc.kv[r] = t.L[r].kv; rc.w[r] = w->radar[r]; }
  rc.nlayers = w->num_radar_layers; rc.Q = Q; rc.T = T; rc.pad_mult = pad_mult; rc.code = code;
  rc.ncls = w->num_classes; rc.M = rows; rc.qscale = 1.0f / sqrtf((float)(C / w->num_heads));
  for (int i = 0; i < 6; ++i) rc.pc[i] = w->pc_range[i];
  rc.all_cls = all_cls_scores; rc.all_box = all_bbox_preds; rc.hits = t.L[0].hits;
  rc.tape = slots;
  rc.tape_stride = (size_t)(t.L[1].qp - t.L[0].qp);
  rc.hits_stride = (size_t)(t.L[1].hits - t.L[0].hits);
  rc.drop = make_drop(dropout_p, dropout_seed, 0u, (unsigned)w->num_radar_tokens_ref);
  return launch_radar_chain(rc, s);
}

int tc_radar_train_bwd(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                       const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                       const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                       void* tape, size_t tape_bytes, float dropout_p, unsigned long long dropout_seed,
                       tc_stream_t stream) {
  TS_TRY(check(w, B, T));
  TC_REQUIRE(grads != nullptr, "radar_train_bwd: grads is NULL");
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_train_bwd: dropout_p=%g", (double)dropout_p);
  const bool drop = dropout_p > 0.0f;
  const unsigned tref = (unsigned)w->num_radar_tokens_ref;
  const float keep_scale = drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  Tape t;
  TC_REQUIRE(tape_layout(w, B, T, tape, tape_bytes, &t) <= tape_bytes, "radar_train_bwd: tape too small");
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, code = w->code_size;
  const int ncls = w->num_classes, RI = w->radar_in_dims;
  const int rows = B * Q, rt = B * T;
  TS_HIP(hipMemsetAsync(t.dmem, 0, (size_t)rt * C * 4, s));
  bool have_dqin = false;                       // gradient w.r.t. this layer's output from the layer above
  bool have_dbox_next = false;
  for (int r = TC_MAX_RADAR_LAYERS - 1; r >= 0; --r) {
    const tc_radar_layer& rl = w->radar[r];
    const tc_radar_layer& gl = grads->radar[r];
    LayerTape& l = t.L[r];
    const float* qin = r == 0 ? hs_last : t.L[r - 1].x3;
    const float* box_prev = r == 0 ? last_box : all_bbox_preds + (size_t)(r - 1) * rows * code;
    const float* dcls = d_all_cls + (size_t)r * rows * ncls;
    // box_r = treg_r (+ prev box {0,1,4}); d box_r = given + what layer r+1 sent down (in t.dbox)
    if (!have_dbox_next) TS_HIP(hipMemsetAsync(t.dbox, 0, (size_t)rows * code * 4, s));
    TS_TRY(add_into(d_all_box + (size_t)r * rows * code, t.dbox, (size_t)rows * code, s));
    // reg branch (Linear ReLU Linear ReLU Linear), dx3 -> dA (=)
    TS_TRY(lin_bwd(l.t1, t.dbox, nullptr, nullptr, rl.final_reg.l4, gl.final_reg.l4, l.t1, t.dB, 0, rows, C, code, s));
    TS_TRY(lin_bwd(l.t0, t.dB, nullptr, nullptr, rl.final_reg.l2, gl.final_reg.l2, l.t0, t.dC, 0, rows, C, C, s));
    TS_TRY(lin_bwd(l.x3, t.dC, nullptr, nullptr, rl.final_reg.l0, gl.final_reg.l0, nullptr, t.dA, 0, rows, C, C, s));
    // the reference of layer r is box_{r-1}: its {0,1,4} columns receive d box_r (HEAD:615-617, 661-662)
    if (r > 0) {
      // t.dbox becomes the incoming gradient of box_{r-1}: keep only columns 0,1,4
      TS_HIP(hipMemsetAsync(t.dB, 0, (size_t)rows * code * 4, s));
      TS_TRY(launch_box_ref_bwd(t.dbox, code, t.dB, rows, s));
      TS_HIP(hipMemcpyAsync(t.dbox, t.dB, (size_t)rows * code * 4, hipMemcpyDeviceToDevice, s));
      have_dbox_next = true;
    }
    // cls branch (Linear LN ReLU Linear LN ReLU Linear), dx3 +=
    TS_TRY(lin_bwd(l.c3, dcls, nullptr, nullptr, rl.final_cls.l6, gl.final_cls.l6, nullptr, t.dB, 0, rows, C, ncls, s));
    TS_TRY(ln_bwd(l.c2, nullptr, rl.final_cls.n4, gl.final_cls.n4, t.dB, l.c3, t.dC, rows, s));
    TS_TRY(lin_bwd(l.c1, t.dC, nullptr, nullptr, rl.final_cls.l3, gl.final_cls.l3, nullptr, t.dB, 0, rows, C, C, s));
    TS_TRY(ln_bwd(l.c0, nullptr, rl.final_cls.n1, gl.final_cls.n1, t.dB, l.c1, t.dC, rows, s));
    TS_TRY(lin_bwd(l.x3, t.dC, nullptr, nullptr, rl.final_cls.l0, gl.final_cls.l0, nullptr, t.dA, 1, rows, C, C, s));
    if (have_dqin) TS_TRY(add_into(t.dqin, t.dA, (size_t)rows * C, s));
    // x3 = LN3(x2 + ff): dz -> dB (grad of x2 and of ff)
    TS_TRY(ln_bwd(l.x2, l.ff, rl.norm3, gl.norm3, t.dA, nullptr, t.dB, rows, s));
    const float* dff = t.dB;                                  // gradient of ffn_out
    if (drop) {     // through rf_dropout3: same mask on the gradient; dz itself still feeds dx2 below
      TS_TRY(launch_dropout(t.dB, nullptr, nullptr, rows, C, make_drop(dropout_p, dropout_seed, 4u * r + 3u, tref), t.dD, s));
      dff = t.dD;
    }
    // l.h = rf_dropout(relu(.)): its zeros are the ReLU zeros and the dropped elements, the kept ones
    // carry 1 / (1 - p)
    TS_TRY(lin_bwd(l.h, dff, nullptr, nullptr, rl.linear2, gl.linear2, l.h, t.dh, 0, rows, F, C, s, keep_scale));
    TS_TRY(lin_bwd(l.x2, t.dh, nullptr, nullptr, rl.linear1, gl.linear1, nullptr, t.dB, 1, rows, C, F, s));   // dx2 = dz + dh W1
    // x2 = LN2(x1): dx1 -> dC
    TS_TRY(ln_bwd(l.x1, nullptr, rl.norm2, gl.norm2, t.dB, nullptr, t.dC, rows, s));
    // x1 = qin + gate * out_proj(ao): d ao -> dA, d qin = dx1
    const float* dproj = t.dC;                                // gradient of out_proj(ao) before the row gate
    if (drop) {     // through rf_dropout2
      TS_TRY(launch_dropout(t.dC, nullptr, nullptr, rows, C, make_drop(dropout_p, dropout_seed, 4u * r + 1u, tref), t.dD, s));
      dproj = t.dD;
    }
    TS_TRY(lin_bwd(l.ao, dproj, nullptr, l.hits, rl.attn.out_proj, gl.attn.out_proj, nullptr, t.dA, 0, rows, C, C, s));
    TS_HIP(hipMemsetAsync(t.dkv, 0, (size_t)rt * 2 * C * 4, s));
    {
      RadarAttnArgs ra = core_args(w, r, t, box_prev, radar_tokens, B, T, pad_mult, dropout_p, dropout_seed);
      const float qs = ra.qscale;
      ra.qscale = 1.0f;
      TS_TRY(launch_radar_attn_bwd(ra, qs, t.dA, t.dqp, t.dkv, s));
    }
    const tc_linear wq{rl.attn.in_proj.w, rl.attn.in_proj.b};
    const tc_linear gq{gl.attn.in_proj.w, gl.attn.in_proj.b};
    const tc_linear wkv{rl.attn.in_proj.w + (size_t)C * C, rl.attn.in_proj.b + C};
    const tc_linear gkv{gl.attn.in_proj.w + (size_t)C * C, gl.attn.in_proj.b + C};
    // d qin = dx1 + dqp Wq  (into dqin for the layer below; nothing below layer 0: the decoder is frozen)
    if (r > 0) {
      TS_HIP(hipMemcpyAsync(t.dqin, t.dC, (size_t)rows * C * 4, hipMemcpyDeviceToDevice, s));
      TS_TRY(lin_bwd(qin, t.dqp, nullptr, nullptr, wq, gq, nullptr, t.dqin, 1, rows, C, C, s));
      have_dqin = true;
    } else {
      TS_TRY(lin_bwd(qin, t.dqp, nullptr, nullptr, wq, gq, nullptr, nullptr, 0, rows, C, C, s));
    }
    TS_TRY(lin_bwd(t.mem, t.dkv, nullptr, nullptr, wkv, gkv, nullptr, t.dmem, 1, rt, C, 2 * C, s));
  }
  // encoders: mem = relu(LN4(u2)) + relu(feat4(f1)); both summands see dmem
  const tc_pos_encoder& pe = w->radar_position_encoder;
  const tc_pos_encoder& gpe = grads->radar_position_encoder;
  TS_TRY(lin_bwd(t.f1, t.dmem, t.f2, nullptr, w->radar_feat4, grads->radar_feat4, t.f1, t.dt128, 0, rt, 128, C, s));
  TS_TRY(lin_bwd(t.f0, t.dt128, nullptr, nullptr, w->radar_feat2, grads->radar_feat2, t.f0, t.dt64, 0, rt, 64, 128, s));
  TS_TRY(lin_bwd(radar_tokens, t.dt64, nullptr, nullptr, w->radar_feat0, grads->radar_feat0, nullptr, nullptr, 0, rt, RI, 64, s));
  // (rows of the token gradients reuse dqp/dkv-sized scratch: rt <= tape sizes by construction)
  float* du = t.dkv;                 // [rt, C] scratch (dkv is [rt, 2C])
  float* du2 = t.dkv + (size_t)rt * C;
  TS_TRY(ln_bwd(t.u2, nullptr, pe.n4, gpe.n4, t.dmem, t.pos, du, rt, s));
  TS_TRY(lin_bwd(t.u1, du, nullptr, nullptr, pe.l3, gpe.l3, nullptr, du2, 0, rt, C, C, s));
  TS_TRY(ln_bwd(t.u0, nullptr, pe.n1, gpe.n1, du2, t.u1, du, rt, s));
  TS_HIP(hipMemsetAsync(t.dw0p, 0, (size_t)C * 4 * 4, s));
  // (the padded xyz operand is rebuilt here: the tape of tc_radar_train_fwd_fused does not carry it)
  TS_HIP(hipMemsetAsync(t.xyz4, 0, (size_t)rt * 4 * 4, s));
  TS_TRY(copy_cols(radar_tokens, RI, t.xyz4, 4, rt, 3, 0, s));
  TS_TRY(launch_linear_bwd_weight(t.xyz4, du, nullptr, nullptr, t.dw0p, const_cast<float*>(gpe.l0.b), rt, 4, C,
                                  1.0f, s));
  TS_TRY(copy_cols(t.dw0p, 4, const_cast<float*>(gpe.l0.w), 3, C, 3, 1, s));
  return 0;
}

// ---- the backward as ONE launch of the row chain for the query side + a handful for the token side + ONE
// grouped weight-gradient launch (round 2: ~130 launches).  Workspace: the transposed packed weights of the
// three layers, the dY tensors the chain stores, dK | dV of the three layers.
namespace {
struct BwdWs {
  tc_radar_layer wT[TC_MAX_RADAR_LAYERS];
  float* dy[DY_COUNT]; size_t dy_stride;
  float* dkv[TC_MAX_RADAR_LAYERS];
  float* dmem;
  float* du0;                  // [rt, C]: gradient of the position encoder's first LayerNorm input
};
size_t bwd_ws_layout(const tc_head_weights* w, int B, int T, void* base, size_t cap, BwdWs* out) {
  const size_t rows = (size_t)B * w->num_query, rt = (size_t)B * T;
  const size_t C = w->embed_dims, F = w->ffn_dims, code = w->code_size, ncls = w->num_classes;
  Arena a(base, cap);
  BwdWs b;
  memset(&b, 0, sizeof(b));
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    tc_radar_layer& l = b.wT[r];
    l.attn.in_proj.w = a.take<float>(packed_floats((int)C, (int)C));
    l.attn.out_proj.w = a.take<float>(packed_floats((int)C, (int)C));
    l.linear1.w = a.take<float>(packed_floats((int)C, (int)F));        // (W1^T: C outputs, F inputs)
    l.linear2.w = a.take<float>(packed_floats((int)F, (int)C));
    l.final_cls.l0.w = a.take<float>(packed_floats((int)C, (int)C));
    l.final_cls.l3.w = a.take<float>(packed_floats((int)C, (int)C));
    l.final_cls.l6.w = a.take<float>(packed_floats((int)C, (int)ncls));
    l.final_reg.l0.w = a.take<float>(packed_floats((int)C, (int)C));
    l.final_reg.l2.w = a.take<float>(packed_floats((int)C, (int)C));
    l.final_reg.l4.w = a.take<float>(packed_floats((int)C, (int)code));
  }
  float* first = nullptr;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    float* s0 = a.take<float>(rows * code);
    if (r == 0) { first = s0; b.dy[DY_DBOX] = s0; }
    float* p1 = a.take<float>(rows * C); float* p2 = a.take<float>(rows * C); float* p3 = a.take<float>(rows * C);
    float* p4 = a.take<float>(rows * C); float* p5 = a.take<float>(rows * C); float* p6 = a.take<float>(rows * F);
    float* p7 = a.take<float>(rows * C); float* p8 = a.take<float>(rows * C);
    float* p9 = a.take<float>(rows * ncls);
    if (r == 0) b.dy[DY_DCLS] = p9;
    if (r == 0) {
      b.dy[DY_DT1] = p1; b.dy[DY_DT0] = p2; b.dy[DY_DC2] = p3; b.dy[DY_DC0] = p4; b.dy[DY_DFF] = p5; b.dy[DY_DH] = p6;
      b.dy[DY_DPROJ] = p7; b.dy[DY_DQP] = p8;
    }
    if (r == 1) b.dy_stride = (size_t)(s0 - first);
  }
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) b.dkv[r] = a.take<float>(rt * 2 * C);   // contiguous with dmem: one memset
  b.dmem = a.take<float>(rt * C);
  b.du0 = a.take<float>(rt * C);     // its own slice: a tape scratch sized by the QUERY rows overran for T > 2 Q
  if (out) *out = b;
  return a.off;
}
}  // namespace

// the transposed packed weights of the three fusion layers (what the backward row chain multiplies by): pack jobs
static int transposed_jobs(const tc_head_weights* w, const BwdWs& ws, PackJob* jobs) {
  const int C = w->embed_dims, F = w->ffn_dims, code = w->code_size, ncls = w->num_classes;
  int n = 0;
  auto add = [&](const float* W, float* P, int N_out, int K_in) {
    PackJob j;
    j.W = W; j.P = P; j.P16 = nullptr; j.N = N_out; j.K = K_in; j.transpose = 1; j.ldw = N_out;
    jobs[n++] = j;
  };
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    const tc_radar_layer& rl = w->radar[r];
    const tc_radar_layer& lt = ws.wT[r];
    // forward W is [N_fwd][K_fwd] (row stride K_fwd); the chain computes dx[K_fwd] = dy[N_fwd] W, i.e. a linear
    // step with N_out = K_fwd outputs and K_in = N_fwd inputs whose weight (n, k) = W[k][n]: ldw = K_fwd = N_out
    add(rl.attn.in_proj.w, const_cast<float*>(lt.attn.in_proj.w), C, C);          // Wq: rows 0..C-1 of in_proj
    add(rl.attn.out_proj.w, const_cast<float*>(lt.attn.out_proj.w), C, C);
    add(rl.linear1.w, const_cast<float*>(lt.linear1.w), C, F);                    // linear1: [F][C]
    add(rl.linear2.w, const_cast<float*>(lt.linear2.w), F, C);                    // linear2: [C][F]
    add(rl.final_cls.l0.w, const_cast<float*>(lt.final_cls.l0.w), C, C);
    add(rl.final_cls.l3.w, const_cast<float*>(lt.final_cls.l3.w), C, C);
    add(rl.final_cls.l6.w, const_cast<float*>(lt.final_cls.l6.w), C, ncls);       // [ncls][C]
    add(rl.final_reg.l0.w, const_cast<float*>(lt.final_reg.l0.w), C, C);
    add(rl.final_reg.l2.w, const_cast<float*>(lt.final_reg.l2.w), C, C);
    add(rl.final_reg.l4.w, const_cast<float*>(lt.final_reg.l4.w), C, code);       // [code][C]
  }
  return n;
}
// (head.hip: tc_radar_train_repack appends these to the forward's re-pack jobs -- ONE launch for both layouts)
int radar_train_transposed_jobs(const tc_head_weights* w, int B, int T, void* workspace, size_t workspace_bytes,
                                PackJob* jobs, int cap) {
  if (check(w, B, T) != 0) return -1;
  BwdWs ws;
  if (workspace == nullptr || bwd_ws_layout(w, B, T, workspace, workspace_bytes, &ws) > workspace_bytes || cap < 10 * TC_MAX_RADAR_LAYERS) {
    set_error("radar_train_repack: backward workspace too small");
    return -1;
  }
  return transposed_jobs(w, ws, jobs);
}

size_t tc_radar_train_bwd_workspace_bytes(const tc_head_weights* w, int B, int T) {
  if (check(w, B, T) != 0) return 0;
  return bwd_ws_layout(w, B, T, nullptr, ~size_t(0), nullptr);
}

// The weight gradients dW = dY^T X (+ bias column sums) of the trainable stack from the tape and the dY tensors the backward
// chain / the token side stored.  group -1: all 38 in one grouped launch; 0 .. TC_MAX_RADAR_LAYERS - 1: the jobs of fusion
// layer (TC_MAX_RADAR_LAYERS - 1 - group) -- the top layer first, the order in which a layer-wise backward would finish
// them --; TC_MAX_RADAR_LAYERS: the token side (radar encoders).
static int weight_gradients(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                            const float* radar_tokens, int B, int T, const Tape& t, const BwdWs& ws, int group, hipStream_t s) {
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, code = w->code_size;
  const int ncls = w->num_classes, RI = w->radar_in_dims;
  const int rows = B * Q, rt = B * T;
  const tc_pos_encoder& gpe = grads->radar_position_encoder;
  float* du = t.dkv;                 // [rt, C] scratch of the tape (the token side's LayerNorm backward wrote it)
  float* du0 = ws.du0;               // [rt, C]
  WeightJob jobs[11 * TC_MAX_RADAR_LAYERS + 5];
  int n = 0;
  auto job = [&](const float* x, const float* dy, const tc_linear& g, int M, int K, int N, const float* relu = nullptr,
                 int ldx = 0) {
    WeightJob j;
    j.x = x; j.dy = dy; j.dw = const_cast<float*>(g.w); j.db = const_cast<float*>(g.b); j.M = M; j.K = K; j.N = N; j.relu = relu;
    j.ldx = ldx;
    jobs[n++] = j;
  };
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    if (group >= 0 && r != TC_MAX_RADAR_LAYERS - 1 - group) continue;
    const tc_radar_layer& gl = grads->radar[r];
    const LayerTape& l = t.L[r];
    const size_t off = (size_t)r * ws.dy_stride;
    const float* qin = r == 0 ? hs_last : t.L[r - 1].x3;
    job(qin, ws.dy[DY_DQP] + off, tc_linear{gl.attn.in_proj.w, gl.attn.in_proj.b}, rows, C, C);
    job(t.mem, ws.dkv[r], tc_linear{gl.attn.in_proj.w + (size_t)C * C, gl.attn.in_proj.b + C}, rt, C, 2 * C);
    job(l.ao, ws.dy[DY_DPROJ] + off, gl.attn.out_proj, rows, C, C);
    job(l.x2, ws.dy[DY_DH] + off, gl.linear1, rows, C, F);
    job(l.h, ws.dy[DY_DFF] + off, gl.linear2, rows, F, C);
    job(l.x3, ws.dy[DY_DC0] + off, gl.final_cls.l0, rows, C, C);
    job(l.c1, ws.dy[DY_DC2] + off, gl.final_cls.l3, rows, C, C);
    job(l.c3, ws.dy[DY_DCLS] + off, gl.final_cls.l6, rows, C, ncls);        // (the guarded copy the chain stored)
    job(l.x3, ws.dy[DY_DT0] + off, gl.final_reg.l0, rows, C, C);
    job(l.t0, ws.dy[DY_DT1] + off, gl.final_reg.l2, rows, C, C);
    job(l.t1, ws.dy[DY_DBOX] + off, gl.final_reg.l4, rows, C, code);
  }
  if (group < 0 || group == TC_MAX_RADAR_LAYERS) {
    job(t.f1, ws.dmem, grads->radar_feat4, rt, 128, C, t.f2);
    job(t.f0, t.dt128, grads->radar_feat2, rt, 64, 128);
    job(radar_tokens, t.dt64, grads->radar_feat0, rt, RI, 64);
    job(t.u1, du, gpe.l3, rt, C, C);
    // radar_position_encoder.0 (Linear(3, C)): X = the tokens' first three columns (xyz), read in place
    job(radar_tokens, du0, gpe.l0, rt, 3, C, nullptr, RI);
  }
  return launch_linear_bwd_weight_group(jobs, n, s);
}

int tc_radar_train_bwd_fused_ex(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                                const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                                const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                                void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                                float* layer_losses_clean, int flags, tc_stream_t stream) {
  TS_TRY(check(w, B, T));
  TC_REQUIRE(grads != nullptr && workspace != nullptr, "radar_train_bwd_fused: null argument");
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_train_bwd_fused: dropout_p=%g", (double)dropout_p);
  TC_REQUIRE(w->code_size <= 10, "radar_train_bwd_fused: code_size=%d", w->code_size);
  Tape t;
  TC_REQUIRE(tape_layout(w, B, T, tape, tape_bytes, &t) <= tape_bytes, "radar_train_bwd_fused: tape too small");
  BwdWs ws;
  TC_REQUIRE(bwd_ws_layout(w, B, T, workspace, workspace_bytes, &ws) <= workspace_bytes,
             "radar_train_bwd_fused: workspace too small");
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, code = w->code_size;
  const int ncls = w->num_classes, RI = w->radar_in_dims;
  const int rows = B * Q, rt = B * T;
  // 1. the transposed packed weights of the three layers: one launch -- unless the caller packed them together with the
  //    forward's copy (tc_radar_train_repack, flags bit 0: the parameters have not changed in between)
  if (!(flags & 1)) {
    PackJob jobs[10 * TC_MAX_RADAR_LAYERS];
    const int n = transposed_jobs(w, ws, jobs);
    TS_TRY(launch_pack_group(jobs, n, s));
  }
  // 2. accumulators
  TS_HIP(hipMemsetAsync(ws.dkv[0], 0, (size_t)((char*)(ws.dmem + (size_t)rt * C) - (char*)ws.dkv[0]), s));   // dK|dV x 3 + dmem
  // 3. the query side of all three layers: one launch
  float* slots[TS_COUNT];
  memset(slots, 0, sizeof(slots));
  const LayerTape& l0 = t.L[0];
  slots[TS_QP] = l0.qp; slots[TS_AO] = l0.ao; slots[TS_X1] = l0.x1; slots[TS_X2] = l0.x2; slots[TS_H] = l0.h;
  slots[TS_SUM] = l0.ff; slots[TS_X3] = l0.x3; slots[TS_C0] = l0.c0; slots[TS_C1] = l0.c1; slots[TS_C2] = l0.c2;
  slots[TS_C3] = l0.c3; slots[TS_T0] = l0.t0; slots[TS_T1] = l0.t1; slots[TS_TREG] = l0.treg;
  RadarBwdChainArgs a;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    a.wT[r] = ws.wT[r]; a.w[r] = w->radar[r]; a.grads[r] = grads->radar[r];
    a.kv[r] = t.L[r].kv; a.dkv[r] = ws.dkv[r];
    const float* box_prev = r == 0 ? last_box : all_bbox_preds + (size_t)(r - 1) * rows * code;
    a.cxy[r] = r == 0 ? t.cxy : box_prev; a.ld_c[r] = r == 0 ? 2 : code; a.box[r] = box_prev;
  }
  a.tape = slots; a.tape_stride = (size_t)(t.L[1].qp - t.L[0].qp);
  a.hits = t.L[0].hits; a.hits_stride = (size_t)(t.L[1].hits - t.L[0].hits);
  a.dy = ws.dy; a.dy_stride = ws.dy_stride;
  a.d_cls = d_all_cls; a.d_box = d_all_box; a.loss_vals = layer_losses; a.loss_out = layer_losses_clean;
  a.tokens = radar_tokens; a.RI = RI; a.T = T; a.pad_mult = pad_mult;
  a.nlayers = TC_MAX_RADAR_LAYERS; a.Q = Q; a.M = rows; a.code = code; a.ncls = ncls;
  a.qscale = 1.0f / sqrtf((float)(C / w->num_heads));
  a.drop = make_drop(dropout_p, dropout_seed, 0u, (unsigned)w->num_radar_tokens_ref);
  {
    // deterministic mode: the chain reads the ranges from a device copy in the last words of the caller's shadow buffer
    // (tc_radar_train_bwd_fused_det)
    const DetAcc& d = current_det();
    DetAcc* dev = current_det_device();
    if (d.shadow[0] != nullptr && d.shadow[1] != nullptr && dev != nullptr) {
      TS_TRY(launch_det_store(d, dev, s));
      a.det_device = dev;
    }
  }
  TS_TRY(launch_radar_chain_bwd(a, s));
  // (deterministic mode: the chain's dK | dV sums sit in their shadow; the token side reads dK | dV next)
  TS_TRY(launch_det_flush(current_det(), 1, s));
  // 4. the token side: dmem = sum_r dkv_r Wkv_r, then the encoders (data gradients only; weights below)
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r)
    TS_TRY(launch_linear_bwd_data(ws.dkv[r], nullptr, nullptr, w->radar[r].attn.in_proj.w + (size_t)C * C, nullptr, ws.dmem,
                                  rt, C, 2 * C, 1.0f, 1, s));
  const tc_pos_encoder& pe = w->radar_position_encoder;
  const tc_pos_encoder& gpe = grads->radar_position_encoder;
  // mem = relu(LN4(u2)) + relu(feat4(f1)): both summands see dmem
  TS_TRY(launch_linear_bwd_data(ws.dmem, t.f2, nullptr, w->radar_feat4.w, t.f1, t.dt128, rt, 128, C, 1.0f, 0, s));
  TS_TRY(launch_linear_bwd_data(t.dt128, nullptr, nullptr, w->radar_feat2.w, t.f0, t.dt64, rt, 64, 128, 1.0f, 0, s));
  float* du = t.dkv;                 // [rt, C] scratch of the tape
  float* du2 = t.dkv + (size_t)rt * C;
  TS_TRY(ln_bwd(t.u2, nullptr, pe.n4, gpe.n4, ws.dmem, t.pos, du, rt, s));
  TS_TRY(launch_linear_bwd_data(du, nullptr, nullptr, pe.l3.w, nullptr, du2, rt, C, C, 1.0f, 0, s));
  float* du0 = ws.du0;               // [rt, C]
  TS_TRY(ln_bwd(t.u0, nullptr, pe.n1, gpe.n1, du2, t.u1, du0, rt, s));
  // 5. every weight gradient: one grouped launch (two with the scalar variant for the 10-wide heads) -- or, flags bit 1,
  //    left to tc_radar_train_bwd_weights: one launch per fusion layer (top layer first) + one for the token side, so that
  //    the caller can start the gradient exchange of a chunk while the next one is computed (round 6)
  if (flags & 2) {
    TC_REQUIRE(current_det().shadow[0] == nullptr, "radar_train_bwd_fused: chunked weight gradients run with float atomics only");
    return 0;
  }
  TS_TRY(weight_gradients(w, grads, hs_last, radar_tokens, B, T, t, ws, -1, s));
  return launch_det_flush(current_det(), 0, s);      // (deterministic mode: the bucket's sums back into the gradients)
}

// The same backward with ORDER-FREE accumulation: every float atomic of the backward -- the weight
// gradients' row chunks, the bias column sums, the LayerNorm parameter gradients of the row chain and of the token side,
// the attention backward's dK | dV -- becomes an integer atomic on a 2^-40 fixed-point shadow (common.hpp DetAcc), the
// split reductions of the token side run unsplit, and two flush launches add the shadows back.  Two calls on the same
// inputs give bit-identical gradients.  grad_base / grad_elems: the span that holds every tensor of `grads` (the flat
// bucket); shadow: shadow_elems >= grad_elems + 3 * B * T * 2 * embed_dims + 8 64-bit words, the first grad_elems + 3 B T 2 C of
// them ZERO on entry (the call leaves them zero); the LAST 8 words of the buffer (shadow + shadow_elems - 8 ...) are
// scratch -- they hold the device copy of the ranges, at a place that does not depend on (B, T): one buffer sized for the
// longest frame serves every shorter one (round 6; until then the copy sat right behind the dK | dV shadows and a
// shorter frame's copy was read as sums by a later, longer one).
int tc_radar_train_bwd_fused_det(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                                 const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                                 const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                                 void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                 float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                                 float* layer_losses_clean, int flags, float* grad_base, size_t grad_elems,
                                 long long* shadow, size_t shadow_elems, tc_stream_t stream) {
  TS_TRY(check(w, B, T));
  TC_REQUIRE(grad_base != nullptr && shadow != nullptr && workspace != nullptr, "radar_train_bwd_fused_det: null argument");
  BwdWs ws;
  TC_REQUIRE(bwd_ws_layout(w, B, T, workspace, workspace_bytes, &ws) <= workspace_bytes,
             "radar_train_bwd_fused_det: workspace too small");
  const size_t dkv_elems = (size_t)TC_MAX_RADAR_LAYERS * B * T * 2 * w->embed_dims;
  constexpr size_t kTail = 8;                                 // a device copy of the ranges in the buffer's last words
  static_assert(sizeof(DetAcc) <= kTail * 8, "the device copy of the ranges fits the tail");
  TC_REQUIRE(shadow_elems >= grad_elems + dkv_elems + kTail, "radar_train_bwd_fused_det: shadow holds %zu words, %zu needed",
             shadow_elems, grad_elems + dkv_elems + kTail);
  TC_REQUIRE(ws.dkv[TC_MAX_RADAR_LAYERS - 1] + (size_t)B * T * 2 * w->embed_dims == ws.dkv[0] + dkv_elems,
             "radar_train_bwd_fused_det: the dK | dV accumulators are not contiguous");
  DetAcc d;
  d.lo[0] = grad_base; d.hi[0] = grad_base + grad_elems; d.shadow[0] = shadow;
  d.lo[1] = ws.dkv[0]; d.hi[1] = ws.dkv[0] + dkv_elems; d.shadow[1] = shadow + grad_elems;
  DetScope scope(d, reinterpret_cast<DetAcc*>(shadow + (shadow_elems - kTail)));
  return tc_radar_train_bwd_fused_ex(w, grads, hs_last, last_box, radar_tokens, B, T, pad_mult, all_bbox_preds, d_all_cls,
                                     d_all_box, tape, tape_bytes, workspace, workspace_bytes, dropout_p, dropout_seed,
                                     layer_losses, layer_losses_clean, flags, stream);
}

int tc_radar_train_bwd_fused(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                             const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                             const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                             void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                             float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                             float* layer_losses_clean, tc_stream_t stream) {
  return tc_radar_train_bwd_fused_ex(w, grads, hs_last, last_box, radar_tokens, B, T, pad_mult, all_bbox_preds, d_all_cls,
                                     d_all_box, tape, tape_bytes, workspace, workspace_bytes, dropout_p, dropout_seed,
                                     layer_losses, layer_losses_clean, 0, stream);
}

// Round 6: the weight gradients of ONE chunk (tc_radar_train_bwd_fused_ex with flags bit 1 left them out): group 0 ..
// TC_MAX_RADAR_LAYERS - 1 = fusion layers top-down, TC_MAX_RADAR_LAYERS = the radar encoders.  Same tape / workspace as
// the backward call it follows on the same stream.
int tc_radar_train_bwd_weights(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                               const float* radar_tokens, int B, int T, void* tape, size_t tape_bytes, void* workspace,
                               size_t workspace_bytes, int group, tc_stream_t stream) {
  TS_TRY(check(w, B, T));
  TC_REQUIRE(grads != nullptr && workspace != nullptr && tape != nullptr, "radar_train_bwd_weights: null argument");
  TC_REQUIRE(group >= 0 && group <= TC_MAX_RADAR_LAYERS, "radar_train_bwd_weights: group=%d (0..%d)", group, TC_MAX_RADAR_LAYERS);
  Tape t;
  TC_REQUIRE(tape_layout(w, B, T, tape, tape_bytes, &t) <= tape_bytes, "radar_train_bwd_weights: tape too small");
  BwdWs ws;
  TC_REQUIRE(bwd_ws_layout(w, B, T, workspace, workspace_bytes, &ws) <= workspace_bytes,
             "radar_train_bwd_weights: workspace too small");
  return weight_gradients(w, grads, hs_last, radar_tokens, B, T, t, ws, group, as_stream(stream));
}

// The multipliers (0 or 1 / (1 - p)) of elements 0..n-1 of a dropout site, for tests and for
// replaying an iteration elsewhere (site = 4 * radar layer + {0 attention probabilities, index
// ((row * 8 + head) * num_radar_tokens_ref + token); 1 rf_dropout2; 2 rf_dropout; 3 rf_dropout3,
// index row * cols + col}).
int tc_dropout_mask(float dropout_p, unsigned long long seed, int site, size_t n, float* out,
                    tc_stream_t stream) {
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f && n < (1ull << 32), "dropout_mask: p=%g n=%zu",
             (double)dropout_p, n);
  hipStream_t s = as_stream(stream);
  const float one = 1.0f;          // dropout of a tensor of ones
  unsigned bits;
  memcpy(&bits, &one, 4);
  TS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out), (int)bits, n, s));
  return launch_dropout(out, nullptr, nullptr, 1, (int)n, make_drop(dropout_p, seed, (unsigned)site, 1500u), out, s);
}

}  // extern "C"
