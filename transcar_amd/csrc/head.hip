// C ABI (include/transcar_hip.h) and the launch sequence of the whole hot path:
// Detr3DHead.forward of the reference (HEAD:248-740) in eval mode = 6 decoder
// layers (self-attn, camera cross-attn, FFN, box refinement) + radar encoders +
// 3 distance-gated radar fusion layers.  Every function only enqueues kernels
// on the caller's stream; all scratch comes from the caller's workspace, so a
// forward is capturable into a hipGraph and replayable.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"

namespace tc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define TC_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

#define TC_HIP(expr)                                          \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) {                                   \
      tc::set_error("%s: %s", #expr, hipGetErrorString(_e)); \
      return (int)_e;                                         \
    }                                                         \
  } while (0)

static int linear(const float* x, int ldx, const tc_linear& w, int M, int K, int N, int act,
                  float* y, int ldy, hipStream_t s, const float* x2 = nullptr,
                  const float* res = nullptr, int ldr = 0, const int* rowgate = nullptr) {
  GemmArgs g;
  g.X = x; g.ldx = ldx; g.X2 = x2; g.x2_cols = x2 ? ((N + 15) / 16) * 16 : 0;
  g.W = w.w; g.ldw = K; g.bias = w.b; g.R = res; g.ldr = ldr; g.rowgate = rowgate;
  g.Y = y; g.ldy = ldy; g.M = M; g.K = K; g.N = N; g.act = act;
  return launch_gemm(g, s);
}

static int layernorm(const float* a, const tc_lnorm& n, float* y, int M, int relu, hipStream_t s) {
  LnArgs l;
  l.a = a; l.gamma = n.g; l.beta = n.b; l.y = y; l.M = M; l.relu = relu;
  return launch_ln256(l, s);
}

// mmcv MultiheadAttention wrapper for decoder self-attention:
// out = x + out_proj(MHA(q = k = x + pos, v = x))
static int self_attn(const tc_mha& w, const float* x, const float* pos, float* out, int B, int Q,
                     int C, int H, float* qk, float* vt, int qpad, float* attn_o, hipStream_t s) {
  const int rows = B * Q;
  GemmArgs g;
  g.X = x; g.ldx = C; g.X2 = pos; g.x2_cols = 2 * C;
  g.W = w.in_proj.w; g.ldw = C; g.bias = w.in_proj.b;
  g.Y = qk; g.ldy = 2 * C;
  g.Yt = vt; g.t_col0 = 2 * C; g.t_ld = qpad; g.t_rows_per_batch = Q;
  g.M = rows; g.K = C; g.N = 3 * C;
  // q * log2(e)/sqrt(head_dim): the attention core exponentiates with 2^x
  g.scale = 1.4426950408889634f / sqrtf((float)(C / H)); g.scale_cols = C;
  TC_TRY(launch_gemm(g, s));
  TC_TRY(launch_self_attn_core(qk, qk + C, 2 * C, vt, qpad, attn_o, C, B, Q, H, s));
  return linear(attn_o, C, w.out_proj, rows, C, C, 0, out, C, s, nullptr, x, C);
}

struct CrossWs { float *logits, *sampled, *t0, *pe0, *pe1; };

// Detr3DCrossAtten.forward; returns the pieces the caller fuses with the next
// LayerNorm: t0 = output_proj(sampled) + query, pe1 = position_encoder.3(...)
static int cross_atten_parts(const tc_linear& aw, const tc_linear& oproj, const tc_pos_encoder& pe,
                             const tc_feats_nhwc* feats, int B, int Q, int C, int ncams,
                             const float* query, const float* pos, const float* l2i,
                             const float* ref, const float* pc, float img_h, float img_w,
                             const CrossWs& ws, unsigned long long* pair_counter, hipStream_t s) {
  const int rows = B * Q;
  const int NL = ncams * feats->num_levels;
  TC_TRY(linear(query, C, aw, rows, C, NL, 0, ws.logits, NL, s, pos));
  CamSampleArgs c;
  c.feats = *feats; c.B = B; c.Q = Q; c.C = C; c.num_cams = ncams;
  c.lidar2img = l2i; c.ref = ref; c.logits = ws.logits;
  for (int i = 0; i < 6; ++i) c.pc[i] = pc[i];
  c.img_h = img_h; c.img_w = img_w; c.out = ws.sampled; c.vis = nullptr; c.pair_counter = pair_counter;
  TC_TRY(launch_cam_sample(c, s));
  TC_TRY(linear(ws.sampled, C, oproj, rows, C, C, 0, ws.t0, C, s, nullptr, query, C));
  TC_TRY(launch_posenc_l1(ref, 3, 1, pe.l0.w, pe.l0.b, pe.n1.g, pe.n1.b, ws.pe0, rows, s));
  return linear(ws.pe0, C, pe.l3, rows, C, C, 0, ws.pe1, C, s);
}

static int check_dims(const tc_head_weights* w) {
  TC_REQUIRE(w != nullptr, "weights pointer is null");
  TC_REQUIRE(w->abi_version == TC_ABI_VERSION, "tc_head_weights.abi_version=%d, library=%d",
             w->abi_version, TC_ABI_VERSION);
  TC_REQUIRE(w->embed_dims == 256 && w->num_heads == 8, "embed_dims=%d num_heads=%d (256/8 supported)",
             w->embed_dims, w->num_heads);
  TC_REQUIRE(w->num_layers >= 1 && w->num_layers <= TC_MAX_LAYERS, "num_layers=%d", w->num_layers);
  TC_REQUIRE(w->num_radar_layers >= 0 && w->num_radar_layers <= TC_MAX_RADAR_LAYERS,
             "num_radar_layers=%d", w->num_radar_layers);
  TC_REQUIRE(w->num_levels == 4, "num_levels=%d (4 supported)", w->num_levels);
  TC_REQUIRE((w->ffn_dims & 31) == 0 && (w->radar_in_dims & 3) == 0, "ffn_dims/radar_in_dims alignment");
  return 0;
}

struct HeadWs {
  float *pos, *x, *qk, *vt, *attn_o, *t0, *t1, *t2, *ffn_h, *logits, *sampled;
  float *init_ref, *inter_refs, *hs, *reg_tmp, *box_m;
  float *tp0, *tp1, *f0, *f1, *f2, *radar_feat, *kv, *qproj, *rattn, *cxy, *addref, *qf;
  float* kv3[TC_MAX_RADAR_LAYERS];
  int* hits;
  int *perm, *hitflag;        // row order of the radar chain (launch_radar_compact) + its scratch
  int qpad;
};

static size_t head_ws_layout(const tc_head_weights* w, int B, int T, void* base, size_t cap,
                             HeadWs* out) {
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, L = w->num_layers;
  const size_t rows = (size_t)B * Q, rt = (size_t)B * T;
  const int qpad = ((Q + 15) / 16) * 16;
  Arena a(base, cap);
  HeadWs h;
  h.qpad = qpad;
  h.pos = a.take<float>(rows * C); h.x = a.take<float>(rows * C);
  h.qk = a.take<float>(rows * 2 * C); h.vt = a.take<float>((size_t)B * C * qpad);
  h.attn_o = a.take<float>(rows * C);
  h.t0 = a.take<float>(rows * C); h.t1 = a.take<float>(rows * C); h.t2 = a.take<float>(rows * C);
  h.ffn_h = a.take<float>(rows * F);
  h.logits = a.take<float>(rows * w->num_cams * w->num_levels);
  h.sampled = a.take<float>(rows * C);
  h.init_ref = a.take<float>(rows * 3); h.inter_refs = a.take<float>((size_t)L * rows * 3);
  h.hs = a.take<float>((size_t)L * rows * C);
  h.reg_tmp = a.take<float>(rows * w->code_size); h.box_m = a.take<float>(rows * w->code_size);
  h.tp0 = a.take<float>(rt * C); h.tp1 = a.take<float>(rt * C);
  h.f0 = a.take<float>(rt * 64); h.f1 = a.take<float>(rt * 128); h.f2 = a.take<float>(rt * C);
  h.radar_feat = a.take<float>(rt * C); h.kv = a.take<float>(rt * 2 * C);
  for (int i = 0; i < TC_MAX_RADAR_LAYERS; ++i) h.kv3[i] = a.take<float>(rt * 2 * C);
  h.qproj = a.take<float>(rows * C); h.rattn = a.take<float>(rows * C);
  h.cxy = a.take<float>(rows * 2); h.addref = a.take<float>(rows * 3);
  h.qf = a.take<float>(rows * C);
  h.hits = a.take<int>((size_t)TC_MAX_RADAR_LAYERS * rows);
  h.perm = a.take<int>(rows); h.hitflag = a.take<int>(rows);
  if (out) *out = h;
  return a.off;
}

// ---- fused path: 12 launches per frame (chain.hip) ---------------------------
static int head_forward_fused(const tc_head_weights* w, const tc_feats_nhwc* feats, int B,
                              const float* lidar2img, float img_h, float img_w,
                              const float* radar_tokens, int T, int pad_mult, float* all_cls_scores,
                              float* all_bbox_preds, const tc_head_aux* aux, const tc_head_options& opt,
                              const HeadWs& h_, hipStream_t s) {
  const int Q = w->num_query, C = w->embed_dims, L = w->num_layers, H = w->num_heads;
  const int code = w->code_size, ncls = w->num_classes;
  const int rows = B * Q, rt = B * T;
  const float attn_qscale = 1.4426950408889634f / sqrtf((float)(C / H));   // 2^x softmax
  // the caller's aux tensors ARE the buffers the chains write (and read back for the next layer): no copies
  // at the end of the forward (round 2: four device-to-device copies per forward with aux, e.g. every training
  // iteration)
  HeadWs h = h_;
  if (aux != nullptr) {
    if (aux->inter_states) h.hs = aux->inter_states;
    if (aux->inter_references) h.inter_refs = aux->inter_references;
    if (aux->last_box) h.box_m = aux->last_box;
  }
  unsigned long long* pairs = aux ? aux->sample_pairs : nullptr;
  const bool radar = w->num_radar_layers > 0;

  // the radar encoders and K/V projections do not depend on the decoder: they ride in the launches of two
  // decoder layers as extra workgroups (chain_dual_kernel): layers 0 and 1 in the one-call forward; the LAST
  // two when the forward comes in two phases (options.phase) -- their K/V feed only the fusion stack, and a
  // caller that still has to build the tokens does so while the device runs the layers before
  const int enc_first = opt.phase != 0 && L > 1 ? L - 2 : 0;
  const int lid_begin = opt.phase == 2 ? enc_first : 0, lid_end = opt.phase == 1 ? enc_first : L;
  RadarEncodeArgs re;
  if (radar) {
    re.tokens = radar_tokens; re.RI = w->radar_in_dims; re.M = rt;
    re.rpe = w->radar_position_encoder; re.f0 = w->radar_feat0; re.f2 = w->radar_feat2;
    re.f4 = w->radar_feat4; re.nlayers = w->num_radar_layers;
    for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
      const tc_mha& m = w->radar[r].attn;
      re.kvproj[r] = tc_linear{m.in_proj.w ? m.in_proj.w + (size_t)C * C : nullptr,
                               m.in_proj.b ? m.in_proj.b + C : nullptr};
      re.kv[r] = h.kv3[r];
    }
    re.radar_feat = h.radar_feat;
    re.w16_delta = w->packed16_delta; re.matrix_path = opt.matrix_path; re.range_status = opt.range_status;
  }

  // train-mode statistics of the frozen decoder: five dropout sites per layer (sites 16 + 8 l + 0..4)
  const bool ddrop = opt.decoder_dropout_p > 0.0f;
  TC_REQUIRE(!ddrop || (unsigned long long)(opt.dropout_seed_stride ? 1 : B) * H * Q * Q < (1ull << 32),
             "decoder dropout: B*H*Q*Q exceeds 32 bits");
  if (radar && ddrop && opt.phase != 1) TC_TRY(launch_radar_encode(re, s));     // no ride in the decoder launches then
  // layer 0 up to its attention output is a constant of the checkpoint (pack time) -- in eval
  // mode: with dropout on the attention probabilities it is not
  const bool folded = !ddrop && w->l0_attn_out != nullptr && w->l0_init_reference != nullptr;
  if (!folded && opt.phase != 2) {
    PrologueArgs pa;
    pa.qe = w->query_embedding; pa.Q = Q; pa.M = rows; pa.refpts = w->reference_points;
    pa.in_proj = w->layers[0].self_attn.in_proj; pa.init_ref = h.init_ref; pa.qk = h.qk; pa.vt = h.vt;
    pa.qpad = h.qpad; pa.qscale = attn_qscale; pa.w16_delta = w->packed16_delta;
    TC_TRY(launch_prologue(pa, s));
  }
  for (int lid = lid_begin; lid < lid_end; ++lid) {
    const bool l0c = folded && lid == 0;
    const float* ref_in = l0c ? w->l0_init_reference
                              : lid == 0 ? h.init_ref : h.inter_refs + (size_t)(lid - 1) * rows * 3;
    DecoderChainArgs d;
    if (ddrop) {
      d.drop = make_drop(opt.decoder_dropout_p, opt.dropout_seed, 16u + 8u * (unsigned)lid, 0u);
      if (opt.dropout_seed_stride != 0) { d.drop.seed_stride = opt.dropout_seed_stride; d.drop.rows_per_sample = (unsigned)Q; }
    }
    d.attn_o = l0c ? w->l0_attn_out : h.attn_o;
    d.attn_mod = l0c ? Q : 0; d.ref_mod = l0c ? Q : 0;
    if (lid == 0) { d.x_in = w->query_embedding + C; d.x_ld = 2 * C; d.x_mod = Q; }
    else { d.x_in = h.hs + (size_t)(lid - 1) * rows * C; d.x_ld = C; d.x_mod = 0; }
    d.qe = w->query_embedding; d.Q = Q;
    d.ref_in = ref_in; d.ref_out = h.inter_refs + (size_t)lid * rows * 3;
    d.box_m = lid == L - 1 ? h.box_m : nullptr;
    d.w = &w->layers[lid];
    d.next_in_proj = lid + 1 < L ? &w->layers[lid + 1].self_attn.in_proj : nullptr;
    d.qscale = attn_qscale;
    d.hs = h.hs + (size_t)lid * rows * C; d.qk = h.qk; d.vt = h.vt; d.qpad = h.qpad;
    d.cam.feats = *feats; d.cam.B = B; d.cam.Q = Q; d.cam.C = C; d.cam.num_cams = w->num_cams;
    d.cam.lidar2img = lidar2img; d.cam.ref = ref_in; d.cam.logits = nullptr;
    for (int i = 0; i < 6; ++i) d.cam.pc[i] = w->pc_range[i];
    d.cam.img_h = img_h; d.cam.img_w = img_w; d.cam.out = nullptr; d.cam.vis = nullptr;
    d.cam.pair_counter = pairs;
    d.code = code; d.M = rows; d.tile_rows = opt.chain_tile_rows; d.matrix_path = opt.matrix_path;
    d.range_status = opt.range_status;
    // the attention core follows the chains' rule (chain.hip tile_rows / use_f16x2): launches that run 16-row tiles on
    // the f16 matrix cores take the staged two-plane core (self_attn.hip, round 4) -- 4- / 8-row launches (one or two
    // frames: too few workgroups for a form without split keys) and TC_MATRIX_F32 the fp32 core.  A
    // frame's arithmetic is therefore fixed by (chain_tile_rows, matrix_path), not by how many frames share a launch.
    // Round 6 (opt-in, tc_head_options.cam_pregather): the staged core's launch also carries the camera gather of THIS
    // layer's chain (pre-gather workgroups: the layer's reference points are final since the previous chain).
    if (!l0c) {
      const int trows = opt.chain_tile_rows ? opt.chain_tile_rows : (rows <= 1024 ? 4 : rows <= 2048 ? 8 : 16);   // (32 counts as 16 here)
      if (trows >= 16 && opt.matrix_path != TC_MATRIX_F32) {
        PreGatherArgs pga;
        const bool pre = !ddrop && opt.cam_pregather == 1;
        if (pre) {
          // scratch: [rows][num_cams][levels][C] level values (only the visible pairs' 4 KiB are ever touched), [rows] masks
          float* pbuf = static_cast<float*>(opt.cam_pregather_ws);
          int* pmask = reinterpret_cast<int*>(pbuf + (size_t)rows * w->num_cams * w->num_levels * C);
          pga.cam = d.cam; pga.M = rows; pga.ref_mod = d.ref_mod; pga.out = pbuf; pga.mask = pmask;
          d.pre = pbuf; d.premask = pmask;
        }
        TC_TRY(launch_self_attn_core_x(h.qk, h.qk + C, 2 * C, h.vt, h.qpad, h.attn_o, C, B, Q, H, s, ddrop ? &d.drop : nullptr,
                                       pre ? &pga : nullptr));
      } else {
        TC_TRY(launch_self_attn_core(h.qk, h.qk + C, 2 * C, h.vt, h.qpad, h.attn_o, C, B, Q, H, s, ddrop ? &d.drop : nullptr));
      }
    }
    // the radar encoders ride in two launches, half each (all in layer 0 when there is only one layer)
    if (radar && !ddrop && lid == enc_first) TC_TRY(launch_decoder_chain_with_encoders(d, re, L > 1 ? 1 : 0, s));
    else if (radar && !ddrop && L > 1 && lid == enc_first + 1) TC_TRY(launch_decoder_chain_with_encoders(d, re, 2, s));
    else TC_TRY(launch_decoder_chain(d, s));
  }
  if (opt.phase == 1) return 0;
  if (aux) {
    if (aux->init_reference) {
      if (folded) {
        for (int b = 0; b < B; ++b)
          TC_HIP(hipMemcpyAsync(aux->init_reference + (size_t)b * Q * 3, w->l0_init_reference,
                                (size_t)Q * 3 * 4, hipMemcpyDeviceToDevice, s));
      } else {
        TC_HIP(hipMemcpyAsync(aux->init_reference, h.init_ref, (size_t)rows * 3 * 4, hipMemcpyDeviceToDevice, s));
      }
    }
  }
  if (!radar) return 0;
  RadarChainArgs rc;
  rc.qf = h.hs + (size_t)(L - 1) * rows * C;
  rc.ref_last = h.inter_refs + (size_t)(L - 1) * rows * 3;
  rc.box_m = h.box_m; rc.tokens = radar_tokens; rc.RI = w->radar_in_dims;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) { rc.kv[r] = h.kv3[r]; rc.w[r] = w->radar[r]; }
  rc.nlayers = w->num_radar_layers; rc.Q = Q; rc.T = T; rc.pad_mult = pad_mult; rc.code = code;
  rc.ncls = ncls; rc.M = rows; rc.qscale = 1.0f / sqrtf((float)(C / H));
  for (int i = 0; i < 6; ++i) rc.pc[i] = w->pc_range[i];
  rc.all_cls = all_cls_scores; rc.all_box = all_bbox_preds; rc.hits = h.hits;
  rc.tile_rows = opt.chain_tile_rows; rc.last_cls_only = opt.last_level_cls_only; rc.matrix_path = opt.matrix_path;
  rc.range_status = opt.range_status;
  if (opt.radar_row_order == 2 || (opt.radar_row_order == 0 && rows > 1024)) {
    // queries with a radar return inside their first gate go first: the other tiles skip the gated part
    TC_TRY(launch_radar_compact(rc.ref_last, rc.box_m, code, 0, w->pc_range, radar_tokens, w->radar_in_dims, B, Q, T,
                                w->radar[0].radius_min, w->radar[0].radius_max, h.hitflag, h.perm, s));
    rc.row_perm = h.perm;
  }
  TC_TRY(launch_radar_chain(rc, s));
  if (aux && aux->radar_hit_counts)
    TC_HIP(hipMemcpyAsync(aux->radar_hit_counts, h.hits, (size_t)w->num_radar_layers * rows * 4,
                          hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- packed weights for the fused chains (pack.hip) --------------------------
struct PackItem { const float* src; int N, K; const float** slot; bool narrow; };

static int collect_pack_items(const tc_head_weights* w, tc_head_weights* v, PackItem* it) {
  const int C = w->embed_dims, F = w->ffn_dims, NL = w->num_cams * w->num_levels;
  const int code = w->code_size, ncls = w->num_classes;
  int n = 0;
  // narrow: the three 10-column heads the chains evaluate as wave-per-row dot products (K_NARROW): they
  // stay in the nn.Linear layout (the view keeps the caller's pointer: nothing to re-pack after an
  // optimizer step); listed so that the item order / counts stay what tc_head_repack_trainable assumes
  auto add = [&](const tc_linear& src, tc_linear& dst, int N, int K, bool narrow = false) {
    it[n].src = src.w; it[n].N = N; it[n].K = K; it[n].slot = &dst.w; it[n].narrow = narrow; ++n;
  };
  add(w->reference_points, v->reference_points, 3, C);
  for (int l = 0; l < w->num_layers; ++l) {
    const tc_decoder_layer& a = w->layers[l]; tc_decoder_layer& b = v->layers[l];
    add(a.self_attn.in_proj, b.self_attn.in_proj, 3 * C, C);
    add(a.self_attn.out_proj, b.self_attn.out_proj, C, C);
    add(a.attention_weights, b.attention_weights, NL, C);
    add(a.output_proj, b.output_proj, C, C);
    add(a.position_encoder.l3, b.position_encoder.l3, C, C);
    add(a.ffn0, b.ffn0, F, C);
    add(a.ffn1, b.ffn1, C, F);
    add(a.reg.l0, b.reg.l0, C, C);
    add(a.reg.l2, b.reg.l2, C, C);
    add(a.reg.l4, b.reg.l4, code, C, true);
  }
  if (w->num_radar_layers > 0) {
    add(w->radar_position_encoder.l3, v->radar_position_encoder.l3, C, C);
    add(w->radar_feat0, v->radar_feat0, 64, w->radar_in_dims);
    add(w->radar_feat2, v->radar_feat2, 128, 64);
    add(w->radar_feat4, v->radar_feat4, C, 128);
    for (int r = 0; r < w->num_radar_layers; ++r) {
      const tc_radar_layer& a = w->radar[r]; tc_radar_layer& b = v->radar[r];
      add(a.attn.in_proj, b.attn.in_proj, 3 * C, C);
      add(a.attn.out_proj, b.attn.out_proj, C, C);
      add(a.linear1, b.linear1, F, C);
      add(a.linear2, b.linear2, C, F);
      add(a.final_cls.l0, b.final_cls.l0, C, C);
      add(a.final_cls.l3, b.final_cls.l3, C, C);
      add(a.final_cls.l6, b.final_cls.l6, ncls, C, true);
      add(a.final_reg.l0, b.final_reg.l0, C, C);
      add(a.final_reg.l2, b.final_reg.l2, C, C);
      add(a.final_reg.l4, b.final_reg.l4, code, C, true);
    }
  }
  return n;
}
constexpr int MAX_PACK_ITEMS = 1 + 10 * TC_MAX_LAYERS + 4 + 10 * TC_MAX_RADAR_LAYERS;

}  // namespace tc

using namespace tc;

extern "C" {

int tc_abi_version(void) { return TC_ABI_VERSION; }
const char* tc_last_error(void) { return g_err; }

int tc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int tc_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, tc_stream_t stream) {
  return launch_nchw_to_nhwc(src, dst, n_img, C, H, W, as_stream(stream));
}

int tc_nchw_to_nhwc_levels(const float* const* src, float* const* dst, int num_levels, int n_img, int C,
                           const int* H, const int* W, tc_stream_t stream) {
  TC_REQUIRE(src != nullptr && dst != nullptr && H != nullptr && W != nullptr, "nchw_to_nhwc_levels: null argument");
  return launch_nchw_to_nhwc_levels(src, dst, num_levels, n_img, C, H, W, as_stream(stream));
}

int tc_radar_build_tokens(const double* raw, const double* times, const int* chan_start, int num_chan,
                          const double* radar_rot, const double* lidar_rot, const double* point_range,
                          float* tokens, int T, int* count, tc_stream_t stream) {
  return launch_radar_ingest(raw, times, chan_start, num_chan, radar_rot, lidar_rot, point_range, tokens, T,
                             count, as_stream(stream));
}

int tc_radar_build_tokens_batch(const double* raw, const double* times, const tc_radar_frame_desc* desc,
                                int P, int cap, float* tokens, int T, int* count, tc_stream_t stream) {
  return launch_radar_ingest_batch(raw, times, desc, P, cap, tokens, T, count, as_stream(stream));
}

int tc_linear_fwd(const float* x, const float* x2, const float* w, const float* b, const float* res,
                  float* y, int M, int K, int N, int act, tc_stream_t stream) {
  tc_linear lw{w, b};
  return linear(x, K, lw, M, K, N, act, y, N, as_stream(stream), x2, res, N);
}

int tc_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta,
                         float* y, int M, int C, int relu, tc_stream_t stream) {
  TC_REQUIRE(C == 256, "add_layernorm: C=%d (256 supported)", C);
  LnArgs l;
  l.a = a; l.b = b; l.gamma = gamma; l.beta = beta; l.y = y; l.M = M; l.relu = relu;
  return launch_ln256(l, as_stream(stream));
}

int tc_refine_reference_fwd(const float* reg_out, int code_size, const float* ref, float* new_ref,
                            int M, tc_stream_t stream) {
  TC_REQUIRE(code_size >= 5, "refine_reference: code_size=%d", code_size);
  const float pc0[6] = {0, 0, 0, 1, 1, 1};
  return launch_ref_update(reg_out, code_size, ref, new_ref, nullptr, pc0, M, as_stream(stream));
}

int tc_cam_sample_fuse_fwd(const tc_feats_nhwc* feats, int B, int Q, int C, int num_cams,
                           const float* lidar2img, const float* ref, const float* attn_logits,
                           const float* pc_range, float img_h, float img_w, float* out,
                           unsigned char* vis_mask, unsigned long long* pair_counter,
                           tc_stream_t stream) {
  TC_REQUIRE(feats != nullptr, "feats is null");
  CamSampleArgs c;
  c.feats = *feats; c.B = B; c.Q = Q; c.C = C; c.num_cams = num_cams;
  c.lidar2img = lidar2img; c.ref = ref; c.logits = attn_logits;
  for (int i = 0; i < 6; ++i) c.pc[i] = pc_range[i];
  c.img_h = img_h; c.img_w = img_w; c.out = out; c.vis = vis_mask; c.pair_counter = pair_counter;
  return launch_cam_sample(c, as_stream(stream));
}

size_t tc_cross_atten_workspace_bytes(int B, int Q, int C, int num_cams, int num_levels) {
  const size_t rows = (size_t)B * Q;
  return arena_slice(rows * num_cams * num_levels, 4) + 4 * arena_slice(rows * C, 4);
}

int tc_cross_atten_fwd(const tc_linear* attention_weights, const tc_linear* output_proj,
                       const tc_pos_encoder* position_encoder, const tc_feats_nhwc* feats, int B,
                       int Q, int C, int num_cams, const float* query, const float* query_pos,
                       const float* lidar2img, const float* ref, const float* pc_range, float img_h,
                       float img_w, float* out, void* workspace, size_t workspace_bytes,
                       tc_stream_t stream) {
  TC_REQUIRE(C == 256, "cross_atten: C=%d (256 supported)", C);
  TC_REQUIRE(workspace_bytes >= tc_cross_atten_workspace_bytes(B, Q, C, num_cams, feats->num_levels),
             "cross_atten: workspace too small");
  const size_t rows = (size_t)B * Q;
  Arena a(workspace, workspace_bytes);
  CrossWs ws;
  ws.logits = a.take<float>(rows * num_cams * feats->num_levels);
  ws.sampled = a.take<float>(rows * C); ws.t0 = a.take<float>(rows * C);
  ws.pe0 = a.take<float>(rows * C); ws.pe1 = a.take<float>(rows * C);
  hipStream_t s = as_stream(stream);
  TC_TRY(cross_atten_parts(*attention_weights, *output_proj, *position_encoder, feats, B, Q, C,
                           num_cams, query, query_pos, lidar2img, ref, pc_range, img_h, img_w, ws,
                           nullptr, s));
  // XFMR:378: output + inp_residual + relu(LN(position_encoder.3(.)))
  LnArgs l;
  l.a = ws.t0; l.c = ws.pe1; l.g2 = position_encoder->n4.g; l.b2 = position_encoder->n4.b;
  l.y = out; l.M = (int)rows;
  return launch_ln256(l, s);
}

size_t tc_self_attn_workspace_bytes(int B, int Q, int C) {
  const size_t rows = (size_t)B * Q;
  const size_t qpad = ((Q + 15) / 16) * 16;
  return arena_slice(rows * 2 * C, 4) + arena_slice((size_t)B * C * qpad, 4) + arena_slice(rows * C, 4);
}

int tc_self_attn_fwd(const tc_mha* w, const float* x, const float* pos, float* out, int B, int Q,
                     int C, int num_heads, void* workspace, size_t workspace_bytes,
                     tc_stream_t stream) {
  TC_REQUIRE(C == 256 && num_heads == 8, "self_attn: C=%d heads=%d (256/8 supported)", C, num_heads);
  TC_REQUIRE(workspace_bytes >= tc_self_attn_workspace_bytes(B, Q, C), "self_attn: workspace too small");
  const size_t rows = (size_t)B * Q;
  const int qpad = ((Q + 15) / 16) * 16;
  Arena a(workspace, workspace_bytes);
  float* qk = a.take<float>(rows * 2 * C);
  float* vt = a.take<float>((size_t)B * C * qpad);
  float* ao = a.take<float>(rows * C);
  return self_attn(*w, x, pos, out, B, Q, C, num_heads, qk, vt, qpad, ao, as_stream(stream));
}

int tc_decoder_layer_tail_fwd(const tc_decoder_layer* layer, const tc_linear* next_in_proj,
                              const tc_feats_nhwc* feats, int B, int Q, int num_cams,
                              int code_size, const float* attn_o, const float* x_in,
                              const float* query_embedding, const float* lidar2img,
                              const float* ref_in, const float* pc_range, float img_h,
                              float img_w, float* hs, float* ref_out, float* qk, float* vt,
                              int qpad, int tile_rows, tc_stream_t stream) {
  TC_REQUIRE(layer != nullptr && feats != nullptr, "decoder_layer_tail: null argument");
  const int C = 256;
  DecoderChainArgs d;
  d.attn_o = attn_o; d.x_in = x_in; d.x_ld = C; d.x_mod = 0;
  d.qe = query_embedding; d.Q = Q; d.ref_in = ref_in; d.ref_out = ref_out; d.box_m = nullptr;
  d.w = layer; d.next_in_proj = next_in_proj;
  d.qscale = 1.4426950408889634f / sqrtf(32.0f);
  d.hs = hs; d.qk = qk; d.vt = vt; d.qpad = qpad;
  d.cam.feats = *feats; d.cam.B = B; d.cam.Q = Q; d.cam.C = C; d.cam.num_cams = num_cams;
  d.cam.lidar2img = lidar2img; d.cam.ref = ref_in; d.cam.logits = nullptr;
  for (int i = 0; i < 6; ++i) d.cam.pc[i] = pc_range[i];
  d.cam.img_h = img_h; d.cam.img_w = img_w; d.cam.out = nullptr; d.cam.vis = nullptr;
  d.cam.pair_counter = nullptr;
  d.code = code_size; d.M = B * Q; d.tile_rows = TC_TILE_ROWS(tile_rows); d.matrix_path = TC_TILE_MATRIX(tile_rows);
  TC_REQUIRE(d.matrix_path <= TC_MATRIX_F16X2 && (tile_rows >> 10) == 0, "decoder_layer_tail: tile_rows=0x%x", tile_rows);
  return launch_decoder_chain(d, as_stream(stream));
}

int tc_sdpa_fwd(const float* q, const float* k, int ld, const float* vt, int ldt, float* out, int ldo,
                int B, int Q, int num_heads, tc_stream_t stream) {
  return launch_self_attn_core(q, k, ld, vt, ldt, out, ldo, B, Q, num_heads, as_stream(stream));
}

size_t tc_sdpa_f16x2_workspace_bytes(int, int, int) { return 0; }     // (the staged form needs none)

int tc_sdpa_fwd_f16x2(const float* qk, const float* vt, int ldt, float* out, int ldo, int B, int Q, int num_heads,
                      void* workspace, size_t workspace_bytes, tc_stream_t stream) {
  (void)workspace; (void)workspace_bytes;          // (an earlier form built f16 planes there; the staged form converts in LDS)
  return launch_self_attn_core_x(qk, qk + num_heads * 32, 2 * num_heads * 32, vt, ldt, out, ldo, B, Q, num_heads, as_stream(stream));
}

size_t tc_radar_xattn_workspace_bytes(int B, int Q, int T, int C) {
  return arena_slice((size_t)B * T * 2 * C, 4) + 2 * arena_slice((size_t)B * Q * C, 4) +
         arena_slice((size_t)B * Q, 4);
}

int tc_radar_gated_xattn_fwd(const tc_mha* w, const float* query, const float* centre_xy,
                             const float* box, int code_size, const float* radar_feat,
                             const float* radar_xy, int B, int Q, int T, int C, int num_heads,
                             int pad_mult, float radius_min, float radius_max, float* out,
                             int* hit_counts, void* workspace, size_t workspace_bytes,
                             tc_stream_t stream) {
  TC_REQUIRE(C == 256 && num_heads == 8, "radar_xattn: C=%d heads=%d (256/8 supported)", C, num_heads);
  TC_REQUIRE(workspace_bytes >= tc_radar_xattn_workspace_bytes(B, Q, T, C), "radar_xattn: workspace too small");
  hipStream_t s = as_stream(stream);
  const int rows = B * Q, rt = B * T;
  Arena a(workspace, workspace_bytes);
  float* kv = a.take<float>((size_t)rt * 2 * C);
  float* qproj = a.take<float>((size_t)rows * C);
  float* rattn = a.take<float>((size_t)rows * C);
  int* hits = hit_counts ? hit_counts : a.take<int>(rows);
  tc_linear wkv{w->in_proj.w + (size_t)C * C, w->in_proj.b + C};
  TC_TRY(linear(radar_feat, C, wkv, rt, C, 2 * C, 0, kv, 2 * C, s));
  GemmArgs g;
  g.X = query; g.ldx = C; g.W = w->in_proj.w; g.ldw = C; g.bias = w->in_proj.b;
  g.Y = qproj; g.ldy = C; g.M = rows; g.K = C; g.N = C;
  g.scale = 1.0f / sqrtf((float)(C / num_heads)); g.scale_cols = C;
  TC_TRY(launch_gemm(g, s));
  RadarAttnArgs r;
  r.qproj = qproj; r.ldq = C; r.kv = kv; r.ldkv = 2 * C; r.centre_xy = centre_xy; r.ld_c = 2;
  r.box = box; r.code = code_size; r.radar_xy = radar_xy; r.ld_xy = 2;
  r.B = B; r.Q = Q; r.T = T; r.C = C; r.H = num_heads; r.pad_mult = pad_mult;
  r.rmin = radius_min; r.rmax = radius_max; r.attn_out = rattn; r.hit_counts = hits;
  TC_TRY(launch_radar_attn(r, s));
  return linear(rattn, C, w->out_proj, rows, C, C, 0, out, C, s, nullptr, query, C, hits);
}

// per-call options of the whole-path entry points (tc_head_forward, tc_radar_fusion_fwd): defaults + validation
static int read_options(const tc_head_options* options, tc_head_options& opt) {
  memset(&opt, 0, sizeof(opt));
  if (options != nullptr) opt = *options;
  TC_REQUIRE(opt.radar_row_order >= 0 && opt.radar_row_order <= 2, "options.radar_row_order=%d (0 automatic, 1 own order, 2 hits first)",
             opt.radar_row_order);
  TC_REQUIRE(opt.chain_tile_rows == 0 || opt.chain_tile_rows == 4 || opt.chain_tile_rows == 8 ||
                 opt.chain_tile_rows == 16 || opt.chain_tile_rows == 32,
             "options.chain_tile_rows=%d (0 = automatic, 4, 8, 16 or 32)", opt.chain_tile_rows);
  TC_REQUIRE(opt.chain_tile_rows != 32 || opt.matrix_path != TC_MATRIX_F32,
             "options.chain_tile_rows=32 exists on the f16x2 matrix path only (matrix_path = f32 was asked for)");
  TC_REQUIRE(opt.decoder_dropout_p >= 0.0f && opt.decoder_dropout_p < 1.0f, "options.decoder_dropout_p=%g",
             (double)opt.decoder_dropout_p);
  TC_REQUIRE(opt.phase >= 0 && opt.phase <= 2, "options.phase=%d (0 whole forward, 1 before the radar tokens, 2 the rest)",
             opt.phase);
  TC_REQUIRE(opt.phase == 0 || !opt.unfused, "options.phase=%d needs the fused path", opt.phase);
  TC_REQUIRE(opt.matrix_path >= TC_MATRIX_AUTO && opt.matrix_path <= TC_MATRIX_F16X2,
             "options.matrix_path=%d (0 automatic, 1 fp32 MFMA, 2 two-plane f16 MFMA)", opt.matrix_path);
  TC_REQUIRE(opt.cam_pregather == 0 || opt.cam_pregather == 1, "options.cam_pregather=%d (0 off, 1 on)", opt.cam_pregather);
  TC_REQUIRE(opt.cam_pregather == 0 || opt.cam_pregather_ws != nullptr, "options.cam_pregather needs cam_pregather_ws");
  return 0;
}

int tc_radar_fusion_fwd(const tc_head_weights* packed_view, const float* hs_last, const float* ref_last,
                        const float* prev_box, const float* radar_tokens, int B, int T, int pad_mult,
                        int first_layer, int num_layers, float* all_cls_scores, float* all_bbox_preds,
                        int* hit_counts, const tc_head_options* options, void* workspace,
                        size_t workspace_bytes, tc_stream_t stream) {
  const tc_head_weights* w = packed_view;
  TC_TRY(check_dims(w));
  TC_REQUIRE(w->l0_attn_out != nullptr, "radar_fusion: packed_view was not produced by tc_head_pack_weights");
  TC_REQUIRE(first_layer >= 0 && num_layers >= 1 && first_layer + num_layers <= w->num_radar_layers,
             "radar_fusion: layers [%d, %d) of %d", first_layer, first_layer + num_layers, w->num_radar_layers);
  TC_REQUIRE(hs_last && prev_box && radar_tokens && all_cls_scores && all_bbox_preds && (first_layer > 0 || ref_last),
             "radar_fusion: null argument");
  TC_REQUIRE(B >= 1 && T >= 1 && pad_mult >= 1, "radar_fusion: B=%d T=%d pad_mult=%d", B, T, pad_mult);
  tc_head_options opt;
  TC_TRY(read_options(options, opt));
  HeadWs h;
  const size_t need = head_ws_layout(w, B, T, workspace, workspace_bytes, &h);
  TC_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu, have %zu", need, workspace_bytes);
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, H = w->num_heads, rows = B * Q, rt = B * T;
  const int code = w->code_size, ncls = w->num_classes;
  // encoders + K/V projections of all layers (stand-alone encoder program)
  RadarEncodeArgs re;
  re.tokens = radar_tokens; re.RI = w->radar_in_dims; re.M = rt;
  re.rpe = w->radar_position_encoder; re.f0 = w->radar_feat0; re.f2 = w->radar_feat2;
  re.f4 = w->radar_feat4; re.nlayers = w->num_radar_layers;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    const tc_mha& m = w->radar[r].attn;
    re.kvproj[r] = tc_linear{m.in_proj.w ? m.in_proj.w + (size_t)C * C : nullptr,
                             m.in_proj.b ? m.in_proj.b + C : nullptr};
    re.kv[r] = h.kv3[r];
  }
  re.radar_feat = h.radar_feat;
  re.w16_delta = w->packed16_delta; re.matrix_path = opt.matrix_path;
  if (!opt.reuse_radar_kv) TC_TRY(launch_radar_encode(re, s));
  RadarChainArgs rc;
  rc.qf = hs_last; rc.ref_last = ref_last; rc.box_m = prev_box; rc.tokens = radar_tokens;
  rc.RI = w->radar_in_dims;
  for (int r = 0; r < num_layers; ++r) { rc.kv[r] = h.kv3[first_layer + r]; rc.w[r] = w->radar[first_layer + r]; }
  rc.nlayers = num_layers; rc.Q = Q; rc.T = T; rc.pad_mult = pad_mult; rc.code = code; rc.ncls = ncls;
  rc.M = rows; rc.qscale = 1.0f / sqrtf((float)(C / H));
  for (int i = 0; i < 6; ++i) rc.pc[i] = w->pc_range[i];
  rc.all_cls = all_cls_scores + (size_t)first_layer * rows * ncls;
  rc.all_box = all_bbox_preds + (size_t)first_layer * rows * code;
  rc.hits = h.hits; rc.tile_rows = opt.chain_tile_rows; rc.last_cls_only = opt.last_level_cls_only; rc.matrix_path = opt.matrix_path;
  rc.range_status = opt.range_status;
  rc.cen_from_box = first_layer > 0;
  if (opt.radar_row_order == 2 || (opt.radar_row_order == 0 && rows > 1024)) {
    TC_TRY(launch_radar_compact(ref_last, prev_box, code, rc.cen_from_box, w->pc_range, radar_tokens, w->radar_in_dims,
                                B, Q, T, w->radar[first_layer].radius_min, w->radar[first_layer].radius_max,
                                h.hitflag, h.perm, s));
    rc.row_perm = h.perm;
  }
  TC_TRY(launch_radar_chain(rc, s));
  if (hit_counts != nullptr)
    TC_HIP(hipMemcpyAsync(hit_counts + (size_t)first_layer * rows, h.hits, (size_t)num_layers * rows * 4,
                          hipMemcpyDeviceToDevice, s));
  return 0;
}

int tc_radar_gate_selfcheck(int n_radii, unsigned long long seed, unsigned long long* mismatches,
                            tc_stream_t stream) {
  return launch_gate_selfcheck(n_radii, seed, mismatches, as_stream(stream));
}

int tc_rowops_selfcheck(int n_blocks, unsigned long long seed, unsigned long long* mismatches, tc_stream_t stream) {
  return launch_rowops_selfcheck(n_blocks, seed, mismatches, as_stream(stream));
}

size_t tc_box_decode_workspace_bytes(int B, int Q, int num_classes) {
  return box_decode_ws_bytes(B, Q, num_classes);
}

int tc_box_decode_topk(const float* cls_scores, const float* bbox_preds, int B, int Q,
                       int num_classes, int code_size, int max_num, const float* post_center_range,
                       float* boxes, float* scores, int* labels, unsigned char* valid,
                       void* workspace, size_t workspace_bytes, tc_stream_t stream) {
  return launch_box_decode(cls_scores, bbox_preds, B, Q, num_classes, code_size, max_num,
                           post_center_range, boxes, scores, labels, valid, workspace,
                           workspace_bytes, as_stream(stream));
}

int tc_box_decode_kept(const float* cls_scores, const float* bbox_preds, int B, int Q, int num_classes, int code_size,
                       int max_num, const float* post_center_range, float score_threshold, int use_threshold,
                       int z_shift, float* kept_boxes, float* kept_scores, long long* kept_labels, int* kept_count,
                       tc_stream_t stream) {
  BoxDecodeKept k{kept_boxes, kept_scores, kept_labels, kept_count, score_threshold, use_threshold, z_shift};
  return launch_box_decode(cls_scores, bbox_preds, B, Q, num_classes, code_size, max_num, post_center_range,
                           nullptr, nullptr, nullptr, nullptr, nullptr, 0, as_stream(stream), &k);
}

size_t tc_head_packed_bytes(const tc_head_weights* w) {
  if (check_dims(w) != 0) return 0;
  tc_head_weights view = *w;
  PackItem items[MAX_PACK_ITEMS];
  const int n = collect_pack_items(w, &view, items);
  size_t total = 0;
  for (int i = 0; i < n; ++i)
    if (!items[i].narrow) total += 3 * arena_slice(packed_floats(items[i].N, items[i].K), 4);   // + the 16x16x4 and the two-plane f16 copies
  // layer-0 constants + the scratch they are computed from (see tc_head_pack_weights)
  const size_t Q = w->num_query, C = w->embed_dims, qpad = ((Q + 15) / 16) * 16;
  total += arena_slice(Q * 3, 4) + arena_slice(Q * C, 4) + arena_slice(Q * 2 * C, 4) + arena_slice(C * qpad, 4);
  return total;
}

int tc_head_pack_weights(const tc_head_weights* w, void* packed, size_t packed_bytes,
                         tc_head_weights* packed_view, tc_stream_t stream) {
  TC_TRY(check_dims(w));
  TC_REQUIRE(packed != nullptr && packed_view != nullptr, "pack_weights: null output");
  TC_REQUIRE(packed_bytes >= tc_head_packed_bytes(w), "pack_weights: buffer too small");
  *packed_view = *w;
  PackItem items[MAX_PACK_ITEMS];
  const int n = collect_pack_items(w, packed_view, items);
  Arena a(packed, packed_bytes);
  // region A: every weight in the 4x4x1 layout; region B: the same weights, same order and slice
  // sizes, in the 16x16x4 layout -- one distance (packed16_delta floats) from any address of a
  // weight (or of a row block of it) to its counterpart; region C: the two-plane f16 copy (4 bytes per weight
  // too), 2 * packed16_delta from region A
  size_t region_a = 0;
  for (int i = 0; i < n; ++i)
    if (!items[i].narrow) region_a += arena_slice(packed_floats(items[i].N, items[i].K), 4);
  const size_t delta = region_a / sizeof(float);
  for (int i = 0; i < n; ++i) {
    TC_REQUIRE(items[i].src != nullptr, "pack_weights: weight %d is null", i);
    if (items[i].narrow) continue;                     // the view keeps the nn.Linear pointer
    float* dst = a.take<float>(packed_floats(items[i].N, items[i].K));
    TC_TRY(launch_pack_linear(items[i].src, items[i].N, items[i].K, dst, dst + delta, dst + 2 * delta, as_stream(stream)));
    *items[i].slot = dst;
  }
  for (int rgn = 0; rgn < 2; ++rgn)
    for (int i = 0; i < n; ++i)
      if (!items[i].narrow) a.take<float>(packed_floats(items[i].N, items[i].K));     // regions B, C
  packed_view->packed16_delta = delta;
  for (int l = 0; l < TC_MAX_LAYERS; ++l) packed_view->layers[l].packed16_delta = delta;
  for (int l = 0; l < TC_MAX_RADAR_LAYERS; ++l) packed_view->radar[l].packed16_delta = delta;
  // Layer 0's self-attention input is the learned embedding alone: reference points,
  // QKV projection and softmax(QK^T)V are constants of the checkpoint.  Evaluate them
  // here with the forward's own kernels (prologue chain + attention core, one batch).
  {
    const int Q = w->num_query, C = w->embed_dims, H = w->num_heads;
    const int qpad = ((Q + 15) / 16) * 16;
    float* init_ref = a.take<float>((size_t)Q * 3);
    float* attn_o = a.take<float>((size_t)Q * C);
    float* qk = a.take<float>((size_t)Q * 2 * C);
    float* vt = a.take<float>((size_t)C * qpad);
    hipStream_t s = as_stream(stream);
    TC_HIP(hipMemsetAsync(vt, 0, (size_t)C * qpad * 4, s));
    PrologueArgs pa;
    pa.qe = w->query_embedding; pa.Q = Q; pa.M = Q; pa.refpts = packed_view->reference_points;
    pa.in_proj = packed_view->layers[0].self_attn.in_proj; pa.init_ref = init_ref; pa.qk = qk; pa.vt = vt;
    pa.qpad = qpad; pa.qscale = 1.4426950408889634f / sqrtf((float)(C / H));
    pa.w16_delta = delta;
    TC_TRY(launch_prologue(pa, s));
    TC_TRY(launch_self_attn_core(qk, qk + C, 2 * C, vt, qpad, attn_o, C, 1, Q, H, s));
    packed_view->l0_init_reference = init_ref;
    packed_view->l0_attn_out = attn_o;
  }
  return 0;
}

int tc_head_repack_trainable_ex(const tc_head_weights* w, tc_head_weights* packed_view, int copies,
                                tc_stream_t stream) {
  TC_TRY(check_dims(w));
  TC_REQUIRE(packed_view != nullptr && packed_view->l0_attn_out != nullptr,
             "repack_trainable: packed_view was not produced by tc_head_pack_weights");
  TC_REQUIRE(copies == 1 || copies == 3, "repack_trainable: copies=%d (1: the 4x4x1 copy, 3: both)", copies);
  // the same item order as tc_head_pack_weights; the decoder's items (frozen under
  // tools/train.py:245-252) and the layer-0 constants keep their packed contents.
  // ONE launch for all of them (round 2: one launch per weight, 28 launches after every optimizer step).
  tc_head_weights scratch = *packed_view;
  PackItem items[MAX_PACK_ITEMS];
  const int n = collect_pack_items(w, &scratch, items);
  const int first = 1 + 10 * w->num_layers;
  PackJob jobs[MAX_PACK_ITEMS];
  int nj = 0;
  for (int i = first; i < n; ++i) {
    TC_REQUIRE(items[i].src != nullptr, "repack_trainable: weight %d is null", i);
    if (items[i].narrow) continue;                     // read in place by the chains
    // scratch is a copy of the packed view: its slot still holds the packed destination
    float* dst = const_cast<float*>(*items[i].slot);
    jobs[nj] = PackJob{items[i].src, dst, copies == 3 ? dst + packed_view->packed16_delta : nullptr,
                       items[i].N, items[i].K};
    jobs[nj++].PH = copies == 3 ? dst + 2 * packed_view->packed16_delta : nullptr;
  }
  if (nj == 0) return 0;
  return launch_pack_group(jobs, nj, as_stream(stream));
}

// train_stack.hip: the pack jobs of the backward's transposed weights (into its workspace)
extern "C" int radar_train_transposed_jobs(const tc_head_weights* w, int B, int T, void* workspace, size_t workspace_bytes,
                                           tc::PackJob* jobs, int cap);

int tc_radar_train_repack(const tc_head_weights* w, tc_head_weights* packed_view, void* bwd_workspace,
                          size_t bwd_workspace_bytes, int B, int T, tc_stream_t stream) {
  TC_TRY(check_dims(w));
  TC_REQUIRE(packed_view != nullptr && packed_view->l0_attn_out != nullptr,
             "radar_train_repack: packed_view was not produced by tc_head_pack_weights");
  tc_head_weights scratch = *packed_view;
  PackItem items[MAX_PACK_ITEMS];
  const int n = collect_pack_items(w, &scratch, items);
  const int first = 1 + 10 * w->num_layers;
  PackJob jobs[MAX_PACK_ITEMS + 10 * TC_MAX_RADAR_LAYERS];
  int nj = 0;
  for (int i = first; i < n; ++i) {
    TC_REQUIRE(items[i].src != nullptr, "radar_train_repack: weight %d is null", i);
    if (items[i].narrow) continue;
    jobs[nj++] = PackJob{items[i].src, const_cast<float*>(*items[i].slot), nullptr, items[i].N, items[i].K};
  }
  const int nt = radar_train_transposed_jobs(w, B, T, bwd_workspace, bwd_workspace_bytes, jobs + nj, 10 * TC_MAX_RADAR_LAYERS);
  if (nt < 0) return -1;
  return launch_pack_group(jobs, nj + nt, as_stream(stream));
}

int tc_head_repack_trainable(const tc_head_weights* w, tc_head_weights* packed_view,
                             tc_stream_t stream) {
  return tc_head_repack_trainable_ex(w, packed_view, 3, stream);
}

size_t tc_cam_pregather_workspace_bytes(const tc_head_weights* w, int B) {
  if (check_dims(w) != 0 || B < 1) return 0;
  const size_t rows = (size_t)B * w->num_query;
  return rows * ((size_t)w->num_cams * w->num_levels * w->embed_dims * sizeof(float) + sizeof(int));
}

size_t tc_head_workspace_bytes(const tc_head_weights* w, int B, int T) {
  if (check_dims(w) != 0) return 0;
  return head_ws_layout(w, B, T, nullptr, ~size_t(0), nullptr);
}

int tc_head_forward(const tc_head_weights* w, const tc_head_weights* packed_view,
                    const tc_feats_nhwc* feats, int B, const float* lidar2img, float img_h,
                    float img_w, const float* radar_tokens, int T, int pad_mult,
                    float* all_cls_scores, float* all_bbox_preds, const tc_head_aux* aux,
                    const tc_head_options* options, void* workspace, size_t workspace_bytes,
                    tc_stream_t stream) {
  TC_TRY(check_dims(w));
  tc_head_options opt;
  TC_TRY(read_options(options, opt));
  TC_REQUIRE(feats != nullptr && feats->num_levels == w->num_levels, "feats: num_levels mismatch");
  TC_REQUIRE(B >= 1, "B=%d", B);
  TC_REQUIRE(w->num_radar_layers == 0 || (radar_tokens != nullptr && T >= 1 && pad_mult >= 1),
             "radar tokens missing (T=%d pad_mult=%d)", T, pad_mult);
  HeadWs h;
  const size_t need = head_ws_layout(w, B, T, workspace, workspace_bytes, &h);
  TC_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu, have %zu", need, workspace_bytes);
  TC_REQUIRE(opt.cam_pregather == 0 || opt.cam_pregather_bytes >= tc_cam_pregather_workspace_bytes(w, B),
             "options.cam_pregather_ws holds %zu bytes, %zu needed", opt.cam_pregather_bytes, tc_cam_pregather_workspace_bytes(w, B));
  hipStream_t s = as_stream(stream);
  const int Q = w->num_query, C = w->embed_dims, F = w->ffn_dims, L = w->num_layers, H = w->num_heads;
  const int code = w->code_size, ncls = w->num_classes;
  const int rows = B * Q, rt = B * T;
  const float* pc = w->pc_range;
  unsigned long long* pairs = aux ? aux->sample_pairs : nullptr;
  if (packed_view != nullptr && !opt.unfused)
    return head_forward_fused(packed_view, feats, B, lidar2img, img_h, img_w, radar_tokens, T, pad_mult,
                              all_cls_scores, all_bbox_preds, aux, opt, h, s);
  TC_REQUIRE(opt.decoder_dropout_p == 0.0f && !opt.last_level_cls_only,
             "the operator-by-operator path implements the default options only");

  // ---- operator-by-operator path (options.unfused): the first build, kept
  // as an in-tree cross-check of the fused chains.  XFMR:119-123
  TC_TRY(launch_split_embed(w->query_embedding, Q, C, B, h.pos, h.x, s));
  TC_TRY(launch_init_ref(w->query_embedding, Q, C, w->reference_points.w, w->reference_points.b,
                         h.init_ref, B, s));
  const float* x = h.x;
  for (int lid = 0; lid < L; ++lid) {
    const tc_decoder_layer& ly = w->layers[lid];
    const float* ref_in = lid == 0 ? h.init_ref : h.inter_refs + (size_t)(lid - 1) * rows * 3;
    float* ref_out = h.inter_refs + (size_t)lid * rows * 3;
    float* hs_l = h.hs + (size_t)lid * rows * C;
    // self_attn, norm
    TC_TRY(self_attn(ly.self_attn, x, h.pos, h.t0, B, Q, C, H, h.qk, h.vt, h.qpad, h.attn_o, s));
    TC_TRY(layernorm(h.t0, ly.norm0, h.t1, rows, 0, s));
    // cross_attn, norm   (XFMR:346-378)
    CrossWs cw{h.logits, h.sampled, h.t0, h.t2, h.attn_o};
    TC_TRY(cross_atten_parts(ly.attention_weights, ly.output_proj, ly.position_encoder, feats, B, Q,
                             C, w->num_cams, h.t1, h.pos, lidar2img, ref_in, pc, img_h, img_w, cw,
                             pairs, s));
    {
      LnArgs l;
      l.a = h.t0; l.c = h.attn_o; l.g2 = ly.position_encoder.n4.g; l.b2 = ly.position_encoder.n4.b;
      l.gamma = ly.norm1.g; l.beta = ly.norm1.b; l.y = h.t1; l.M = rows;
      TC_TRY(launch_ln256(l, s));
    }
    // ffn, norm
    TC_TRY(linear(h.t1, C, ly.ffn0, rows, C, F, 1, h.ffn_h, F, s));
    TC_TRY(linear(h.ffn_h, F, ly.ffn1, rows, F, C, 0, h.t0, C, s, nullptr, h.t1, C));
    TC_TRY(layernorm(h.t0, ly.norm2, hs_l, rows, 0, s));
    x = hs_l;
    // XFMR:190-203 box refinement of the reference points
    TC_TRY(linear(hs_l, C, ly.reg.l0, rows, C, C, 1, h.t0, C, s));
    TC_TRY(linear(h.t0, C, ly.reg.l2, rows, C, C, 1, h.t2, C, s));
    TC_TRY(linear(h.t2, C, ly.reg.l4, rows, C, code, 0, h.reg_tmp, code, s));
    TC_TRY(launch_ref_update(h.reg_tmp, code, ref_in, ref_out, lid == L - 1 ? h.box_m : nullptr, pc,
                             rows, s));
  }
  if (aux) {
    if (aux->inter_states)
      TC_HIP(hipMemcpyAsync(aux->inter_states, h.hs, (size_t)L * rows * C * 4, hipMemcpyDeviceToDevice, s));
    if (aux->init_reference)
      TC_HIP(hipMemcpyAsync(aux->init_reference, h.init_ref, (size_t)rows * 3 * 4, hipMemcpyDeviceToDevice, s));
    if (aux->inter_references)
      TC_HIP(hipMemcpyAsync(aux->inter_references, h.inter_refs, (size_t)L * rows * 3 * 4,
                            hipMemcpyDeviceToDevice, s));
    if (aux->last_box)
      TC_HIP(hipMemcpyAsync(aux->last_box, h.box_m, (size_t)rows * code * 4, hipMemcpyDeviceToDevice, s));
  }
  if (w->num_radar_layers == 0) return 0;

  // radar encoders, HEAD:531-536
  const int RI = w->radar_in_dims;
  const tc_pos_encoder& rpe = w->radar_position_encoder;
  TC_TRY(launch_posenc_l1(radar_tokens, RI, 0, rpe.l0.w, rpe.l0.b, rpe.n1.g, rpe.n1.b, h.tp0, rt, s));
  TC_TRY(linear(h.tp0, C, rpe.l3, rt, C, C, 0, h.tp1, C, s));
  TC_TRY(linear(radar_tokens, RI, w->radar_feat0, rt, RI, 64, 1, h.f0, 64, s));
  TC_TRY(linear(h.f0, 64, w->radar_feat2, rt, 64, 128, 1, h.f1, 128, s));
  TC_TRY(linear(h.f1, 128, w->radar_feat4, rt, 128, C, 1, h.f2, C, s));
  {
    LnArgs l;
    l.a = h.tp1; l.gamma = rpe.n4.g; l.beta = rpe.n4.b; l.relu = 1; l.d = h.f2; l.y = h.radar_feat;
    l.M = rt;
    TC_TRY(launch_ln256(l, s));
  }
  // HEAD:543-547, 596-598
  TC_TRY(launch_radar_ref_l1(h.inter_refs + (size_t)(L - 1) * rows * 3, pc, h.cxy, h.addref, rows, s));
  const float* qf = h.hs + (size_t)(L - 1) * rows * C;
  const float qscale = 1.0f / sqrtf((float)(C / H));
  for (int r = 0; r < w->num_radar_layers; ++r) {
    const tc_radar_layer& rl = w->radar[r];
    float* cls_out = all_cls_scores + (size_t)r * rows * ncls;
    float* box_out = all_bbox_preds + (size_t)r * rows * code;
    const float* box_prev = r == 0 ? h.box_m : all_bbox_preds + (size_t)(r - 1) * rows * code;
    int* hits = h.hits + (size_t)r * rows;
    tc_linear wkv{rl.attn.in_proj.w + (size_t)C * C, rl.attn.in_proj.b + C};
    TC_TRY(linear(h.radar_feat, C, wkv, rt, C, 2 * C, 0, h.kv, 2 * C, s));
    GemmArgs g;
    g.X = qf; g.ldx = C; g.W = rl.attn.in_proj.w; g.ldw = C; g.bias = rl.attn.in_proj.b;
    g.Y = h.qproj; g.ldy = C; g.M = rows; g.K = C; g.N = C; g.scale = qscale; g.scale_cols = C;
    TC_TRY(launch_gemm(g, s));
    RadarAttnArgs ra;
    ra.qproj = h.qproj; ra.ldq = C; ra.kv = h.kv; ra.ldkv = 2 * C;
    ra.centre_xy = r == 0 ? h.cxy : box_prev; ra.ld_c = r == 0 ? 2 : code;
    ra.box = box_prev; ra.code = code; ra.radar_xy = radar_tokens; ra.ld_xy = RI;
    ra.B = B; ra.Q = Q; ra.T = T; ra.C = C; ra.H = H; ra.pad_mult = pad_mult;
    ra.rmin = rl.radius_min; ra.rmax = rl.radius_max; ra.attn_out = h.rattn; ra.hit_counts = hits;
    TC_TRY(launch_radar_attn(ra, s));
    // HEAD:581-586
    TC_TRY(linear(h.rattn, C, rl.attn.out_proj, rows, C, C, 0, h.t0, C, s, nullptr, qf, C, hits));
    TC_TRY(layernorm(h.t0, rl.norm2, h.t1, rows, 0, s));
    TC_TRY(linear(h.t1, C, rl.linear1, rows, C, F, 1, h.ffn_h, F, s));
    TC_TRY(linear(h.ffn_h, F, rl.linear2, rows, F, C, 0, h.t0, C, s, nullptr, h.t1, C));
    TC_TRY(layernorm(h.t0, rl.norm3, h.qf, rows, 0, s));
    qf = h.qf;
    // final_cls / final_reg, HEAD:592-600
    TC_TRY(linear(h.qf, C, rl.final_cls.l0, rows, C, C, 0, h.t0, C, s));
    TC_TRY(layernorm(h.t0, rl.final_cls.n1, h.t2, rows, 1, s));
    TC_TRY(linear(h.t2, C, rl.final_cls.l3, rows, C, C, 0, h.t0, C, s));
    TC_TRY(layernorm(h.t0, rl.final_cls.n4, h.t2, rows, 1, s));
    TC_TRY(linear(h.t2, C, rl.final_cls.l6, rows, C, ncls, 0, cls_out, ncls, s));
    TC_TRY(linear(h.qf, C, rl.final_reg.l0, rows, C, C, 1, h.t0, C, s));
    TC_TRY(linear(h.t0, C, rl.final_reg.l2, rows, C, C, 1, h.t2, C, s));
    TC_TRY(linear(h.t2, C, rl.final_reg.l4, rows, C, code, 0, h.reg_tmp, code, s));
    if (r == 0)
      TC_TRY(launch_box_add_ref(h.reg_tmp, code, h.addref, 3, h.addref + 2, 3, box_out, nullptr, rows, s));
    else
      TC_TRY(launch_box_add_ref(h.reg_tmp, code, box_prev, code, box_prev + 4, code, box_out, nullptr,
                                rows, s));
  }
  if (aux && aux->radar_hit_counts)
    TC_HIP(hipMemcpyAsync(aux->radar_hit_counts, h.hits, (size_t)w->num_radar_layers * rows * 4,
                          hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- training entry points (train.hip) ---------------------------------------
int tc_linear_gated_fwd(const float* x, const float* w, const float* b, const float* res,
                        const int* row_gate, float* y, int M, int K, int N, tc_stream_t stream) {
  tc_linear lw{w, b};
  return linear(x, K, lw, M, K, N, 0, y, N, as_stream(stream), nullptr, res, N, row_gate);
}

int tc_linear_bwd_data(const float* dy, const float* y_relu, const int* row_gate, const float* w,
                       const float* x_relu, float* dx, int M, int K, int N, float alpha,
                       int accumulate, tc_stream_t stream) {
  return launch_linear_bwd_data(dy, y_relu, row_gate, w, x_relu, dx, M, K, N, alpha, accumulate,
                                as_stream(stream));
}

int tc_linear_bwd_weight(const float* x, const float* dy, const float* y_relu, const int* row_gate,
                         float* dw, float* db, int M, int K, int N, float alpha,
                         tc_stream_t stream) {
  return launch_linear_bwd_weight(x, dy, y_relu, row_gate, dw, db, M, K, N, alpha, as_stream(stream));
}

int tc_add_layernorm_bwd(const float* a, const float* b, const float* gamma, const float* dy,
                         const float* y_relu, float* dz, float* dgamma, float* dbeta, int M, int C,
                         tc_stream_t stream) {
  TC_REQUIRE(C == 256, "layernorm_bwd: C=%d (256 supported)", C);
  return launch_ln256_bwd(a, b, gamma, dy, y_relu, dz, dgamma, dbeta, M, as_stream(stream));
}

int tc_radar_reference_l1(const float* ref, const float* pc_range, float* centre_xy, float* add_ref,
                          int M, tc_stream_t stream) {
  return launch_radar_ref_l1(ref, pc_range, centre_xy, add_ref, M, as_stream(stream));
}

int tc_box_add_ref_fwd(const float* reg_out, int code_size, const float* ref_xy, int ld_xy,
                       const float* ref_z, int ld_z, float* box, int M, tc_stream_t stream) {
  return launch_box_add_ref(reg_out, code_size, ref_xy, ld_xy, ref_z, ld_z, box, nullptr, M,
                            as_stream(stream));
}

int tc_box_add_ref_bwd(const float* d_box, int code_size, float* d_prev_box, int M,
                       tc_stream_t stream) {
  return launch_box_ref_bwd(d_box, code_size, d_prev_box, M, as_stream(stream));
}

static RadarAttnArgs radar_core_args(const float* qproj, float q_scale, const float* kv,
                                     const float* centre_xy, int ld_c, const float* box,
                                     int code_size, const float* radar_xy, int ld_xy, int B, int Q,
                                     int T, int C, int num_heads, int pad_mult, float radius_min,
                                     float radius_max) {
  RadarAttnArgs r;
  r.qproj = qproj; r.ldq = C; r.kv = kv; r.ldkv = 2 * C; r.centre_xy = centre_xy; r.ld_c = ld_c;
  r.box = box; r.code = code_size; r.radar_xy = radar_xy; r.ld_xy = ld_xy;
  r.B = B; r.Q = Q; r.T = T; r.C = C; r.H = num_heads; r.pad_mult = pad_mult;
  r.rmin = radius_min; r.rmax = radius_max; r.attn_out = nullptr; r.hit_counts = nullptr;
  r.qscale = q_scale;
  return r;
}

int tc_radar_attn_core_fwd(const float* qproj, float q_scale, const float* kv,
                           const float* centre_xy, int ld_c, const float* box, int code_size,
                           const float* radar_xy, int ld_xy, int B, int Q, int T, int C,
                           int num_heads, int pad_mult, float radius_min, float radius_max,
                           float* attn_out, int* hit_counts, float dropout_p,
                           unsigned long long dropout_seed, int dropout_site, tc_stream_t stream) {
  TC_REQUIRE(attn_out != nullptr && hit_counts != nullptr, "radar_attn_core: NULL output");
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_attn_core: dropout_p=%g", (double)dropout_p);
  RadarAttnArgs r = radar_core_args(qproj, q_scale, kv, centre_xy, ld_c, box, code_size, radar_xy,
                                    ld_xy, B, Q, T, C, num_heads, pad_mult, radius_min, radius_max);
  r.attn_out = attn_out; r.hit_counts = hit_counts;
  r.drop = make_drop(dropout_p, dropout_seed, (unsigned)dropout_site, 1500u);
  return launch_radar_attn(r, as_stream(stream));
}

int tc_radar_attn_core_bwd(const float* qproj, float q_scale, const float* kv,
                           const float* centre_xy, int ld_c, const float* box, int code_size,
                           const float* radar_xy, int ld_xy, int B, int Q, int T, int C,
                           int num_heads, int pad_mult, float radius_min, float radius_max,
                           const float* attn_out, const float* d_attn, float* dq, float* dkv,
                           float dropout_p, unsigned long long dropout_seed, int dropout_site,
                           tc_stream_t stream) {
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "radar_attn_core_bwd: dropout_p=%g", (double)dropout_p);
  RadarAttnArgs r = radar_core_args(qproj, 1.0f, kv, centre_xy, ld_c, box, code_size, radar_xy,
                                    ld_xy, B, Q, T, C, num_heads, pad_mult, radius_min, radius_max);
  r.attn_out = const_cast<float*>(attn_out);
  r.drop = make_drop(dropout_p, dropout_seed, (unsigned)dropout_site, 1500u);
  return launch_radar_attn_bwd(r, q_scale, d_attn, dq, dkv, as_stream(stream));
}

int tc_dropout(const float* x, int rows, int cols, float dropout_p, unsigned long long seed, int site,
               float* out, tc_stream_t stream) {
  TC_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "dropout: p=%g", (double)dropout_p);
  return launch_dropout(x, nullptr, nullptr, rows, cols, make_drop(dropout_p, seed, (unsigned)site, 1500u), out,
                        as_stream(stream));
}

int tc_sq_norm(const float* g, size_t n, float* out, tc_stream_t stream) {
  return launch_sqnorm(g, n, out, as_stream(stream));
}

int tc_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale,
                  float max_norm, const float* sq_norm, tc_stream_t stream) {
  return launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_norm,
                      sq_norm, as_stream(stream));
}

}  // extern "C"
