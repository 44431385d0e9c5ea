/*
 * transcar_hip.h -- C ABI of the MI355X-native (gfx950) TransCAR fusion decoder.
 *
 * The reference (pangsu0613/TransCAR) has no native code and no FFI: its
 * boundary is Python (`nn.Module.forward` signatures + state_dict keys, see
 * SURVEY.md section 8(b)).  This header is the boundary a maintainer binds
 * with ctypes from those Python call sites (INTEGRATION.md shows the stubs).
 * Every entry point names the reference interface it replaces; paths are
 * relative to the reference root:
 *   XFMR  = projects/mmdet3d_plugin/models/utils/detr3d_transformer.py
 *   HEAD  = projects/mmdet3d_plugin/models/dense_heads/detr3d_head.py
 *   CODER = projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py
 *   UTIL  = projects/mmdet3d_plugin/core/bbox/util.py
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 row-major arrays unless a
 *     parameter is documented as "host";
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues
 *     work on that stream (no allocation, no synchronisation, capturable in
 *     a hipGraph);
 *   - return value: 0 on success, otherwise a hipError_t (or -1 for an
 *     argument error); tc_last_error() gives the message;
 *   - nn.Linear weights are [out, in] row-major exactly as in the checkpoint.
 */
#ifndef TRANSCAR_HIP_H
#define TRANSCAR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_MAX_LEVELS 4
#define TC_MAX_LAYERS 8
#define TC_MAX_RADAR_LAYERS 3
#define TC_ABI_VERSION 12

typedef void* tc_stream_t;

/* ---- parameter views (pointers into the caller's state_dict tensors) ---- */
typedef struct { const float* w; const float* b; } tc_linear;   /* nn.Linear   */
typedef struct { const float* g; const float* b; } tc_lnorm;    /* nn.LayerNorm, eps 1e-5 */
/* Linear(3,C)-LN-ReLU-Linear(C,C)-LN-ReLU  (XFMR:285-292, HEAD:173-180) */
typedef struct { tc_linear l0; tc_lnorm n1; tc_linear l3; tc_lnorm n4; } tc_pos_encoder;
/* Linear-LN-ReLU x2 + Linear(C,num_classes)  (HEAD:200-206, HEAD:74-82) */
typedef struct { tc_linear l0; tc_lnorm n1; tc_linear l3; tc_lnorm n4; tc_linear l6; } tc_cls_branch;
/* Linear-ReLU x2 + Linear(C,code_size)  (HEAD:208-213, HEAD:84-90) */
typedef struct { tc_linear l0; tc_linear l2; tc_linear l4; } tc_reg_branch;
/* torch.nn.MultiheadAttention: packed in_proj [3C,C] rows (Wq;Wk;Wv) + out_proj */
typedef struct { tc_linear in_proj; tc_linear out_proj; } tc_mha;

/* one DetrTransformerDecoderLayer (CFG:65-82) + its reg_branches[lid] (XFMR:191) */
typedef struct {
  tc_mha self_attn;                 /* attentions.0.attn                      */
  tc_lnorm norm0;
  tc_linear attention_weights;      /* attentions.1.attention_weights [N*L,C] */
  tc_linear output_proj;            /* attentions.1.output_proj               */
  tc_pos_encoder position_encoder;  /* attentions.1.position_encoder          */
  tc_lnorm norm1;
  tc_linear ffn0;                   /* ffns.0.layers.0.0 [F,C]                */
  tc_linear ffn1;                   /* ffns.0.layers.1   [C,F]                */
  tc_lnorm norm2;
  tc_reg_branch reg;                /* reg_branches.{lid}                     */
  /* Set by tc_head_pack_weights in the packed view only (0 in the caller's struct): every packed
   * linear weight has a second copy, laid out for the 16-row tiles' 16x16x4 MFMA, this many FLOATS
   * behind the first (the same distance for all weights of a packed buffer). */
  size_t packed16_delta;
} tc_decoder_layer;

/* one radar fusion layer (HEAD:538-611 / :613-668 / :670-729) */
typedef struct {
  tc_mha attn;                      /* rf_multihead_attn{,2,3}                */
  tc_lnorm norm2;                   /* rf_norm2{,_2,_3}                       */
  tc_linear linear1;                /* rf_linear1{,_2,_3} [F,C]               */
  tc_linear linear2;                /* rf_linear2{,_2,_3} [C,F]               */
  tc_lnorm norm3;                   /* rf_norm3{,_2,_3}                       */
  tc_cls_branch final_cls;          /* final_cls{,2,3}                        */
  tc_reg_branch final_reg;          /* final_reg{,2,3}                        */
  float radius_min, radius_max;     /* clamp: (1,2),(1,2),(0.5,1)  HEAD:567,635,693 */
  size_t packed16_delta;            /* packed view only, as in tc_decoder_layer */
} tc_radar_layer;

/* All parameters Detr3DHead.forward reads (HEAD:43-238), in eval mode. */
typedef struct {
  int abi_version;                  /* TC_ABI_VERSION                         */
  int num_query, embed_dims, num_heads, ffn_dims, num_layers;
  int num_cams, num_levels, num_classes, code_size;
  int radar_in_dims, num_radar_layers, num_radar_tokens_ref; /* 36, 3, 1500   */
  float pc_range[6];
  const float* query_embedding;     /* [Q, 2C]: (query_pos | query) XFMR:119  */
  tc_linear reference_points;       /* transformer.reference_points [3,C]     */
  tc_decoder_layer layers[TC_MAX_LAYERS];
  tc_pos_encoder radar_position_encoder;
  tc_linear radar_feat0, radar_feat2, radar_feat4;  /* radar_feat_encoder.{0,2,4} */
  tc_radar_layer radar[TC_MAX_RADAR_LAYERS];
  /* Set by tc_head_pack_weights in the packed view only (leave NULL in the caller's
   * struct).  Layer 0's self-attention sees nothing but the learned query embedding
   * (XFMR:119-123, config CFG:65-72: q = k = query + query_pos, v = query), so its
   * initial reference points and its attention output do not depend on the frame:
   * they are evaluated once per checkpoint, not once per forward. */
  const float* l0_init_reference;   /* [Q,3]  sigmoid(reference_points(query_pos))      */
  const float* l0_attn_out;         /* [Q,C]  softmax(q k^T / sqrt(d)) v of layer 0     */
  size_t packed16_delta;            /* packed view only, as in tc_decoder_layer         */
} tc_head_weights;

/* multi-view FPN feature maps, channels-last: level l is [B*num_cams, H, W, C] */
typedef struct {
  int num_levels;
  const float* data[TC_MAX_LEVELS];
  int H[TC_MAX_LEVELS];
  int W[TC_MAX_LEVELS];
} tc_feats_nhwc;

/* optional intermediate outputs of tc_head_forward (NULL pointers are skipped) */
typedef struct {
  float* inter_states;        /* hs            [L, B, Q, C]  (HEAD:274 layout)        */
  float* init_reference;      /*               [B, Q, 3]                              */
  float* inter_references;    /*               [L, B, Q, 3]                           */
  int*   radar_hit_counts;    /* per radar layer [R, B, Q]: radar tokens inside the gate */
  float* last_box;            /* `tmp` of the last decoder level, xy/z in metres
                                 (HEAD:287-293) [B, Q, code_size]: gate geometry of radar layer 1 */
  unsigned long long* sample_pairs; /* [1]: += number of visible (query,cam) pairs sampled */
} tc_head_aux;

/* per-call options of tc_head_forward (NULL = all defaults = the reference's eval forward) */
typedef struct {
  int chain_tile_rows;      /* rows of a workgroup's tile in the fused row chains: 0 = automatic
                               (4 up to 1024 rows per launch, 8 up to 2048, 16 up to 4096, 32 beyond -- 16 when
                               matrix_path is TC_MATRIX_F32), 4, 8, 16 or 32 (32: f16x2 matrix path only; one
                               workgroup of 8 waves per CU, activations held in LDS as two f16 planes) */
  int unfused;              /* 1: operator-by-operator launch sequence (~160 launches), the
                               in-tree cross-check of the fused chains */
  int last_level_cls_only;  /* 1 (inference opt-in): final_cls / final_cls2 are not evaluated --
                               get_bboxes decodes level 3 only (HEAD:1003-1023) and levels 1-2
                               feed only their BOX to the next gate; all_cls_scores[0:2] are
                               left untouched.  0 keeps the reference's [3,B,Q,10] output */
  int reuse_radar_kv;       /* tc_radar_fusion_fwd only: 1 = skip the radar encoders, the K/V
                               projections of the previous call in the same workspace are reused
                               (timing the fusion chain on its own) */
  /* train-mode statistics of the FROZEN decoder: tools/train.py:245-252 only clears
   * requires_grad, so the dropout (p = 0.1, CFG:68-80, XFMR:378) of the decoder layers stays
   * active during training.  0 = eval (off).  Masks are counter-based (common.hpp drop_keep):
   * site = 16 + 8 * layer + {0 attention probabilities, 1 self-attention output, 2 cross-
   * attention output, 3 FFN hidden, 4 FFN output}. */
  float decoder_dropout_p;
  int radar_row_order;      /* the radar chain's row order: 0 = automatic, 1 = the queries' own order, 2 = queries
                               with a radar return inside their first gate first, so that the row tiles
                               without any skip the gated attention part (two small extra launches: worth it
                               beyond one frame per launch, which is what automatic does).  Outputs are
                               bit-identical either way */
  unsigned long long dropout_seed;
  int phase;                /* the fused forward in two calls (same arguments, same workspace, same stream):
                               0 = the whole forward; 1 = only what does not read `radar_tokens` -- the prologue and
                               decoder layers 0 .. L-3 (the radar encoders then ride in the launches of the LAST two
                               decoder layers instead of the first two: their K/V feed only the fusion stack);
                               2 = the rest.  Between 1 and 2
                               the caller builds the tokens (host packing, H2D, tc_radar_build_tokens*) while the
                               device already runs the decoder: Detr3DHead.forward does exactly that */
  int matrix_path;          /* arithmetic of the 16-row tiles' linear steps (frames batched into one launch; 4- and
                               8-row tiles always run the exact fp32 MFMA): TC_MATRIX_AUTO (0) = TC_MATRIX_F16X2,
                               the operands as two f16 planes each, three v_mfma_f32_16x16x32_f16 per 32 k with
                               fp32 accumulation -- fp32-accurate (closer to fp64 than the fp32 FMA chain,
                               profiles/r4_split_mfma_probe.txt), activations of a linear step must stay below
                               4.19e6 in magnitude (an overflow shows as inf / NaN, never silently);
                               TC_MATRIX_F32 (1) = v_mfma_f32_16x16x4_f32, exact fp32 FMA chains */
  unsigned long long dropout_seed_stride; /* decoder dropout with B > 1: 0 = one mask index space over the batch (seed
                               = dropout_seed); != 0: sample b draws the masks it would draw LAUNCHED ALONE with the seed
                               dropout_seed + b * stride (element indices relative to the sample) -- a batch of
                               look-ahead frames through the frozen decoder is then bit-identical, frame by frame, to
                               the frames launched one at a time (FusionTrainer(prefetch_depth=...)) */
  int* range_status;        /* ABI 10, device pointer or NULL: a sticky word the kernels of the f16x2 matrix path OR 1
                               into when a linear step produces a non-finite value -- which is what an operand beyond
                               the f16 planes' range (|activation| >= 65504 * 2^6 = 4.19e6, |weight| >= 65504) turns
                               into: inf / NaN in the step's result of the affected rows.  Later steps may turn those
                               into finite-looking outputs again (a sigmoid, the clamp of a reference point, a
                               row gate of 0): the FLAG is the signal, not the outputs.  The caller clears it;
                               Detr3DHead reads it where it already reads results back (get_bboxes) and, on
                               TC_MATRIX_AUTO, runs its next forwards on TC_MATRIX_F32 */
  int cam_pregather;        /* ABI 12 (round 6), opt-in.  feature_sampling (XFMR:381-422) of decoder layer l needs only the
                               layer's reference points, which are final when layer l - 1 ends -- and the attention core of
                               layer l runs in between.  1: launches that take the f16x2 attention core (16- / 32-row
                               tiles) carry the gather as extra workgroups of that launch (projection, the 16 taps of every
                               visible (query, camera) pair, the four bilinear level values: 4 KiB per pair into
                               cam_pregather_ws) and the chain's sampling step only weighs and sums them (XFMR:367-373):
                               the same products in the same order, bit-identical outputs.  Measured (DESIGN.md section
                               5, round 6): decoder chain 100 -> 89.5 us per nine frames, attention-core launch 42 -> 50.4
                               us; one launch sequence at a time 1 % faster, three in flight 2.8 % SLOWER (the launch
                               moves 170 MB more) -- hence 0 = the chain gathers itself (default; layer 0 and the 4- /
                               8-row launches always do) */
  void* cam_pregather_ws;   /* cam_pregather = 1: device scratch of tc_cam_pregather_workspace_bytes(w, B) bytes, owned by
                               the caller for the duration of the call (one per stream in flight) */
  size_t cam_pregather_bytes;
} tc_head_options;
#define TC_MATRIX_AUTO 0
#define TC_MATRIX_F32 1
#define TC_MATRIX_F16X2 2
/* tc_decoder_layer_tail_fwd's tile_rows carries the matrix path in bits 8..9: rows | (TC_MATRIX_* << 8) */
#define TC_TILE_ROWS(v) ((v) & 0xFF)
#define TC_TILE_MATRIX(v) (((v) >> 8) & 3)

/* ---- library ---- */
int tc_abi_version(void);
const char* tc_last_error(void);
/* number of HIP devices visible; does not initialise a context */
int tc_device_count(void);

/* ---- layout ---- */
/* FPN handoff (DET:62-66 hands the head NCHW [B,N,C,H,W]): NCHW -> NHWC. */
int tc_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W,
                    tc_stream_t stream);
/* The same for all FPN levels of a frame in ONE launch.  src / dst / H / W are HOST
 * arrays of num_levels (<= TC_MAX_LEVELS) entries; level l is [n_img, C, H[l], W[l]]. */
int tc_nchw_to_nhwc_levels(const float* const* src, float* const* dst, int num_levels,
                           int n_img, int C, const int* H, const int* W, tc_stream_t stream);

/* ---- radar ingest on the device (HEAD:301-536; numpy on the main thread in the reference) ----
 * Raw devkit rows of ONE sample -> the [T,36] float32 token matrix tc_head_forward reads.
 *   raw        device [N,18] float64, point-major: the 18 fields of RadarPointCloud (x y z dyn_prop
 *              id rcs vx vy vx_comp vy_comp is_quality_valid ambig_state x_rms y_rms invalid_state
 *              pdh0 vx_rms vy_rms), the radars concatenated in the reference's channel order
 *              (RADAR_FRONT, _FRONT_LEFT, _FRONT_RIGHT, _BACK_LEFT, _BACK_RIGHT, HEAD:310-451)
 *   times      device [N] float64 sweep time lags (from_file_multisweep's second result)
 *   chan_start HOST [num_chan+1]: rows chan_start[c] .. chan_start[c+1] belong to radar c
 *   radar_rot  HOST [num_chan,9] row-major rotation radar -> ego of each radar's calibrated sensor
 *   lidar_rot  HOST [9] rotation lidar -> ego (applied transposed, HEAD:317-327)
 *   point_range HOST [6] float64 (HEAD:304: -51.2 -51.2 -5 51.2 51.2 3), strict inequalities in
 *              float64 as in the reference (a point at exactly 51.2 is dropped)
 *   tokens     device [T,36]: kept points in order, then rows of 500.0 (HEAD:523-530)
 *   count      device [1] (may be NULL): number of points the range filter kept.  With T < 1500
 *              row T-1 is ALWAYS a 500.0 pad row (it carries pad_mult): at most T-1 points are
 *              written, count > T-1 means the frame did not fit (use T = 1500 for such frames;
 *              the reference keeps the first 1500)
 * With tokens [T,36] the head is called with pad_mult = 1500 - T + 1. */
int tc_radar_build_tokens(const double* raw, const double* times, const int* chan_start, int num_chan,
                          const double* radar_rot, const double* lidar_rot, const double* point_range,
                          float* tokens, int T, int* count, tc_stream_t stream);

/* The same for P samples in ONE launch with the per-sample parameters in DEVICE memory, so that a captured
 * hipGraph can replay it on new frames (FramePipeline: the first node of a lane's graph; Detr3DHead.forward
 * with raw sweeps in img_metas).  Replaces the same reference lines (HEAD:301-536).
 *   raw    device [P, cap, 18] float64, sample b's points at rows [0, chan_start[num_chan]) of its slab
 *   times  device [P, cap] float64
 *   desc   device [P]: chan_start / num_chan / rotations / point_range of each sample (clamped to [0, cap])
 *   tokens device [P, T, 36]; count device [P] (may be NULL) -- semantics as tc_radar_build_tokens */
#define TC_MAX_RADAR_CHANNELS 8
typedef struct {
  int chan_start[TC_MAX_RADAR_CHANNELS + 1];
  int num_chan;
  double radar_rot[TC_MAX_RADAR_CHANNELS * 9];
  double lidar_rot[9];
  double point_range[6];
} tc_radar_frame_desc;
int tc_radar_build_tokens_batch(const double* raw, const double* times, const tc_radar_frame_desc* desc,
                                int P, int cap, float* tokens, int T, int* count, tc_stream_t stream);

/* ---- operators, one per reference call site ---- */

/* nn.Linear (+ optional fused pieces).  y[M,N] = act((x (+x2)) W^T + b) (+res)
 * act: 0 none, 1 ReLU, 2 sigmoid (XFMR:122-123).  x2/res may be NULL. */
int tc_linear_fwd(const float* x, const float* x2, const float* w, const float* b,
                  const float* res, float* y, int M, int K, int N, int act,
                  tc_stream_t stream);

/* y = LayerNorm(a (+ b))*g + beta, optional ReLU; eps 1e-5 (mmcv `norms[i]`,
 * HEAD:583,586).  b may be NULL. */
int tc_add_layernorm_fwd(const float* a, const float* b, const float* gamma,
                         const float* beta, float* y, int M, int C, int relu,
                         tc_stream_t stream);

/* Iterative reference-point refinement of Detr3DTransformerDecoder.forward
 * (XFMR:195-203): new_ref = sigmoid(reg_out[...,{0,1,4}] + inverse_sigmoid(ref)).
 * reg_out [M,code_size], ref/new_ref [M,3]. */
int tc_refine_reference_fwd(const float* reg_out, int code_size, const float* ref,
                            float* new_ref, int M, tc_stream_t stream);

/* feature_sampling + the weighting/reduction of Detr3DCrossAtten.forward
 * (XFMR:365-373, XFMR:381-422): projection of the reference points into every
 * camera, visibility mask, bilinear 4-level sampling (align_corners=False,
 * zeros padding), NaN->0, sigmoid(attn_logits)*mask weighting, sum over
 * (cam, level).  out [B,Q,C];  vis_mask [B,Q,num_cams] (bytes) may be NULL. */
int tc_cam_sample_fuse_fwd(const tc_feats_nhwc* feats, int B, int Q, int C, int num_cams,
                           const float* lidar2img /*[B,N,4,4]*/, const float* ref /*[B,Q,3]*/,
                           const float* attn_logits /*[B,Q,N*L]*/,
                           const float* pc_range /*host[6]*/, float img_h, float img_w,
                           float* out, unsigned char* vis_mask,
                           unsigned long long* pair_counter /*may be NULL*/,
                           tc_stream_t stream);

/* Detr3DCrossAtten.forward (XFMR:302-378), eval mode: returns
 * output_proj(sampled) + query + position_encoder(inverse_sigmoid(ref)).
 * query/query_pos/out are [B,Q,C] (the reference's [Q,B,C] with B folded
 * first).  workspace: tc_cross_atten_workspace_bytes(). */
size_t tc_cross_atten_workspace_bytes(int B, int Q, int C, int num_cams, int num_levels);
int tc_cross_atten_fwd(const tc_linear* attention_weights, const tc_linear* output_proj,
                       const tc_pos_encoder* position_encoder,
                       const tc_feats_nhwc* feats, int B, int Q, int C, int num_cams,
                       const float* query, const float* query_pos,
                       const float* lidar2img, const float* ref,
                       const float* pc_range /*host[6]*/, float img_h, float img_w,
                       float* out, void* workspace, size_t workspace_bytes,
                       tc_stream_t stream);

/* mmcv MultiheadAttention wrapper as used for decoder self-attention
 * (CFG:68-72): out = x + out_proj(MHA(q=k=x+pos, v=x)).
 * x/pos/out [B,Q,C]. */
size_t tc_self_attn_workspace_bytes(int B, int Q, int C);
int tc_self_attn_fwd(const tc_mha* w, const float* x, const float* pos, float* out,
                     int B, int Q, int C, int num_heads,
                     void* workspace, size_t workspace_bytes, tc_stream_t stream);

/* One decoder layer AFTER its attention core, as one fused launch (the row-chain
 * kernel of tc_head_forward; mmcv layer order CFG:81-82, XFMR:346-378, 190-203):
 *   x1 = norm0(x_in + out_proj(attn_o)); camera cross-attention (attention
 *   weights, sampling, output_proj, position encoder); norm1; FFN; norm2 -> hs;
 *   reg branch -> ref_out = refined reference points; next layer's q,k (qk) and
 *   transposed v (vt) when next_in_proj != NULL.
 * `layer` / `next_in_proj` must come from a tc_head_pack_weights view (packed
 * weights).  attn_o, x_in, hs [B*Q,C]; query_embedding [Q,2C]; ref_in/ref_out
 * [B*Q,3]; qk [B*Q,2C]; vt [B,C,qpad]. */
int tc_decoder_layer_tail_fwd(const tc_decoder_layer* layer, const tc_linear* next_in_proj,
                              const tc_feats_nhwc* feats, int B, int Q, int num_cams,
                              int code_size, const float* attn_o, const float* x_in,
                              const float* query_embedding, const float* lidar2img,
                              const float* ref_in, const float* pc_range /*host[6]*/,
                              float img_h, float img_w, float* hs, float* ref_out,
                              float* qk, float* vt, int qpad,
                              int tile_rows /* 0 automatic, 4, 8, 16; | TC_MATRIX_* << 8 */, tc_stream_t stream);

/* The attention core of the above on already projected operands (the kernel
 * the roofline is quoted on): out = softmax(q k^T) v per (batch, head).
 *   q, k [B*Q, ld] token-major, head h at columns h*32.. (q pre-scaled by
 *   log2(e)/sqrt(32): the kernel exponentiates with 2^x);
 *   vt [B, H*32, ldt] = V transposed, ldt >= round_up(Q,16);
 *   out [B*Q, ldo]. */
int tc_sdpa_fwd(const float* q, const float* k, int ld, const float* vt, int ldt,
                float* out, int ldo, int B, int Q, int num_heads, tc_stream_t stream);
/* The same core on the f16 matrix cores with fp32 accuracy (round 4; what tc_head_forward runs in launches with
 * 16-row tiles unless options.matrix_path = TC_MATRIX_F32, with the train-mode dropout on the probabilities too): q | k rows [B*Q, 2C]
 * token-major (q at columns 0..C-1, pre-scaled as above, k at C..2C-1), vt as above.  Every operand is used as two
 * f16 planes (hi = f16(x), lo = f16(x - hi); built in registers / LDS inside the kernel), every product is three
 * v_mfma_f32_16x16x32_f16 with fp32 accumulation.  |q|, |k|, |v| must stay below 65 504; an element's low plane is
 * exact to 2^-25 ABSOLUTE (f16 subnormals), i.e. the result is fp32-accurate for operands of magnitude >~ 2^-3 and
 * absolutely accurate to ~1e-7 x |the other operand| below that.  `workspace` is unused (may be null): the query
 * returns 0. */
size_t tc_sdpa_f16x2_workspace_bytes(int B, int Q, int num_heads);
int tc_sdpa_fwd_f16x2(const float* qk, const float* vt, int ldt, float* out, int ldo, int B, int Q, int num_heads,
                      void* workspace, size_t workspace_bytes, tc_stream_t stream);

/* Distance-gated radar cross-attention, one fusion layer's attention step
 * (HEAD:549-581 / :619-653 / :675-711): three-circle gate around
 * (centre, front, rear), masked nn.MultiheadAttention of the gated queries
 * over the radar tokens, residual add on the rows that have at least one hit.
 *   query      [B,Q,C]   query_feat before the layer
 *   centre_xy  [B,Q,2]   metres;  box [B,Q,code]: log-length at [3], sin/cos at [6],[7]
 *   radar_feat [B,T,C]   encoded tokens; radar_xy [B,T,2]
 *   pad_mult   multiplicity of the LAST token (the reference always attends
 *              over 1500 tokens, HEAD:526; identical pad tokens beyond T are
 *              folded into token T-1)
 *   out        [B,Q,C] = query (+ attention output on hit rows)
 *   hit_counts [B,Q] (may be NULL) */
size_t tc_radar_xattn_workspace_bytes(int B, int Q, int T, int C);
int tc_radar_gated_xattn_fwd(const tc_mha* w, const float* query, const float* centre_xy,
                             const float* box, int code_size,
                             const float* radar_feat, const float* radar_xy,
                             int B, int Q, int T, int C, int num_heads, int pad_mult,
                             float radius_min, float radius_max,
                             float* out, int* hit_counts,
                             void* workspace, size_t workspace_bytes, tc_stream_t stream);

/* The radar part of Detr3DHead.forward on its own (HEAD:531-729), from given decoder
 * outputs: radar encoders + K/V projections, then fusion layers [first_layer,
 * first_layer + num_layers) -- the same fused chain kernels tc_head_forward runs.
 *   hs_last   [B,Q,C]    query_feat entering layer `first_layer` (HEAD:539: hs[-1] for layer 0)
 *   ref_last  [B,Q,3]    normalised reference points of the last decoder layer; read only
 *                        when first_layer == 0 (gate centre of layer 1, HEAD:543-547, z quirk
 *                        of HEAD:596-598); may be NULL otherwise
 *   prev_box  [B,Q,code] box whose length / sin / cos shape the gate: the decoder's last `tmp`
 *                        (HEAD:287-293) for first_layer == 0, else the previous fusion layer's
 *                        all_bbox_preds slice, which is also the gate centre (HEAD:615-617)
 *   all_cls_scores / all_bbox_preds [R,B,Q,*]: slices first_layer.. are written
 *   hit_counts [R,B,Q] (may be NULL): slices first_layer.. are written
 *   workspace: tc_head_workspace_bytes(packed_view, B, T) bytes.
 * Used by the per-layer teacher-forced parity tests (tests/test_gpu_teacher_forced.py). */
int tc_radar_fusion_fwd(const tc_head_weights* packed_view, const float* hs_last,
                        const float* ref_last, const float* prev_box, const float* radar_tokens,
                        int B, int T, int pad_mult, int first_layer, int num_layers,
                        float* all_cls_scores, float* all_bbox_preds, int* hit_counts,
                        const tc_head_options* options, void* workspace, size_t workspace_bytes,
                        tc_stream_t stream);

/* Self-check of the gate predicate (HEAD:568-571: cdist < radius for any of three circles).  The
 * kernels compare SQUARED distances against t* = the smallest float whose correctly-rounded square root
 * reaches the radius -- sqrtf is monotone, so that is the reference's comparison, without a square root
 * per (query, token, circle).  This entry evaluates both forms on n_radii radii of the clamp range
 * [0.5, 2] (the clamp values themselves included) x (every float within 256 ulps of radius^2 + 4096 random
 * ones) and adds the number of disagreements to *mismatches (device, 8 bytes, zeroed by the caller). */
int tc_radar_gate_selfcheck(int n_radii, unsigned long long seed, unsigned long long* mismatches,
                            tc_stream_t stream);

/* Self-check of the row-local arithmetic of the 16- / 32-row chains (round 6; no reference line: an implementation
 * invariant).  Three pieces were rewritten for fewer vector instructions and must give the same BITS as their plain forms:
 * [0] LayerNorm of a wave's four rows through one packed reduction tree (against one row at a time), [1] the split of a
 * value into two f16 planes, [2] a plane pair back to fp32.  n_blocks workgroups draw their own rows / values (magnitudes
 * over 40 binades, constant rows, signed zeros, subnormals, f16 overflow, infinities, NaN); the numbers of differing results
 * are ADDED to mismatches[0..2] (device, 24 bytes, zeroed by the caller). */
int tc_rowops_selfcheck(int n_blocks, unsigned long long seed, unsigned long long* mismatches, tc_stream_t stream);

/* NMSFreeCoder.decode_single + get_bboxes z-shift (CODER:39-90, UTIL:26-52,
 * HEAD:1018): sigmoid, top-`max_num` of Q*num_classes scores, gather,
 * denormalise, centre-range mask, z -= h/2.
 *   boxes [B,max_num,9], scores [B,max_num], labels [B,max_num] (int),
 *   valid [B,max_num] (bytes: 1 = inside post_center_range) */
size_t tc_box_decode_workspace_bytes(int B, int Q, int num_classes);
int tc_box_decode_topk(const float* cls_scores /*[B,Q,num_classes]*/,
                       const float* bbox_preds /*[B,Q,code]*/, int B, int Q,
                       int num_classes, int code_size, int max_num,
                       const float* post_center_range /*host[6]*/,
                       float* boxes, float* scores, int* labels, unsigned char* valid,
                       void* workspace, size_t workspace_bytes, tc_stream_t stream);
/* The same selection, returned the way NMSFreeCoder.decode_single returns it (CODER:62-84): only the rows inside
 * post_center_range (tested on the un-shifted centre) and, with use_threshold, with score > score_threshold --
 * compacted in descending score order -- and their number.  The caller reads kept_count[b] (one small D2H for the
 * batch) and takes the first kept_count[b] rows: no mask select, no gather.
 *   kept_boxes [B,max_num,9], kept_scores [B,max_num], kept_labels [B,max_num] (64-bit: torch.long, CODER:54),
 *   kept_count [B]; rows kept_count[b] .. max_num-1 are left unwritten.
 *   z_shift 1: z -= h/2 (get_bboxes, HEAD:1018); 0: gravity-centre z as the coder itself returns it. */
int tc_box_decode_kept(const float* cls_scores, const float* bbox_preds, int B, int Q, int num_classes,
                       int code_size, int max_num, const float* post_center_range /*host[6]*/,
                       float score_threshold, int use_threshold, int z_shift,
                       float* kept_boxes, float* kept_scores, long long* kept_labels, int* kept_count,
                       tc_stream_t stream);

/* ---- the whole hot path: Detr3DHead.forward (HEAD:248-740), eval mode ----
 *   radar_tokens [B,T,36]: rows of the 36 hand-built features (HEAD:499-510),
 *                padded with 500.0 rows (HEAD:527); T <= 1500
 *   pad_mult     see tc_radar_gated_xattn_fwd
 *   all_cls_scores / all_bbox_preds [3,B,Q,10] */
size_t tc_head_workspace_bytes(const tc_head_weights* w, int B, int T);
/* scratch of tc_head_options.cam_pregather = 1 (ABI 12): B * Q * (num_cams * num_levels * C floats + one int) */
size_t tc_cam_pregather_workspace_bytes(const tc_head_weights* w, int B);
/* One-time re-layout of the nn.Linear weights into MFMA fragment order for the
 * fused row-chain kernels (1 KiB-coalesced weight streaming): call once per
 * checkpoint load / after an optimizer step changes the weights.  `packed` is a
 * device buffer of tc_head_packed_bytes(w) bytes that must outlive
 * `packed_view`, a copy of `w` whose linear-weight pointers point into it. */
size_t tc_head_packed_bytes(const tc_head_weights* w);
int tc_head_pack_weights(const tc_head_weights* w, void* packed, size_t packed_bytes,
                         tc_head_weights* packed_view, tc_stream_t stream);
/* After an optimizer step under the reference's freeze list (tools/train.py:245-252):
 * re-pack only the radar encoder / fusion layer / final_* weights into the existing
 * packed buffer; the frozen decoder's weights and the layer-0 constants stay. */
int tc_head_repack_trainable(const tc_head_weights* w, tc_head_weights* packed_view,
                             tc_stream_t stream);
/* The same with a choice of copies: 1 = only the 4x4x1 layout (what the 4- and 8-row chains of a training
 * iteration read, tc_radar_train_fwd_fused), 3 = both layouts.  One launch for all trainable weights. */
int tc_head_repack_trainable_ex(const tc_head_weights* w, tc_head_weights* packed_view, int copies,
                                tc_stream_t stream);
/* packed_view == NULL selects the operator-by-operator launch sequence (~160
 * launches instead of 16); results agree to fp32 rounding. */
int tc_head_forward(const tc_head_weights* w, const tc_head_weights* packed_view,
                    const tc_feats_nhwc* feats, int B,
                    const float* lidar2img /*[B,N,4,4]*/, float img_h, float img_w,
                    const float* radar_tokens, int T, int pad_mult,
                    float* all_cls_scores, float* all_bbox_preds,
                    const tc_head_aux* aux /*may be NULL*/,
                    const tc_head_options* options /*may be NULL*/,
                    void* workspace, size_t workspace_bytes, tc_stream_t stream);


/* ======================================================================
 * Training.  tools/train.py:245-252
 * freezes the DETR3D decoder, so one iteration differentiates the radar
 * encoders, the three gated radar fusion layers and final_cls* / final_reg*
 * (HEAD:531-729).  The frozen decoder runs through tc_head_forward (aux gives
 * hs, references, last_box); the trainable stack runs operator by operator so
 * that the host autograd (the reference's own boundary for training: plain
 * nn.Modules, `loss.backward()`) can save activations, with these backward
 * entry points behind torch.autograd.Function (transcar_amd/autograd_ops.py).
 * Gradient outputs named "+=" accumulate with fp32 atomics: zero them first.
 * ====================================================================== */

/* y = res + (row_gate[m] > 0 ? x W^T + b : 0): the row-subset update of
 * HEAD:581 (`query_feat[nan_row_index] += tgt2`).  row_gate [M] may be NULL. */
int tc_linear_gated_fwd(const float* x, const float* w, const float* b, const float* res,
                        const int* row_gate, float* y, int M, int K, int N, tc_stream_t stream);

/* nn.Linear backward.  dY~ = dY masked by the layer's own ReLU (y_relu = its
 * saved output, NULL if none) and by row_gate (NULL if none).
 *   data:   dx[M,K] (=|+=) alpha * dY~ W     (x_relu: zero dx where the saved
 *           ReLU output that fed this layer is <= 0; may be NULL)
 *   weight: dw[N,K] += alpha * dY~^T x ; db[N] += alpha-less column sums of dY~
 *           (db may be NULL) */
int tc_linear_bwd_data(const float* dy, const float* y_relu, const int* row_gate,
                       const float* w, const float* x_relu, float* dx, int M, int K, int N,
                       float alpha, int accumulate, tc_stream_t stream);
int tc_linear_bwd_weight(const float* x, const float* dy, const float* y_relu,
                         const int* row_gate, float* dw, float* db, int M, int K, int N,
                         float alpha, tc_stream_t stream);

/* backward of tc_add_layernorm_fwd: y = LN(a (+b)) (ReLU if y_relu != NULL, the
 * saved output).  dz [M,C] = grad of a (and of b); dgamma/dbeta [C] += . C = 256 */
int tc_add_layernorm_bwd(const float* a, const float* b, const float* gamma, const float* dy,
                         const float* y_relu, float* dz, float* dgamma, float* dbeta, int M,
                         int C, tc_stream_t stream);

/* HEAD:544-547 / 596-598: from the normalised last reference [M,3]:
 * centre_xy [M,2] in metres and the add-reference (x_m, y_m, z NORMALISED). */
int tc_radar_reference_l1(const float* ref, const float* pc_range /*host[6]*/, float* centre_xy,
                          float* add_ref, int M, tc_stream_t stream);

/* HEAD:599-600 (661-662, 719-720): box = reg_out; box[0:2] += ref_xy; box[4] += ref_z.
 * ref_xy at ref_xy[m*ld_xy + {0,1}], ref_z at ref_z[m*ld_z].  Backward into the
 * previous layer's box (layers 2, 3: the reference IS the previous box):
 * d_prev[{0,1,4}] += d_box[{0,1,4}]. */
int tc_box_add_ref_fwd(const float* reg_out, int code_size, const float* ref_xy, int ld_xy,
                       const float* ref_z, int ld_z, float* box, int M, tc_stream_t stream);
int tc_box_add_ref_bwd(const float* d_box, int code_size, float* d_prev_box, int M,
                       tc_stream_t stream);

/* The gated attention core of tc_radar_gated_xattn_fwd on projected operands:
 *   qproj [B*Q,C] (unscaled: q_scale = 1/sqrt(C/H) is applied inside),
 *   kv [B*T,2C] = K | V, centre at centre_xy[m*ld_c + {0,1}], box [B*Q,code],
 *   token xy at radar_xy[(b*T+t)*ld_xy + {0,1}]
 *   -> attn_out [B*Q,C] (zero rows without a hit), hit_counts [B*Q].
 * Backward: dq [B*Q,C] (=), dkv [B*T,2C] (+=). */
int tc_radar_attn_core_fwd(const float* qproj, float q_scale, const float* kv,
                           const float* centre_xy, int ld_c, const float* box, int code_size,
                           const float* radar_xy, int ld_xy, int B, int Q, int T, int C,
                           int num_heads, int pad_mult, float radius_min, float radius_max,
                           float* attn_out, int* hit_counts, float dropout_p,
                           unsigned long long dropout_seed, int dropout_site, tc_stream_t stream);
int tc_radar_attn_core_bwd(const float* qproj, float q_scale, const float* kv,
                           const float* centre_xy, int ld_c, const float* box, int code_size,
                           const float* radar_xy, int ld_xy, int B, int Q, int T, int C,
                           int num_heads, int pad_mult, float radius_min, float radius_max,
                           const float* attn_out, const float* d_attn, float* dq, float* dkv,
                           float dropout_p, unsigned long long dropout_seed, int dropout_site,
                           tc_stream_t stream);
/* out[i] = keep(seed, site, i) ? x[i] / (1 - p) : 0 (nn.Dropout in train mode with the counter-based
 * masks described at tc_radar_train_fwd; in place allowed; applied to a gradient with the same
 * (seed, site) it is the backward).  dropout_p / seed / site of the attention core: the mask on the
 * attention probabilities (nn.MultiheadAttention(dropout=0.1), HEAD:129); p = 0 turns it off. */
int tc_dropout(const float* x, int rows, int cols, float dropout_p, unsigned long long seed, int site,
               float* out, tc_stream_t stream);

/* Optimizer on the flat fp32 bucket of the trainable parameters (one RCCL
 * all-reduce).  tc_sq_norm: out is TC_SQ_NORM_PARTIALS floats (zeroed by the caller);
 * out[b] += the sum of g^2 over the elements workgroup b owns -- no atomics, so the ranks of a job get the same
 * bits from the same all-reduced bucket; sum(out[]) = sum g^2.  tc_adamw_step: torch.optim.AdamW step with
 * mmcv's grad clip (CFG:214 max_norm=35; coef = max_norm/(norm+1e-6) if < 1)
 * evaluated on the device from sq_norm[0 .. TC_SQ_NORM_PARTIALS), added up in index order (NULL or
 * max_norm <= 0: no clip); g is multiplied by grad_scale first (1/world_size after a SUM all-reduce). */
#define TC_SQ_NORM_PARTIALS 256
int tc_sq_norm(const float* g, size_t n, float* out, tc_stream_t stream);
int tc_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, float max_norm, const float* sq_norm, tc_stream_t stream);


/* The trainable stack (radar encoders + three fusion layers + final_cls / final_reg,
 * HEAD:531-729) as two calls: the forward keeps its activations on a caller-provided
 * tape (tc_radar_train_tape_bytes), the backward walks it and ADDS the parameter
 * gradients into `grads`, a tc_head_weights whose pointers are the gradient buffers
 * of the corresponding parameters (only the radar / final_* / radar encoder entries
 * are used; zero them first).  Inputs of the stack come from tc_head_forward's aux:
 *   hs_last [B,Q,C] = inter_states[-1], ref_last [B,Q,3] = inter_references[-1],
 *   last_box [B,Q,code].  `w` holds the UNPACKED nn.Linear weights.
 *   d_all_cls / d_all_box: gradients of the loss w.r.t. the two outputs.
 * dropout_p > 0: the four dropout sites of each fusion layer as in train mode of the
 *   reference (HEAD:129-171, 581-585: attention probabilities, rf_dropout2, rf_dropout,
 *   rf_dropout3; p = 0.1 there).  Masks are counter-based -- a function of (dropout_seed,
 *   site, element index), regenerated by the backward: pass the SAME p and seed to both,
 *   a fresh seed per iteration.  tc_dropout_mask writes the multipliers (0 or 1/(1-p)) of
 *   elements 0..n-1 of site 4*layer + {0 probabilities [(row*8+head)*num_radar_tokens_ref + token],
 *   1 rf_dropout2, 2 rf_dropout, 3 rf_dropout3 [row*cols + col]}. */
size_t tc_radar_train_tape_bytes(const tc_head_weights* w, int B, int T);
int tc_radar_train_fwd(const tc_head_weights* w, const float* hs_last, const float* ref_last,
                       const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                       float* all_cls_scores, float* all_bbox_preds, void* tape, size_t tape_bytes,
                       float dropout_p, unsigned long long dropout_seed, tc_stream_t stream);
/* tc_radar_train_fwd as launches of the fused row chains: ONE launch for the encoders + K|V (token rows), ONE for
 * the three fusion layers (query rows; the dropout sites in the epilogues / the attention core, every tape tensor
 * stored as it is produced) + the reference set-up.  Same tape, same outputs (to fp32 rounding of the different
 * summation order), same (seed, site, index) dropout masks: tc_radar_train_bwd takes either tape.
 * `packed_view`: the head's packed weights holding the CURRENT parameters (after an optimizer step:
 * tc_head_repack_trainable_ex(w, view, 1, stream), one launch).  Replaces HEAD:531-729 in train mode. */
int tc_radar_train_fwd_fused(const tc_head_weights* packed_view, const float* hs_last, const float* ref_last,
                             const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                             float* all_cls_scores, float* all_bbox_preds, void* tape, size_t tape_bytes,
                             float dropout_p, unsigned long long dropout_seed, tc_stream_t stream);
int tc_radar_train_bwd(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                       const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                       const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                       void* tape, size_t tape_bytes, float dropout_p, unsigned long long dropout_seed,
                       tc_stream_t stream);
/* tc_radar_train_bwd with the query side of all three fusion layers as ONE launch of the backward row chain
 * (data gradients are row-local; every dY a weight gradient needs is stored as it is produced; LayerNorm
 * parameter gradients leave as one atomic per channel and workgroup), the token side (dK|dV -> encoders) as a
 * handful of launches, and EVERY weight / bias gradient in one grouped GEMM launch (two: a scalar variant for
 * the 10-wide heads) -- ~16 launches instead of ~130.  `workspace`: tc_radar_train_bwd_workspace_bytes (the
 * transposed packed weights, re-built here from the current parameters in one launch, the stored dY tensors,
 * dK|dV).  Takes the tape of either forward.  Gradients are ADDED into `grads` as by tc_radar_train_bwd. */
size_t tc_radar_train_bwd_workspace_bytes(const tc_head_weights* w, int B, int T);
int tc_radar_train_bwd_fused(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                             const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                             const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                             void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                             float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                             float* layer_losses_clean, tc_stream_t stream);
/* The same with `flags`: bit 0 = the workspace already holds the transposed packed weights of the CURRENT parameters
 * (tc_radar_train_repack packed them in the launch that re-packed the forward's copy): the backward's own pack launch
 * is skipped.  Bit 1 (ABI 12) = the weight gradients dW = dY^T X are NOT formed by this call: the caller launches them
 * chunk by chunk with tc_radar_train_bwd_weights (below) and starts the gradient exchange of a chunk behind its launch. */
int tc_radar_train_bwd_fused_ex(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                                const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                                const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                                void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                                float* layer_losses_clean, int flags, tc_stream_t stream);
/* ABI 12 (round 6): the weight gradients of one chunk after tc_radar_train_bwd_fused_ex(flags bit 1) on the same stream,
 * same tape / workspace: group 0 .. TC_MAX_RADAR_LAYERS - 1 = fusion layers from the TOP one down (the order a layer-wise
 * backward finishes them, the order DDP's buckets fill in the reference: tools/train.py:253-260), TC_MAX_RADAR_LAYERS =
 * the radar encoders.  Every group is one grouped launch (two with the 10-wide heads); together they add exactly what the
 * single grouped launch of tc_radar_train_bwd_fused_ex adds. */
int tc_radar_train_bwd_weights(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                               const float* radar_tokens, int B, int T, void* tape, size_t tape_bytes, void* workspace,
                               size_t workspace_bytes, int group, tc_stream_t stream);
/* ABI 11: the same backward with ORDER-FREE accumulation -- what torch calls a deterministic algorithm.  The backward adds
 * partial sums from many workgroups into one element in five places (weight-gradient row chunks, bias column sums,
 * LayerNorm parameter gradients of the row chain and of the token side, dK | dV of the attention backward); with float
 * atomics the rounding depends on the arrival order and two runs differ in the last bits.  Here every such add is an
 * INTEGER atomic on a 64-bit fixed-point shadow of its target (units of 2^-40: exact for |v| >= 2^-16, 9e-13 absolute
 * below; |v| < 8.4e6), the token side's split reductions run unsplit, and two small launches add the shadows back:
 * bit-identical gradients from identical inputs, on any schedule.  (tools/train.py:238-260 trains with torch's float
 * atomics, i.e. without this guarantee; FusionTrainer(deterministic=True) and the tests that compare runs use it.)
 * grad_base / grad_elems: the span that holds EVERY tensor of `grads` (the flat bucket); shadow: shadow_elems >= grad_elems +
 * 3 * B * T * 2 * embed_dims + 8 64-bit words; the first grad_elems + 3 B T 2 embed_dims must be zero on entry -- the
 * call leaves them zero -- and the LAST 8 words of the buffer (shadow[shadow_elems - 8 ..]) are scratch, wherever (B, T)
 * put the end of the sums: one buffer sized for the longest frame serves every shorter one (round 6).  A partial sum the
 * fixed-point word cannot hold (NaN, inf, |v| >= 2^23) is added to the float target itself: non-finite or exploding
 * gradients stay visible, as with the float atomics of tc_radar_train_bwd_fused. */
int tc_radar_train_bwd_fused_det(const tc_head_weights* w, const tc_head_weights* grads, const float* hs_last,
                                 const float* last_box, const float* radar_tokens, int B, int T, int pad_mult,
                                 const float* all_bbox_preds, const float* d_all_cls, const float* d_all_box,
                                 void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                 float dropout_p, unsigned long long dropout_seed, const float* layer_losses,
                                 float* layer_losses_clean, int flags, float* grad_base, size_t grad_elems,
                                 long long* shadow, size_t shadow_elems, tc_stream_t stream);
/* One launch for BOTH weight layouts a fused training iteration needs from the current parameters: the 4x4x1 packed
 * copy of the trainable weights inside `packed_view` (tc_head_repack_trainable_ex(w, view, 1)) and the transposed
 * packed weights of the three fusion layers inside the backward's workspace (tc_radar_train_bwd_workspace_bytes(w, B,
 * T)); call it once after the optimizer step, then tc_radar_train_fwd_fused and tc_radar_train_bwd_fused_ex(flags 1). */
int tc_radar_train_repack(const tc_head_weights* w, tc_head_weights* packed_view, void* bwd_workspace,
                          size_t bwd_workspace_bytes, int B, int T, tc_stream_t stream);
/* layer_losses: optional device [num_radar_layers, 2] = (loss_cls, loss_bbox) of each level as tc_detr_loss_fwd_bwd
 * wrote them: a level whose loss is not finite sends no gradient down and non-finite gradient elements count
 * as 0 (HEAD:915-916 zeroes a NaN loss) -- the guard of transcar_amd/device_loss.py inside the
 * backward instead of eight elementwise launches in front of it.  NULL: the gradients are taken as given.
 * layer_losses_clean: optional device [num_radar_layers, 2], receives layer_losses with NaN -> 0
 * (HEAD:915-916; an infinite loss stays): the values of the iteration's loss dict, written by the same launch. */
int tc_dropout_mask(float dropout_p, unsigned long long seed, int site, size_t n, float* out,
                    tc_stream_t stream);

/* (ABI 5's process-global tc_set_chain_tile_rows is gone: the tile height is per call --
 * tc_head_options.chain_tile_rows, tc_decoder_layer_tail_fwd's tile_rows.  With frames in flight
 * 8-row tiles at one frame per launch trade latency for throughput.) */


/* ---- targets and losses on the device (HEAD:742-917; ASSIGN:106-125; COST:15-26) ----
 * tc_normalize_bbox: UTIL:4-24 on n ground-truth boxes [n,9] -> [n,10].
 * tc_match_cost: Hungarian cost for every decoder output in one launch,
 *   cost[l,b,q,g] = cls_weight * FocalLossCost(cls[l,b,q], label[b,g]) + reg_weight * |box[l,b,q,:10] - gt_norm[b,g]|_1
 *   (0 for g >= gt_counts[b]); gt_norm [B,Gmax,10], gt_labels [B,Gmax], gt_counts [B] (device).
 *   The assignment itself stays scipy's linear_sum_assignment on the host, as in the reference.
 * tc_detr_loss_fwd_bwd: with assigned[l,b,q] = matched gt index or -1: sigmoid focal loss
 *   (mmdet FocalLoss, background = no target) and L1 loss of the matched box codes
 *   (code_weights [code] on the device; rows with a non-finite target are skipped), each
 *   divided by avg_factors[2l] (classification) / avg_factors[2l+1] (boxes), device memory:
 *   max(mean over ranks of the number of positives, 1) (HEAD:889-902):
 *   losses[l] = {loss_cls, loss_bbox} (+=: zero first) and the gradients d_all_cls, d_all_box (=). */
int tc_normalize_bbox(const float* gt_boxes, int n, float* out, tc_stream_t stream);
int tc_match_cost(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                  int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                  const int* gt_counts, int Gmax, float cls_weight, float reg_weight, float alpha,
                  float gamma, float eps, float* cost, tc_stream_t stream);
int tc_detr_loss_fwd_bwd(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                         int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                         int Gmax, const int* assigned, const float* avg_factors,
                         const float* code_weights, float alpha, float gamma, float cls_loss_weight,
                         float bbox_loss_weight, float* losses, float* d_all_cls, float* d_all_box,
                         tc_stream_t stream);

/* tc_lsa_assign (round 4): the Hungarian assignment ITSELF on the device -- what ASSIGN:117-125 does with a D2H copy
 * and scipy.optimize.linear_sum_assignment on the host (0.30 ms of a 0.96 ms training iteration's critical path).
 * The algorithm scipy implements (shortest augmenting paths, Jonker-Volgenant in Crouse's rectangular form: the
 * ground-truth boxes are the rows), in float64 like scipy, one wavefront per (output, sample).
 *   cost      [num_outputs, B, Q, Gmax] from tc_match_cost; gt_counts [B] (device); Q <= 1024, Gmax <= 128, Gmax <= Q
 *   assigned  [num_outputs, B, Q] int: matched gt index or -1 (as HungarianAssigner3D's assigned_gt_inds - 1)
 *   num_pos   [num_outputs, 2] float or NULL: += the number of matched boxes, in both columns (zero first) -- the
 *             `avg_factors` of tc_detr_loss_fwd_bwd_counts (one rank), or the operand of the ranks' all-reduce
 *   status    [1] int or NULL: += 1 for every (output, sample) with a non-finite cost (scipy raises there): that
 *             sample stays unassigned
 * The optimum is unique unless two assignments cost exactly the same; on an exact tie this kernel prefers the
 * lowest unassigned query, scipy the one met last in its work list (same total cost). */
int tc_lsa_assign(const float* cost, const int* gt_counts, int num_outputs, int B, int Q, int Gmax, int* assigned,
                  float* num_pos, int* status, tc_stream_t stream);
/* ABI 10: ... and poison_losses [num_outputs, 2] (the zeroed loss accumulators tc_detr_loss_fwd_bwd* adds to) or NULL:
 * NaN goes into the pair of an output one of whose samples could not be assigned, so that the non-finite-loss guard of
 * the backward zeroes that output's gradients instead of training every query of the sample towards "background".
 * The reference stops there (scipy raises ValueError, ASSIGN:117-125); FusionTrainer raises too, one iteration late
 * (the status word travels back without a synchronisation). */
int tc_lsa_assign_ex(const float* cost, const int* gt_counts, int num_outputs, int B, int Q, int Gmax, int* assigned,
                     float* num_pos, int* status, float* poison_losses, tc_stream_t stream);
/* tc_detr_loss_fwd_bwd with avg_factors = raw counts: the normalisers are max(count, 1) */
int tc_detr_loss_fwd_bwd_counts(const float* all_cls, const float* all_box, int num_outputs, int B, int Q,
                                int num_classes, int code_size, const float* gt_norm, const int* gt_labels,
                                int Gmax, const int* assigned, const float* avg_factors,
                                const float* code_weights, float alpha, float gamma, float cls_loss_weight,
                                float bbox_loss_weight, float* losses, float* d_all_cls, float* d_all_box,
                                tc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRANSCAR_HIP_H */
