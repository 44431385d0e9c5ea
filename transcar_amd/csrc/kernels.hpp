// Host-side launchers of the gfx950 kernels (one translation unit each).
#pragma once
#include "common.hpp"
#include "../../include/transcar_hip.h"

namespace tc {

// ---- gemm.hip: token-wise linear on the f32 MFMA (v_mfma_f32_16x16x4_f32) --
struct GemmArgs {
  const float* X = nullptr;  int ldx = 0;      // [M,K]
  const float* X2 = nullptr; int x2_cols = 0;  // prologue: X+X2 for output cols < x2_cols
  const float* W = nullptr;  int ldw = 0;      // [N,K] (nn.Linear weight)
  const float* bias = nullptr;
  const float* R = nullptr;  int ldr = 0;      // residual added last (may be null)
  const int* rowgate = nullptr;                // [M]: 0 => drop the linear term (y = R)
  float* Y = nullptr;        int ldy = 0;      // [M,N] (cols < t_col0 when Yt is set)
  float* Yt = nullptr;                         // transposed store of cols >= t_col0:
  int t_col0 = 0, t_ld = 0, t_rows_per_batch = 1;  //   Yt[(b*(N-t_col0)+c)*t_ld + q]
  int M = 0, K = 0, N = 0;
  int act = 0;                                 // 0 none, 1 relu, 2 sigmoid
  float scale = 1.0f; int scale_cols = 0;      // (acc+bias)*scale for cols < scale_cols
};
int launch_gemm(const GemmArgs& a, hipStream_t s);

// ---- rowops.hip: wave-per-row kernels -------------------------------------
// y = LN(a (+b gated by rowgate) (+ relu(LN(c; g2,b2))))*g + beta, optional relu
struct LnArgs {
  const float* a = nullptr;
  const float* b = nullptr;          // optional addend
  const float* c = nullptr;          // optional addend that first goes through LN(g2,b2)+ReLU
  const float* g2 = nullptr; const float* b2 = nullptr;
  const float* gamma = nullptr; const float* beta = nullptr;  // null => no final LN (plain sum)
  float* y = nullptr;
  int M = 0; int relu = 0;
  const float* d = nullptr;          // optional addend after the final LN/ReLU (y += d)
  int d_relu = 0;                    // apply relu to d first
};
int launch_ln256(const LnArgs& a, hipStream_t s);

// y = relu(LN(W0 p + b0)), p = inverse_sigmoid(src[row,0:3]) or raw src[row*ld+0:3]
int launch_posenc_l1(const float* src, int ld, int inv_sigmoid, const float* w0, const float* b0,
                     const float* g, const float* beta, float* y, int M, hipStream_t s);

// initial reference points: sigmoid(query_pos W^T + b), XFMR:122-123.  out [B,Q,3]
int launch_init_ref(const float* query_embedding, int Q, int C, const float* w, const float* b,
                    float* ref, int B, hipStream_t s);

// expand query_embedding [Q,2C] into pos [B,Q,C], x [B,Q,C] (XFMR:119-121)
int launch_split_embed(const float* query_embedding, int Q, int C, int B, float* pos, float* x,
                       hipStream_t s);

// XFMR:195-203: new_ref = sigmoid(tmp[...,{0,1,4}] + inverse_sigmoid(ref)); also (optionally)
// the head's surviving `tmp` of HEAD:287-293 (xy,z in metres) into box_m [M,code]
int launch_ref_update(const float* tmp, int code, const float* ref, float* new_ref,
                      float* box_m, const float* pc_range6_host, int M, hipStream_t s);

// radar layer epilogue: box = reg_out; box[0:2] += ref_xy; box[4] += ref_z  (HEAD:599-600 etc.)
// and next-layer reference (xy = box[0:2], z = box[4])  (HEAD:615-617)
int launch_box_add_ref(const float* reg_out, int code, const float* ref_xy, int ld_xy,
                       const float* ref_z, int ld_z, float* box, float* next_ref3, int M,
                       hipStream_t s);

// layer-1 radar reference (HEAD:544-547, 596-598): from normalised ref [M,3]:
// centre_xy (metres) [M,2] and the quirky add-reference (x_m, y_m, z_NORMALISED) [M,3]
int launch_radar_ref_l1(const float* ref, const float* pc_range6_host, float* centre_xy,
                        float* addref3, int M, hipStream_t s);

// ---- cam_sample.hip --------------------------------------------------------
struct CamSampleArgs {
  tc_feats_nhwc feats;
  int B, Q, C, num_cams;
  const float* lidar2img; const float* ref; const float* logits;
  float pc[6]; float img_h, img_w;
  float* out; unsigned char* vis; unsigned long long* pair_counter;
};
int launch_cam_sample(const CamSampleArgs& a, hipStream_t s);

// ---- chain.hip: fused row-chain kernels (16 rows per workgroup) ------------
struct PrologueArgs {
  const float* qe; int Q, M;                 // query_embedding [Q,512]; M = B*Q rows
  tc_linear refpts, in_proj;                 // transformer.reference_points, layer-0 in_proj
  float* init_ref; float* qk; float* vt; int qpad; float qscale;
  size_t w16_delta = 0;                      // packed view's packed16_delta (16-row tiles read W + delta)
};
int launch_prologue(const PrologueArgs& a, hipStream_t s);

struct DecoderChainArgs {
  const float* attn_o;                       // [M,256]
  int attn_mod = 0, ref_mod = 0;             // attn_o / ref_in rows taken modulo this when > 0
  const float* x_in; int x_ld, x_mod;        // layer input rows (row % x_mod when x_mod > 0)
  const float* qe; int Q;
  const float* ref_in; float* ref_out; float* box_m;
  const tc_decoder_layer* w;
  const tc_linear* next_in_proj;             // next layer's in_proj (null after the last layer)
  float qscale;
  float* hs; float* qk; float* vt; int qpad;
  CamSampleArgs cam;                         // feats, lidar2img, pc, img size (ref/logits/out unused)
  int code, M;
  int tile_rows = 0;                         // 0: automatic (4 up to 1024 rows, 8 up to 2048, 16 beyond), 4, 8, 16
  int matrix_path = 0;                       // 16-row tiles: TC_MATRIX_AUTO / _F32 / _F16X2 (tc_head_options.matrix_path)
  // train-mode statistics of the frozen decoder (thr 0 = eval): drop.site is the site of THIS
  // layer's attention probabilities (16 + 8 * layer); the chain's four sites are drop.site + 1
  // (self-attention output), + 2 (cross-attention output, XFMR:378), + 3 (FFN hidden), + 4 (FFN output)
  DropK drop = DropK{0, 0, 1.0f, 0, 0, 0, 0, 0};
  int* range_status = nullptr;               // f16x2 kernels: sticky non-finite flag (tc_head_options.range_status)
  // round 6: the sampling step reads what the pre-gather workgroups of the attention-core launch in front stored
  // (PreGatherArgs::out / mask of the SAME reference points) instead of gathering itself; null: gathers itself
  const float* pre = nullptr; const int* premask = nullptr;
};
int launch_decoder_chain(const DecoderChainArgs& a, hipStream_t s);

struct RadarEncodeArgs {
  const float* tokens; int RI, M;            // [M, RI]
  tc_pos_encoder rpe; tc_linear f0, f2, f4;
  int nlayers; tc_linear kvproj[TC_MAX_RADAR_LAYERS]; float* kv[TC_MAX_RADAR_LAYERS];
  float* radar_feat;                         // optional [M,256]
  size_t w16_delta = 0;
  float* const* tape = nullptr;              // training forward: tape tensors by TapeSlot (chain.hip TSel order)
  int matrix_path = 0;                       // as DecoderChainArgs (the stand-alone encoder program runs 16-row tiles)
  int* range_status = nullptr;               // f16x2 kernels: sticky non-finite flag (tc_head_options.range_status)
};
int launch_radar_encode(const RadarEncodeArgs& a, hipStream_t s);
// a decoder layer and the radar encoders as ONE launch (no side stream / graph branch)
int launch_decoder_chain_with_encoders(const DecoderChainArgs& d, const RadarEncodeArgs& e, int part,
                                       hipStream_t s);

struct RadarChainArgs {
  const float* qf; const float* ref_last; const float* box_m;
  const float* tokens; int RI;
  const float* kv[TC_MAX_RADAR_LAYERS];
  tc_radar_layer w[TC_MAX_RADAR_LAYERS];
  int nlayers, Q, T, pad_mult, code, ncls, M;
  float qscale; float pc[6];
  float* all_cls; float* all_box; int* hits;
  int tile_rows = 0;
  int matrix_path = 0;                       // as DecoderChainArgs
  int last_cls_only = 0;                     // skip final_cls of all but the last layer (inference opt-in)
  int cen_from_box = 0;                      // w[0] is not fusion layer 1: gate centre from box_m (HEAD:615-617)
  const int* row_perm = nullptr;             // optional [M]: tile position -> row (launch_radar_compact)
  // forward of a training iteration (tc_radar_train_fwd_fused): tape tensors of fusion layer 0 by TapeSlot,
  // layer r at + r * tape_stride floats; hit counts of layer r at hits + r * hits_stride; dropout seed / p
  float* const* tape = nullptr; size_t tape_stride = 0, hits_stride = 0;
  DropK drop = DropK{0, 0, 1.0f, 0, 1500, 0, 0, 0};
  int* range_status = nullptr;               // f16x2 kernels: sticky non-finite flag (tc_head_options.range_status)
};
// Backward of the three fusion layers for the query rows as ONE launch of the row chain (chain.hip
// PROG_RADAR_BWD): data gradients row-local, every dY a weight gradient needs stored (DySlot order), LayerNorm
// parameter gradients added into `grads`, dK | dV into dkv[r] (atomics; zero them first).
enum DySlot { DY_DBOX = 0, DY_DT1, DY_DT0, DY_DC2, DY_DC0, DY_DFF, DY_DH, DY_DPROJ, DY_DQP, DY_DCLS, DY_COUNT };
struct RadarBwdChainArgs {
  tc_radar_layer wT[TC_MAX_RADAR_LAYERS];      // TRANSPOSED packed weights (pack.hip, PackJob::transpose)
  tc_radar_layer w[TC_MAX_RADAR_LAYERS];       // the forward's parameters (LayerNorm gamma, radii)
  tc_radar_layer grads[TC_MAX_RADAR_LAYERS];   // gradient destinations (LayerNorm g / b are used here)
  const float* kv[TC_MAX_RADAR_LAYERS]; float* dkv[TC_MAX_RADAR_LAYERS];
  const float* cxy[TC_MAX_RADAR_LAYERS]; int ld_c[TC_MAX_RADAR_LAYERS]; const float* box[TC_MAX_RADAR_LAYERS];
  float* const* tape; size_t tape_stride, hits_stride; const int* hits;
  float* const* dy; size_t dy_stride;
  const float* d_cls; const float* d_box;      // [layers, M, ncls / code]
  const float* loss_vals = nullptr;            // optional [layers, 2]: non-finite losses / elements send nothing down
  float* loss_out = nullptr;                   // optional [layers, 2]: loss_vals, NaN -> 0
  const float* tokens; int RI, T, pad_mult;
  int nlayers, Q, M, code, ncls;
  float qscale;
  DropK drop;
  int tile_rows = 0;
  const DetAcc* det_device = nullptr;          // deterministic mode: a device copy of the thread's DetAcc (launch_det_store)
};
int launch_radar_chain_bwd(const RadarBwdChainArgs& a, hipStream_t s);
// order of the tape pointer arrays above (= chain.hip TSel)
enum TapeSlot { TS_QP = 0, TS_AO, TS_X1, TS_X2, TS_H, TS_SUM, TS_X3, TS_C0, TS_C1, TS_C2, TS_C3, TS_T0, TS_T1, TS_TREG,
                TS_U0, TS_U1, TS_U2, TS_POS, TS_F0, TS_F1, TS_F2, TS_MEM, TS_COUNT };
int launch_radar_chain(const RadarChainArgs& a, hipStream_t s);

// ---- radar_compact.hip: row order for the radar chain (queries with a radar hit first) --------
// flags [B*Q] scratch, perm [B*Q]: perm[b*Q + i] = a row of sample b
int launch_radar_compact(const float* ref_last, const float* box, int code, int cen_from_box,
                         const float* pc6_host, const float* tokens, int RI, int B, int Q, int T,
                         float rmin, float rmax, int* flags, int* perm, hipStream_t s);

int launch_gate_selfcheck(int n_radii, unsigned long long seed, unsigned long long* mismatches, hipStream_t s);
int launch_rowops_selfcheck(int n_blocks, unsigned long long seed, unsigned long long* mismatches, hipStream_t s);   // chain.hip

// ---- self_attn.hip ---------------------------------------------------------
// q,k: [B*Q, ld] token-major with head h at column h*32; vt: [B, C, ldt] (V transposed)
// drop (may be null / thr 0 = eval): dropout on the attention probabilities, index ((b*H+h)*Q+i)*Q+j
int launch_self_attn_core(const float* q, const float* k, int ld, const float* vt, int ldt,
                          float* out, int ldo, int B, int Q, int H, hipStream_t s,
                          const DropK* drop = nullptr);
// the same on the f16 matrix cores, fp32-accurate (round 4): two-plane f16 operands (hi, lo), fp32 accumulate; K / V^T are
// split and staged through LDS inside the kernel, every wave walks all keys.  Same operands as launch_self_attn_core.
// pregather (round 6; may be null): extra workgroups of the same launch gather the camera taps of the decoder chain that
// FOLLOWS (rowdev.hpp cam_pregather_rows): `cam` as for the chain (ref = that layer's reference points), out
// [M][num_cams][num_levels][256] floats, mask [M] ints.  Eval launches only (drop == null / thr 0).
struct PreGatherArgs {
  CamSampleArgs cam; int M, ref_mod; float* out; int* mask;
};
int launch_self_attn_core_x(const float* q, const float* k, int ld, const float* vt, int ldt, float* out, int ldo,
                            int B, int Q, int H, hipStream_t s, const DropK* drop = nullptr,
                            const PreGatherArgs* pregather = nullptr);

// ---- radar_attn.hip --------------------------------------------------------
struct RadarAttnArgs {
  const float* qproj; int ldq;         // [B*Q, C] projected+scaled queries
  const float* kv; int ldkv;           // [B*T, 2C]: K | V
  const float* centre_xy; int ld_c;    // centre at centre_xy[row*ld_c + {0,1}]
  const float* box; int code;          // [B*Q,code]
  const float* radar_xy; int ld_xy;    // token xy at radar_xy[(b*T+t)*ld_xy + {0,1}]
  int B, Q, T, C, H, pad_mult;
  float rmin, rmax;
  float* attn_out;                     // [B*Q, C] (zero rows where no hit)
  int* hit_counts;                     // [B*Q]
  float qscale = 1.0f;                 // applied to qproj on load (1 when pre-scaled)
  DropK drop = DropK{0, 0, 1.0f, 0, 1500, 0, 0, 0};   // training: dropout on the attention probabilities (thr 0 = off)
};
int launch_radar_attn(const RadarAttnArgs& a, hipStream_t s);
int launch_dropout(const float* x, const float* res, const int* gate, int rows, int cols, const DropK& d,
                   float* out, hipStream_t s);

// ---- train.hip: backward of the trainable (radar) part + optimizer ----------
int launch_linear_bwd_data(const float* dy, const float* relu_out, const int* row_gate,
                           const float* w, const float* in_relu_mask, float* dx, int M, int K,
                           int N, float alpha, int accumulate, hipStream_t s);
int launch_linear_bwd_weight(const float* x, const float* dy, const float* relu_out,
                             const int* row_gate, float* dw, float* db, int M, int K, int N,
                             float alpha, hipStream_t s);
// every weight gradient of an iteration in one launch (masks already applied to dy): dw += dy^T x, db += colsum dy
struct WeightJob { const float* x; const float* dy; float* dw; float* db; int M, K, N; const float* relu = nullptr; /* dy is zeroed where relu <= 0 */
                   int ldx = 0; /* row stride of x (0: K) -- the first K columns of a wider matrix */ };
int launch_linear_bwd_weight_group(const WeightJob* jobs, int n, hipStream_t s);
// Deterministic accumulation (common.hpp DetAcc): the launchers of the backward (bwd GEMMs, LayerNorm backward, the
// backward row chain) hand the calling thread's current DetAcc to their kernels.  tc_radar_train_bwd_fused_det sets it
// for the duration of ITS call (DetScope) -- per call, per thread: not a process-wide switch.
DetAcc& current_det();
// ... and where the DEVICE copy of it lives for the backward row chain (launch_det_store): the LAST words of the caller's
// shadow buffer -- a place that does not move with (B, T) (round 6, ADVICE r5: behind the dK | dV shadows it moved with T,
// and a shorter frame left its pointer words inside a longer frame's shadow range)
DetAcc*& current_det_device();
struct DetScope {
  DetAcc saved; DetAcc* saved_dev;
  DetScope(const DetAcc& d, DetAcc* dev) : saved(current_det()), saved_dev(current_det_device()) { current_det() = d; current_det_device() = dev; }
  ~DetScope() { current_det() = saved; current_det_device() = saved_dev; }
  DetScope(const DetScope&) = delete;
  DetScope& operator=(const DetScope&) = delete;
};
// targets[i] += shadow[i] * 2^-40; shadow[i] = 0, for range r of d (one launch)
int launch_det_flush(const DetAcc& d, int range, hipStream_t s);
// *dst (device memory) = d, stream-ordered: the backward row chain reads the ranges from there
int launch_det_store(const DetAcc& d, DetAcc* dst, hipStream_t s);
int launch_ln256_bwd(const float* a, const float* b, const float* gamma, const float* dy,
                     const float* relu_out, float* dz, float* dgamma, float* dbeta, int M,
                     hipStream_t s);
int launch_radar_attn_bwd(const RadarAttnArgs& a, float qscale, const float* d_attn, float* dq,
                          float* dkv, hipStream_t s);
int launch_box_ref_bwd(const float* d_box, int code, float* d_prev, int M, hipStream_t s);
int launch_sqnorm(const float* g, size_t n, float* out, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1,
                 float b2, float eps, float wd, int step, float grad_scale, float max_norm,
                 const float* sqnorm, hipStream_t s);

// ---- pack.hip: one-time weight re-layout for the fused chains ---------------
size_t packed_floats(int N, int K);
// P16 (may be null): the copy for the 16-row tiles' 16x16x4 MFMA (pack.hip)
// PH (may be null): the two-plane f16 copy for the 16-row tiles on the f16 matrix cores
int launch_pack_linear(const float* W, int N, int K, float* P, float* P16, float* PH, hipStream_t s);
// several weights in one launch (P16 of an item may be null: only the 4x4x1 copy)
//   transpose: the matrix to pack is W^T -- element (n, k) = W[k * ldw + n] (the backward row chain: dx = dy W)
struct PackJob { const float* W; float* P; float* P16; int N, K; int transpose = 0, ldw = 0; float* PH = nullptr; };
int launch_pack_group(const PackJob* jobs, int n, hipStream_t s);

// ---- transpose.hip ---------------------------------------------------------
int launch_nchw_to_nhwc(const float* src, float* dst, int n_img, int C, int H, int W, hipStream_t s);
int launch_nchw_to_nhwc_levels(const float* const* src, float* const* dst, int num_levels, int n_img,
                               int C, const int* H, const int* W, hipStream_t s);

// ---- radar_ingest.hip ------------------------------------------------------
int launch_radar_ingest(const double* raw, const double* times, const int* chan_start_host, int num_chan,
                        const double* radar_rot_host, const double* lidar_rot_host,
                        const double* point_range_host, float* tokens, int T, int* count, hipStream_t s);
int launch_radar_ingest_batch(const double* raw, const double* times, const tc_radar_frame_desc* desc, int P, int cap,
                              float* tokens, int T, int* count, hipStream_t s);

// ---- decode.hip ------------------------------------------------------------
// the rows NMSFreeCoder keeps, compacted in score order (CODER:62-84): [B,max_num,9] / [B,max_num] / [B,max_num]
// int64 (the reference's labels are torch.long) and count [B]; rows count[b] .. max_num-1 are not written
struct BoxDecodeKept {
  float* boxes; float* scores; long long* labels; int* count;
  float score_threshold; int use_threshold;     // CODER:63-64, 73-74: strict `>`; off for None / 0
  int z_shift;                                  // 1: z -= h / 2 (get_bboxes, HEAD:1018); 0: the coder's own output
};
int launch_box_decode(const float* cls, const float* box, int B, int Q, int ncls, int code,
                      int max_num, const float* pcr6_host, float* boxes, float* scores, int* labels,
                      unsigned char* valid, void* ws, size_t ws_bytes, hipStream_t s, const BoxDecodeKept* kept = nullptr);
size_t box_decode_ws_bytes(int B, int Q, int ncls);

}  // namespace tc
