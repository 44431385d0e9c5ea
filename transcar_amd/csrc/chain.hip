// Fused row-chain kernel: the launch-count fix for the B = 1 hot path.
//
// Everything in Detr3DHead.forward except the 900x900 self-attention is
// ROW-LOCAL: a query's out_proj, LayerNorms, camera sampling, FFN, box
// refinement, next-layer QKV projection, and the whole 3-layer radar fusion
// stack only ever touch that query's own 256 channels (plus read-only feature
// maps / radar K,V).  The first build ran them as ~160 dependent launches per
// frame; on MI355X a dependent launch costs ~4-5 us even when the kernel is
// trivial, so the frame was launch-bound (profiles/r1_v1_*).  Here a workgroup
// of 4 waves owns R = 4 / 8 / 16 rows (queries or radar tokens), keeps them in
// LDS and walks a whole chain; weights stream from L2 into MFMA operands.
//
// ONE kernel, four programs (step tables in constant memory), so a frame is
// 16 launches instead of 163:
//   PROG_PROLOGUE   embedding split, initial reference points, layer-0 QKV
//   PROG_DECODER    per decoder layer, everything after the attention core up
//                   to and including the next layer's QKV
//   PROG_RADAR_ENC  radar MLP encoders + the K/V projections of the three
//                   fusion layers (independent of the decoder: side stream)
//   PROG_RADAR      the three radar fusion layers incl. gated attention,
//                   class / box MLPs and reference bookkeeping
//
// Design notes, every one of them measured on MI355X (tools/*_probe.hip and
// in-kernel s_memtime stamps; DESIGN.md "Row chains"):
//  * v_mfma_f32_4x4x1_16B_f32 instead of 16x16x4.  With 16x16x4 a row tile must
//    hold 16 queries = 57 workgroups for 900 queries, i.e. 57 CUs' worth of f32
//    matrix pipe, and the chain was bound by exactly that.  The 4x4x1 form is a
//    [4 queries] x [64 output columns] outer product per instruction with the A
//    operand BROADCAST from one 4-lane block (cbsz = 4, abid = row group): 4-row
//    tiles give 225 workgroups at full MFMA efficiency (9.0-9.3 cycles per MFMA
//    measured in the loop, 9.6 in isolation), and with R = 8 / 16 each weight
//    register feeds 2 / 4 MFMAs.  Exact fp32: f32 FMA chains, two k-interleaved
//    accumulators per row group so the pipe is issue- not latency-bound.
//  * Weights are PRE-PACKED (pack.hip) so one wave-instruction reads 1 KiB
//    contiguous: from the nn.Linear [N][K] layout a CU streams fragments at
//    36 GB/s, packed at 124 GB/s; loads of item i+1 fly under the MFMAs of item i.
//  * The A operand goes LDS -> registers with 16 UNGUARDED ds_read_b128 before
//    the MFMA burst: a bounds check per read turned them into 16 dependent round
//    trips (1350 cycles per item, more than the MFMAs).
//  * __builtin_amdgcn_sched_barrier(0) around load groups: hipcc's scheduler
//    otherwise sinks each load next to its first use (1-2 loads in flight).
//  * Instruction-cache capacity is NOT an issue (128 KB of straight-line MFMAs
//    runs at full rate: tools/icache_probe.hip); the step table exists to keep
//    ONE code image for all four programs.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "kernels.hpp"
#ifdef TC_CHAIN_STAMPS
__device__ long long g_cam_stamps[4][8];      // camera sampling of workgroup 100, per wave
#define CAM_STAMP(slot)                                                                        \
  do {                                                                                         \
    if (blockIdx.x == 100 && (threadIdx.x & 63) == 0)                                          \
      g_cam_stamps[(threadIdx.x >> 6) & 3][(slot)] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#endif
#include "rowdev.hpp"
// Round 6: the camera taps are loaded with the NON-TEMPORAL hint (rowdev.hpp cam_tap_ld).  A tap row is read once; with
// plain loads the 134 MB a nine-frame layer gathers went through the L2s as ordinary lines and pushed out what IS
// reused -- the layer's packed weights (3.2 MB per XCD, streamed by 32 workgroups each) and, a launch later, the next
// kernels' operands.  Same loads, same values (bit-identical outputs); measured on one box, alternating runs
// (profiles/r6_taps_nt_ab.txt): driver command 9 408 -> 9 638 frames/s (+2.4 %), 216-step windows 10 296 -> 10 409 (+1.1 %),
// one launch sequence at a time 8 081 -> 8 465 (+4.7 %).
// The same hint on the other read-once streams of a chain -- the attention output, the layer input, the radar chain's
// query rows (TC_ONCE_LD): another +1.0 % (driver command 9 595 -> 9 693) / +1.6 % (216-step windows 10 300 -> 10 468).  The
// hint on the write-once STORES (hs, q | k, the attention output) costs 1-1.5 % instead: they stay plain
// (profiles/r6_taps_nt_ab.txt; tools/experiments/README.md).
#define TC_ONCE_LD(p) ldg4_stream(p)
#ifdef TC_CHAIN_DUMP
// diagnostic build only (make DUMP=1): the LDS destination of every step of the radar program, [step][row][256] floats
__device__ float* g_chain_dump;
__device__ long long g_chain_dump_floats;
#endif

namespace tc {

namespace {

constexpr int CH_NW = 4;              // waves of a workgroup with 4- / 8- / 16-row tiles ...
constexpr int CH_NT = CH_NW * 64;
constexpr int CH_NW_MAX = 8;          // ... and with 32-row tiles (round 5: one workgroup of 8 waves per CU)
constexpr int nw_of(int R) { return R == 32 ? 8 : 4; }
constexpr int LD5 = 516;   // 512 + 4 floats
constexpr int LD2 = 260;   // 256 + 4
constexpr int LDL = 36;

// a step of the table with every pointer / size resolved (built once per launch, on the host)
struct StepRes {
  const float* p0; const float* p1; const float* p2; const float* p3;   // linear: W, bias; LN: g, b, g2, b2
  float* gd; float* gt;          // global destination (source for K_LOAD) / transposed destination
  int K, N, gld, gmod;
  short kind, src, src2, dst, res, act, flags, sync, rep, si;
  int src_off, src2_off;         // linear: A operand buffers as float offsets into shared memory (-1: none)
  int pad_;
};
// what the epilogue of a linear step needs, with the LDS buffers as float offsets from
// the start of shared memory (-1: none): rebuilt from here in ~150 cycles per tile
struct EpiRec {
  float* gd; float* gt;
  int dst_off, res_off, dst_ld, res_ld;
  int gld, act, flags, N;
  int has_bias, woff;
  int drop_site;               // train-mode decoder dropout on this step's output: site + 1 (0: none)
  int pad1;
};
// per wave: the next linear step of the same run in which the wave owns a column tile
struct PreRec { const float* first; int nidx; int pad; };
// One step, resolved ON THE HOST at launch time and handed over in the kernel-argument
// segment; the workgroup copies the records to LDS with one round of parallel vector loads
// and reads a step from there (ds_read + v_readfirstlane into SGPRs).  Measured
// alternatives: resolved by the workgroup itself (table in constant memory -> kernel
// arguments -> LDS records) every launch began with ~9400 cycles of dependent loads, 4 us
// of a 56 us decoder layer; read per step with scalar loads straight from the
// kernel-argument segment, every step paid 400-800 cycles of scalar-cache misses (+6 us).
template <int NW> struct StepAllT { StepRes r; EpiRec e; PreRec p[NW]; };
using StepAll = StepAllT<CH_NW>;
static_assert(sizeof(StepAllT<CH_NW>) % 16 == 0 && sizeof(StepAllT<CH_NW_MAX>) % 16 == 0, "records are copied 16 bytes at a time");
template <int N, int NW = CH_NW> struct Recs { StepAllT<NW> s[N]; };

// Activations of the R rows: FOUR [R][256 + 4] units (round 1 had seven: 120 KB at R = 16, one
// workgroup per CU).  Unit 0 (B_X) is the residual stream; units 1-3 are temporaries placed by the
// step tables below from the programs' liveness; the 512-wide FFN hidden tile (B_A, row stride
// 512 + 4) spans units 1 and 2.  query_pos is not held at all: the LayerNorm in front of the two
// linears that need x + query_pos writes that sum next to x (F_LN_XP).  R = 16: 66.6 KB + records
// = 76 KB -- two workgroups per CU.
// radar: the gate's hit tokens per row, 64 per word (T <= 64 * HM_WORDS), live from K_RADAR_GATE to
// K_RADAR_ATTN in the row's slice of `l` (free until the layer's last two steps) -- the radar program at
// R = 16 is 81.4 KB: 544 bytes more and only one workgroup fits a CU
constexpr int HM_WORDS = 8;
static_assert(HM_WORDS * 8 <= LDL * 4 && (LDL * 4) % 8 == 0, "hit masks live in a row of the logit buffer");
template <int R, int NREC>
struct ChainLds {
  StepAllT<nw_of(R)> recs[NREC];
  float unit[4][R][LD2];
  float l[R][LDL];
  float box[R][12];
  float cen[R][4];
  int gate[R];
  int rowg[R];                 // radar: the global row of tile position i (ChainDev::row_perm, else m0 + i)
};
static_assert(LD5 * 1 <= 2 * LD2, "the 512-wide tile must fit two units");

// timing experiments (skip epilogues / barriers / LayerNorm / sampling) exist in the STAMPS
// build only; the production library has no such switch
#ifdef TC_CHAIN_STAMPS
#define CHAIN_DBG(v) (v)
#else
#define CHAIN_DBG(v) 0
#endif

#define MFMA44(a, b, c, grp) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 4, (grp), 0)

// ---- step tables -----------------------------------------------------------
enum Buf : short { B_NONE = -1, B_A = 0 /* units 1+2, row stride LD5 */, B_X /* unit 0 */, B_U1, B_U2, B_U3, B_L };
enum Kind : short {
  K_END = 0, K_LOAD, K_LINEAR, K_LN, K_POSENC, K_SAMPLE, K_REFUPD, K_TOKENS, K_RADAR_ATTN, K_BOXADD, K_RADAR_GATE,
  K_NARROW,   // y[R, N <= 12] = x[R, 256] W^T + b: one 16x16 MFMA sub-tile, k split over the waves, W in the nn.Linear layout
  K_NOP,  // a step switched off at run time (no next layer): only its barrier remains
  // backward row chain (PROG_RADAR_BWD)
  K_LOADG,     // dst[R][0..63] = global [M, N <= 64] rows (zero filled); F_CARRY: + the box gradient carried down
  K_MASKCOPY,  // dst = dropout-keep(site K - 1) (.) [row gate] (.) src, stored to gd
  K_LN_BWD,    // LayerNorm backward of dy (src) at z = tape p1 (+ p2) -> dz (dst), dgamma / dbeta atomics
  K_ATTN_BWD   // gated attention backward of one row: d(attention output) (src) -> d(projected query) (dst)
};
enum NSpecial : short { N_LOGITS = -1, N_CODE = -2, N_CLS = -3 };
enum Flags : short {
  F_GATE = 1,         // linear: row gate from LDS (radar hit counts)
  F_SCALEQ = 2,       // linear: scale the first 256 output columns by qscale
  F_WOFF = 4,         // linear: weights start 512 rows into the pair (V part of a packed in_proj)
  F_LN_RELU = 8,      // ln: relu after the outer LN
  F_SKIP_NONEXT = 16, // skip the step when there is no next layer
  F_WAVE1 = 32,       // linear: column tile t goes to wave (t + 1) % 4 -- a narrow step (one tile)
                      //   then runs BESIDE the preceding narrow step, which keeps wave 0 busy
  F_NOT_W0 = 64,      // posenc: rows go to waves 1..3 only (wave 0 is in a narrow linear step)
  F_LN_XP = 128,      // ln: also write (result + query_pos row) into the buffer named by `res`
  F_IFHIT = 256,      // radar: the step only matters for rows with a radar hit -- skipped when no row of the
                      //   tile has one (a linear step then copies `res` to `dst`: x + gate * (...) = x)
  F_GPRE = 512,       // linear: the global store happens BEFORE the residual add (training tape: the FFN output
                      //   rf_dropout3(linear2(.)) itself, not x + it)
  F_CMASK = 1024,     // linear (backward): zero the outputs where the tape tensor `gt` [M, N] is <= 0 (ReLU')
  F_CMASK_SCALE = 2048,  // ... and multiply the kept ones by the dropout scale (the tape tensor is a dropped-out ReLU)
  F_CARRY = 4096,     // K_LOADG: the layer's first step (box gradient carry, hit counts of the layer)
  F_NOT_LAYER0 = 8192, // the step does not exist for fusion layer 1 (nothing below it is trainable)
  F_PRESYNC = 16384   // K_NARROW: the barrier in front of the step is the step's own, BEHIND its weight loads (the step before
                      // ends without one): the loads' round trip runs under the wait for the slowest wave
};
// global tensors, indices into ChainK::g
enum GSel : short {
  G_NONE = -1, G_HS = 0, G_QK, G_VT, G_INITREF, G_ATTN_O, G_XIN, G_POS, G_CLS, G_KV0, G_KV1, G_KV2,
  G_QF, G_RFEAT, G_COUNT
};

struct StepDesc {
  short kind, wp, wp2, K, N, src, src2, dst, res, act, flags, gsel, gtsel, sync;
};
//   kind      wp wp2   K    N        src   src2    dst    res  act flags gsel gtsel sync
// decoder layer pairs: 0 in_proj 1 out_proj 2 norm0 3 attw 4 output_proj 5 pe.l0 6 pe.n1 7 pe.l3
//   8 pe.n4 9 norm1 10 ffn0 11 ffn1 12 norm2 13 reg.l0 14 reg.l2 15 reg.l4 ; 16 next in_proj
constexpr StepDesc PROG_DECODER_T[] = {
    {K_LOAD, 0, 0, 0, 0, B_NONE, B_NONE, B_U1, B_NONE, 0, 0, G_ATTN_O, G_NONE, 0},
    {K_LOAD, 0, 0, 0, 0, B_NONE, B_NONE, B_X, B_NONE, 0, 0, G_XIN, G_NONE, 1},
    {K_LINEAR, 1, -1, 256, 256, B_U1, B_NONE, B_U2, B_X, 0, 0, G_NONE, G_NONE, 1},        // x + out_proj(attn)
    {K_LN, 2, -1, 0, 0, B_U2, B_NONE, B_X, B_U1, 0, F_LN_XP, G_NONE, G_NONE, 1},          // norm0; x + pos -> U1
    {K_LINEAR, 3, -1, 256, N_LOGITS, B_U1, B_NONE, B_L, B_NONE, 0, 0, G_NONE, G_NONE, 0}, // attention_weights(x + pos)
    {K_POSENC, 5, 6, 0, 0, B_NONE, B_NONE, B_U2, B_NONE, 0, F_NOT_W0, G_NONE, G_NONE, 1}, // pe.0-2 (waves 1-3)
    {K_SAMPLE, 0, 0, 0, 0, B_L, B_NONE, B_U3, B_NONE, 0, 0, G_NONE, G_NONE, 1},           // camera sampling
    {K_LINEAR, 7, -1, 256, 256, B_U2, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 1},     // pe.3 (barrier: U2 is re-used next)
    {K_LINEAR, 4, -1, 256, 256, B_U3, B_NONE, B_U2, B_X, 0, 0, G_NONE, G_NONE, 1},        // x + output_proj
    {K_LN, 9, 8, 0, 0, B_U2, B_U1, B_X, B_NONE, 0, 0, G_NONE, G_NONE, 1},                 // norm1(U2 + relu(LN(U1)))
    {K_LINEAR, 10, -1, 256, 512, B_X, B_NONE, B_A, B_NONE, 1, 0, G_NONE, G_NONE, 1},      // ffn0
    {K_LINEAR, 11, -1, 512, 256, B_A, B_NONE, B_U3, B_X, 0, 0, G_NONE, G_NONE, 1},        // x + ffn1
    {K_LN, 12, -1, 0, 0, B_U3, B_NONE, B_X, B_U1, 0, F_LN_XP, G_HS, G_NONE, 1},           // norm2 -> hs; x + pos -> U1
    {K_LINEAR, 13, -1, 256, 256, B_X, B_NONE, B_U2, B_NONE, 1, 0, G_NONE, G_NONE, 0},     // reg.0
    {K_LINEAR, 16, -1, 256, 512, B_U1, B_NONE, B_NONE, B_NONE, 0, F_SCALEQ | F_SKIP_NONEXT, G_QK, G_NONE, 0},
    {K_LINEAR, 16, -1, 256, 256, B_X, B_NONE, B_NONE, B_NONE, 0, F_WOFF | F_SKIP_NONEXT, G_NONE, G_VT, 1},
    {K_LINEAR, 14, -1, 256, 256, B_U2, B_NONE, B_U3, B_NONE, 1, 0, G_NONE, G_NONE, 0},    // reg.2
    {K_NARROW, 15, -1, 256, N_CODE, B_U3, B_U1, B_L, B_NONE, 0, F_PRESYNC, G_NONE, G_NONE, 1},    // reg.4 (partial sums: U1)
    {K_REFUPD, 0, 0, 0, 0, B_L, B_NONE, B_NONE, B_NONE, 0, 0, G_NONE, G_NONE, 0},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// prologue pairs: 0 reference_points 16 layer-0 in_proj
constexpr StepDesc PROG_PROLOGUE_T[] = {
    {K_LOAD, 0, 0, 0, 0, B_NONE, B_NONE, B_U1, B_NONE, 0, 0, G_POS, G_NONE, 0},
    {K_LOAD, 0, 0, 0, 0, B_NONE, B_NONE, B_X, B_NONE, 0, 0, G_XIN, G_NONE, 1},
    {K_LINEAR, 0, -1, 256, 3, B_U1, B_NONE, B_NONE, B_NONE, 2, 0, G_INITREF, G_NONE, 0},
    {K_LINEAR, 16, -1, 256, 512, B_X, B_U1, B_NONE, B_NONE, 0, F_SCALEQ, G_QK, G_NONE, 0},
    {K_LINEAR, 16, -1, 256, 256, B_X, B_NONE, B_NONE, B_NONE, 0, F_WOFF, G_NONE, G_VT, 0},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// radar encoder pairs: 0 rpe.l0 1 rpe.n1 2 rpe.l3 3 rpe.n4 4 f0 5 f2 6 f4 7..9 kv proj of layer 0..2
// units: the token tile (64 columns) lives in B_A (units 1-2) until feat.0 and the position
// encoder's first layer have read it; the encoded tokens end in U2
constexpr StepDesc PROG_RADAR_ENC_T[] = {
    {K_TOKENS, 0, 0, 0, 0, B_NONE, B_NONE, B_A, B_NONE, 0, 0, G_NONE, G_NONE, 1},
    {K_POSENC, 0, 1, 0, 0, B_A, B_NONE, B_U3, B_NONE, 0, 0, G_NONE, G_NONE, 0},           // raw xyz from the tile
    {K_LINEAR, 4, -1, 36, 64, B_A, B_NONE, B_X, B_NONE, 1, 0, G_NONE, G_NONE, 1},         // feat.0 (K = RI)
    {K_LINEAR, 2, -1, 256, 256, B_U3, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 0},     // rpe.3
    {K_LINEAR, 5, -1, 64, 128, B_X, B_NONE, B_U2, B_NONE, 1, 0, G_NONE, G_NONE, 1},       // feat.2
    {K_LINEAR, 6, -1, 128, 256, B_U2, B_NONE, B_X, B_NONE, 1, 0, G_NONE, G_NONE, 1},      // feat.4
    {K_LN, 3, -1, 0, 0, B_U1, B_NONE, B_U2, B_X, 0, F_LN_RELU, G_NONE, G_NONE, 1},        // relu(LN(U1)) + feat
    {K_LINEAR, 7, -1, 256, 512, B_U2, B_NONE, B_NONE, B_NONE, 0, 0, G_KV0, G_NONE, 0},
    {K_LINEAR, 8, -1, 256, 512, B_U2, B_NONE, B_NONE, B_NONE, 0, 0, G_KV1, G_NONE, 0},
    {K_LINEAR, 9, -1, 256, 512, B_U2, B_NONE, B_NONE, B_NONE, 0, 0, G_KV2, G_NONE, 0},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// The encoders in two halves, carried by the launches of decoder layers 0 and 1 (16 workgroups of
// 16-row tiles each): as one program they ran 70 us beside a 58 us decoder layer and set its
// duration.  Half A: encoders + K/V of radar layer 0, encoded tokens to global; half B: K/V of
// radar layers 1 and 2.
constexpr StepDesc PROG_RADAR_ENC_A_T[] = {
    {K_TOKENS, 0, 0, 0, 0, B_NONE, B_NONE, B_A, B_NONE, 0, 0, G_NONE, G_NONE, 1},
    {K_POSENC, 0, 1, 0, 0, B_A, B_NONE, B_U3, B_NONE, 0, 0, G_NONE, G_NONE, 0},
    {K_LINEAR, 4, -1, 36, 64, B_A, B_NONE, B_X, B_NONE, 1, 0, G_NONE, G_NONE, 1},
    {K_LINEAR, 2, -1, 256, 256, B_U3, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 0},
    {K_LINEAR, 5, -1, 64, 128, B_X, B_NONE, B_U2, B_NONE, 1, 0, G_NONE, G_NONE, 1},
    {K_LINEAR, 6, -1, 128, 256, B_U2, B_NONE, B_X, B_NONE, 1, 0, G_NONE, G_NONE, 1},
    {K_LN, 3, -1, 0, 0, B_U1, B_NONE, B_U2, B_X, 0, F_LN_RELU, G_RFEAT, G_NONE, 1},
    {K_LINEAR, 7, -1, 256, 512, B_U2, B_NONE, B_NONE, B_NONE, 0, 0, G_KV0, G_NONE, 0},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
constexpr StepDesc PROG_RADAR_ENC_B_T[] = {
    {K_LOAD, 0, 0, 0, 0, B_NONE, B_NONE, B_X, B_NONE, 0, 0, G_RFEAT, G_NONE, 1},
    {K_LINEAR, 8, -1, 256, 512, B_X, B_NONE, B_NONE, B_NONE, 0, 0, G_KV1, G_NONE, 0},
    {K_LINEAR, 9, -1, 256, 512, B_X, B_NONE, B_NONE, B_NONE, 0, 0, G_KV2, G_NONE, 0},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// radar layer pairs (+14*r): 0 attn.in_proj 1 attn.out_proj 2 norm2 3 linear1 4 linear2 5 norm3
//   6 cls.0 7 cls.n1 8 cls.3 9 cls.n4 10 cls.6 11 reg.0 12 reg.2 13 reg.4
// The gate goes first: ~76 % of the queries have no radar return inside their circles (G5 rig: 214 / 219 /
// 69 of 900 have one), and for a tile without a single hit the q projection, the attention and the
// out_proj are x + 0 * (...): skipped (F_IFHIT).  launch_radar_compact orders the rows so that such tiles
// are the rule, not a 0.76^R accident.
constexpr StepDesc PROG_RADAR_LAYER_T[] = {
    {K_RADAR_GATE, 0, 0, 0, 0, B_NONE, B_NONE, B_NONE, B_NONE, 0, 0, G_NONE, G_NONE, 1},     // hit counts of the R rows
    {K_LINEAR, 0, -1, 256, 256, B_X, B_NONE, B_U1, B_NONE, 0, F_SCALEQ | F_IFHIT, G_NONE, G_NONE, 1},  // q projection
    {K_RADAR_ATTN, 0, 0, 0, 0, B_U1, B_NONE, B_U2, B_NONE, 0, F_IFHIT, G_NONE, G_NONE, 1},
    {K_LINEAR, 1, -1, 256, 256, B_U2, B_NONE, B_U3, B_X, 0, F_GATE | F_IFHIT, G_NONE, G_NONE, 1},      // x + gate*out_proj
    {K_LN, 2, -1, 0, 0, B_U3, B_NONE, B_X, B_NONE, 0, 0, G_NONE, G_NONE, 1},                 // rf_norm2
    {K_LINEAR, 3, -1, 256, 512, B_X, B_NONE, B_A, B_NONE, 1, 0, G_NONE, G_NONE, 1},
    {K_LINEAR, 4, -1, 512, 256, B_A, B_NONE, B_U3, B_X, 0, 0, G_NONE, G_NONE, 1},
    {K_LN, 5, -1, 0, 0, B_U3, B_NONE, B_X, B_NONE, 0, 0, G_NONE, G_NONE, 1},                 // rf_norm3
    {K_LINEAR, 6, -1, 256, 256, B_X, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 0},         // final_cls.0
    {K_LINEAR, 11, -1, 256, 256, B_X, B_NONE, B_U2, B_NONE, 1, 0, G_NONE, G_NONE, 1},        // final_reg.0
    {K_LN, 7, -1, 0, 0, B_U1, B_NONE, B_U1, B_NONE, 0, F_LN_RELU, G_NONE, G_NONE, 0},        // in place
    {K_LINEAR, 12, -1, 256, 256, B_U2, B_NONE, B_U3, B_NONE, 1, 0, G_NONE, G_NONE, 1},       // final_reg.2
    {K_LINEAR, 8, -1, 256, 256, B_U1, B_NONE, B_U2, B_NONE, 0, 0, G_NONE, G_NONE, 1},        // final_cls.3
    {K_LN, 9, -1, 0, 0, B_U2, B_NONE, B_U2, B_NONE, 0, F_LN_RELU, G_NONE, G_NONE, 0},        // in place
    // the two 10-column heads: one MFMA sub-tile each, k split over the four waves (K_NARROW)
    {K_NARROW, 13, -1, 256, N_CODE, B_U3, B_U1, B_L, B_NONE, 0, F_PRESYNC, G_NONE, G_NONE, 0},       // final_reg.4 (partial sums: U1)
    {K_NARROW, 10, -1, 256, N_CLS, B_U2, B_U1, B_NONE, B_NONE, 0, F_PRESYNC, G_CLS, G_NONE, 1},      // final_cls.6
    {K_BOXADD, 0, 0, 0, 0, B_L, B_NONE, B_NONE, B_NONE, 0, 0, G_NONE, G_NONE, 1},
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

// PROG_RADAR_TRAIN / PROG_RADAR_ENC_TRAIN: the same step tables as PROG_RADAR / PROG_RADAR_ENC run as the
// FORWARD OF A TRAINING ITERATION (tc_radar_train_fwd_fused): every activation the backward needs is stored on
// the tape as it is produced (the steps' global destinations), the four dropout sites of a fusion layer are
// applied in the epilogues / the attention core (counter-based masks), no step is skipped for hit-free tiles
// (the tape must be complete) and q stays unscaled on the tape (the attention core scales it).
// Backward of one fusion layer for the rows of a tile (HEAD:538-729 reversed; pairs as PROG_RADAR_LAYER_T but
// holding the TRANSPOSED packed weights: dx = dy W is the linear step y' = dy (W^T)^T).  Data gradients are
// row-local; every dY a weight gradient needs is stored as it is produced (the grouped GEMM of train.hip forms
// dW = dY^T X afterwards); LayerNorm parameter gradients leave as per-workgroup sums (atomics).  Across layers
// the gradient of the layer input stays in unit X, the box gradient ({0, 1, 4}) in the row's box record.
constexpr StepDesc PROG_RADAR_BWD_T[] = {
    {K_LOADG, 0, 0, 0, N_CODE, B_NONE, B_NONE, B_U1, B_NONE, 0, F_CARRY, G_NONE, G_NONE, 1},           // 0 d box (+ carry)
    {K_LINEAR, 13, -1, N_CODE, 256, B_U1, B_NONE, B_U2, B_NONE, 0, F_CMASK, G_NONE, G_NONE, 1},          // 1 reg.4^T, relu'(t1)
    {K_LINEAR, 12, -1, 256, 256, B_U2, B_NONE, B_U1, B_NONE, 0, F_CMASK, G_NONE, G_NONE, 1},             // 2 reg.2^T, relu'(t0)
    {K_LINEAR, 11, -1, 256, 256, B_U1, B_NONE, B_U3, B_X, 0, 0, G_NONE, G_NONE, 1},                      // 3 reg.0^T + d(layer out) from above
    {K_LOADG, 0, 0, 0, N_CLS, B_NONE, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 1},                    // 4 d cls
    {K_LINEAR, 10, -1, N_CLS, 256, B_U1, B_NONE, B_U2, B_NONE, 0, 0, G_NONE, G_NONE, 1},                 // 5 cls.6^T
    {K_LN_BWD, 9, -1, 0, 0, B_U2, B_NONE, B_U1, B_X, 0, F_LN_RELU, G_NONE, G_NONE, 1},                   // 6 cls n4 (scratch X)
    {K_LINEAR, 8, -1, 256, 256, B_U1, B_NONE, B_U2, B_NONE, 0, 0, G_NONE, G_NONE, 1},                    // 7 cls.3^T
    {K_LN_BWD, 7, -1, 0, 0, B_U2, B_NONE, B_U1, B_X, 0, F_LN_RELU, G_NONE, G_NONE, 1},                   // 8 cls n1 (scratch X)
    {K_LINEAR, 6, -1, 256, 256, B_U1, B_NONE, B_X, B_U3, 0, 0, G_NONE, G_NONE, 1},                       // 9 cls.0^T: X = d x3
    {K_LN_BWD, 5, -1, 0, 0, B_X, B_NONE, B_X, B_U1, 0, 0, G_NONE, G_NONE, 1},                            // 10 norm3 in place: X = dz (scratch U1)
    {K_MASKCOPY, 0, 0, 3, 0, B_X, B_NONE, B_U3, B_NONE, 0, 0, G_NONE, G_NONE, 1},                        // 11 rf_dropout3: d ffn_out
    {K_LINEAR, 4, -1, 256, 512, B_U3, B_NONE, B_A, B_NONE, 0, F_CMASK | F_CMASK_SCALE, G_NONE, G_NONE, 1},  // 12 linear2^T, relu' and rf_dropout via h
    {K_LINEAR, 3, -1, 512, 256, B_A, B_NONE, B_U3, B_X, 0, 0, G_NONE, G_NONE, 1},                        // 13 linear1^T + dz = d x2
    {K_LN_BWD, 2, -1, 0, 0, B_U3, B_NONE, B_X, B_U1, 0, 0, G_NONE, G_NONE, 1},                           // 14 norm2: X = d x1 (scratch U1)
    {K_MASKCOPY, 0, 0, 1, 0, B_X, B_NONE, B_U1, B_NONE, 0, F_GATE, G_NONE, G_NONE, 1},                   // 15 gate, rf_dropout2: d out_proj(ao)
    {K_LINEAR, 1, -1, 256, 256, B_U1, B_NONE, B_U2, B_NONE, 0, 0, G_NONE, G_NONE, 1},                    // 16 out_proj^T: d ao
    {K_ATTN_BWD, 0, 0, 0, 0, B_U2, B_NONE, B_U1, B_NONE, 0, 0, G_NONE, G_NONE, 1},                       // 17 d q proj (+ dK | dV atomics)
    {K_LINEAR, 0, -1, 256, 256, B_U1, B_NONE, B_X, B_X, 0, F_NOT_LAYER0, G_NONE, G_NONE, 1},             // 18 Wq^T + d x1 = d(layer in)
    {K_END, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
// the dY tensors stored by the steps above (layer r at + r * dy_stride floats) ...
enum DSel : short { D_NONE = -1, D_DBOX = 0, D_DT1, D_DT0, D_DC2, D_DC0, D_DFF, D_DH, D_DPROJ, D_DQP, D_DCLS, D_COUNT };
constexpr short BWD_STORE[19] = {D_DBOX, D_DT1, D_DT0, D_NONE, D_DCLS, D_NONE, D_DC2, D_NONE, D_DC0, D_NONE, D_NONE, D_DFF,
                                 D_DH, D_NONE, D_NONE, D_DPROJ, D_NONE, D_DQP, D_NONE};
enum Program : int { PROG_PROLOGUE = 0, PROG_DECODER, PROG_RADAR_ENC, PROG_RADAR, PROG_RADAR_ENC_A, PROG_RADAR_ENC_B,
                     PROG_RADAR_TRAIN, PROG_RADAR_ENC_TRAIN, PROG_RADAR_BWD };
constexpr bool prog_is_radar(int p) { return p == PROG_RADAR || p == PROG_RADAR_TRAIN; }     // (forward programs)
constexpr bool prog_is_enc_full(int p) { return p == PROG_RADAR_ENC || p == PROG_RADAR_ENC_TRAIN; }
// tape tensors of a fusion layer (layer r: base + r * tape_stride floats) ...
enum TSel : short { T_NONE = -1, T_QP = 0, T_AO, T_X1, T_X2, T_H, T_FF, T_X3, T_C0, T_C1, T_C2, T_C3, T_T0, T_T1, T_TREG,
                    // ... and of the encoders
                    E_U0, E_U1, E_U2, E_POS, E_F0, E_F1, E_F2, E_MEM, T_COUNT };
// tape tensor written by step `si` of the radar layer table / the encoder table (second: a step's other output)
constexpr short RADAR_TAPE[17] = {T_NONE, T_QP, T_AO, T_X1, T_X2, T_H, T_FF, T_X3, T_C0, T_T0, T_C1, T_T1, T_C2, T_C3,
                                  T_TREG, T_NONE, T_NONE};
constexpr short ENC_TAPE[10] = {T_NONE, E_U1, E_F0, E_U2, E_F1, E_F2, E_MEM, T_NONE, T_NONE, T_NONE};
constexpr short ENC_TAPE2[10] = {T_NONE, E_U0, T_NONE, T_NONE, T_NONE, T_NONE, E_POS, T_NONE, T_NONE, T_NONE};
// ... and the tape tensor a step reads (ReLU' mask of a linear step / z of a LayerNorm), second: z's other summand
// or the LayerNorm's ReLU output
constexpr short BWD_TAPE[19] = {T_NONE, T_T1, T_T0, T_NONE, T_NONE, T_NONE, T_C2, T_NONE, T_C0, T_NONE, T_X2, T_NONE,
                                T_H, T_NONE, T_X1, T_NONE, T_NONE, T_NONE, T_NONE};
constexpr short BWD_TAPE2[19] = {T_NONE, T_NONE, T_NONE, T_NONE, T_NONE, T_NONE, T_C3, T_NONE, T_C1, T_NONE, T_FF, T_NONE,
                                 T_NONE, T_NONE, T_NONE, T_NONE, T_NONE, T_NONE, T_NONE};

constexpr int MAX_PAIRS = 48;
constexpr int RADAR_PAIRS = 14;

// what the device reads ...
struct ChainDev {
  int program, M, Q, code, ncls, nlogits, has_next, nlayers;
  int total, early_n;          // resolved steps; leading K_LOAD steps (issued before anything else)
  float* g[G_COUNT]; int g_ld[G_COUNT]; int g_mod[G_COUNT];   // global tensors by GSel
  float qscale; int qpad;
  DropK drop;                  // decoder program, DROP instantiation only (thr 0: off)
  int dbg;                     // STAMPS build only: TRANSCAR_CHAIN_DBG (timing experiments, wrong results)
  int tile_rows;               // requested row-tile height (0: automatic); host side only
  int matrix_path;             // TC_MATRIX_* for the 16-row tiles; host side only
  int last_cls_only;           // radar program: class MLPs of the last layer only; host side only
  // decoder
  const float* ref_in; int ref_mod; float* ref_out; float* box_m;
  CamK cam; unsigned long long* pair_counter;
  const float* pre; const int* premask;     // round 6: the sampling step's pre-gathered level values / visibility masks, or null
  // radar
  const float* tokens; int RI, T, pad_mult;
  const float* ref_last; const float* box_in;
  int cen_from_box;            // the first layer run is not fusion layer 1: gate centre = previous box
  float rmin[TC_MAX_RADAR_LAYERS], rmax[TC_MAX_RADAR_LAYERS];
  float* all_box; int* hits;
  const int* row_perm;         // radar: optional row order (launch_radar_compact), rows stay in their sample
  // training forward (PROG_RADAR_TRAIN / PROG_RADAR_ENC_TRAIN)
  float* tape[T_COUNT]; size_t tape_stride; size_t hits_stride;   // hits of layer r at hits + r * hits_stride (0: M)
  DropK rdrop;                 // radar dropout: seed / thr / scale / tokens_ref (site = 4 * layer + {0..3})
  // backward (PROG_RADAR_BWD), by fusion layer
  const float* bwd_cxy[TC_MAX_RADAR_LAYERS]; int bwd_ldc[TC_MAX_RADAR_LAYERS];   // gate centre of the layer's forward
  const float* bwd_box[TC_MAX_RADAR_LAYERS];                                      // ... and its box (previous level)
  int* range_status;           // f16x2 kernels (MM = 1): tc_head_options.range_status or null
  const float* loss_vals;      // [layers, 2] (cls, bbox) losses of the iteration or null: a layer whose loss is not
                               // finite sends no gradient down (HEAD:915-916 zeroes such a loss), non-finite elements are 0
  float* loss_out;             // or null: loss_vals with NaN -> 0 for the iteration's loss dict (workgroup 0 writes it)
  const DetAcc* detp;          // backward: deterministic accumulation of dgamma / dbeta / dK | dV (common.hpp) -- a
                               // DEVICE copy of the ranges, or null (by value its 12 scalars cost the backward
                               // chain 29 more spilled SGPRs: +7 us per launch in the default mode)
};
// ... plus what only the host-side resolver needs
struct ChainK : ChainDev {
  tc_linear pairs[MAX_PAIRS];
  tc_linear gpairs[MAX_PAIRS];  // backward: gradient destinations of the LayerNorm pairs
  float* dy[D_COUNT]; size_t dy_stride = 0;          // backward: dY stores of layer 0, layer r at + r * dy_stride
  const float* d_cls = nullptr; const float* d_box = nullptr;   // backward: given gradients [layers, M, ncls / code]
  float* dkv[TC_MAX_RADAR_LAYERS] = {nullptr, nullptr, nullptr};
  size_t w16_delta = 0;        // packed16_delta of the packed view: 16-row tiles read their own weight copy
};

constexpr int table_steps(int prog) {
  return (prog == PROG_DECODER ? (int)(sizeof(PROG_DECODER_T) / sizeof(StepDesc))
          : prog == PROG_PROLOGUE ? (int)(sizeof(PROG_PROLOGUE_T) / sizeof(StepDesc))
          : prog_is_enc_full(prog) ? (int)(sizeof(PROG_RADAR_ENC_T) / sizeof(StepDesc))
          : prog == PROG_RADAR_ENC_A ? (int)(sizeof(PROG_RADAR_ENC_A_T) / sizeof(StepDesc))
          : prog == PROG_RADAR_ENC_B ? (int)(sizeof(PROG_RADAR_ENC_B_T) / sizeof(StepDesc))
          : prog == PROG_RADAR_BWD ? (int)(sizeof(PROG_RADAR_BWD_T) / sizeof(StepDesc))
                                     : (int)(sizeof(PROG_RADAR_LAYER_T) / sizeof(StepDesc))) - 1;
}
constexpr int rec_cap(int prog) { return table_steps(prog) * ((prog_is_radar(prog) || prog == PROG_RADAR_BWD) ? TC_MAX_RADAR_LAYERS : 1); }
static_assert(sizeof(BWD_STORE) / sizeof(short) == table_steps(PROG_RADAR_BWD), "backward maps follow the step table");
static_assert(sizeof(RADAR_TAPE) / sizeof(short) == table_steps(PROG_RADAR) && sizeof(ENC_TAPE) / sizeof(short) == table_steps(PROG_RADAR_ENC),
              "tape maps follow the step tables");
inline const StepDesc* prog_table(int prog) {
  return prog == PROG_DECODER ? PROG_DECODER_T
         : prog == PROG_PROLOGUE ? PROG_PROLOGUE_T
         : prog_is_enc_full(prog) ? PROG_RADAR_ENC_T
         : prog == PROG_RADAR_ENC_A ? PROG_RADAR_ENC_A_T
         : prog == PROG_RADAR_ENC_B ? PROG_RADAR_ENC_B_T
         : prog == PROG_RADAR_BWD ? PROG_RADAR_BWD_T : PROG_RADAR_LAYER_T;
}

// runtime view of a linear step
struct LinSpec {
  const float* W; const float* bias; int K, N;
  const float* src; int src_ld;
  const float* src2; int src2_ld;
  float* dst; int dst_ld;
  const float* res; int res_ld;
  const int* gate;
  int act; float scale; int scale_cols;
  float* gdst; int gdst_ld;
  float* gt; int gt_ld, gt_rpb;
  int m0, M;
  int woff;                    // wave w owns column tiles ((w - woff) & 3) + 4 i
  int dbg;
  int sub_on;
  int drop_site;               // DROP instantiations: site + 1 of this step's output dropout (0: none)
  unsigned long long drop_seed; unsigned drop_thr; float drop_scale;
  unsigned long long drop_stride; int drop_q;     // per-sample seeds (DropK::seed_stride / rows_per_sample; 0: off)
  const int* rowg;             // radar: LDS table tile position -> global row for the gdst stores (null: m0 + i)
  int gpre;                    // F_GPRE
  const float* cmask; float cscale;   // F_CMASK: y = cmask[row, col] > 0 ? y * cscale : 0 (cmask is [M, N])
  int dst_pl, res_pl;          // 32-row kernels: the LDS destination / residual holds planes (act_ld4 / act_st4)
  int* range_flag;             // f16x2 kernels: sticky device word, OR-ed with 1 when the step produces a non-finite value
};

// One work item = (64-column output tile, 64-deep k block): 16 x 16-byte weight
// loads per lane (1 KiB contiguous per wave-instruction from the packed layout
// P[tile][k/4][lane][4]: lane n owns output column 64*tile + n), then
// 64 * R/4 MFMAs.  Lane q < R supplies row q of the activations (A operand of
// row group q/4, broadcast to the 16 column blocks by cbsz/abid).
constexpr int KB = 64;
struct WBuf { float4 b[16]; };

#ifdef TC_CHAIN_STAMPS
__device__ long long g_chain_sub[CH_NW_MAX][64];
__device__ int g_sub_step;      // table index of the linear step to dissect
#define SUB_STAMP(slot)                                                                       \
  do {                                                                                        \
    if (blockIdx.x == 100 && (threadIdx.x & 63) == 0 && s.sub_on && (slot) < 64)              \
      g_chain_sub[threadIdx.x >> 6][(slot)] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
#else
#define SUB_STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ void wload(WBuf& wb, const float* w, int nq) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    wb.b[i] = i < nq ? ld4(w + i * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int NG>
struct Acc { f32x4 v[NG][2]; };

template <int NG>
__device__ __forceinline__ void acc_zero(Acc<NG>& a) {
#pragma unroll
  for (int g = 0; g < NG; ++g) { a.v[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; a.v[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
}

// The A operand of one item (R rows x 64 k) is ONE ds_read_b128 per row group: lane
// 4j + q holds row q, k chunk j (4 consecutive k), and the MFMA's abid selects block j
// as the 4-lane group that is broadcast to the 16 column blocks (cbsz = 4).  (Earlier
// builds had every lane < R hold its whole row: 16 reads and 64 VGPRs per item, and the
// weight buffers spilled to AGPRs.)  No guards on the read: the LDS tile is zero filled
// up to the next multiple of 64 wherever K is not one, and the packed weights are zero
// there.  Exact fp32: f32 FMA chains, two k-interleaved accumulators per row group so
// the pipe is issue- not latency-bound.
//
// The 16 weight loads of the NEXT item (1 KiB per wave-instruction) are issued one per
// group of 4*NG MFMAs of the current item.  Issued as one burst they fill the
// vector-memory queue, the wave blocks at the queue and its MFMAs wait behind the
// loads in program order: loads and MFMAs then run back to back (2330 cycles per
// item at R = 4); interleaved they overlap (1410; tools/chainpipe_probe.hip).
template <int NG, bool PF, int J>
struct ItemSteps {
  static __device__ __forceinline__ void run(Acc<NG>& acc, const WBuf& wb, const float4* ar, WBuf& nx,
                                             const float* np) {
    if (PF) nx.b[J] = ld4(np + J * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc.v[g][0] = MFMA44(ar[g].x, wb.b[J].x, acc.v[g][0], J);
      acc.v[g][1] = MFMA44(ar[g].y, wb.b[J].y, acc.v[g][1], J);
      acc.v[g][0] = MFMA44(ar[g].z, wb.b[J].z, acc.v[g][0], J);
      acc.v[g][1] = MFMA44(ar[g].w, wb.b[J].w, acc.v[g][1], J);
    }
    __builtin_amdgcn_sched_barrier(0);
    ItemSteps<NG, PF, J + 1>::run(acc, wb, ar, nx, np);
  }
};
template <int NG, bool PF>
struct ItemSteps<NG, PF, 16> {
  static __device__ __forceinline__ void run(Acc<NG>&, const WBuf&, const float4*, WBuf&, const float*) {}
};

template <int NG, bool PF>
__device__ __forceinline__ void wcompute(Acc<NG>& acc, const WBuf& wb, const float4* ar, WBuf& nx,
                                         const float* np) {
  ItemSteps<NG, PF, 0>::run(acc, wb, ar, nx, np);
}

// lane n holds y[4g + i][64*tile + n] in acc.v[g][*][i]
// Phases instead of a per-row chain of branches: all LDS reads of a kind are issued
// together and waited for once (per row they serialised: ~4 x 150 cycles per tile).
template <int NG, bool DROP>
__device__ __forceinline__ void lin_epilogue(const LinSpec& s, int tile, const Acc<NG>& acc, int lane, float bv) {
  const int col = tile * 64 + lane;
  if (col >= s.N) return;
  const float sc = (col < s.scale_cols) ? s.scale : 1.0f;
  const float bias = s.bias != nullptr ? bv : 0.0f;
  float y[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) y[g][i] = ((acc.v[g][0][i] + acc.v[g][1][i]) + bias) * sc;
  if (s.act == 1) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[g][i] = relu_(y[g][i]);
  } else if (s.act == 2) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[g][i] = sigmoidf_(y[g][i]);
  }
  if constexpr (DROP) {      // backward row chain (a DROP instantiation): ReLU' (and the dropout scale) from the tape
    if (s.cmask != nullptr) {
      float mk[NG][4];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) mk[g][i] = ldg1(s.cmask + (size_t)min(s.m0 + 4 * g + i, s.M - 1) * s.N + col);
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) y[g][i] = mk[g][i] > 0.0f ? y[g][i] * s.cscale : 0.0f;
    }
  }
  if (s.gate != nullptr) {
    int gt_[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) gt_[g][i] = s.gate[4 * g + i];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) if (gt_[g][i] == 0) y[g][i] = 0.0f;
  }
  if (DROP) {
    // nn.Dropout on this step's output (before the residual): element index row * N + col
    if (s.drop_site != 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned row = (unsigned)(s.m0 + 4 * g + i);
          unsigned long long seed = s.drop_seed;
          if (s.drop_q > 0) { const unsigned b = row / (unsigned)s.drop_q; row -= b * (unsigned)s.drop_q; seed += b * s.drop_stride; }
          const unsigned idx = row * (unsigned)s.N + (unsigned)col;
          y[g][i] = drop_keep(seed, (unsigned)(s.drop_site - 1), idx, s.drop_thr) ? y[g][i] * s.drop_scale : 0.0f;
        }
    }
  }
  // F_GPRE (training tape): the value that goes to global memory is the one BEFORE the residual add (kept in
  // registers; a second global-store site in this function trips a hipcc back-end error)
  // -- only in the DROP instantiations (the training programs): the inference kernels are unchanged
  float yp[NG][4];
  if constexpr (DROP) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) yp[g][i] = y[g][i];
  }
  if (s.res != nullptr) {
    float rr[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) rr[g][i] = s.res[(4 * g + i) * s.res_ld + col];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[g][i] += rr[g][i];
  }
  if (s.dst != nullptr) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) s.dst[(4 * g + i) * s.dst_ld + col] = y[g][i];
  }
  if (s.gdst != nullptr) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (s.m0 + 4 * g + i < s.M) {
          const int grow = s.rowg != nullptr ? s.rowg[4 * g + i] : s.m0 + 4 * g + i;
          float v = y[g][i];
          if constexpr (DROP) { if (s.gpre) v = yp[g][i]; }
          stg1(s.gdst + (size_t)grow * s.gdst_ld + col, v);
        }
  }
  if (s.gt != nullptr) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int row0 = s.m0 + 4 * g;
      if (row0 + 3 < s.M && (s.gt_rpb & 3) == 0) {
        const int bb = row0 / s.gt_rpb, q = row0 - bb * s.gt_rpb;
        st4(s.gt + ((size_t)bb * s.N + col) * s.gt_ld + q, make_float4(y[g][0], y[g][1], y[g][2], y[g][3]));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + i;
          if (row < s.M) {
            const int bb = row / s.gt_rpb, q = row - bb * s.gt_rpb;
            stg1(s.gt + ((size_t)bb * s.N + col) * s.gt_ld + q, y[g][i]);
          }
        }
      }
    }
  }
}

// ---- 16-row tiles: v_mfma_f32_16x16x4 on the P16 copy of the weights (pack.hip) --------------------
// A single wave issues a v_mfma_f32_4x4x1 every 12.3 cycles, not 8 (tools/issue_probe.hip; 9.5 with
// four waves per SIMD), a 16x16x4 every 34 of its 32: with all 16 rows of an MFMA in use the larger
// instruction is 1.2-1.45x the matrix rate.  Per item (64 columns x 64 k) still 16 weight loads of
// 1 KiB per wave, refilled in place; 64 MFMAs (4 column sub-tiles x 16 k groups of 4) instead of 256;
// the A operand of a 16-wide k group is one ds_read_b128 (lane 16g + c: row c, k 16 kg + 4g .. + 3).
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
struct Acc16 { f32x4 v[4]; };     // v[j][i] = y[c][64 tile + 16 j + 4 g + i] at lane 16 g + c (TRANSPOSED product, below)

__device__ __forceinline__ float f4_at(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// The product is formed TRANSPOSED (round 3): the weights are the MFMA's first operand, the activations its
// second -- per lane the same two values as before, swapped (a 16x16x4 takes one value of each operand from a
// lane: row / column = lane % 16, k = lane / 16 for both), the same dot products in the same k order, so the
// results are bit-identical -- but the accumulator then holds D'[n][m] = y[m][n]: a lane owns FOUR CONSECUTIVE
// COLUMNS of ONE row instead of four rows of one column, and the epilogue's residual reads, LDS stores and
// global stores are 16-byte accesses (4 per lane and tile instead of 16 scalar ones each; the epilogues were
// 8 % of the decoder chain, tools/r3_ablate.sh).
template <int J>
struct ItemSteps16 {
  // np: wave-uniform address of the next item, lo: the lane's float offset (scalar base + 32-bit
  // vector offset: the address arithmetic of the 16 loads stays on the scalar unit)
  static __device__ __forceinline__ void run(Acc16& acc, WBuf& wb, const float4* ar, const float* np, unsigned lo) {
    const float a = f4_at(ar[J >> 2], J & 3);
    acc.v[0] = MFMA16(wb.b[J].x, a, acc.v[0]);
    acc.v[1] = MFMA16(wb.b[J].y, a, acc.v[1]);
    acc.v[2] = MFMA16(wb.b[J].z, a, acc.v[2]);
    acc.v[3] = MFMA16(wb.b[J].w, a, acc.v[3]);
    __builtin_amdgcn_sched_barrier(0);
    wb.b[J] = ld4(np + (size_t)(lo + J * 256u));   // fragment J of the next item, into fragment J's registers
    __builtin_amdgcn_sched_barrier(0);
    ItemSteps16<J + 1>::run(acc, wb, ar, np, lo);
  }
};
template <>
struct ItemSteps16<16> {
  static __device__ __forceinline__ void run(Acc16&, WBuf&, const float4*, const float*, unsigned) {}
};

// lin_epilogue for the 16x16 accumulator layout: lane 16g + c holds row c, columns 64 tile + 16j + 4g .. + 3,
// j = 0..3.  bvl: the bias of column 64 tile + lane (one coalesced load under the MFMAs).  It is added by the
// matrix pipe: bvl IS the first operand of a rank-1 update -- lane 16g + c supplies A[m' = c][kk = g] =
// bias[16 g + c] -- and with B[kk][n'] = (kk == j) the product is bias[16 j + m'] in every column n': one more
// v_mfma per sub-tile, D = fma(bias, 1, acc) + 0 + 0 + 0, the same single rounding as the scalar add it replaces
// (16 crossbar moves per tile otherwise, or 16 live registers).  N is a multiple of 4 wherever a step has an LDS destination or a residual
// (256 / 512 / 24 / 64 / 128); the 3-column head of the prologue only stores to global memory, element by element.
template <bool PL> __device__ __forceinline__ float4 act_ld4(const float* row, int col);      // (defined with the f16 planes below)
template <bool PL> __device__ __forceinline__ void act_st4(float* row, int col, const float4& v);
// NJ: sub-tiles of 16 columns in the wave's tile (4: the 64-column tiles of the 16-row kernels; 2: the 32-column tiles
// of the 32-row kernels); colbase: the tile's first column; bvl: the bias of column colbase + lane (lanes >= 16 NJ: unused).
// The rows are s.src / dst / res / gate / rowg / m0 + 0 .. 15: the 32-row kernels pass a view of their second row group.
// NRG (32-row kernels: 2): the accumulators of NRG row groups of 16 rows go through the phases TOGETHER -- accv[NJ rg + jj]
// is sub-tile jj of rows 16 rg .. 16 rg + 15 -- so that every LDS read of the tile is in flight before the first
// LDS store (one call per row group serialised them: the compiler cannot tell the stores from the next group's reads).
// bias4 (32-row kernels): the lane's own four bias values per sub-tile, loaded as float4 at the start of the tile -- added
// here with one rounding each, exactly what the rank-1 MFMA of the 16-row kernels does with `bvl` (which those keep:
// one register instead of 4 NJ across their item loop).
template <bool DROP, int NJ = 4, bool PL = false, int NRG = 1>
__device__ __forceinline__ void lin_epilogue16(const LinSpec& s, int colbase, const f32x4 (&accv)[NJ * NRG], int lane, float bvl,
                                               const float4* bias4 = nullptr) {
  constexpr int V = NJ * NRG;
  const int c = lane & 15, g = lane >> 4;
  const int colb = colbase + 4 * g;              // + 16 jj
  float y[V][4];
  f32x4 ab[V];
#pragma unroll
  for (int v = 0; v < V; ++v) ab[v] = accv[v];
  if (s.bias != nullptr) {
    if (bias4 != nullptr) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 b4 = bias4[v % NJ];
        ab[v][0] += b4.x; ab[v][1] += b4.y; ab[v][2] += b4.z; ab[v][3] += b4.w;
      }
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) ab[v] = MFMA16(bvl, g == (v % NJ) ? 1.0f : 0.0f, ab[v]);
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float sc = (colb + 16 * (v % NJ) < s.scale_cols) ? s.scale : 1.0f;     // scale_cols is 0 or 256
#pragma unroll
    for (int i = 0; i < 4; ++i) y[v][i] = ab[v][i] * sc;
  }
  if (s.range_flag != nullptr) {
    // the f16 planes overflow at |activation| >= 4.19e6 / |weight| >= 65504: the product then is inf or NaN -- one sticky
    // word says so (tc_head_options.range_status)
    bool bad = false;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) bad |= !(fabsf(y[v][i]) <= 3.0e38f);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(s.range_flag, 1);
  }
  if (s.act == 1) {
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[v][i] = relu_(y[v][i]);
  } else if (s.act == 2) {
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[v][i] = sigmoidf_(y[v][i]);
  }
  if (s.gate != nullptr) {
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg)
      if (s.gate[16 * rg + c] == 0) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) y[NJ * rg + jj][i] = 0.0f;
      }
  }
  if (DROP) {
    if (s.drop_site != 0) {
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) {
        unsigned row = (unsigned)(s.m0 + 16 * rg + c);
        unsigned long long seed = s.drop_seed;
        if (s.drop_q > 0) { const unsigned b = row / (unsigned)s.drop_q; row -= b * (unsigned)s.drop_q; seed += b * s.drop_stride; }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          // the lane's four consecutive columns of sub-tile jj: one hash (N and colb are multiples of 4)
          const unsigned m = drop_keep4(seed, (unsigned)(s.drop_site - 1), row * (unsigned)s.N + (unsigned)(colb + 16 * jj), s.drop_thr);
#pragma unroll
          for (int i = 0; i < 4; ++i) y[NJ * rg + jj][i] = ((m >> i) & 1u) ? y[NJ * rg + jj][i] * s.drop_scale : 0.0f;
        }
      }
    }
  }
  if (s.res != nullptr) {
    float4 rr[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int col = colb + 16 * (v % NJ), row = 16 * (v / NJ) + c;
      rr[v] = col >= s.N ? make_float4(0.f, 0.f, 0.f, 0.f)
              : (PL && s.res_pl) ? act_ld4<PL>(s.res + row * s.res_ld, col)
                                 : *reinterpret_cast<const float4*>(s.res + row * s.res_ld + col);
    }
#pragma unroll
    for (int v = 0; v < V; ++v) { y[v][0] += rr[v].x; y[v][1] += rr[v].y; y[v][2] += rr[v].z; y[v][3] += rr[v].w; }
  }
  const bool vec = (s.N & 3) == 0;               // wave-uniform
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int col = colb + 16 * (v % NJ), rloc = 16 * (v / NJ) + c;
    if (col >= s.N) continue;
    const float4 y4 = make_float4(y[v][0], y[v][1], y[v][2], y[v][3]);
    if (s.dst != nullptr) {                                                                // (N % 4 == 0 here)
      if (PL && s.dst_pl) act_st4<PL>(s.dst + rloc * s.dst_ld, col, y4);
      else *reinterpret_cast<float4*>(s.dst + rloc * s.dst_ld + col) = y4;
    }
    if (s.gdst != nullptr && s.m0 + rloc < s.M) {
      const int grow = s.rowg != nullptr ? s.rowg[rloc] : s.m0 + rloc;
      float* gp = s.gdst + (size_t)grow * s.gdst_ld + col;
      if (vec && (s.gdst_ld & 3) == 0) st4(gp, y4);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (col + i < s.N) stg1(gp + i, y[v][i]);
      }
    }
    if (s.gt != nullptr && s.m0 + rloc < s.M) {
      // the transposed V of the next layer's attention, [b][column][query]: the 16 lanes of a group write 16
      // consecutive queries of one column
      const int row = s.m0 + rloc;
      const int bb = row / s.gt_rpb, q = row - bb * s.gt_rpb;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (col + i < s.N) stg1(s.gt + ((size_t)bb * s.N + col + i) * s.gt_ld + q, y[v][i]);
    }
  }
  // Global stores read their data registers late and gfx9 tracks that with vmcnt, which retires in
  // order: hipcc, seeing stores that MAY have been issued (this epilogue sits in the item loop), put
  // s_waitcnt vmcnt(1) / vmcnt(0) in front of the first operand reads of EVERY item -- i.e. every item
  // waited for the weight loads issued at the end of the previous one (~600 cycles of 2600).  An
  // explicit drain here, in the two steps of a layer that store, tells it nothing is pending there.
  if (s.gdst != nullptr || s.gt != nullptr) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
}

// linear_step for R = 16 (same contract: w0 may arrive preloaded with the step's first item, the last
// item fetches `next_first`): ONE weight buffer refilled in place -- fragment J of the next item is
// loaded into fragment J's registers right behind its four MFMAs (the matrix pipe reads its operands at
// issue, the load writes them hundreds of cycles later; 64 MFMAs = 2048 cycles per item leave every
// fragment an item time to arrive).  Without a second buffer the kernel fits 256 registers: two
// workgroups per CU.
template <bool DROP, bool SRC2, typename SpecFn>
__device__ __forceinline__ bool linear_step16(const LinSpec& s, WBuf& w0, bool preloaded, const float* next_first,
                                              SpecFn make_spec, int step_idx) {
  const int lane = threadIdx.x & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - s.woff) & (CH_NW - 1);
  const int ntiles = (s.N + 63) >> 6;
  const int kpad = (s.K + 63) & ~63;
  const int nkb = kpad / KB;
  const int my_tiles = wave < ntiles ? (ntiles - wave + CH_NW - 1) / CH_NW : 0;
  const int nitems = my_tiles * nkb;
  // lane 16g + c reads row c, the 4 consecutive k of group g in every 16-wide k group
  const float* arow = s.src + (lane & 15) * s.src_ld + 4 * (lane >> 4);
  const float* a2row = SRC2 && s.src2 ? s.src2 + (lane & 15) * s.src2_ld + 4 * (lane >> 4) : nullptr;
  const float* wbase = s.W + (size_t)wave * 64 * kpad;       // wave-uniform; the lane adds lo
  const unsigned lo = 4u * lane;
  const size_t tile_stride = (size_t)CH_NW * 64 * kpad;
  Acc16 acc;
  float bvl = 0.f;
#ifdef TC_CHAIN_STAMPS
  constexpr bool stamp_items = true;
#else
  constexpr bool stamp_items = false;
#endif
  SUB_STAMP(1);
  if (CHAIN_DBG(s.dbg) & 32) return false;
  if (!preloaded) {
    wload(w0, wbase + lo, 16);
    __builtin_amdgcn_sched_barrier(0);
  }
  SUB_STAMP(2);
  // Nothing but this step's first weight item may be in flight when the item loop starts: a global
  // store of an earlier step (LayerNorm -> hs, ...) still reading its data registers is enough for
  // hipcc to put s_waitcnt vmcnt(1) / vmcnt(0) in front of the operand reads of EVERY item (vmcnt
  // retires in order: each item then waited for the weight loads issued at the end of the one before).
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
  int tt = 0, kb = 0;
  const float* wcur = wbase;
#pragma unroll 1
  for (int it = 0; it < nitems; ++it) {
    const float* np = wcur;
    int nt = tt, nk = kb;
    if (++nk == nkb) { nk = 0; ++nt; np = wbase + (size_t)nt * tile_stride; }
    else np += (KB / 4) * 256;
    const bool last = it + 1 >= nitems;
    const float* nload = last ? (next_first != nullptr ? next_first : wbase) : np;
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(20 + 5 * kb);
    if (kb == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc.v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the tile's 64 biases, one per lane, in flight under the MFMAs; an unconditional load (see linear_step)
      const float* bsrc = s.bias != nullptr ? s.bias : s.W;
      bvl = ldg1(bsrc + min((wave + tt * CH_NW) * 64 + lane, s.N - 1));
    }
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(22 + 5 * kb);
    float4 ar[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      ar[kg] = *reinterpret_cast<const float4*>(arow + kb * KB + 16 * kg);
      if (SRC2 && a2row != nullptr) ar[kg] = add4(ar[kg], *reinterpret_cast<const float4*>(a2row + kb * KB + 16 * kg));
    }
    ItemSteps16<0>::run(acc, w0, ar, nload, lo);
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(23 + 5 * kb);
    if (kb == nkb - 1) {
      int tile = wave + tt * CH_NW;
      int sidx = step_idx;
      asm volatile("" : "+s"(tile), "+s"(sidx));         // see linear_step: the epilogue rebuilds its view
      if (!(CHAIN_DBG(s.dbg) & 1)) {
        const LinSpec e = make_spec(sidx);
        lin_epilogue16<DROP>(e, tile * 64, acc.v, lane, bvl);
      }
    }
    wcur = np; tt = nt; kb = nk;
    __builtin_amdgcn_sched_barrier(0);
  }
  return next_first != nullptr;
}

// ---- 16-row tiles on the f16 MATRIX CORES, fp32-accurate (round 4) ---------------------------------------
// The f32 MFMA is the vector pipe (64 flop / clk / SIMD, and it holds the VALU's issue slots); the f16 / bf16
// MFMAs run on the matrix cores at 16 x that rate.  Operands as TWO f16 planes each,
//     x = (x1 + 2^-11 x2') / s,   x1 = f16(s x),  x2' = f16((s x - x1) 2^11)        (round to nearest)
// (s = 2^-6: |x| < 4.19e6 cannot overflow; the residual is exact in fp32 and the 2^11 keeps it a normal f16, so
// the pair represents s x to <= 2^-24 |s x| wherever |s x| >= 2^-14 -- one ulp of fp32 -- and to < 1.5e-11
// absolute below that), weights likewise with s = 1 (pack.hip PH).  x w = x1 w1 + 2^-11 (x1 w2' + x2' w1)
// + 2^-22 x2' w2': three v_mfma_f32_16x16x32_f16 per 32 k (products of f16 values are exact in the fp32
// accumulator), the last term (<= 2^-24 |x w|) is dropped.  Measured against fp64 on a 912 x 256 x 256
// product (tools/split_mfma_probe.hip, profiles/r4_split_mfma_probe.txt): rms error 2.6e-6 vs 6.2e-6 for the
// fp32 FMA chain of v_mfma_f32_16x16x4_f32 (the matrix core adds its 32 products before it rounds).
// The item loop then is bound by the weight stream (~57 B / clk / CU), not by the pipe: 2 226 cycles per item
// against 4 702, which is why the planes are f16 x 2 (4 bytes per weight, as fp32) and not bf16 x 3 (6 bytes:
// 3 429 cycles, same probe).  The activations are split IN REGISTERS per item -- the LDS of a 16-row tile has
// no room for planes at two workgroups per CU; ~30 VALU instructions per 12 MFMAs, and the f16 MFMA leaves
// the VALU half of its issue slots.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
constexpr float H_ACT_SCALE = 1.0f / 64.0f;      // s
constexpr float H_LO_SCALE = 2048.0f;            // 2^11
struct Acc16H { f32x4 hi[4]; f32x4 lo[4]; };     // hi: s x1 w1; lo: 2^11 s (x1 w2 + x2 w1)

__device__ __forceinline__ unsigned pk_h2(float a, float b) {
  const f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// Two scaled values t = s x -> their halves of the two planes: hi = f16(t), lo = f16((t - hi) 2^11).  Round 6: written so
// that hipcc emits FIVE vector instructions per pair instead of ten -- v_cvt_pk_f16_f32; t - hi as v_fma_mix_f32 reading
// the half straight out of the packed register (the multiplier -1 made opaque: a literal is canonicalised to
// v_cvt_f32_f16 + v_sub_f32, and without the opaque `hi` the halves are converted a second time, one by one); the scaling and
// the conversion of lo as v_fma_mixlo / mixhi_f16 (r 2^11 + 0: exact in fp32, one rounding to f16 -- as v_mul_f32 +
// v_cvt_pk_f16_f32 had).  Same values bit for bit (t - hi is exact either way; never -0).  The row-local steps of the
// 16- / 32-row chains are VALU-bound -- a fifth of a decoder layer's vector instructions were these splits
// (profiles/r6_valu_by_step.txt).
__device__ __forceinline__ void split_t2(float t0, float t1, unsigned& hi, unsigned& lo) {
  hi = pk_h2(t0, t1);
  asm("" : "+v"(hi));
  float m1 = -1.0f;
  asm("" : "+s"(m1));
  const f16x2 h = __builtin_bit_cast(f16x2, hi);
  const float r0 = __builtin_fmaf((float)h[0], m1, t0), r1 = __builtin_fmaf((float)h[1], m1, t1);
  const f16x2 l = {(_Float16)__builtin_fmaf(r0, H_LO_SCALE, 0.0f), (_Float16)__builtin_fmaf(r1, H_LO_SCALE, 0.0f)};
  lo = __builtin_bit_cast(unsigned, l);
}
// 8 consecutive k of one row -> the lane's operand fragments of the two planes
__device__ __forceinline__ void split_h(const float4& a, const float4& b, float4& p1, float4& p2) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned q1[4], q2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_t2(x[2 * i] * H_ACT_SCALE, x[2 * i + 1] * H_ACT_SCALE, q1[i], q2[i]);
  p1 = make_float4(__uint_as_float(q1[0]), __uint_as_float(q1[1]), __uint_as_float(q1[2]), __uint_as_float(q1[3]));
  p2 = make_float4(__uint_as_float(q2[0]), __uint_as_float(q2[1]), __uint_as_float(q2[2]), __uint_as_float(q2[3]));
}

// ---- the 32-row kernels keep their activations in LDS AS PLANES (round 5) -------------------------------------------
// With 8 waves per workgroup every wave would split all 32 rows of an item for itself (176 vector instructions per 24
// MFMAs: measured, the item loop then is VALU bound and SLOWER than two 16-row workgroups).  So the split happens where a
// value is WRITTEN, once: the four activation units (X, U1..U3, A) of a 32-row workgroup hold, per row and per 8
// consecutive columns, 16 bytes of hi plane and 16 bytes of lo plane -- the 32 bytes the 8 floats would take, and exactly
// the two operand fragments a lane of v_mfma_f32_16x16x32_f16 needs (two ds_read_b128, no arithmetic in the item loop).
// Every other reader (LayerNorm, residual adds, the narrow heads, the attention's query) reconstructs
// x = 64 hi + lo / 32 (one rounding; the pair carries s x to 2^-24 relative, so a store / load round trip moves a value
// by at most one unit in its last place).  The logits / box / centre side buffers stay fp32.
template <bool PL>
__device__ __forceinline__ float4 act_ld4(const float* row, int col) {       // col % 4 == 0; `row`: start of the LDS row
  if constexpr (!PL) {
    return *reinterpret_cast<const float4*>(row + col);
  } else {
    const char* p = reinterpret_cast<const char*>(row) + (col >> 3) * 32 + (col & 7) * 2;
    const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + 16);
    const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
    const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
    // x = (hi + lo 2^-11) / s: the sum as ONE v_fma_mix_f32 with both halves read as f16, then the exact power-of-two
    // un-scaling -- two instructions per value (round 6; as fma(lo, 2^-11 / s, hi / s) hipcc needed three: v_cvt_f32_f16,
    // v_mul_f32, v_fma_mix_f32).  Scaling by 2^6 commutes with the one rounding: same bits.
    constexpr float US = 1.0f / H_ACT_SCALE, UL = 1.0f / H_LO_SCALE;
    return make_float4(fmaf((float)l0[0], UL, (float)h0[0]) * US, fmaf((float)l0[1], UL, (float)h0[1]) * US,
                       fmaf((float)l1[0], UL, (float)h1[0]) * US, fmaf((float)l1[1], UL, (float)h1[1]) * US);
  }
}
template <bool PL>
__device__ __forceinline__ void act_st4(float* row, int col, const float4& v) {
  if constexpr (!PL) {
    *reinterpret_cast<float4*>(row + col) = v;
  } else {
    const float t0 = v.x * H_ACT_SCALE, t1 = v.y * H_ACT_SCALE, t2 = v.z * H_ACT_SCALE, t3 = v.w * H_ACT_SCALE;
    uint2 h, l;
    split_t2(t0, t1, h.x, l.x);
    split_t2(t2, t3, h.y, l.y);
    char* p = reinterpret_cast<char*>(row) + (col >> 3) * 32 + (col & 7) * 2;
    *reinterpret_cast<uint2*>(p) = h;
    *reinterpret_cast<uint2*>(p + 16) = l;
  }
}

// The epilogue of a 32 x 32 tile of the 32-row kernels (both row groups, both 16-column sub-tiles: v = 2 rg + jj is rows
// 16 rg + c, columns colbase + 16 jj + 4 g ..).  Same arithmetic, in the same order, as lin_epilogue16 behind the
// hi / lo recombination -- but carried out on t = s x (s = H_ACT_SCALE, a power of two: every step commutes with it
// bit for bit), which is what the planes store: the un-scaling of the accumulators, of a residual read from planes and
// the re-scaling in front of the split cancel (3 multiplies per element less).  The tile lies inside N (the caller
// sends ragged tiles -- the 24 attention logits -- through lin_epilogue16): no per-lane column tests, so all four
// residual reads are in flight together, ahead of the arithmetic, and nothing between them and the stores branches on
// a lane.
struct Acc32H;
template <bool DROP>
__device__ __forceinline__ void lin_epilogue32(const LinSpec& s, int colbase, const f32x4 (&thi)[4], const f32x4 (&tlo)[4], int lane,
                                               const float4 (&bias4)[2]) {
  const int c = lane & 15, g = lane >> 4;
  const int colb = colbase + 4 * g;
  constexpr float US = 1.0f / H_ACT_SCALE;
  // residual: raw reads first
  uint2 rh[4], rl[4];
  float4 rf[4];
  const bool res_planes = s.res != nullptr && s.res_pl, res_f32 = s.res != nullptr && !s.res_pl;
  if (res_planes) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int col = colb + 16 * (v & 1), row = 16 * (v >> 1) + c;
      const char* p = reinterpret_cast<const char*>(s.res + row * s.res_ld) + (col >> 3) * 32 + (col & 7) * 2;
      rh[v] = *reinterpret_cast<const uint2*>(p); rl[v] = *reinterpret_cast<const uint2*>(p + 16);
    }
  } else if (res_f32) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      rf[v] = *reinterpret_cast<const float4*>(s.res + (16 * (v >> 1) + c) * s.res_ld + colb + 16 * (v & 1));
  }
  float t[4][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) t[v][i] = fmaf(tlo[v][i], 1.0f / H_LO_SCALE, thi[v][i]);
  if (s.bias != nullptr) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 b4 = bias4[v & 1];
      t[v][0] = fmaf(b4.x, H_ACT_SCALE, t[v][0]); t[v][1] = fmaf(b4.y, H_ACT_SCALE, t[v][1]);
      t[v][2] = fmaf(b4.z, H_ACT_SCALE, t[v][2]); t[v][3] = fmaf(b4.w, H_ACT_SCALE, t[v][3]);
    }
  }
  if (colbase < s.scale_cols) {                    // scale_cols is 0 or 256, colbase a multiple of 32: wave-uniform
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[v][i] *= s.scale;
  }
  if (s.range_flag != nullptr) {
    bool bad = false;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) bad |= !(fabsf(t[v][i]) <= 3.0e38f * H_ACT_SCALE);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(s.range_flag, 1);
  }
  if (s.act == 1) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[v][i] = relu_(t[v][i]);
  } else if (s.act == 2) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[v][i] = sigmoidf_(t[v][i] * US) * H_ACT_SCALE;
  }
  if (s.gate != nullptr) {
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
      if (s.gate[16 * rg + c] == 0) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int i = 0; i < 4; ++i) t[2 * rg + jj][i] = 0.0f;
      }
  }
  if (DROP) {
    if (s.drop_site != 0) {
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        unsigned row = (unsigned)(s.m0 + 16 * rg + c);
        unsigned long long seed = s.drop_seed;
        if (s.drop_q > 0) { const unsigned b = row / (unsigned)s.drop_q; row -= b * (unsigned)s.drop_q; seed += b * s.drop_stride; }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const unsigned m = drop_keep4(seed, (unsigned)(s.drop_site - 1), row * (unsigned)s.N + (unsigned)(colb + 16 * jj), s.drop_thr);
#pragma unroll
          for (int i = 0; i < 4; ++i) t[2 * rg + jj][i] = ((m >> i) & 1u) ? t[2 * rg + jj][i] * s.drop_scale : 0.0f;
        }
      }
    }
  }
  if (res_planes) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const f16x2 h0 = __builtin_bit_cast(f16x2, rh[v].x), h1 = __builtin_bit_cast(f16x2, rh[v].y);
      const f16x2 l0 = __builtin_bit_cast(f16x2, rl[v].x), l1 = __builtin_bit_cast(f16x2, rl[v].y);
      t[v][0] += fmaf((float)l0[0], 1.0f / H_LO_SCALE, (float)h0[0]); t[v][1] += fmaf((float)l0[1], 1.0f / H_LO_SCALE, (float)h0[1]);
      t[v][2] += fmaf((float)l1[0], 1.0f / H_LO_SCALE, (float)h1[0]); t[v][3] += fmaf((float)l1[1], 1.0f / H_LO_SCALE, (float)h1[1]);
    }
  } else if (res_f32) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      t[v][0] = fmaf(rf[v].x, H_ACT_SCALE, t[v][0]); t[v][1] = fmaf(rf[v].y, H_ACT_SCALE, t[v][1]);
      t[v][2] = fmaf(rf[v].z, H_ACT_SCALE, t[v][2]); t[v][3] = fmaf(rf[v].w, H_ACT_SCALE, t[v][3]);
    }
  }
  if (s.dst != nullptr) {
    if (s.dst_pl) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int col = colb + 16 * (v & 1), rloc = 16 * (v >> 1) + c;
        uint2 h, l;
        split_t2(t[v][0], t[v][1], h.x, l.x);
        split_t2(t[v][2], t[v][3], h.y, l.y);
        char* p = reinterpret_cast<char*>(s.dst + rloc * s.dst_ld) + (col >> 3) * 32 + (col & 7) * 2;
        *reinterpret_cast<uint2*>(p) = h;
        *reinterpret_cast<uint2*>(p + 16) = l;
      }
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<float4*>(s.dst + (16 * (v >> 1) + c) * s.dst_ld + colb + 16 * (v & 1)) =
            make_float4(t[v][0] * US, t[v][1] * US, t[v][2] * US, t[v][3] * US);
    }
  }
  if (s.gdst != nullptr || s.gt != nullptr) {
    const bool vec = (s.gdst_ld & 3) == 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int col = colb + 16 * (v & 1), rloc = 16 * (v >> 1) + c;
      if (s.m0 + rloc >= s.M) continue;
      const float4 y4 = make_float4(t[v][0] * US, t[v][1] * US, t[v][2] * US, t[v][3] * US);
      if (s.gdst != nullptr) {
        const int grow = s.rowg != nullptr ? s.rowg[rloc] : s.m0 + rloc;
        float* gp = s.gdst + (size_t)grow * s.gdst_ld + col;
        if (vec) st4(gp, y4);
        else { stg1(gp, y4.x); stg1(gp + 1, y4.y); stg1(gp + 2, y4.z); stg1(gp + 3, y4.w); }
      }
      if (s.gt != nullptr) {
        const int row = s.m0 + rloc;
        const int bb = row / s.gt_rpb, q = row - bb * s.gt_rpb;
        float* tp = s.gt + ((size_t)bb * s.N + col) * s.gt_ld + q;
        stg1(tp, y4.x); stg1(tp + s.gt_ld, y4.y); stg1(tp + 2 * (size_t)s.gt_ld, y4.z); stg1(tp + 3 * (size_t)s.gt_ld, y4.w);
      }
    }
    // (no vmcnt(0) drain here, unlike lin_epilogue16: the 32-row item loop loads into the buffer it does not compute
    // from, hipcc's own waits behind these stores are no worse than a drain -- measured +0.3 % without it)
  }
}

// One HALF item (64 columns x 32 k): 8 fragments of 1 KiB -- (sub-tile j, plane p) at wb.b[2 j + p] --, 12 MFMAs.
// TWO register buffers of one half item each (b[0..7], b[8..15]): while half item i issues its MFMAs from one, the
// fragments of half item i + 1 are loaded into the other -- group by group behind the MFMAs of the same group, so
// that a register is overwritten a whole half item (12 MFMAs of this wave) after its last read.
// Round 3's f32 loop refills a fragment IN PLACE, right behind its MFMAs.  The first build of this loop did the same and
// the radar program returned a few wrong rows per launch at two workgroups per CU.  Round 5 found out why
// (profiles/r5_refill_hazard.txt): the in-place refill itself is EXACT (tools/refill_hazard_probe.hip), but while a wave
// has loads landing in registers its matrix-core MFMAs have just read, a NEIGHBOURING wave on the SIMD gets 0 in lanes
// 48..63 of the low result of `v_pk_mul_f32 ... op_sel:[0,1]` -- the form hipcc's SLP vectoriser gave the radar attention's
// pv * v products (tools/pk_hazard_probe.hip reproduces it outside the library).  The library avoids both sides: no such
// instruction is formed (-fno-slp-vectorize, tools/isa_lint.py in the test suite), and the fragments of half item i + 1
// go to the OTHER buffer, a whole half item (12 MFMAs of this wave) away from the registers' last read -- beside this loop
// the probe's victims saw nothing in 2.5e9 executions.
template <int J, int BUF>
struct ItemSteps16H {
  static __device__ __forceinline__ void run(Acc16H& acc, WBuf& wb, const float4& x1, const float4& x2, const float* np,
                                             unsigned lo) {
    constexpr int C0 = 8 * BUF + 2 * J, N0 = 8 * (1 - BUF) + 2 * J;
    acc.lo[J] = MFMA16H(wb.b[C0 + 1], x1, acc.lo[J]);
    acc.lo[J] = MFMA16H(wb.b[C0], x2, acc.lo[J]);
    acc.hi[J] = MFMA16H(wb.b[C0], x1, acc.hi[J]);
    __builtin_amdgcn_sched_barrier(0);
    wb.b[N0] = ld4(np + (size_t)(lo + (2 * J) * 256u));
    wb.b[N0 + 1] = ld4(np + (size_t)(lo + (2 * J + 1) * 256u));
    __builtin_amdgcn_sched_barrier(0);
    ItemSteps16H<J + 1, BUF>::run(acc, wb, x1, x2, np, lo);
  }
};
template <int BUF>
struct ItemSteps16H<4, BUF> {
  static __device__ __forceinline__ void run(Acc16H&, WBuf&, const float4&, const float4&, const float*, unsigned) {}
};

// linear_step16 on the PH copy of the weights: same contract (w0 may arrive preloaded with the step's first HALF item
// in b[0..7], the last half item fetches `next_first`), same accumulator layout, same epilogue.  A step has an even
// number of half items (K is padded to 64), so every step starts and ends on buffer 0.
template <bool DROP, typename SpecFn>
__device__ __forceinline__ bool linear_step16h(const LinSpec& s, WBuf& w0, bool preloaded, const float* next_first,
                                               SpecFn make_spec, int step_idx) {
  const int lane = threadIdx.x & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - s.woff) & (CH_NW - 1);
  const int ntiles = (s.N + 63) >> 6;
  const int kpad = (s.K + 63) & ~63;
  const int nhalf = kpad / 32;
  const int my_tiles = wave < ntiles ? (ntiles - wave + CH_NW - 1) / CH_NW : 0;
  const int nitems = my_tiles * nhalf;
  // lane 16g + c reads row c, the 8 consecutive k of group g in every 32-wide half item
  const float* arow = s.src + (lane & 15) * s.src_ld + 8 * (lane >> 4);
  const float* wbase = s.W + (size_t)wave * 64 * kpad;       // wave-uniform; the lane adds lo
  const unsigned lo = 4u * lane;
  const size_t tile_stride = (size_t)CH_NW * 64 * kpad;
  Acc16H acc;
  float bvl = 0.f;
  SUB_STAMP(1);
  if (CHAIN_DBG(s.dbg) & 32) return false;
  if (!preloaded) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w0.b[i] = ld4(wbase + (size_t)(lo + i * 256u));
    __builtin_amdgcn_sched_barrier(0);
  }
  SUB_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): see linear_step16
  int tt = 0, kh = 0;
  const float* wcur = wbase;
#pragma unroll 1
  for (int it = 0; it < nitems; it += 2) {
    // ---- even half item: buffer 0 -> MFMAs, buffer 1 <- the next (odd) half item of the same tile
    {
      if (kh == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc.hi[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.lo[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* bsrc = s.bias != nullptr ? s.bias : s.W;
        bvl = ldg1(bsrc + min((wave + tt * CH_NW) * 64 + lane, s.N - 1));
      }
      float4 x1, x2;
      split_h(*reinterpret_cast<const float4*>(arow + kh * 32), *reinterpret_cast<const float4*>(arow + kh * 32 + 4), x1, x2);
      ItemSteps16H<0, 0>::run(acc, w0, x1, x2, wcur + 8 * 256, lo);
      wcur += 8 * 256; ++kh;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- odd half item: buffer 1 -> MFMAs, buffer 0 <- the half item after it (next tile, or the wave's next step; at
    // the end of a run of steps a dead re-read of this step's first half item: a load behind a branch would make
    // hipcc drain the stream at the join)
    {
      const float* np = wcur;
      int nt = tt, nk = kh;
      if (++nk == nhalf) { nk = 0; ++nt; np = wbase + (size_t)nt * tile_stride; }
      else np += 8 * 256;
      const bool last = it + 2 >= nitems;
      const float* nload = last ? (next_first != nullptr ? next_first : wbase) : np;
      float4 x1, x2;
      split_h(*reinterpret_cast<const float4*>(arow + kh * 32), *reinterpret_cast<const float4*>(arow + kh * 32 + 4), x1, x2);
      ItemSteps16H<0, 1>::run(acc, w0, x1, x2, nload, lo);
      if (kh == nhalf - 1) {
        int tile = wave + tt * CH_NW;
        int sidx = step_idx;
        asm volatile("" : "+s"(tile), "+s"(sidx));         // see linear_step: the epilogue rebuilds its view
        if (!(CHAIN_DBG(s.dbg) & 1)) {
          Acc16 y;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              y.v[j][i] = fmaf(acc.lo[j][i], 1.0f / (H_LO_SCALE * H_ACT_SCALE), acc.hi[j][i] * (1.0f / H_ACT_SCALE));
          const LinSpec e = make_spec(sidx);
          lin_epilogue16<DROP>(e, tile * 64, y.v, lane, bvl);
        }
      }
      wcur = np; tt = nt; kh = nk;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return next_first != nullptr;
}


// ---- 32-row tiles on the f16 matrix cores (round 5) --------------------------------------------------------
// What bounds the 16-row f16 loop is the weight stream: every 16-row workgroup pulls the layer's 3.18 MB of planes through
// its CU's vector-memory path (two workgroups per CU: ~60 GB/s per CU, 64 B/clk at best), and the matrix cores idle 85 %
// of the time.  Here ONE workgroup of EIGHT waves owns 32 rows, one per CU (140-152 KB of LDS): a wave owns 32-COLUMN
// tiles (two of the 16-column sub-tiles of the same packed PH copy: fragments [kk][2 half .. + 1][p] of every 64-deep
// item, still 1 KiB per wave-instruction) and BOTH 16-row groups, so a fetched fragment feeds 6 MFMAs instead of 3 and
// the CU streams the layer's weights once per 32 rows -- while the row-local steps (LayerNorm, camera sampling, gate,
// attention) still see 4 rows per wave on 8 waves, as in two 16-row workgroups.  Same products in the same k order as
// linear_step16h: bit-identical results.  Item = 32 columns x 64 k = 8 fragments ((kk, jj, p) at b[8 BUF + 4 kk + 2 jj + p]),
// 24 MFMAs; two item buffers, the fragments of item i + 1 go to the other buffer group by group (see above: never in place).
struct Acc32H { f32x4 hi[2][2]; f32x4 lo[2][2]; };     // [row group][sub-tile jj]

template <int BUF>
__device__ __forceinline__ void item32h(Acc32H& acc, WBuf& wb, const float* arow0, const float* arow1, int koff,
                                        const float* np, unsigned lo) {
  constexpr int C = 8 * BUF, N = 8 * (1 - BUF);
  // the operand fragments of the two row groups: the 16 bytes of hi plane and the 16 bytes of lo plane of the lane's 8 k,
  // BOTH 32-k halves of the item up front (eight ds_read_b128: the second half's latency runs under the first half's MFMAs)
  float4 x1[2][2], x2[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    x1[kk][0] = *reinterpret_cast<const float4*>(arow0 + koff + 32 * kk); x2[kk][0] = *reinterpret_cast<const float4*>(arow0 + koff + 32 * kk + 4);
    x1[kk][1] = *reinterpret_cast<const float4*>(arow1 + koff + 32 * kk); x2[kk][1] = *reinterpret_cast<const float4*>(arow1 + koff + 32 * kk + 4);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int f = 4 * kk + 2 * jj;
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        acc.lo[rg][jj] = MFMA16H(wb.b[C + f + 1], x1[kk][rg], acc.lo[rg][jj]);
        acc.lo[rg][jj] = MFMA16H(wb.b[C + f], x2[kk][rg], acc.lo[rg][jj]);
        acc.hi[rg][jj] = MFMA16H(wb.b[C + f], x1[kk][rg], acc.hi[rg][jj]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the wave's fragments of the next item: (kk, jj, p) at 8 kk + 2 jj + p fragments behind np (np carries the half)
      wb.b[N + f] = ld4(np + (size_t)(lo + (8 * kk + 2 * jj) * 256u));
      wb.b[N + f + 1] = ld4(np + (size_t)(lo + (8 * kk + 2 * jj + 1) * 256u));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// linear_step for R = 32 (8 waves; same contract as linear_step16h: w0 may arrive preloaded with the step's first item in
// b[0..7], the last item of an even count fetches `next_first`; an odd count -- the radar encoders' two K = 64 steps --
// ends on buffer 1 and hands nothing over)
template <bool DROP, typename SpecFn>
__device__ __forceinline__ bool linear_step32h(const LinSpec& s, WBuf& w0, bool preloaded, const float* next_first,
                                               SpecFn make_spec, int step_idx) {
  constexpr int NW = 8;
  const int lane = threadIdx.x & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - s.woff) & (NW - 1);
  const int ntiles = (s.N + 31) >> 5;                  // 32-column tiles
  const int kpad = (s.K + 63) & ~63;
  const int nkb = kpad / KB;
  const int my_tiles = wave < ntiles ? (ntiles - wave + NW - 1) / NW : 0;
  const int nitems = my_tiles * nkb;
  // lane 16g + c reads rows c and 16 + c, the 8 consecutive k of group g in every 32-wide half of the item
  const float* arow0 = s.src + (lane & 15) * s.src_ld + 8 * (lane >> 4);
  const float* arow1 = arow0 + 16 * s.src_ld;
  const float* wbase = s.W + (size_t)(wave >> 1) * 64 * kpad + (size_t)(wave & 1) * 4 * 256;   // wave-uniform; the lane adds lo
  const unsigned lo = 4u * lane;
  const size_t tile_stride = (size_t)(NW / 2) * 64 * kpad;
  Acc32H acc;
  float4 bias4[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  SUB_STAMP(1);
  if (CHAIN_DBG(s.dbg) & 32) return false;
  if (!preloaded) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w0.b[i] = ld4(wbase + (size_t)(lo + i * 256u));
      w0.b[4 + i] = ld4(wbase + (size_t)(lo + (8 + i) * 256u));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  SUB_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): see linear_step16
  int tt = 0, kb = 0;
  const float* wcur = wbase;
  // one item: accumulators / bias at the start of a tile, the MFMAs from buffer BUF, the epilogue at the end of a tile
  auto run = [&](auto buf, const float* np_last) {
    constexpr int BUF = decltype(buf)::value;
    const float* np = wcur;
    int nt = tt, nk = kb;
    if (++nk == nkb) { nk = 0; ++nt; np = wbase + (size_t)nt * tile_stride; }
    else np += 16 * 256;
    const float* nload = np_last != nullptr ? np_last : np;
    if (kb == 0) {
#pragma unroll
      for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) { acc.hi[rg][jj] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.lo[rg][jj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      // the lane's four consecutive bias values of either sub-tile, in flight under the MFMAs; an unconditional load
      // (see linear_step; columns beyond N are never stored: any valid address will do)
      const float* bsrc = s.bias != nullptr ? s.bias : s.W;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        bias4[jj] = ld4(bsrc + min((wave + tt * NW) * 32 + 16 * jj + 4 * (lane >> 4), max(s.N - 4, 0)));
    }
    item32h<BUF>(acc, w0, arow0, arow1, kb * KB, nload, lo);
    SUB_STAMP(3 + tt * nkb + kb);
    if (kb == nkb - 1) {
      int tile = wave + tt * NW;
      int sidx = step_idx;
      asm volatile("" : "+s"(tile), "+s"(sidx));         // see linear_step: the epilogue rebuilds its view
      if (!(CHAIN_DBG(s.dbg) & 1)) {
        const LinSpec e = make_spec(sidx);
        if (tile * 32 + 32 <= e.N) {                 // (wave-uniform)
          const f32x4 thi[4] = {acc.hi[0][0], acc.hi[0][1], acc.hi[1][0], acc.hi[1][1]};
          const f32x4 tlo[4] = {acc.lo[0][0], acc.lo[0][1], acc.lo[1][0], acc.lo[1][1]};
          lin_epilogue32<DROP>(e, tile * 32, thi, tlo, lane, bias4);
        } else {
          f32x4 y[4];                                  // [2 rg + jj]
#pragma unroll
          for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                y[2 * rg + jj][i] = fmaf(acc.lo[rg][jj][i], 1.0f / (H_LO_SCALE * H_ACT_SCALE), acc.hi[rg][jj][i] * (1.0f / H_ACT_SCALE));
          lin_epilogue16<DROP, 2, true, 2>(e, tile * 32, y, lane, 0.0f, bias4);
        }
      }
    }
    if (nk == 0) SUB_STAMP(16 + tt);               // (behind the tile's epilogue)
    wcur = np; tt = nt; kb = nk;
    __builtin_amdgcn_sched_barrier(0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  const float* fallback = wbase;
  int it = 0;
#pragma unroll 1
  for (; it + 1 < nitems; it += 2) {
    run(B0{}, nullptr);
    const bool last = it + 2 >= nitems;
    run(B1{}, last ? (next_first != nullptr ? next_first : fallback) : nullptr);
  }
  if (it < nitems) {               // odd item count: the step ends on buffer 1, nothing is handed over
    run(B0{}, fallback);
    return false;
  }
  return next_first != nullptr;
}

// y[R, N] = epilogue(src[R, K] W^T): called by all CH_NT threads, no internal barrier.
//
// A wave's work items (its column tiles x 64-deep k blocks) alternate between two
// register buffers: while item i issues its MFMAs, the 16 weight loads (16 KiB per
// wave) of item i+1 are issued in between them.  Inside a run of consecutive
// linear / LayerNorm steps the pipeline does not drain at a step boundary: the
// last item of a step fetches `next_first`, the first item of the wave's next
// linear step (the weight stream depends on the step table only, not on data),
// which then arrives in `w0` (`preloaded`).  Returns true when w0 holds that item.
// SRC2: the A operand is src + src2 (the prologue's x + query_pos only: a run-time test of the pointer
// put branches between the operand reads of every item, and hipcc waits vmcnt(0) at their joins, i.e.
// for the item's whole weight fetch)
template <int R, bool DROP, bool SRC2, int MM, typename SpecFn>
__device__ __forceinline__ bool linear_step(const LinSpec& s, WBuf& w0, bool preloaded,
                                            const float* next_first, SpecFn make_spec, int step_idx) {
  // 16-row tiles: ONE weight buffer refilled in place, 16x16x4 f32 MFMAs (linear_step16) or the two-plane f16
  // form on the matrix cores (linear_step16h, MM = 1)
  if constexpr (R == 32) return linear_step32h<DROP>(s, w0, preloaded, next_first, make_spec, step_idx);
  if constexpr (R == 16 && MM == 1) return linear_step16h<DROP>(s, w0, preloaded, next_first, make_spec, step_idx);
  if constexpr (R == 16) return linear_step16<DROP, SRC2>(s, w0, preloaded, next_first, make_spec, step_idx);
  constexpr int NG = R / 4;
  const int lane = threadIdx.x & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - s.woff) & (CH_NW - 1);
  const int ntiles = (s.N + 63) >> 6;
  const int kpad = (s.K + 63) & ~63;                 // packed tile = 64 * kpad floats
  const int nkb = kpad / KB;
  const int my_tiles = wave < ntiles ? (ntiles - wave + CH_NW - 1) / CH_NW : 0;
  const int nitems = my_tiles * nkb;
  // lane 4j + q reads row q (of each row group), k chunk j
  const float* arow = s.src + (lane & 3) * s.src_ld + 4 * (lane >> 2);
  const float* a2row = SRC2 && s.src2 ? s.src2 + (lane & 3) * s.src2_ld + 4 * (lane >> 2) : nullptr;
  const float* wbase = s.W + 4 * lane + (size_t)wave * 64 * kpad;
  const size_t tile_stride = (size_t)CH_NW * 64 * kpad;
  Acc<NG> acc;
  float bv = 0.0f;
#ifdef TC_CHAIN_STAMPS
  constexpr bool stamp_items = true;
#else
  constexpr bool stamp_items = false;
#endif
  // item = (tile tt, k block kb), kb fastest; tracked incrementally (no divisions)
  int tt = 0, kb = 0;
  const float* wcur = wbase;
  auto run = [&](const WBuf& wb, WBuf& nx, auto pf, const float* np_last) {
    constexpr bool PF = decltype(pf)::value;
    const float* np = wcur;
    int nt = tt, nk = kb;
    if (++nk == nkb) { nk = 0; ++nt; np = wbase + (size_t)nt * tile_stride; }
    else np += (KB / 4) * 256;
    const float* nload = np_last != nullptr ? np_last : np;
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(20 + 5 * kb);
    if (kb == 0) {
      acc_zero<NG>(acc);
      // bias: in flight under the MFMAs.  Unconditional load (clamped column, any valid
      // address when there is no bias): behind a branch hipcc waits vmcnt(0) at the join,
      // i.e. for the whole weight stream in flight (900 cycles per step, measured)
      const int col = min((wave + tt * CH_NW) * 64 + lane, s.N - 1);
      bv = ldg1((s.bias != nullptr ? s.bias : s.W) + col);
    }
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(22 + 5 * kb);
    float4 ar[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      ar[g] = *reinterpret_cast<const float4*>(arow + 4 * g * s.src_ld + kb * KB);
      if (SRC2 && a2row != nullptr)
        ar[g] = add4(ar[g], *reinterpret_cast<const float4*>(a2row + 4 * g * s.src2_ld + kb * KB));
    }
    wcompute<NG, PF>(acc, wb, ar, nx, nload);
    if (stamp_items && tt == 0 && kb < 2) SUB_STAMP(23 + 5 * kb);
    if (kb == nkb - 1) {
      // The epilogue REBUILDS its view of the step from the LDS record (behind an opaque
      // asm on the step / tile index): kept live across the item loop, its ~25 uniform
      // values (destinations, strides, flags) push the kernel past the SGPR budget and
      // every step starts with ~1200 cycles of v_readlane / v_writelane spill traffic.
      int tile = wave + tt * CH_NW;
      int sidx = step_idx;
      asm volatile("" : "+s"(tile), "+s"(sidx));
      if (!(CHAIN_DBG(s.dbg) & 1)) {
        const LinSpec e = make_spec(sidx);
        lin_epilogue<NG, DROP>(e, tile, acc, lane, bv);
      }
    }
    wcur = np; tt = nt; kb = nk;
  };
  using Yes = std::integral_constant<bool, true>;
  SUB_STAMP(1);
  if (CHAIN_DBG(s.dbg) & 32) return false;
  if (!preloaded) {
    wload(w0, wbase, 16);
    __builtin_amdgcn_sched_barrier(0);
  }
  SUB_STAMP(2);
  WBuf w1;
  // Two copies of the item body only (w0 -> w1, w1 -> w0): every item prefetches -- the next
  // item of the step, the first item of the wave's next linear step (`next_first`), or, when
  // there is neither, a harmless re-read (16 KiB at the end of a run of steps).
  const float* fallback = wbase;
  int i = 0;
#pragma unroll 1
  for (; i + 1 < nitems; i += 2) {
    run(w0, w1, Yes{}, nullptr);
    __builtin_amdgcn_sched_barrier(0);
    SUB_STAMP(3 + i);
    const bool last = i + 2 >= nitems;
    run(w1, w0, Yes{}, last ? (next_first != nullptr ? next_first + 4 * lane : fallback) : nullptr);
    __builtin_amdgcn_sched_barrier(0);
    SUB_STAMP(4 + i);
  }
  if (i < nitems) {                  // odd item count (radar feature encoder): no cross-step fetch
    run(w0, w1, Yes{}, fallback);
    SUB_STAMP(3 + i);
    return false;
  }
  return next_first != nullptr;
}

template <int R, int NREC>
__device__ __forceinline__ float* buf_ptr(ChainLds<R, NREC>& S, int id) {
  switch (id) {
    case B_A: return &S.unit[1][0][0];
    case B_X: return &S.unit[0][0][0];
    case B_U1: return &S.unit[1][0][0];
    case B_U2: return &S.unit[2][0][0];
    case B_U3: return &S.unit[3][0][0];
    case B_L: return &S.l[0][0];
    default: return nullptr;
  }
}
__device__ __forceinline__ int buf_ld(int id) { return id == B_A ? LD5 : id == B_L ? LDL : LD2; }

#ifdef TC_CHAIN_STAMPS
// debug build only (make STAMPS=1): s_memtime after every step of workgroup 100 -- and of a second workgroup
// (g_stamp_block2, tc_debug_stamp_block2: block 100's partner on its CU in the stagger experiment below)
__device__ long long g_chain_stamps[CH_NW_MAX][64];
__device__ long long g_chain_stamps2[CH_NW_MAX][64];
__device__ int g_stamp_block2 = -1;
#define STEP_STAMP()                                                                          \
  do {                                                                                        \
    if (blockIdx.x == 100 && lane == 0 && stamp_i < 64)                                       \
      g_chain_stamps[wave][stamp_i] = __builtin_amdgcn_s_memtime();                           \
    if ((int)blockIdx.x == g_stamp_block2 && lane == 0 && stamp_i < 64)                       \
      g_chain_stamps2[wave][stamp_i] = __builtin_amdgcn_s_memtime();                          \
    ++stamp_i;                                                                                \
  } while (0)
#else
#define STEP_STAMP() do {} while (0)
#endif

__device__ __forceinline__ int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// a resolved step from LDS into SGPRs (every field is uniform over the workgroup:
// left in VGPRs they cost the item loop ~40 registers and the weight buffers spill)
template <typename T>
__device__ __forceinline__ T load_uniform(const T& src) {
  constexpr int NWORDS = sizeof(T) / 4;
  static_assert(sizeof(T) % 4 == 0, "record size");
  union U { T r; int w[NWORDS]; __device__ U() {} } u;
  const int* p = reinterpret_cast<const int*>(&src);
#pragma unroll
  for (int i = 0; i < NWORDS; ++i) u.w[i] = __builtin_amdgcn_readfirstlane(p[i]);
  return u.r;
}

#ifdef TC_CHAIN_STAMPS
#define START_STAMP(slot)                                                                     \
  do {                                                                                        \
    if (blockIdx.x == 100 && threadIdx.x == 0) g_chain_sub[0][(slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// entry / exit time of every workgroup (wave 0): dispatch skew and the slowest workgroups
__device__ long long g_wg_span[1024][2];
#define WG_STAMP(which)                                                                       \
  do {                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_span[blockIdx.x][(which)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// ROUND 6 EXPERIMENT (tools/stagger_probe.py): two 16-row workgroups share a CU; the SECOND arrival
// on a CU starts TRANSCAR_CHAIN_DBG bits 8..15 x 4096 cycles late, so that its row-local phases (epilogues, LayerNorm,
// sampling: VALU / LDS / latency) run beside the first one's item loops (matrix cores) and vice versa.  With more
// workgroups than slots (18 / 27 frames per launch) a later workgroup takes the slot -- and the phase -- of the one
// that ended: the offset persists over the launch.  Arrival number per CU: key = XCC id + the SE / SH / CU fields of
// HW_ID; g_wg_cu[block] = {key, ticket} for the write-up.  The launcher zeroes the tickets (tc_debug_stagger_reset).
__device__ unsigned g_cu_ticket[2048];
__device__ int g_wg_cu[2048][2];
#else
#define START_STAMP(slot) do {} while (0)
#define WG_STAMP(which) do {} while (0)
#endif

// PROG is a compile-time parameter: each program's kernel contains only the step kinds it
// uses.  (One code image for all four programs made the R = 4 kernel spill 9 dwords to a
// private segment under the combined pressure of the camera-sampling and radar-attention
// bodies; the specialised kernels are smaller and were 3.5 % faster per frame.)
// PRE (round 6; decoder program on the f16x2 path only): the sampling step reads the level values the pre-gather workgroups
// of the attention-core launch stored -- its own instantiations, so that the kernels that gather for themselves keep
// their code and registers exactly (with both paths in one kernel the direct-gather launches lost 2 us to 12 more spills)
template <int R, int PROG, bool DROP = false, int MM = 0, bool PRE = false>
__device__ __forceinline__ void chain_body(const ChainDev& k, const StepAllT<nw_of(R)>* __restrict__ recs, const int block) {
  constexpr int NW = nw_of(R), NT = NW * 64;        // waves / threads of the workgroup
  constexpr bool PL = R == 32;                      // the activation units hold planes (act_ld4 / act_st4)
  static_assert((MM == 0 && R <= 16) || (MM == 1 && R >= 16 && PROG != PROG_PROLOGUE), "the f16 two-plane path exists for 16- and 32-row tiles (and is the only one at 32)");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  using Lds = ChainLds<R, rec_cap(PROG)>;
  Lds& S = *reinterpret_cast<Lds*>(smem_raw);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = block * R;
  // The records, kernel-argument segment -> registers (-> LDS below): issued before anything
  // depends on a kernel argument -- the first touch of the segment costs ~2600 cycles, and
  // this way the copy and the scalar loads of the arguments share that one latency.
  constexpr int REC16 = rec_cap(PROG) * (int)(sizeof(StepAllT<NW>) / 16);
  constexpr int REC_TRIPS = (REC16 + NT - 1) / NT;
  int4 rec_v[REC_TRIPS];
#pragma unroll
  for (int t = 0; t < REC_TRIPS; ++t) {
    const int i = min((int)threadIdx.x + t * NT, REC16 - 1);   // unconditional: stays in registers
    rec_v[t] = reinterpret_cast<const int4*>(recs)[i];
  }
  const int M = k.M;
  if (CHAIN_DBG(k.dbg) & 64) return;
  START_STAMP(40);
  WG_STAMP(0);
#ifdef TC_CHAIN_STAMPS
  if constexpr (R == 16 && PROG == PROG_DECODER) {
    const int stg = (k.dbg >> 8) & 0xFF;
    if (stg > 0 || (k.dbg & (1 << 16))) {         // (bit 16: draw and record the tickets without a delay)
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID
      const unsigned key = ((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu);
      unsigned tk = 0;
      if (threadIdx.x == 0) {
        tk = atomicAdd(&g_cu_ticket[key], 1u);
        if (blockIdx.x < 2048) { g_wg_cu[blockIdx.x][0] = (int)key; g_wg_cu[blockIdx.x][1] = (int)tk; }
      }
      tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);       // wave 0 only: the others meet it at the first barrier
      if (wave == 0 && tk == 1u) {
#pragma unroll 1
        for (int i = 0; i < stg; ++i) __builtin_amdgcn_s_sleep(64);
      }
    }
  }
#endif
  const int total = k.total;
  // The leading global-to-LDS loads of a program (a decoder layer starts with three) go out
  // together: one memory latency, one barrier.
  constexpr int EARLY_MAX = 3, EARLY_RW = (R + NW - 1) / NW;
  const int early_n = k.early_n;
  {
    float4 early_v[EARLY_MAX][EARLY_RW];
#pragma unroll
    for (int j = 0; j < EARLY_MAX; ++j) {
      if (j < early_n) {
        const float* gsrc = recs[j].r.gd;
        const int ld = recs[j].r.gld, mod = recs[j].r.gmod;
#pragma unroll
        for (int ri = 0; ri < EARLY_RW; ++ri) {
          int grow = min(m0 + wave + ri * NW, M - 1);
          if (mod > 0) grow = grow % mod;
          early_v[j][ri] = TC_ONCE_LD(gsrc + (size_t)grow * ld + 4 * lane);
        }
      }
    }
    START_STAMP(41);
#pragma unroll
    for (int j = 0; j < EARLY_MAX; ++j) {
      if (j < early_n) {
        float* dst = reinterpret_cast<float*>(smem_raw) + recs[j].e.dst_off;
#pragma unroll
        for (int ri = 0; ri < EARLY_RW; ++ri) {
          const int row = wave + ri * NW;
          if (row < R) act_st4<PL>(dst + row * LD2, 4 * lane, early_v[j][ri]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < REC_TRIPS; ++t) {
    const int i = threadIdx.x + t * NT;
    if (i < REC16) reinterpret_cast<int4*>(&S.recs[0])[i] = rec_v[t];
  }
  START_STAMP(43);

  if (prog_is_radar(PROG)) {     // HEAD:539, 543-547, 596-598
    // tile position -> row: the identity, or the order of launch_radar_compact (hit rows first)
    auto row_of = [&](int row) {
      const int pos = min(m0 + row, M - 1);
      return k.row_perm != nullptr ? k.row_perm[pos] : pos;
    };
    if (threadIdx.x < R) S.rowg[threadIdx.x] = row_of(threadIdx.x);
    for (int row = wave; row < R; row += NW) {
      const int grow = row_of(row);
      act_st4<PL>(&S.unit[0][row][0], 4 * lane, TC_ONCE_LD(k.g[G_QF] + (size_t)grow * 256 + 4 * lane));
    }
    {
      // one (row, column) per thread: columns 0..code-1 the previous box, 12..14 the gate centre
      const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
      if (row < R) {
        const int grow = row_of(row);
        if (j < k.code) S.box[row][j] = k.box_in[(size_t)grow * k.code + j];
        if (j >= 12 && j < 15) {
          const int c = j - 12;
          if (k.cen_from_box) {      // HEAD:615-617 / 671-673: xy = box[0:2], z = box[4]
            S.cen[row][c] = k.box_in[(size_t)grow * k.code + (c == 2 ? 4 : c)];
          } else {
            const float r = k.ref_last[(size_t)grow * 3 + c];
            const float* pc = k.cam.pc;
            // z stays normalised (HEAD:598 indexes an empty slice)
            S.cen[row][c] = c < 2 ? __fadd_rn(__fmul_rn(r, pc[3 + c] - pc[c]), pc[c]) : r;
          }
        }
      }
    }
  }
  if constexpr (PROG == PROG_RADAR_BWD) {
    // nothing comes down to the top layer: the carried gradients (layer input in unit X, box in the row records) start at zero
    for (int i = threadIdx.x; i < R * LD2; i += NT) (&S.unit[0][0][0])[i] = 0.0f;
    if (threadIdx.x < R * 12) (&S.box[0][0])[threadIdx.x] = 0.0f;
    if (threadIdx.x < R) S.gate[threadIdx.x] = 0;
  }
  __syncthreads();

  START_STAMP(46);
#ifdef TC_CHAIN_STAMPS
  int stamp_i = 0;
  STEP_STAMP();
#endif

  // dst = [relu] LN(a (+ relu(LN(c; p2,p3)))) (+ d): wave w owns rows w, w+4, ...
  // All rows of a wave (R/4: 1, 2 or 4) go through the phases together: one fetch of gamma / beta per
  // wave, the rows' independent reduction chains interleave, stores last (row by row the 16-row tiles
  // paid four gamma / beta round trips and four serial reduction chains per LayerNorm).
  auto do_ln = [&](const StepRes& r) {
    constexpr int NR = R / NW;
    const float* a = buf_ptr(S, r.src); const int lda = buf_ld(r.src);
    const float* c = buf_ptr(S, r.src2); const int ldc = buf_ld(r.src2);
    const bool xp = (r.flags & F_LN_XP) != 0;          // `res` names an OUTPUT then: result + query_pos
    const float* dd = xp ? nullptr : buf_ptr(S, r.res);
    float* dst2 = xp ? buf_ptr(S, r.res) : nullptr;
    float* dst = buf_ptr(S, r.dst);
    float* gdst = r.gd;
    float4 gg, bb;
    if (CHAIN_DBG(k.dbg) & (1 << 17)) {                // timing experiment: a LayerNorm without its parameter round trip
      gg = make_float4(1.f, 1.f, 1.f, 1.f); bb = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      gg = ld4(r.p0 + 4 * lane); bb = ld4(r.p1 + 4 * lane);
    }
    float4 v[NR], pos4[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = wave + NW * i;
      pos4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xp && !(CHAIN_DBG(k.dbg) & (1 << 18))) {      // in flight under the reductions
        int prow = min(m0 + row, M - 1);
        if (k.g_mod[G_POS] > 0) prow = prow % k.g_mod[G_POS];
        pos4[i] = ld4(k.g[G_POS] + (size_t)prow * k.g_ld[G_POS] + 4 * lane);
      }
      v[i] = act_ld4<PL>(a + row * lda, 4 * lane);
    }
    if (c != nullptr) {
      const float4 g2 = ld4(r.p2 + 4 * lane), b2 = ld4(r.p3 + 4 * lane);
      float4 cv[NR];
#pragma unroll
      for (int i = 0; i < NR; ++i) cv[i] = act_ld4<PL>(c + (wave + NW * i) * ldc, 4 * lane);
      ln_rows<NR>(cv, g2, b2);
#pragma unroll
      for (int i = 0; i < NR; ++i) v[i] = add4(v[i], relu4(cv[i]));
    }
    ln_rows<NR>(v, gg, bb);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = wave + NW * i;
      if (r.flags & F_LN_RELU) v[i] = relu4(v[i]);
      if constexpr (PROG == PROG_RADAR_ENC_TRAIN) {      // tape: pos = relu(LN(u2)) before the feature sum
        if (r.gt != nullptr && m0 + row < M) st4(r.gt + (size_t)(m0 + row) * 256 + 4 * lane, v[i]);
      }
      if (dd != nullptr) v[i] = add4(v[i], act_ld4<PL>(dd + row * LD2, 4 * lane));
      act_st4<PL>(dst + row * LD2, 4 * lane, v[i]);
      if (xp) act_st4<PL>(dst2 + row * LD2, 4 * lane, add4(v[i], pos4[i]));
      if (gdst != nullptr && m0 + row < M) st4(gdst + (size_t)(m0 + row) * 256 + 4 * lane, v[i]);
    }
  };
  // the part of a linear step the item loop needs ...
  auto lin_spec = [&](const StepRes& r) {
    LinSpec s;
    s.K = r.K; s.N = r.N;
    s.W = r.p0; s.bias = r.p1;
    const float* lbase = reinterpret_cast<const float*>(smem_raw);
    s.src = lbase + r.src_off; s.src_ld = buf_ld(r.src);
    s.src2 = r.src2_off >= 0 ? lbase + r.src2_off : nullptr; s.src2_ld = buf_ld(r.src2);
    s.dst = nullptr; s.dst_ld = 0; s.res = nullptr; s.res_ld = 0; s.gate = nullptr;
    s.act = 0; s.scale = 1.0f; s.scale_cols = 0;
    s.gdst = nullptr; s.gdst_ld = 0; s.gt = nullptr; s.gt_ld = 0; s.gt_rpb = 1;
    s.m0 = m0; s.M = M;
    s.woff = (r.flags & F_WAVE1) ? 1 : 0;
    s.dbg = k.dbg;
    s.sub_on = 0;
    s.drop_site = 0; s.drop_seed = 0; s.drop_thr = 0; s.drop_scale = 1.0f; s.drop_stride = 0; s.drop_q = 0;
    s.rowg = nullptr; s.gpre = 0; s.cmask = nullptr; s.cscale = 1.0f;
    s.dst_pl = 0; s.res_pl = 0; s.range_flag = nullptr;
    return s;
  };
  // ... and the part its epilogue needs, rebuilt per tile from the 64-byte LDS record
  auto epi_spec = [&](int j) {
    const EpiRec e = load_uniform<EpiRec>(S.recs[j].e);
    float* base = reinterpret_cast<float*>(smem_raw);
    LinSpec s;
    s.K = 0; s.N = e.N; s.W = nullptr;
    s.bias = e.has_bias ? base : nullptr;           // only tested against nullptr in the epilogue
    s.src = nullptr; s.src_ld = 0; s.src2 = nullptr; s.src2_ld = 0;
    s.dst = e.dst_off >= 0 ? base + e.dst_off : nullptr; s.dst_ld = e.dst_ld;
    s.res = e.res_off >= 0 ? base + e.res_off : nullptr; s.res_ld = e.res_ld;
    s.gate = (e.flags & F_GATE) ? &S.gate[0] : nullptr;
    s.act = e.act;
    s.scale = (e.flags & F_SCALEQ) ? k.qscale : 1.0f; s.scale_cols = (e.flags & F_SCALEQ) ? 256 : 0;
    s.gdst = e.gd; s.gdst_ld = e.gld;
    s.gt = (e.flags & F_CMASK) ? nullptr : e.gt; s.gt_ld = k.qpad; s.gt_rpb = k.Q;
    s.cmask = (e.flags & F_CMASK) ? e.gt : nullptr;
    s.cscale = (e.flags & F_CMASK_SCALE) ? k.rdrop.scale : 1.0f;
    s.m0 = m0; s.M = M;
    s.woff = e.woff;
    s.dbg = 0;
    s.sub_on = 0;
    s.drop_site = 0; s.drop_seed = 0; s.drop_thr = 0; s.drop_scale = 1.0f; s.drop_stride = 0; s.drop_q = 0;
    s.rowg = (prog_is_radar(PROG) && k.row_perm != nullptr) ? &S.rowg[0] : nullptr;
    s.gpre = (e.flags & F_GPRE) ? 1 : 0;
    s.dst_pl = PL && e.dst_ld != LDL; s.res_pl = PL && e.res_ld != LDL;      // (the logit buffer stays fp32)
    s.range_flag = MM == 1 ? k.range_status : nullptr;
    if (DROP) {
      if constexpr (PROG == PROG_RADAR_TRAIN) {      // radar dropout sites 4 r + {1, 2, 3} (HEAD:581-585)
        s.drop_site = e.drop_site; s.drop_seed = k.rdrop.seed; s.drop_thr = k.rdrop.thr; s.drop_scale = k.rdrop.scale;
      } else {
        s.drop_site = e.drop_site; s.drop_seed = k.drop.seed; s.drop_thr = k.drop.thr; s.drop_scale = k.drop.scale;
        s.drop_stride = k.drop.seed_stride; s.drop_q = (int)k.drop.rows_per_sample;
      }
    }
    return s;
  };

  // radar program: does any row of this tile have a radar return inside its gate (this layer)?
  auto tile_has_hit = [&]() {
    int any = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) any |= S.gate[i];
    return __builtin_amdgcn_readfirstlane(any) > 0;
  };

#ifdef TC_CHAIN_DUMP
  // called by every thread after a step (radar program only): waits for the step's writers, copies the first 256
  // columns of its LDS destination to g_chain_dump[step][row], waits again
  auto dump_step = [&](int step, int dst_id) {
    if constexpr (PROG == PROG_RADAR) {
      __syncthreads();
      float* out = g_chain_dump;
      const float* src = buf_ptr(S, dst_id);
      if (out != nullptr && src != nullptr && dst_id != B_L && (long long)(step + 1) * M * 256 <= g_chain_dump_floats) {
        const int ld = buf_ld(dst_id);
        for (int row = wave; row < R; row += NW)
          if (m0 + row < M)
            st4(out + ((size_t)step * M + S.rowg[row]) * 256 + 4 * lane, *reinterpret_cast<const float4*>(src + row * ld + 4 * lane));
      }
      __syncthreads();
    }
  };
#define DUMP_STEP(step, dst) dump_step((step), (dst))
#else
#define DUMP_STEP(step, dst) do {} while (0)
#endif
  int idx = (CHAIN_DBG(k.dbg) & 16) ? total : early_n;          // the leading loads are already in LDS
#ifdef TC_CHAIN_STAMPS
  for (int j = 0; j < 2 * early_n; ++j) STEP_STAMP();
#endif
#pragma unroll 1
  while (idx < total) {
    const int kind = ufirst(S.recs[idx].r.kind);
    if (kind == K_LINEAR || kind == K_LN || kind == K_NOP) {
      // ---- a run of linear / LayerNorm steps: the weight pipeline stays primed across them
      WBuf w0;
      int pre_idx = -1;                      // step whose first item is in flight in w0
#ifdef TC_CHAIN_STAMPS
      int ln_rep = 0;
#endif
#pragma unroll 1
      for (;;) {
        const StepRes r = load_uniform<StepRes>(S.recs[idx].r);
        const int kd = r.kind;
        bool skip = false;
        if constexpr (PROG == PROG_RADAR) {           // (the training forward skips nothing: the tape must be complete)
          if (kd == K_LINEAR && (r.flags & F_IFHIT)) skip = !tile_has_hit();
        }
        if (skip) {
          // every row gate is 0: dst = res + 0 * (...) (a step without a residual only feeds skipped steps).
          // w0 may hold this step's first item: pre_idx then names a step that is not the next one -> fresh load
          if (r.res != B_NONE && r.dst != B_NONE) {
            const float* rs = buf_ptr(S, r.res); const int rld = buf_ld(r.res);
            float* dd = buf_ptr(S, r.dst);
            for (int row = wave; row < R; row += NW)
              *reinterpret_cast<float4*>(dd + row * LD2 + 4 * lane) = *reinterpret_cast<const float4*>(rs + row * rld + 4 * lane);
          }
        } else if (kd == K_LINEAR) {
          LinSpec s = lin_spec(r);
#ifdef TC_CHAIN_STAMPS
          s.sub_on = (r.si == g_sub_step && r.rep == 0);
          SUB_STAMP(0);
#endif
          if (((wave - s.woff) & (NW - 1)) < (R == 32 ? (s.N + 31) >> 5 : (s.N + 63) >> 6)) {   // else: no column tile here, w0 keeps waiting
            const PreRec pr = load_uniform<PreRec>(S.recs[idx].p[wave]);
            const bool have = linear_step<R, DROP, PROG == PROG_PROLOGUE, MM>(s, w0, pre_idx == idx, pr.first, epi_spec, idx);
            pre_idx = have ? pr.nidx : -1;
          }
        } else if (kd == K_LN) {
          if (!(CHAIN_DBG(k.dbg) & 4)) do_ln(r);
#ifdef TC_CHAIN_STAMPS
          // timing experiment (bits 19..20): the step 1 + n times through the SAME code -- what a repeat costs is the
          // LayerNorm with its instructions already fetched
          if (ln_rep < ((k.dbg >> 19) & 3)) { ++ln_rep; continue; }
          ln_rep = 0;
#endif
        } else if (kd != K_NOP) {
          break;
        }
        STEP_STAMP();
        if (r.sync && !(CHAIN_DBG(k.dbg) & 2)) __syncthreads();
        STEP_STAMP();
        DUMP_STEP(idx, r.dst);
        if (++idx >= total) break;
      }
      continue;
    }
    const StepRes r = load_uniform<StepRes>(S.recs[idx].r);
    const int rep = r.rep;
    switch (kind) {
      case K_LOAD: { if constexpr (PROG == PROG_DECODER || PROG == PROG_PROLOGUE || PROG == PROG_RADAR_ENC_B) {
        // this and the directly following K_LOAD steps (the decoder starts with three) go
        // out together: one memory latency instead of three, one barrier
        constexpr int MAXL = 3, RW = (R + NW - 1) / NW;
        int n = 1;
        while (n < MAXL && idx + n < total && ufirst(S.recs[idx + n].r.kind) == K_LOAD) ++n;
        float4 v[MAXL][RW];
        float* dsts[MAXL];
        int any_sync = 0;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
          if (j < n) {
            const StepRes rj = load_uniform<StepRes>(S.recs[idx + j].r);
            const float* gsrc = rj.gd;
            const int ld = rj.gld, mod = rj.gmod;
            dsts[j] = buf_ptr(S, rj.dst);
            any_sync |= rj.sync;
#pragma unroll
            for (int ri = 0; ri < RW; ++ri) {
              int grow = min(m0 + wave + ri * NW, M - 1);
              if (mod > 0) grow = grow % mod;
              v[j][ri] = TC_ONCE_LD(gsrc + (size_t)grow * ld + 4 * lane);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
          if (j < n) {
#pragma unroll
            for (int ri = 0; ri < RW; ++ri) {
              const int row = wave + ri * NW;
              if (row < R) act_st4<PL>(dsts[j] + row * LD2, 4 * lane, v[j][ri]);
            }
          }
        }
        for (int j = 0; j < n; ++j) { STEP_STAMP(); STEP_STAMP(); }
        if (any_sync) __syncthreads();
        idx += n;
        continue;
      }
      } break;
      case K_TOKENS: { if constexpr (prog_is_enc_full(PROG) || PROG == PROG_RADAR_ENC_A) {   // radar token tile, zero padded to 64 columns
        if constexpr (PL) {           // four columns per thread (the tile is the A operand of feat.0: planes)
          for (int i = threadIdx.x; i < R * 16; i += NT) {
            const int row = i >> 4, c4 = 4 * (i & 15);
            const int grow = min(m0 + row, M - 1);
            const float* tk = k.tokens + (size_t)grow * k.RI;
            act_st4<PL>(&S.unit[1][0][0] + row * LD5, c4,
                        make_float4(c4 < k.RI ? tk[c4] : 0.0f, c4 + 1 < k.RI ? tk[c4 + 1] : 0.0f, c4 + 2 < k.RI ? tk[c4 + 2] : 0.0f,
                                    c4 + 3 < k.RI ? tk[c4 + 3] : 0.0f));
          }
        } else {
        for (int i = threadIdx.x; i < R * 64; i += NT) {
          const int row = i >> 6, c = i & 63;
          const int grow = min(m0 + row, M - 1);
          (&S.unit[1][0][0])[row * LD5 + c] = c < k.RI ? k.tokens[(size_t)grow * k.RI + c] : 0.0f;
        }
        }
      } break;
      } break;
      case K_POSENC: { if constexpr (PROG == PROG_DECODER || prog_is_enc_full(PROG) || PROG == PROG_RADAR_ENC_A) {   // Linear(3,256) + LN + ReLU of inverse_sigmoid(ref) or of raw token xyz
        float* dst = buf_ptr(S, r.dst);
        const bool skip0 = (r.flags & F_NOT_W0) != 0;      // wave 0 is busy with the narrow linear step before
        const int row_first = skip0 ? wave - 1 : wave, row_step = skip0 ? NW - 1 : NW;
        // decoder: lane i fetches and inverts the reference point of row i -- one round trip and one
        // inverse_sigmoid per ROW of the tile (before: per row of the loop below, evaluated by all 64 lanes)
        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
        if (r.src != B_A && row_first >= 0) {
          int grow = min(m0 + min(lane, R - 1), M - 1);
          if (k.ref_mod > 0) grow = grow % k.ref_mod;
          q0 = inverse_sigmoidf_(k.ref_in[(size_t)grow * 3 + 0]);
          q1 = inverse_sigmoidf_(k.ref_in[(size_t)grow * 3 + 1]);
          q2 = inverse_sigmoidf_(k.ref_in[(size_t)grow * 3 + 2]);
          if constexpr (PROG == PROG_DECODER) {
            // kept for the layer's last step (K_REFUPD adds the regression deltas to exactly these values): the centre
            // records are free in the decoder program -- one global round trip and three logs less at the end of every tile
            if (wave == (skip0 ? 1 : 0) && lane < R) { S.cen[lane][0] = q0; S.cen[lane][1] = q1; S.cen[lane][2] = q2; }
          }
        }
        auto xyz = [&](int row, float& p0, float& p1, float& p2) {
          if (r.src == B_A) {
            if constexpr (PL) {          // (the token tile holds planes: the raw xyz come from the tokens themselves)
              const float* tk = k.tokens + (size_t)min(m0 + row, M - 1) * k.RI; p0 = tk[0]; p1 = tk[1]; p2 = tk[2];
            } else { const float* tk = &S.unit[1][0][0] + row * LD5; p0 = tk[0]; p1 = tk[1]; p2 = tk[2]; }
          }
          else { p0 = lane_f(q0, row); p1 = lane_f(q1, row); p2 = lane_f(q2, row); }
        };
        int row = row_first;
        if constexpr (PROG != PROG_RADAR_ENC_TRAIN) {
          // Round 6: a wave's rows four at a time -- W0 / b0 / gamma / beta fetched once, the LayerNorms through the packed
          // reductions (rowdev.hpp ln_rows<4>).  Same expressions per row as posenc_l0_row: bit-identical.  (Row by row
          // this step was 10.5 K cycles of a 32-row decoder layer for the seven waves that share it: VALU-bound.)
          if (row >= 0 && row + 3 * row_step < R) {
            const float* w0 = uptr(r.p0); const float* b0 = uptr(r.p1);
            float wc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = 4 * lane + i;
              wc[i][0] = ldg1(w0 + c * 3 + 0); wc[i][1] = ldg1(w0 + c * 3 + 1); wc[i][2] = ldg1(w0 + c * 3 + 2); wc[i][3] = ldg1(b0 + c);
            }
            const float4 gg = ld4(uptr(r.p2) + 4 * lane), bb = ld4(uptr(r.p3) + 4 * lane);
            for (; row + 3 * row_step < R; row += 4 * row_step) {
              float4 v4[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float p0, p1, p2;
                xyz(row + j * row_step, p0, p1, p2);
                v4[j] = make_float4(wc[0][0] * p0 + wc[0][1] * p1 + wc[0][2] * p2 + wc[0][3], wc[1][0] * p0 + wc[1][1] * p1 + wc[1][2] * p2 + wc[1][3],
                                    wc[2][0] * p0 + wc[2][1] * p1 + wc[2][2] * p2 + wc[2][3], wc[3][0] * p0 + wc[3][1] * p1 + wc[3][2] * p2 + wc[3][3]);
              }
              ln_rows<4>(v4, gg, bb);
#pragma unroll
              for (int j = 0; j < 4; ++j) act_st4<PL>(dst + (row + j * row_step) * LD2, 4 * lane, relu4(v4[j]));
            }
          }
        }
        for (; row < R && row >= 0; row += row_step) {
          float p0, p1, p2;
          xyz(row, p0, p1, p2);
          if constexpr (PROG == PROG_RADAR_ENC_TRAIN) {
            // tape: the pre-LayerNorm values u0 (r.gt) and u1 = relu(LN(u0)) (r.gd)
            float4 pre;
            const float4 u1 = posenc_l0_row(p0, p1, p2, uptr(r.p0), uptr(r.p1), uptr(r.p2), uptr(r.p3), lane, &pre);
            act_st4<PL>(dst + row * LD2, 4 * lane, u1);
            if (m0 + row < M) {
              st4(r.gt + (size_t)(m0 + row) * 256 + 4 * lane, pre);
              st4(r.gd + (size_t)(m0 + row) * 256 + 4 * lane, u1);
            }
          } else {
            act_st4<PL>(dst + row * LD2, 4 * lane,
                        posenc_l0_row(p0, p1, p2, uptr(r.p0), uptr(r.p1), uptr(r.p2), uptr(r.p3), lane));
          }
        }
      } break;
      } break;
      case K_SAMPLE: { if constexpr (PROG == PROG_DECODER) {   // camera sampling of this block's queries
        if (CHAIN_DBG(k.dbg) & 8) break;
        int pairs = 0;
        CAM_STAMP(5);
        if constexpr (PRE) {
          // Round 6: the taps were gathered and bilinearly reduced by the pre-gather workgroups of the attention-core
          // launch in front of this one (rowdev.hpp cam_pregather_rows: same projection, same tap geometry, same
          // cam_level_value).  What is left: a row's visibility mask, 4 KiB contiguous per visible (row, camera) pair,
          // sigmoid(attention_weights) . level value, summed in cam_sample_core's order -- bit-identical.  The FIRST
          // visible camera of all the wave's rows is fetched in one round trip (1.03 visible cameras per row on the
          // bench's rig); further cameras of a row follow one pair at a time.
          constexpr int NR = R / NW;
          const int NC = k.cam.num_cams;
          int mk = 0;
          if (lane < NR) mk = k.premask[min(m0 + wave + NW * lane, M - 1)];
          float4 v[NR][4];
          int vmask[NR], cam0[NR];
          const float* rowp[NR];
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            vmask[i] = __builtin_amdgcn_readlane(mk, i);
            cam0[i] = vmask[i] ? __ffs(vmask[i]) - 1 : 0;
            const int grow = min(m0 + wave + NW * i, M - 1);
            rowp[i] = k.pre + ((size_t)grow * NC) * (4 * 256) + 4 * lane;
#pragma unroll
            for (int l = 0; l < 4; ++l) v[i][l] = cam_tap_ld(rowp[i] + ((size_t)cam0[i] * 4 + l) * 256);   // (no visible camera: read, never used; read once: non-temporal)
          }
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const int row = wave + NW * i;
            const float sg_lane = sigmoidf_(S.l[row][min(lane, NC * 4 - 1)]);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int rest = vmask[i];
            if (rest) {                                  // (wave-uniform)
              float4 camacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int l = 0; l < 4; ++l) {
                const float a = lane_f(sg_lane, cam0[i] * 4 + l);
                camacc.x += v[i][l].x * a; camacc.y += v[i][l].y * a; camacc.z += v[i][l].z * a; camacc.w += v[i][l].w * a;
              }
              acc.x += camacc.x; acc.y += camacc.y; acc.z += camacc.z; acc.w += camacc.w;
              rest &= rest - 1;
#pragma unroll 1
              while (rest) {
                const int cam = __ffs(rest) - 1;
                rest &= rest - 1;
                float4 u[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) u[l] = cam_tap_ld(rowp[i] + ((size_t)cam * 4 + l) * 256);
                float4 ca = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                  const float a = lane_f(sg_lane, cam * 4 + l);
                  ca.x += u[l].x * a; ca.y += u[l].y * a; ca.z += u[l].z * a; ca.w += u[l].w * a;
                }
                acc.x += ca.x; acc.y += ca.y; acc.z += ca.z; acc.w += ca.w;
              }
            }
            act_st4<PL>(buf_ptr(S, r.dst) + row * LD2, 4 * lane, acc);
            if (m0 + row < M) pairs += __popc(vmask[i]);
          }
        } else {
        // (launchers: PRE instantiations exist where k.pre is set -- decoder program, f16x2 path)
        // The projections of ALL the wave's rows first, 16 lanes per row (lane 16 i + c: row wave + 4 i, camera c):
        // one round trip for the reference points and one for the lidar2img rows per WAVE instead of per row
        // (stamps: 3 300 of a row's 7 700 cycles were its projection -- two dependent global loads and a
        // division in front of the first tap).  Same arithmetic per (row, camera): bit-identical.
        float pu, pv;
        unsigned long long vm;
        int Hl, Wl;
        cam_level_dims<4>(k.cam, lane, Hl, Wl);      // once per step, in flight with the projections (not once per row)
        {
          const int i = lane >> 4, c = lane & 15;
          const int prow = wave + NW * min(i, R / NW - 1);
          const int grow = min(m0 + prow, M - 1);
          const bool act = i < R / NW && c < k.cam.num_cams;
          vm = __ballot(cam_project_lane(k.cam, k.ref_mod > 0 ? grow % k.ref_mod : grow, grow / k.Q,
                                         min(c, k.cam.num_cams - 1), act, pu, pv));
        }
#pragma unroll 1
        for (int i = 0; i < R / NW; ++i) {
          const int row = wave + NW * i;
          const int grow = min(m0 + row, M - 1);
          const unsigned long long vmask = (vm >> (16 * i)) & 0xFFFFull;
          const float4 o = cam_sample_core<4>(k.cam, grow / k.Q, &S.l[row][0], lane, vmask, pu, pv,
                                              [](int, int, int, const float* ptr) { return cam_tap_ld(ptr); }, 16 * i, Hl, Wl);
          act_st4<PL>(buf_ptr(S, r.dst) + row * LD2, 4 * lane, o);
          if (m0 + row < M) pairs += __popcll(vmask);
        }
        }
        CAM_STAMP(6);
        if (k.pair_counter != nullptr && lane == 0 && pairs > 0)
          atomicAdd(k.pair_counter, (unsigned long long)pairs);
        CAM_STAMP(7);
      } break;
      } break;
      case K_REFUPD: { if constexpr (PROG == PROG_DECODER) {   // XFMR:195-203, HEAD:287-293
        // one (row, box column) per thread: as a per-row loop on R threads this took ~2900 cycles
        const int row = threadIdx.x >> 4, j = threadIdx.x & 15;
        if (row < R && m0 + row < M && j < k.code) {
          const int grow = m0 + row;
          float val = S.l[row][j];
          const int c = j == 0 ? 0 : j == 1 ? 1 : j == 4 ? 2 : -1;      // box columns cx, cy, cz <- reference x, y, z
          if (c >= 0) {
            const float n = sigmoidf_(val + S.cen[row][c]);      // inverse_sigmoid(reference), computed by K_POSENC
            k.ref_out[(size_t)grow * 3 + c] = n;
            const float* pc = k.cam.pc;
            val = n * (pc[3 + c] - pc[c]) + pc[c];
          }
          if (k.box_m != nullptr) k.box_m[(size_t)grow * k.code + j] = val;
        }
      } break;
      } break;
      case K_NARROW: { if constexpr (PROG == PROG_DECODER || prog_is_radar(PROG)) {
        // The 10-column heads (reg.4, final_reg.4, final_cls.6): as a 64-column tile of the item loop they
        // kept ONE wave busy for four items (13 000 cycles at 16 rows) while three waited at the barrier.
        // Here: ONE 16-column MFMA sub-tile, the 16 k groups of 16 split over the four waves (16
        // v_mfma_f32_16x16x4 each), partial sums through the LDS buffer named by `src2`, wave 0 adds them in
        // a fixed order.  W is read in the nn.Linear layout [N][256] (not packed): lane 16g + c's float4 at
        // W[c][16 kg + 4g ..] IS the B operand of the k group's four MFMAs.  Same code at every tile height
        // (rows >= R repeat row R - 1, never stored).
        // 32-row tiles (8 waves): waves 0-3 the first 16 rows, waves 4-7 the second (rgw), the k groups over kq
        const int N = r.N;                                // <= 12 (launchers check code / num_classes)
        const int c = lane & 15, g = lane >> 4;
        const int rgw = wave >> 2, kq = wave & 3;
        const float* Wn = uptr(r.p0) + (size_t)min(c, N - 1) * 256 + 4 * g;
        const float* src = buf_ptr(S, r.src) + min(16 * rgw + c, R - 1) * buf_ld(r.src);
        float4 av[4], bw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = ld4(Wn + 16 * (4 * kq + q));
        const float bias = r.p1 != nullptr ? ldg1(uptr(r.p1) + min(c, N - 1)) : 0.0f;
        if (r.flags & F_PRESYNC) __syncthreads();         // (the source tile is complete behind this barrier)
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = act_ld4<PL>(src, 4 * g + 16 * (4 * kq + q));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc = MFMA16(av[q].x, bw[q].x, acc); acc = MFMA16(av[q].y, bw[q].y, acc);
          acc = MFMA16(av[q].z, bw[q].z, acc); acc = MFMA16(av[q].w, bw[q].w, acc);
        }
        float* part = buf_ptr(S, r.src2);                 // [4 waves][16 rows][16 columns]
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(wave * 16 + 4 * g + i) * 16 + c] = acc[i];
        __syncthreads();
        if (kq == 0 && c < N) {
          float y[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int prow = 4 * g + i;                   // row of the group's partial tiles
            float v = part[(wave * 16 + prow) * 16 + c];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += part[((wave + w) * 16 + prow) * 16 + c];
            y[i] = v + bias;
          }
          // (the LDS and the global stores in ONE predicated loop body trip a hipcc back-end error:
          // "Illegal instruction detected: Operand has incorrect register class")
          if (r.dst != B_NONE) {
            float* dd = buf_ptr(S, r.dst) + c;
            const int dld = buf_ld(r.dst);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (16 * rgw + 4 * g + i < R) dd[(16 * rgw + 4 * g + i) * dld] = y[i];
          }
          if (r.gd != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 16 * rgw + 4 * g + i;
              if (row < R && m0 + row < M) {
                int grow = m0 + row;
                if constexpr (prog_is_radar(PROG)) grow = S.rowg[row];
                stg1(r.gd + (size_t)grow * r.gld + c, y[i]);
              }
            }
          }
        }
      } break;
      } break;
      case K_RADAR_GATE: { if constexpr (prog_is_radar(PROG)) {   // the gate of HEAD:549-567 alone: hit counts + masks
        // the tokens' xy once per wave when they fit four words (T <= 256, the packed default; a tile's
        // rows are one sample's, two at a sample boundary); masks are kept up to 64 * HM_WORDS tokens
        constexpr int GW = 4;
        // the per-row constants of the gate (expf, clamp, sqrt_threshold: ~300 instructions) once per row, one
        // thread each, into the spare columns of the row's box / centre records (code <= 10 of 12, 3 of 4)
        // (the tokens' xy go out first: their latency runs under the per-row constants and the barrier)
        const bool keep = k.T <= 64 * HM_WORDS, cached = k.T <= 64 * GW;
        const int b0 = S.rowg[wave] / k.Q;
        float ty0[GW], ty1[GW];
        if (cached) {
          const float* rxy = k.tokens + (size_t)b0 * k.T * k.RI;
#pragma unroll
          for (int w = 0; w < GW; ++w) {
            const int t = min(64 * w + lane, k.T - 1);
            ty0[w] = rxy[(size_t)t * k.RI]; ty1[w] = rxy[(size_t)t * k.RI + 1];
          }
        }
        if (threadIdx.x < R) {
          const int i = threadIdx.x;
          const GateGeom::Pre pp = GateGeom::precompute(S.box[i][3], S.box[i][6], S.box[i][7], k.rmin[rep], k.rmax[rep]);
          S.box[i][10] = pp.ox; S.box[i][11] = pp.oy; S.cen[i][3] = pp.tstar;
        }
        __syncthreads();
#pragma unroll 1
        for (int row = wave; row < R; row += NW) {
          const int grow = S.rowg[row];
          const int b = grow / k.Q;
          int count = 0;
          if (cached && b == b0) {
            const GateGeom gg(S.cen[row][0], S.cen[row][1], GateGeom::Pre{S.box[row][10], S.box[row][11], S.cen[row][3]});
#pragma unroll
            for (int w = 0; w < GW; ++w) {
              if (64 * w < k.T) {                            // wave-uniform
                const unsigned long long mask =
                    __ballot(64 * w + lane < k.T && gg.hit(ty0[w], ty1[w], sqnorm2(ty0[w], ty1[w])));
                count += __popcll(mask);
                if (k.T - 1 >= 64 * w && k.T - 1 < 64 * w + 64 && ((mask >> (k.T - 1 - 64 * w)) & 1ull)) count += k.pad_mult - 1;
                if (lane == 0) reinterpret_cast<unsigned long long*>(&S.l[row][0])[w] = mask;
              }
            }
          } else {
            count = radar_gate_count(S.cen[row][0], S.cen[row][1], S.box[row][3], S.box[row][6], S.box[row][7],
                                     k.rmin[rep], k.rmax[rep], k.tokens + (size_t)b * k.T * k.RI, k.RI, k.T,
                                     k.pad_mult, lane, keep ? reinterpret_cast<unsigned long long*>(&S.l[row][0]) : nullptr);
          }
          if (lane == 0) {
            S.gate[row] = count;
            if (m0 + row < M) k.hits[(size_t)rep * (k.hits_stride ? k.hits_stride : (size_t)M) + grow] = count;
          }
        }
      } break;
      } break;
      case K_RADAR_ATTN: { if constexpr (prog_is_radar(PROG)) {   // distance-gated attention (HEAD:549-579)
        if (PROG == PROG_RADAR && (r.flags & F_IFHIT) && !tile_has_hit()) break;      // its output only feeds the (skipped) out_proj
#pragma unroll 1
        for (int row = wave; row < R; row += NW) {
          const int grow = S.rowg[row];
          const int b = grow / k.Q;
          float4 q4 = act_ld4<PL>(buf_ptr(S, r.src) + row * LD2, 4 * lane);
          const float* kv = (rep == 0 ? k.g[G_KV0] : rep == 1 ? k.g[G_KV1] : k.g[G_KV2]);
          int count = 0;
          const GateGeom gg(S.cen[row][0], S.cen[row][1], GateGeom::Pre{S.box[row][10], S.box[row][11], S.cen[row][3]});
          const unsigned long long* hm = k.T <= 64 * HM_WORDS ? reinterpret_cast<const unsigned long long*>(&S.l[row][0]) : nullptr;
          float4 o;
          if constexpr (PROG == PROG_RADAR_TRAIN) {
            // q is unscaled here (it goes to the tape as the projection's output); dropout on the probabilities,
            // site 4 * layer (nn.MultiheadAttention dropout, HEAD:129), index by the GLOBAL row
            q4.x *= k.qscale; q4.y *= k.qscale; q4.z *= k.qscale; q4.w *= k.qscale;
            DropK dk = k.rdrop;
            dk.site = 4u * (unsigned)rep;
            o = radar_attn_row_g<true>(gg, q4, k.tokens + (size_t)b * k.T * k.RI, k.RI, kv + (size_t)b * k.T * 512, 512,
                                       k.T, k.pad_mult, lane, count, dk, grow, hm);
            if (m0 + row < M) st4(r.gd + (size_t)grow * 256 + 4 * lane, o);          // tape: attention output
          } else {
            o = radar_attn_row_g(gg, q4, k.tokens + (size_t)b * k.T * k.RI, k.RI, kv + (size_t)b * k.T * 512, 512,
                                 k.T, k.pad_mult, lane, count, DropK(), 0, hm);
          }
          act_st4<PL>(buf_ptr(S, r.dst) + row * LD2, 4 * lane, o);
          if (lane == 0) S.gate[row] = count;        // the same count as K_RADAR_GATE's (same predicate)
        }
      } break;
      } break;
      case K_LOADG: { if constexpr (PROG == PROG_RADAR_BWD) {
        // dst[row][0..63] = src[grow][0..N) (zero filled: the A operand of a K = N linear step); F_CARRY (the
        // layer's first step): + the {0, 1, 4} columns of the box gradient the layer above sent down (the row's
        // box record), their sum stored for the weight gradient and carried on; the layer's hit counts -> gate
        const int N = r.N;
        float* dst = buf_ptr(S, r.dst);
        const float* src = uptr(r.p0);
        const bool carry = (r.flags & F_CARRY) != 0;
        for (int i = threadIdx.x; i < R * 64; i += NT) {
          const int row = i >> 6, c = i & 63;
          const int grow = min(m0 + row, M - 1);
          float v = c < N ? ldg1(src + (size_t)grow * N + c) : 0.0f;
          if (k.loss_vals != nullptr) {
            const float lv = k.loss_vals[2 * (k.nlayers - 1 - rep) + (carry ? 1 : 0)];
            if (!(fabsf(lv) <= 3.0e38f) || !(fabsf(v) <= 3.0e38f)) v = 0.0f;        // NaN / inf
          }
          if (carry && c < N) v += S.box[row][c];
          if (c < N && r.gd != nullptr && m0 + row < M) stg1(r.gd + (size_t)grow * N + c, v);
          dst[row * LD2 + c] = v;
        }
        if (carry) {
          __syncthreads();                 // every thread has read the old carry
          const int layer = k.nlayers - 1 - rep;
          if (blockIdx.x == 0 && threadIdx.x < 2 && k.loss_vals != nullptr && k.loss_out != nullptr) {
            // HEAD:915-916 (`loss[torch.isnan(loss)] = 0`): NaN -> 0, an infinite loss stays
            const float lv = k.loss_vals[2 * layer + threadIdx.x];
            k.loss_out[2 * layer + threadIdx.x] = lv != lv ? 0.0f : lv;
          }
          if (threadIdx.x < R * 16) {
            const int row = threadIdx.x >> 4, c = threadIdx.x & 15;
            if (c < 12) S.box[row][c] = (c == 0 || c == 1 || c == 4) ? dst[row * LD2 + c] : 0.0f;
            if (c == 15) S.gate[row] = k.hits[(size_t)layer * (k.hits_stride ? k.hits_stride : (size_t)M) + min(m0 + row, M - 1)];
          }
        }
      } break;
      } break;
      case K_MASKCOPY: { if constexpr (PROG == PROG_RADAR_BWD) {
        // dst = keep(seed, site, row * 256 + col) (.) [row gate] (.) src: a dropout (and the hit gate) applied to
        // a gradient -- the same mask as the forward's (HEAD:581-585), regenerated
        const float* src = buf_ptr(S, r.src);
        float* dst = buf_ptr(S, r.dst);
        const unsigned site = 4u * (unsigned)(k.nlayers - 1 - rep) + (unsigned)r.K;
        const bool gated = (r.flags & F_GATE) != 0;
        for (int row = wave; row < R; row += NW) {
          const int grow = min(m0 + row, M - 1);
          float4 v = *reinterpret_cast<const float4*>(src + row * LD2 + 4 * lane);
          const unsigned idx = (unsigned)grow * 256u + 4u * (unsigned)lane;
          const float sc = k.rdrop.scale;
          const unsigned dm = drop_keep4(k.rdrop.seed, site, idx, k.rdrop.thr);
          v.x = (dm & 1u) ? v.x * sc : 0.0f;
          v.y = (dm & 2u) ? v.y * sc : 0.0f;
          v.z = (dm & 4u) ? v.z * sc : 0.0f;
          v.w = (dm & 8u) ? v.w * sc : 0.0f;
          if (gated && S.gate[row] <= 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(dst + row * LD2 + 4 * lane) = v;
          if (r.gd != nullptr && m0 + row < M) st4(r.gd + (size_t)grow * 256 + 4 * lane, v);
        }
      } break;
      } break;
      case K_LN_BWD: { if constexpr (PROG == PROG_RADAR_BWD) {
        // y = [relu] LN(z) * gamma + beta, z = p1 (+ p2 unless F_LN_RELU: then p2 is y itself, the ReLU mask):
        // dz from dy (src), the same statistics as the forward (rowdev.hpp ln_row); dgamma (gt) / dbeta (p3) leave
        // as ONE atomic per channel and workgroup (the waves' sums meet in the LDS unit named by `res`)
        const float* dyb = buf_ptr(S, r.src);
        float* dzb = buf_ptr(S, r.dst);
        float* scratch = buf_ptr(S, r.res);
        const bool relu = (r.flags & F_LN_RELU) != 0;
        const float4 g = ld4(uptr(r.p0) + 4 * lane);
        float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int row = wave; row < R; row += NW) {
          const int grow = min(m0 + row, M - 1);
          const size_t o = (size_t)grow * 256 + 4 * lane;
          float4 z = ld4(uptr(r.p1) + o);
          float4 dy = *reinterpret_cast<const float4*>(dyb + row * LD2 + 4 * lane);
          if (r.p2 != nullptr) {
            const float4 t = ld4(uptr(r.p2) + o);
            if (relu) {
              if (t.x <= 0.f) dy.x = 0.f;
              if (t.y <= 0.f) dy.y = 0.f;
              if (t.z <= 0.f) dy.z = 0.f;
              if (t.w <= 0.f) dy.w = 0.f;
            } else {
              z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w;
            }
          }
          const float mean = wave_sum(z.x + z.y + z.z + z.w) * (1.0f / 256.0f);
          const float4 d = make_float4(z.x - mean, z.y - mean, z.z - mean, z.w - mean);
          const float q = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
          const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
          const float4 xh = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);
          if (m0 + row < M) {
            ag.x += dy.x * xh.x; ag.y += dy.y * xh.y; ag.z += dy.z * xh.z; ag.w += dy.w * xh.w;
            ab.x += dy.x; ab.y += dy.y; ab.z += dy.z; ab.w += dy.w;
          }
          const float4 dx = make_float4(dy.x * g.x, dy.y * g.y, dy.z * g.z, dy.w * g.w);
          const float s1 = wave_sum(dx.x + dx.y + dx.z + dx.w) * (1.0f / 256.0f);
          const float s2 = wave_sum(dx.x * xh.x + dx.y * xh.y + dx.z * xh.z + dx.w * xh.w) * (1.0f / 256.0f);
          const float4 dz = make_float4(rstd * (dx.x - s1 - xh.x * s2), rstd * (dx.y - s1 - xh.y * s2),
                                        rstd * (dx.z - s1 - xh.z * s2), rstd * (dx.w - s1 - xh.w * s2));
          *reinterpret_cast<float4*>(dzb + row * LD2 + 4 * lane) = dz;
          if (r.gd != nullptr && m0 + row < M) st4(r.gd + o, dz);
        }
        // the waves' partial sums: [NW][256] floats fit one unit at every tile height (R >= 4)
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
          __syncthreads();                       // (part 0: dy / the scratch unit's previous contents are done with)
          *reinterpret_cast<float4*>(scratch + wave * 256 + 4 * lane) = part == 0 ? ag : ab;
          __syncthreads();
          float* gdst = part == 0 ? r.gt : const_cast<float*>(uptr(r.p3));
          if (gdst != nullptr) {
            float t = scratch[threadIdx.x];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += scratch[w * 256 + threadIdx.x];
            if (k.detp != nullptr) acc_add(*k.detp, gdst + threadIdx.x, t);
            else unsafeAtomicAdd(gdst + threadIdx.x, t);
          }
        }
      } break;
      } break;
      case K_ATTN_BWD: { if constexpr (PROG == PROG_RADAR_BWD) {
        const int layer = k.nlayers - 1 - rep;
        const float* dao = buf_ptr(S, r.src);
        float* dqb = buf_ptr(S, r.dst);
        DropK dk = k.rdrop;
        dk.site = 4u * (unsigned)layer;
        for (int row = wave; row < R; row += NW) {
          const int grow = min(m0 + row, M - 1);
          const int b = grow / k.Q;
          float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
          if (S.gate[row] > 0 && m0 + row < M) {            // no radar return in the gate: no gradient through the attention
            const float* cxy = k.bwd_cxy[layer] + (size_t)grow * k.bwd_ldc[layer];
            const float* bx = k.bwd_box[layer] + (size_t)grow * k.code;
            float4 q4 = ld4(uptr(r.p0) + (size_t)grow * 256 + 4 * lane);
            q4.x *= k.qscale; q4.y *= k.qscale; q4.z *= k.qscale; q4.w *= k.qscale;
            const float4 o4 = ld4(uptr(r.p1) + (size_t)grow * 256 + 4 * lane);
            const float4 dO = *reinterpret_cast<const float4*>(dao + row * LD2 + 4 * lane);
            dq = radar_attn_bwd_row(cxy[0], cxy[1], bx[3], bx[6], bx[7], k.rmin[layer], k.rmax[layer], q4,
                                    k.tokens + (size_t)b * k.T * k.RI, k.RI, uptr(r.p2) + (size_t)b * k.T * 512,
                                    const_cast<float*>(uptr(r.p3)) + (size_t)b * k.T * 512, 512, k.T, k.pad_mult, dO, o4,
                                    dk, grow, lane,
                                    k.detp != nullptr ? det_shadow_of(*k.detp, uptr(r.p3) + (size_t)b * k.T * 512) : nullptr);
            dq.x *= k.qscale; dq.y *= k.qscale; dq.z *= k.qscale; dq.w *= k.qscale;
          }
          *reinterpret_cast<float4*>(dqb + row * LD2 + 4 * lane) = dq;
          if (r.gd != nullptr && m0 + row < M) st4(r.gd + (size_t)grow * 256 + 4 * lane, dq);
        }
      } break;
      } break;
      case K_BOXADD: { if constexpr (prog_is_radar(PROG)) {   // box = reg + reference (HEAD:599-600, 664-665, 722-723); next ref (HEAD:615-617)
        const int row = threadIdx.x >> 4, j = threadIdx.x & 15;     // one (row, box column) per thread
        if (row < R && j < k.code) {
          const int c = j == 0 ? 0 : j == 1 ? 1 : j == 4 ? 2 : -1;
          float bx = S.l[row][j];
          if (c >= 0) { bx += S.cen[row][c]; S.cen[row][c] = bx; }   // only this thread touches cen[row][c]
          S.box[row][j] = bx;
          if (m0 + row < M) k.all_box[((size_t)rep * M + S.rowg[row]) * k.code + j] = bx;
        }
      } break;
      } break;
      default: break;
    }
    STEP_STAMP();
    if (r.sync) __syncthreads();
    STEP_STAMP();
    DUMP_STEP(idx, r.dst);
    ++idx;
  }
  WG_STAMP(1);
}

template <int R, int PROG, bool DROP = false, int MM = 0, bool PRE = false>
__global__ __launch_bounds__(nw_of(R) * 64, 2) void chain_kernel(ChainDev k, Recs<rec_cap(PROG), nw_of(R)> recs) {
  chain_body<R, PROG, DROP, MM, PRE>(k, recs.s, blockIdx.x);
}

// Two programs in one launch: workgroups [0, na) run the decoder layer `ka` on RA-row
// tiles, the rest the radar encoders `kb` on RB-row tiles.  Decoder layer 0 carries the
// encoders this way: as a branch of the hipGraph on a side stream, the fork and the join
// each left a ~10 us hole in the replayed frame (profiles: rocprofv3 kernel trace).
template <int RA, int RB, int PROGB, int MM = 0, bool PRE = false>
__global__ __launch_bounds__(nw_of(RA) * 64, 2) void chain_dual_kernel(ChainDev ka, ChainDev kb, int na,
                                                           Recs<rec_cap(PROG_DECODER), nw_of(RA)> ra,
                                                           Recs<rec_cap(PROGB), nw_of(RB)> rb) {
  static_assert(nw_of(RA) == nw_of(RB), "both programs of a launch run with the same workgroup size");
  if ((int)blockIdx.x < na) chain_body<RA, PROG_DECODER, false, MM, PRE>(ka, ra.s, blockIdx.x);
  else chain_body<RB, PROGB, false, MM>(kb, rb.s, (int)blockIdx.x - na);
}

// ---- host side: the step table of a program -> resolved records --------------------------
template <int R, int PROG>
int lds_off(int id) {     // float offset of an LDS buffer from the start of shared memory
  using Lds = ChainLds<R, rec_cap(PROG)>;
  const int unit0 = (int)(offsetof(Lds, unit) / 4), ustride = R * LD2;
  switch (id) {
    case B_A: return unit0 + ustride;
    case B_X: return unit0;
    case B_U1: return unit0 + ustride;
    case B_U2: return unit0 + 2 * ustride;
    case B_U3: return unit0 + 3 * ustride;
    case B_L: return (int)(offsetof(Lds, l) / 4);
    default: return -1;
  }
}
template <int R, int PROG>
constexpr size_t chain_lds_bytes() { return sizeof(ChainLds<R, rec_cap(PROG)>); }
inline int buf_ld_h(int id) { return id == B_A ? LD5 : id == B_L ? LDL : LD2; }

template <int R, int PROG, int MM = 0>
void resolve_program(ChainK& k, StepAllT<nw_of(R)>* out) {
  constexpr int NW = nw_of(R);
  const StepDesc* table = prog_table(PROG);
  constexpr int nsteps = table_steps(PROG);
  const int nrep = (prog_is_radar(PROG) || PROG == PROG_RADAR_BWD) ? k.nlayers : 1;
  const int total = nsteps * nrep;
  memset(out, 0, sizeof(StepAllT<NW>) * rec_cap(PROG));
  for (int idx = 0; idx < total; ++idx) {
    const int rep = idx / nsteps, si = idx - rep * nsteps;
    const int layer = PROG == PROG_RADAR_BWD ? k.nlayers - 1 - rep : rep;          // the backward walks the layers down
    const int pair0 = (prog_is_radar(PROG) || PROG == PROG_RADAR_BWD) ? layer * RADAR_PAIRS : 0;
    const StepDesc d = table[si];
    StepRes& r = out[idx].r;
    r.K = d.K; r.N = d.N;
    r.kind = d.kind; r.src = d.src; r.src2 = d.src2; r.dst = d.dst; r.res = d.res; r.act = d.act;
    r.flags = d.flags; r.sync = d.sync; r.rep = (short)rep; r.si = (short)si;
    if ((d.flags & F_SKIP_NONEXT) && !k.has_next) r.kind = K_NOP;
    // inference opt-in (tc_head_options.last_level_cls_only): get_bboxes decodes the last level
    // only and levels 1-2 hand only their BOX to the next gate (HEAD:615-617, 1003-1023), so the
    // class MLPs (pairs 6..10: final_cls.0 / n1 / .3 / n4 / .6) of the earlier layers are dropped
    if (PROG == PROG_RADAR && k.last_cls_only && rep + 1 < nrep && d.wp >= 6 && d.wp <= 10 &&
        (d.kind == K_LINEAR || d.kind == K_LN || d.kind == K_NARROW))
      r.kind = K_NOP;
    if (d.gsel != G_NONE) { r.gd = k.g[d.gsel]; r.gld = k.g_ld[d.gsel]; r.gmod = k.g_mod[d.gsel]; }
    if (d.gtsel != G_NONE) r.gt = k.g[d.gtsel];
    bool taped = false;
    if (PROG == PROG_RADAR_TRAIN) {
      // q stays unscaled (tape), nothing is skipped
      if (d.flags & F_SCALEQ) r.flags = (short)(r.flags & ~F_SCALEQ);
      r.flags = (short)(r.flags & ~F_IFHIT);
      if (d.kind == K_LINEAR && d.wp == 4) r.flags = (short)(r.flags | F_GPRE);     // tape: the FFN output itself
      const short ts = RADAR_TAPE[si];
      if (ts != T_NONE && r.gd == nullptr) { r.gd = k.tape[ts] + (size_t)rep * k.tape_stride; r.gld = 256; taped = true; }
    }
    if (PROG == PROG_RADAR_BWD) {
      if ((d.flags & F_NOT_LAYER0) && layer == 0) r.kind = K_NOP;
      const short ds = BWD_STORE[si], ts = BWD_TAPE[si], ts2 = BWD_TAPE2[si];
      if (ds != D_NONE && k.dy[ds] != nullptr) { r.gd = k.dy[ds] + (size_t)layer * k.dy_stride; taped = true; }
      const float* t1 = ts != T_NONE ? k.tape[ts] + (size_t)layer * k.tape_stride : nullptr;
      const float* t2 = ts2 != T_NONE ? k.tape[ts2] + (size_t)layer * k.tape_stride : nullptr;
      if (d.kind == K_LINEAR && (d.flags & F_CMASK)) r.gt = const_cast<float*>(t1);
      if (d.kind == K_LN_BWD) {
        const tc_linear n = k.pairs[pair0 + d.wp], gn = k.gpairs[pair0 + d.wp];
        r.p0 = n.w; r.p1 = t1; r.p2 = t2; r.gt = const_cast<float*>(gn.w); r.p3 = gn.b;
      }
      if (d.kind == K_LOADG) {
        r.N = d.N == N_CODE ? k.code : k.ncls;
        r.p0 = (d.N == N_CODE ? k.d_box : k.d_cls) + (size_t)layer * k.M * r.N;
      }
      if (d.kind == K_MASKCOPY) r.K = d.K;
      if (d.kind == K_ATTN_BWD) {
        r.p0 = k.tape[T_QP] + (size_t)layer * k.tape_stride; r.p1 = k.tape[T_AO] + (size_t)layer * k.tape_stride;
        r.p2 = k.g[G_KV0 + layer]; r.p3 = k.dkv[layer];
      }
    }
    if (PROG == PROG_RADAR_ENC_TRAIN) {
      const short ts = ENC_TAPE[si], ts2 = ENC_TAPE2[si];
      if (ts != T_NONE && r.gd == nullptr) { r.gd = k.tape[ts]; r.gld = 256; taped = true; }
      if (ts2 != T_NONE) r.gt = k.tape[ts2];
    }
    if (d.kind == K_LINEAR) {
      const tc_linear pr = k.pairs[pair0 + d.wp];
      r.K = d.K == 36 ? k.RI : d.K == N_CODE ? k.code : d.K == N_CLS ? k.ncls : d.K;
      r.N = d.N == N_LOGITS ? k.nlogits : d.N == N_CODE ? k.code : d.N == N_CLS ? k.ncls : d.N;
      const int woff = (d.flags & F_WOFF) ? 512 : 0;
      r.p0 = pr.w + (size_t)woff * ((r.K + 63) & ~63);   // packed: a 64-row tile = 64 * kpad floats
      if (R >= 16) r.p0 += (MM == 1 ? 2 : 1) * k.w16_delta;   // the 16x16x4 copy / the two-plane f16 copy behind it (pack.hip); launch_r checks delta != 0
      r.p1 = pr.b ? pr.b + woff : nullptr;
      if (PROG == PROG_RADAR_BWD) r.p1 = nullptr;         // dx = dy W: no bias
      if (d.gsel == G_CLS) r.gd += (size_t)rep * k.M * k.ncls;
      if (taped) r.gld = r.N;                              // the taped tensor is [M, N]
    } else if (d.kind == K_NARROW) {
      // NOT packed (tc_head_pack_weights leaves these three heads in the nn.Linear layout): W [N][256]
      const tc_linear pr = k.pairs[pair0 + d.wp];
      r.N = d.N == N_CODE ? k.code : d.N == N_CLS ? k.ncls : d.N;
      r.p0 = pr.w; r.p1 = pr.b;
      if (d.gsel == G_CLS) r.gd += (size_t)rep * k.M * k.ncls;
      if (taped) r.gld = r.N;
    } else if (d.kind == K_LN || d.kind == K_POSENC) {
      const tc_linear n = k.pairs[pair0 + d.wp];
      r.p0 = n.w; r.p1 = n.b;
      if (d.kind == K_POSENC || d.src2 != B_NONE) {
        const tc_linear n2 = k.pairs[pair0 + d.wp2];
        r.p2 = n2.w; r.p3 = n2.b;
      }
    }
    r.src_off = lds_off<R, PROG>(r.src); r.src2_off = lds_off<R, PROG>(r.src2);
    EpiRec& e = out[idx].e;
    e.gd = r.gd; e.gt = r.gt; e.gld = r.gld; e.act = r.act; e.flags = r.flags; e.N = r.N;
    e.has_bias = r.p1 != nullptr; e.woff = (r.flags & F_WAVE1) ? 1 : 0;
    e.drop_site = 0;
    if (PROG == PROG_DECODER && k.drop.thr != 0 && d.kind == K_LINEAR) {
      // mmcv MultiheadAttention's output dropout, Detr3DCrossAtten.dropout (XFMR:378), the FFN's two
      const int off = d.wp == 1 ? 1 : d.wp == 4 ? 2 : d.wp == 10 ? 3 : d.wp == 11 ? 4 : 0;
      if (off) e.drop_site = (int)k.drop.site + off + 1;
    }
    if (PROG == PROG_RADAR_TRAIN && d.kind == K_LINEAR) {
      // rf_dropout2 on out_proj's output (before gate-free residual, HEAD:581), rf_dropout on relu(linear1),
      // rf_dropout3 on linear2's output (HEAD:584-585): sites 4 r + {1, 2, 3}
      const int off = d.wp == 1 ? 1 : d.wp == 3 ? 2 : d.wp == 4 ? 3 : 0;
      if (off) e.drop_site = 4 * rep + off + 1;
    }
    e.dst_off = lds_off<R, PROG>(r.dst); e.dst_ld = buf_ld_h(r.dst);
    e.res_off = lds_off<R, PROG>(r.res); e.res_ld = buf_ld_h(r.res);
  }
  // per (step, wave): the next linear step inside the same run of light steps where the
  // wave owns a column tile, and that step's first weight item
  for (int idx = 0; idx < total; ++idx) {
    for (int w = 0; w < NW; ++w) {
      PreRec& pr = out[idx].p[w];
      pr.first = nullptr; pr.nidx = -1;
      for (int j = idx + 1; j < total; ++j) {
        const StepRes& n = out[j].r;
        if (n.kind == K_LN || n.kind == K_NOP) continue;
        if (n.kind != K_LINEAR) break;
        const int vw = (w - ((n.flags & F_WAVE1) ? 1 : 0)) & (NW - 1);
        const size_t kpad = (size_t)((n.K + 63) & ~63);
        if (R == 32) {
          // 32-row tiles: a wave owns 32-COLUMN tiles (linear_step32h): tile vw = half (vw & 1) of the 64-column tile
          // vw >> 1 of the packed planes -- fragments [kk][2 half .. 2 half + 1][p] of every 64-deep item
          if (vw >= ((n.N + 31) >> 5)) continue;
          pr.nidx = j;
          pr.first = n.p0 + (size_t)(vw >> 1) * 64 * kpad + (size_t)(vw & 1) * 4 * 256;
        } else {
          if (vw >= ((n.N + 63) >> 6)) continue;
          pr.nidx = j;
          pr.first = n.p0 + (size_t)vw * 64 * kpad;
        }
        break;
      }
    }
  }
  k.total = total;
  int early = 0;
  while (early < 3 && early < total && out[early].r.kind == K_LOAD) ++early;
  k.early_n = early;
}

template <int RA, int RB, int PROGB, int MM = 0, bool PRE = false>
int launch_dual_r(const ChainK& ka_, const ChainK& kb_, hipStream_t s, const char* what) {
  constexpr size_t lds = chain_lds_bytes<RA, PROG_DECODER>() > chain_lds_bytes<RB, PROGB>() ? chain_lds_bytes<RA, PROG_DECODER>() : chain_lds_bytes<RB, PROGB>();
  static DeviceOnce once;
  if (const int once_dev = once.need(); once_dev >= 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_dual_kernel<RA, RB, PROGB, MM, PRE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("chain: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    once.done(once_dev);
  }
  ChainK ka = ka_, kb = kb_;
  TC_REQUIRE((RA < 16 || ka.w16_delta != 0) && (RB < 16 || kb.w16_delta != 0),
             "%s: 16- / 32-row tiles need weights from a tc_head_pack_weights view (packed16_delta is 0)", what);
  Recs<rec_cap(PROG_DECODER), nw_of(RA)> ra;
  Recs<rec_cap(PROGB), nw_of(RB)> rb;
  resolve_program<RA, PROG_DECODER, MM>(ka, ra.s);
  resolve_program<RB, PROGB, MM>(kb, rb.s);
  const int na = (ka.M + RA - 1) / RA, nb = (kb.M + RB - 1) / RB;
  hipLaunchKernelGGL((chain_dual_kernel<RA, RB, PROGB, MM, PRE>), dim3(na + nb), dim3(nw_of(RA) * 64), lds, s,
                     static_cast<const ChainDev&>(ka), static_cast<const ChainDev&>(kb), na, ra, rb);
  return check_launch(what);
}

template <int R, int PROG, bool DROP = false, int MM = 0, bool PRE = false>
int launch_r(const ChainK& k_, hipStream_t s, const char* what) {
  static DeviceOnce once;
  if (const int once_dev = once.need(); once_dev >= 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<R, PROG, DROP, MM, PRE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)chain_lds_bytes<R, PROG>());
    if (e != hipSuccess) { set_error("chain: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    once.done(once_dev);
  }
  ChainK k = k_;
  TC_REQUIRE(R < 16 || k.w16_delta != 0,
             "%s: 16- / 32-row tiles need weights from a tc_head_pack_weights view (packed16_delta is 0)", what);
  Recs<rec_cap(PROG), nw_of(R)> recs;
  resolve_program<R, PROG, MM>(k, recs.s);
  constexpr size_t lds = chain_lds_bytes<R, PROG>();
  hipLaunchKernelGGL((chain_kernel<R, PROG, DROP, MM, PRE>), dim3((k.M + R - 1) / R), dim3(nw_of(R) * 64), lds, s,
                     static_cast<const ChainDev&>(k), recs);
  return check_launch(what);
}

#ifdef TC_CHAIN_STAMPS
static int g_dbg_override = -1;     // tc_debug_set_chain_dbg: TRANSCAR_CHAIN_DBG at run time (tools/stagger_probe.py)
#endif
void init_k(ChainK& k) {
  memset(&k, 0, sizeof(k));
#ifdef TC_CHAIN_STAMPS
  static const int dbg = [] { const char* e = getenv("TRANSCAR_CHAIN_DBG"); return e ? atoi(e) : 0; }();
  k.dbg = g_dbg_override >= 0 ? g_dbg_override : dbg;
#endif
}

// Row-tile height (measured, bench.py --pair 1/2/4/8): 4 rows while that gives about one
// workgroup per CU (one frame: 225 workgroups), 8 rows up to two frames per launch, 16 beyond.
// All three run two workgroups per CU: the 16-row tiles since round 2 (four LDS units = 76 KB,
// one weight buffer refilled in place = 239 VGPRs; before: 120 KB / 304 VGPRs, one per CU, and
// they lost to 8 rows everywhere).  4 frames per launch x 3 lanes: 4214 (8 rows) -> 4597 frames/s.
// The height is a per-call argument (tc_head_options.chain_tile_rows / tile_rows of
// tc_decoder_layer_tail_fwd): no process-global state.
// Round 5: 32 rows (one workgroup of 8 waves per CU, activations as planes) once the 16-row tiles would need two
// workgroups per CU (more than 4096 rows: five frames or more) and the matrix path is not pinned to f32: the CU then pulls
// a layer's weights through its L2 port once per 32 rows without relying on two workgroups sharing the L1 in lockstep
// (nine frames: decoder chain 110 -> 102 us, radar chain 233 -> 214 us; profiles/r5_*).
int tile_rows(const ChainK& k) {
  if (k.tile_rows) return k.tile_rows;
  return k.M <= 1024 ? 4 : k.M <= 2048 ? 8 : (k.M > 4096 && k.matrix_path != TC_MATRIX_F32) ? 32 : 16;
}

// The matrix path of the 16-row tiles (tc_head_options.matrix_path): automatic = the two-plane f16 form on the
// matrix cores (measured: DESIGN.md section 5 "Round 4"); TC_MATRIX_F32 keeps the exact fp32 FMA chains of
// v_mfma_f32_16x16x4_f32.  4- and 8-row tiles always compute in fp32 (their item loops are bound by the weight
// stream, and both copies are 4 bytes per weight).
bool use_f16x2(const ChainK& k) { return k.matrix_path != TC_MATRIX_F32; }

template <int PROG>
int launch_rows(const ChainK& k, hipStream_t s, const char* what) {
  const int rows = tile_rows(k);
  if (rows == 4) return launch_r<4, PROG>(k, s, what);
  if (rows == 8) return launch_r<8, PROG>(k, s, what);
  if (rows == 32) {
    TC_REQUIRE(use_f16x2(k), "%s: 32-row tiles exist on the f16x2 matrix path only", what);
    if constexpr (PROG == PROG_DECODER) { if (k.pre != nullptr) return launch_r<32, PROG, false, 1, true>(k, s, what); }
    return launch_r<32, PROG, false, 1>(k, s, what);
  }
  if constexpr (PROG == PROG_DECODER) { if (k.pre != nullptr && use_f16x2(k)) return launch_r<16, PROG, false, 1, true>(k, s, what); }
  if (use_f16x2(k)) return launch_r<16, PROG, false, 1>(k, s, what);
  return launch_r<16, PROG>(k, s, what);
}

int launch(const ChainK& k, hipStream_t s, const char* what) {
  switch (k.program) {
    case PROG_DECODER:
      if (k.drop.thr != 0) {            // train-mode dropout: its own instantiations -- 4- and 8-row tiles, and the
        //                                  16-row tiles of the f16x2 path (several frames per launch: the trainer's
        //                                  batched look-ahead of the frozen decoder, round 4)
        TC_REQUIRE((unsigned long long)(k.drop.rows_per_sample ? k.drop.rows_per_sample : k.M) * 512ull < (1ull << 32),
                   "decoder_chain: dropout index space");
        const int rows = tile_rows(k);
        if (rows == 32) {
          TC_REQUIRE(use_f16x2(k), "%s: 32-row tiles exist on the f16x2 matrix path only", what);
          return launch_r<32, PROG_DECODER, true, 1>(k, s, what);
        }
        if (rows == 16 && use_f16x2(k)) return launch_r<16, PROG_DECODER, true, 1>(k, s, what);
        return rows == 4 ? launch_r<4, PROG_DECODER, true>(k, s, what) : launch_r<8, PROG_DECODER, true>(k, s, what);
      }
      return launch_rows<PROG_DECODER>(k, s, what);
    case PROG_RADAR: return launch_rows<PROG_RADAR>(k, s, what);
    // training forward: 4-row tiles up to 1024 rows (one frame per GPU, CFG:188), 8 beyond; always the DROP
    // instantiation (thr 0 keeps everything)
    case PROG_RADAR_TRAIN:
      TC_REQUIRE((unsigned long long)k.M * 512ull < (1ull << 32), "radar_train: dropout index space");
      return tile_rows(k) == 4 ? launch_r<4, PROG_RADAR_TRAIN, true>(k, s, what)
                               : launch_r<8, PROG_RADAR_TRAIN, true>(k, s, what);
    case PROG_RADAR_ENC_TRAIN: return launch_r<4, PROG_RADAR_ENC_TRAIN>(k, s, what);
    case PROG_RADAR_BWD:
      TC_REQUIRE((unsigned long long)k.M * 512ull < (1ull << 32), "radar_bwd: dropout index space");
      return tile_rows(k) == 4 ? launch_r<4, PROG_RADAR_BWD, true>(k, s, what)
                               : launch_r<8, PROG_RADAR_BWD, true>(k, s, what);
    // the prologue runs once per checkpoint on Q rows (tc_head_pack_weights); stand-alone radar
    // encoders take the fewest workgroups (16-row tiles): 225 + 64 workgroups of 4-row tiles
    // did not fit 256 CUs next to a decoder layer
    case PROG_PROLOGUE: return launch_r<4, PROG_PROLOGUE>(k, s, what);
    default: return use_f16x2(k) ? launch_r<16, PROG_RADAR_ENC, false, 1>(k, s, what) : launch_r<16, PROG_RADAR_ENC>(k, s, what);
  }
}

}  // namespace

#ifdef TC_CHAIN_STAMPS
extern "C" int tc_debug_chain_stamps(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chain_stamps), sizeof(long long) * CH_NW_MAX * 64);
}
extern "C" int tc_debug_chain_stamps2(int block2, long long* host_out) {
  if (host_out == nullptr) return (int)hipMemcpyToSymbol(HIP_SYMBOL(tc::g_stamp_block2), &block2, sizeof(int));
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tc::g_chain_stamps2), sizeof(long long) * tc::CH_NW_MAX * 64);
}
extern "C" int tc_debug_set_chain_dbg(int v) { tc::g_dbg_override = v; return 0; }
// zero the per-CU arrival tickets (stream-ordered) / read {cu key, ticket} of every workgroup of the last launch
extern "C" int tc_debug_stagger_reset(void* stream) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(tc::g_cu_ticket));
  if (e == hipSuccess) e = hipMemsetAsync(p, 0, sizeof(unsigned) * 2048, static_cast<hipStream_t>(stream));
  return (int)e;
}
extern "C" int tc_debug_wg_cu(int* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tc::g_wg_cu), sizeof(int) * 2048 * 2);
}
extern "C" int tc_debug_wg_spans(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg_span), sizeof(long long) * 1024 * 2);
}
extern "C" int tc_debug_cam_stamps(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cam_stamps), sizeof(long long) * 4 * 8);
}
extern "C" int tc_debug_chain_sub(int step, long long* host_out) {
  if (host_out == nullptr) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sub_step), &step, sizeof(int));
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_chain_sub), sizeof(long long) * CH_NW_MAX * 64);
}
#endif

#ifdef TC_CHAIN_DUMP
extern "C" int tc_debug_set_chain_dump(void* buf, long long floats) {
  float* p = static_cast<float*>(buf);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_chain_dump), &p, sizeof(p));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_chain_dump_floats), &floats, sizeof(floats));
  return (int)e;
}
#endif
#ifdef TC_DIAG_BUILD
// marks a diagnostic build (make DUMP=1): transcar_amd/_lib.py refuses it without TRANSCAR_ALLOW_STAMPS=1
extern "C" int tc_debug_diag_build() { return 2; }
#endif

void fill_camk(const CamSampleArgs& a, CamK& p);   // cam_sample.hip

int launch_prologue(const PrologueArgs& a, hipStream_t s) {
  ChainK k;
  init_k(k);
  k.program = PROG_PROLOGUE; k.M = a.M; k.Q = a.Q; k.has_next = 1; k.w16_delta = a.w16_delta;
  k.pairs[0] = a.refpts; k.pairs[16] = a.in_proj;
  k.g[G_POS] = const_cast<float*>(a.qe); k.g_ld[G_POS] = 512; k.g_mod[G_POS] = a.Q;
  k.g[G_XIN] = const_cast<float*>(a.qe) + 256; k.g_ld[G_XIN] = 512; k.g_mod[G_XIN] = a.Q;
  k.g[G_INITREF] = a.init_ref; k.g_ld[G_INITREF] = 3;
  k.g[G_QK] = a.qk; k.g_ld[G_QK] = 512; k.g[G_VT] = a.vt;
  k.qscale = a.qscale; k.qpad = a.qpad;
  return launch(k, s, "chain(prologue)");
}

static int make_decoder_k(const DecoderChainArgs& a, ChainK& k) {
  TC_REQUIRE(a.code <= 12 && a.cam.num_cams * a.cam.feats.num_levels <= 32, "decoder_chain: code/logit width");
  TC_REQUIRE(a.cam.num_cams <= 8, "decoder_chain: num_cams=%d (<= 8)", a.cam.num_cams);
  for (int l = 0; l < a.cam.feats.num_levels; ++l)      // pixel indices are 32-bit in the kernel
    TC_REQUIRE((long long)a.cam.B * a.cam.num_cams * a.cam.feats.H[l] * a.cam.feats.W[l] < (1ll << 31),
               "decoder_chain: level %d has too many pixels for one call", l);
  init_k(k);
  k.program = PROG_DECODER; k.M = a.M; k.Q = a.Q; k.code = a.code;
  k.nlogits = a.cam.num_cams * a.cam.feats.num_levels;
  const tc_decoder_layer& w = *a.w;
  const tc_pos_encoder& pe = w.position_encoder;
  k.pairs[0] = w.self_attn.in_proj; k.pairs[1] = w.self_attn.out_proj;
  k.pairs[2] = tc_linear{w.norm0.g, w.norm0.b};
  k.pairs[3] = w.attention_weights; k.pairs[4] = w.output_proj; k.pairs[5] = pe.l0;
  k.pairs[6] = tc_linear{pe.n1.g, pe.n1.b}; k.pairs[7] = pe.l3; k.pairs[8] = tc_linear{pe.n4.g, pe.n4.b};
  k.pairs[9] = tc_linear{w.norm1.g, w.norm1.b}; k.pairs[10] = w.ffn0; k.pairs[11] = w.ffn1;
  k.pairs[12] = tc_linear{w.norm2.g, w.norm2.b};
  k.pairs[13] = w.reg.l0; k.pairs[14] = w.reg.l2; k.pairs[15] = w.reg.l4;
  k.has_next = a.next_in_proj != nullptr;
  if (k.has_next) k.pairs[16] = *a.next_in_proj;
  k.w16_delta = w.packed16_delta;
  k.g[G_ATTN_O] = const_cast<float*>(a.attn_o); k.g_ld[G_ATTN_O] = 256; k.g_mod[G_ATTN_O] = a.attn_mod;
  k.g[G_XIN] = const_cast<float*>(a.x_in); k.g_ld[G_XIN] = a.x_ld; k.g_mod[G_XIN] = a.x_mod;
  k.g[G_POS] = const_cast<float*>(a.qe); k.g_ld[G_POS] = 512; k.g_mod[G_POS] = a.Q;
  k.g[G_HS] = a.hs; k.g_ld[G_HS] = 256;
  k.g[G_QK] = a.qk; k.g_ld[G_QK] = 512; k.g[G_VT] = a.vt;
  k.qscale = a.qscale; k.qpad = a.qpad;
  k.ref_in = a.ref_in; k.ref_mod = a.ref_mod; k.ref_out = a.ref_out; k.box_m = a.box_m;
  fill_camk(a.cam, k.cam);
  k.pair_counter = a.cam.pair_counter;
  TC_REQUIRE((a.pre == nullptr) == (a.premask == nullptr), "decoder_chain: pre-gathered values and masks come together");
  k.pre = a.pre; k.premask = a.premask;
  TC_REQUIRE(a.tile_rows == 0 || a.tile_rows == 4 || a.tile_rows == 8 || a.tile_rows == 16 || a.tile_rows == 32,
             "decoder_chain: tile_rows=%d (0 = automatic, 4, 8, 16 or 32)", a.tile_rows);
  k.tile_rows = a.tile_rows;
  k.matrix_path = a.matrix_path;
  k.drop = a.drop;
  k.range_status = a.range_status;
  return 0;
}

int launch_decoder_chain(const DecoderChainArgs& a, hipStream_t s) {
  ChainK k;
  int rc = make_decoder_k(a, k);
  if (rc != 0) return rc;
  return launch(k, s, "chain(decoder)");
}

static int make_radar_enc_k(const RadarEncodeArgs& a, ChainK& k, int part = 0) {
  TC_REQUIRE(a.RI <= 64 && (a.RI & 3) == 0, "radar_encode: radar_in_dims=%d", a.RI);
  TC_REQUIRE(a.nlayers == TC_MAX_RADAR_LAYERS, "radar_encode: %d radar layers (3 supported)", a.nlayers);
  init_k(k);
  k.program = part == 1 ? PROG_RADAR_ENC_A : part == 2 ? PROG_RADAR_ENC_B : PROG_RADAR_ENC;
  if (a.tape != nullptr) {
    TC_REQUIRE(part == 0, "radar_encode: the training forward runs the whole encoder program");
    k.program = PROG_RADAR_ENC_TRAIN;
    for (int i = 0; i < T_COUNT; ++i) k.tape[i] = a.tape[i];
  }
  k.M = a.M; k.Q = 1; k.RI = a.RI; k.tokens = a.tokens; k.has_next = 1;
  k.w16_delta = a.w16_delta; k.matrix_path = a.matrix_path; k.range_status = a.range_status;
  TC_REQUIRE(part == 0 || a.radar_feat != nullptr, "radar_encode: split programs need the radar_feat buffer");
  k.g[G_RFEAT] = a.radar_feat; k.g_ld[G_RFEAT] = 256;
  k.pairs[0] = a.rpe.l0; k.pairs[1] = tc_linear{a.rpe.n1.g, a.rpe.n1.b}; k.pairs[2] = a.rpe.l3;
  k.pairs[3] = tc_linear{a.rpe.n4.g, a.rpe.n4.b}; k.pairs[4] = a.f0; k.pairs[5] = a.f2; k.pairs[6] = a.f4;
  for (int r = 0; r < TC_MAX_RADAR_LAYERS; ++r) {
    k.pairs[7 + r] = a.kvproj[r];
    k.g[G_KV0 + r] = a.kv[r]; k.g_ld[G_KV0 + r] = 512;
  }
  return 0;
}

int launch_radar_encode(const RadarEncodeArgs& a, hipStream_t s) {
  ChainK k;
  int rc = make_radar_enc_k(a, k);
  if (rc != 0) return rc;
  return launch(k, s, "chain(radar_encode)");
}

// a decoder layer and (a part of) the radar encoders in one launch;
// part 0: the whole encoder program, 1 / 2: its halves (PROG_RADAR_ENC_A / _B)
int launch_decoder_chain_with_encoders(const DecoderChainArgs& d, const RadarEncodeArgs& e, int part,
                                       hipStream_t s) {
  ChainK kd, ke;
  int rc = make_decoder_k(d, kd);
  if (rc != 0) return rc;
  rc = make_radar_enc_k(e, ke, part);
  if (rc != 0) return rc;
  TC_REQUIRE(kd.drop.thr == 0, "decoder dropout: launch the encoders on their own (launch_radar_encode)");
  const int rows = tile_rows(kd);
  const char* what = "chain(decoder + radar_encode)";
#define TC_DUAL(RA, RB, MM, ...)                                                              \
  (part == 1 ? launch_dual_r<RA, RB, PROG_RADAR_ENC_A, MM, ##__VA_ARGS__>(kd, ke, s, what)       \
   : part == 2 ? launch_dual_r<RA, RB, PROG_RADAR_ENC_B, MM, ##__VA_ARGS__>(kd, ke, s, what)     \
               : launch_dual_r<RA, RB, PROG_RADAR_ENC, MM, ##__VA_ARGS__>(kd, ke, s, what))
  // 4-row decoder tiles run two workgroups per CU (256 VGPRs): the encoder rows then use 4-row
  // tiles too -- a 16-row body in the same kernel would spill ~100 registers at that budget,
  // and 225 + T/4 workgroups fit the chip at two per CU
  if (rows == 4) return TC_DUAL(4, 4, 0);
  if (rows == 8) return TC_DUAL(8, 8, 0);
  if (rows == 32) {
    TC_REQUIRE(use_f16x2(kd), "%s: 32-row tiles exist on the f16x2 matrix path only", what);
    if (kd.pre != nullptr) return TC_DUAL(32, 32, 1, true);
    return TC_DUAL(32, 32, 1);
  }
  if (use_f16x2(kd) && kd.pre != nullptr) return TC_DUAL(16, 16, 1, true);
  if (use_f16x2(kd)) return TC_DUAL(16, 16, 1);
  return TC_DUAL(16, 16, 0);
#undef TC_DUAL
}

int launch_radar_chain(const RadarChainArgs& a, hipStream_t s) {
  // code <= 10: columns 10, 11 of a row's box record carry the gate's precomputed offsets (K_RADAR_GATE)
  TC_REQUIRE(a.code <= 10 && a.ncls <= 32, "radar_chain: code=%d ncls=%d", a.code, a.ncls);
  TC_REQUIRE(a.nlayers >= 1 && a.nlayers <= TC_MAX_RADAR_LAYERS, "radar_chain: nlayers=%d", a.nlayers);
  ChainK k;
  init_k(k);
  k.program = PROG_RADAR; k.M = a.M; k.Q = a.Q; k.code = a.code; k.ncls = a.ncls; k.nlayers = a.nlayers;
  k.w16_delta = a.w[0].packed16_delta;
  k.row_perm = a.row_perm;
  k.has_next = 1;
  for (int r = 0; r < a.nlayers; ++r) {
    const tc_radar_layer& w = a.w[r];
    tc_linear* p = &k.pairs[r * RADAR_PAIRS];
    p[0] = w.attn.in_proj; p[1] = w.attn.out_proj; p[2] = tc_linear{w.norm2.g, w.norm2.b};
    p[3] = w.linear1; p[4] = w.linear2; p[5] = tc_linear{w.norm3.g, w.norm3.b};
    p[6] = w.final_cls.l0; p[7] = tc_linear{w.final_cls.n1.g, w.final_cls.n1.b}; p[8] = w.final_cls.l3;
    p[9] = tc_linear{w.final_cls.n4.g, w.final_cls.n4.b}; p[10] = w.final_cls.l6;
    p[11] = w.final_reg.l0; p[12] = w.final_reg.l2; p[13] = w.final_reg.l4;
    k.rmin[r] = w.radius_min; k.rmax[r] = w.radius_max;
    k.g[G_KV0 + r] = const_cast<float*>(a.kv[r]);
  }
  k.g[G_QF] = const_cast<float*>(a.qf);
  k.g[G_CLS] = a.all_cls; k.g_ld[G_CLS] = a.ncls;
  k.qscale = a.qscale;
  k.tokens = a.tokens; k.RI = a.RI; k.T = a.T; k.pad_mult = a.pad_mult;
  k.ref_last = a.ref_last; k.box_in = a.box_m; k.cen_from_box = a.cen_from_box;
  TC_REQUIRE(a.cen_from_box || a.ref_last != nullptr, "radar_chain: ref_last is null");
  for (int i = 0; i < 6; ++i) k.cam.pc[i] = a.pc[i];
  k.all_box = a.all_box; k.hits = a.hits;
  TC_REQUIRE(a.tile_rows == 0 || a.tile_rows == 4 || a.tile_rows == 8 || a.tile_rows == 16 || a.tile_rows == 32,
             "radar_chain: tile_rows=%d (0 = automatic, 4, 8, 16 or 32)", a.tile_rows);
  k.tile_rows = a.tile_rows; k.last_cls_only = a.last_cls_only; k.matrix_path = a.matrix_path;
  k.range_status = a.range_status;
  if (a.tape != nullptr) {                    // forward of a training iteration: tape + dropout
    TC_REQUIRE(a.row_perm == nullptr && !a.last_cls_only, "radar_chain: the training forward takes the rows in their own order");
    k.program = PROG_RADAR_TRAIN;
    for (int i = 0; i < T_COUNT; ++i) k.tape[i] = a.tape[i];
    k.tape_stride = a.tape_stride; k.hits_stride = a.hits_stride; k.rdrop = a.drop;
    if (k.tile_rows >= 16 || (k.tile_rows == 0 && a.M > 2048)) k.tile_rows = 8;
  }
  return launch(k, s, "chain(radar)");
}

// Backward of the three fusion layers for the query rows: one launch (PROG_RADAR_BWD).
int launch_radar_chain_bwd(const RadarBwdChainArgs& a, hipStream_t s) {
  TC_REQUIRE(a.code <= 10 && a.ncls <= 32 && a.nlayers >= 1 && a.nlayers <= TC_MAX_RADAR_LAYERS,
             "radar_chain_bwd: code=%d ncls=%d nlayers=%d", a.code, a.ncls, a.nlayers);
  TC_REQUIRE(a.tape != nullptr && a.dy != nullptr && a.d_cls != nullptr && a.d_box != nullptr && a.hits != nullptr,
             "radar_chain_bwd: null argument");
  ChainK k;
  init_k(k);
  k.program = PROG_RADAR_BWD; k.M = a.M; k.Q = a.Q; k.code = a.code; k.ncls = a.ncls; k.nlayers = a.nlayers;
  k.has_next = 1;
  for (int r = 0; r < a.nlayers; ++r) {
    const tc_radar_layer& w = a.wT[r];             // transposed packed weights
    const tc_radar_layer& n = a.w[r];              // LayerNorm parameters (and radii) of the forward
    const tc_radar_layer& g = a.grads[r];
    tc_linear* p = &k.pairs[r * RADAR_PAIRS];
    tc_linear* gp = &k.gpairs[r * RADAR_PAIRS];
    p[0] = w.attn.in_proj; p[1] = w.attn.out_proj; p[2] = tc_linear{n.norm2.g, n.norm2.b};
    p[3] = w.linear1; p[4] = w.linear2; p[5] = tc_linear{n.norm3.g, n.norm3.b};
    p[6] = w.final_cls.l0; p[7] = tc_linear{n.final_cls.n1.g, n.final_cls.n1.b}; p[8] = w.final_cls.l3;
    p[9] = tc_linear{n.final_cls.n4.g, n.final_cls.n4.b}; p[10] = w.final_cls.l6;
    p[11] = w.final_reg.l0; p[12] = w.final_reg.l2; p[13] = w.final_reg.l4;
    gp[2] = tc_linear{g.norm2.g, g.norm2.b}; gp[5] = tc_linear{g.norm3.g, g.norm3.b};
    gp[7] = tc_linear{g.final_cls.n1.g, g.final_cls.n1.b}; gp[9] = tc_linear{g.final_cls.n4.g, g.final_cls.n4.b};
    k.rmin[r] = n.radius_min; k.rmax[r] = n.radius_max;
    k.g[G_KV0 + r] = const_cast<float*>(a.kv[r]);
    k.dkv[r] = a.dkv[r];
    k.bwd_cxy[r] = a.cxy[r]; k.bwd_ldc[r] = a.ld_c[r]; k.bwd_box[r] = a.box[r];
  }
  for (int i = 0; i < T_COUNT; ++i) k.tape[i] = a.tape[i];
  k.tape_stride = a.tape_stride; k.hits_stride = a.hits_stride;
  for (int i = 0; i < D_COUNT; ++i) k.dy[i] = a.dy[i];
  k.dy_stride = a.dy_stride;
  k.d_cls = a.d_cls; k.d_box = a.d_box; k.loss_vals = a.loss_vals; k.loss_out = a.loss_out;
  k.hits = const_cast<int*>(a.hits);
  k.qscale = a.qscale; k.tokens = a.tokens; k.RI = a.RI; k.T = a.T; k.pad_mult = a.pad_mult;
  k.rdrop = a.drop;
  k.detp = a.det_device;
  k.tile_rows = a.tile_rows == 8 || (a.tile_rows == 0 && a.M > 1024) ? 8 : 4;
  return launch(k, s, "chain(radar backward)");
}

// ---- tc_rowops_selfcheck (round 6): the instruction-count forms of the row-local arithmetic against the plain ones ----------
// Round 6 rewrote three pieces of the 16- / 32-row chains for fewer vector instructions and claims the same BITS:
//   [0] ln_rows<4> (four rows through one packed reduction tree, one 1 / sqrt on the packed variances) against four
//       ln_rows<1> (wave_sum per row);
//   [1] split_t2 (v_fma_mix / v_fma_mixlo forms) against hi = f16(t), lo = f16((t - hi) 2^11) written out;
//   [2] act_ld4<true> (one two-f16 v_fma_mix + the power-of-two un-scaling) against fma(lo, 2^-11 / s, hi / s).
// Every workgroup draws its own rows / values (random magnitudes over 40 binades, constant rows, signed zeros, f16
// overflow, subnormals, infinities, NaN) and adds the number of differing results to mismatches[0..2] (two NaNs agree).
namespace {
__device__ __forceinline__ unsigned long long sc_mix(unsigned long long h) {
  h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
  return h;
}
__device__ __forceinline__ float sc_value(unsigned long long h, int kind) {
  const float u = (float)(h >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;       // [-1, 1)
  const int e = (int)((h >> 8) & 63) - 44;                                       // 2^-44 .. 2^19
  switch (kind & 15) {
    case 0: return 0.0f;
    case 1: return -0.0f;
    case 2: return __builtin_bit_cast(float, (unsigned)(h & 0x007FFFFFu));              // fp32 subnormal
    case 3: return ldexpf(u, -20);                                                      // f16 subnormal range after the split
    case 4: return u * 70000.0f * 64.0f;                                                // around the planes' range
    case 5: return (h & 1) ? INFINITY : -INFINITY;
    case 6: return __builtin_bit_cast(float, 0x7FC00000u | (unsigned)(h & 0xFFFFu));    // NaN
    default: return ldexpf(u, e);
  }
}
__device__ __forceinline__ bool sc_differ(float a, float b) {
  if (a != a && b != b) return false;
  return __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b);
}
__global__ __launch_bounds__(256) void rowops_selfcheck_kernel(unsigned long long seed, unsigned long long* mismatches) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
  const unsigned long long base = sc_mix(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 4 + wave + 1));
  // [0] LayerNorm of four rows: per row a magnitude, an offset (mean >> deviation is the hard case) and a flavour
  {
    float4 v[4], w1[4];
    const unsigned long long hg = sc_mix(base + 77 + (unsigned long long)lane);
    const float4 gg = make_float4(sc_value(hg, 8), sc_value(sc_mix(hg + 1), 8), sc_value(sc_mix(hg + 2), 8), sc_value(sc_mix(hg + 3), 8));
    const float4 bb = make_float4(sc_value(sc_mix(hg + 4), 8), sc_value(sc_mix(hg + 5), 8), sc_value(sc_mix(hg + 6), 8), sc_value(sc_mix(hg + 7), 8));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long hr = sc_mix(base + 1000ull * (unsigned long long)(i + 1));
      const float mag = ldexpf(1.0f, (int)(hr & 31) - 16), off = (hr & 32) ? mag * (float)((hr >> 6) & 1023) : 0.0f;
      const int flavour = (int)((hr >> 20) & 7);       // 0: a constant row, 1: one outlier, else random
      float x[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned long long he = sc_mix(hr + 16ull * (unsigned long long)lane + (unsigned long long)c);
        const float u = (float)(he >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        x[c] = flavour == 0 ? off + mag : flavour == 1 ? off + ((lane == 17 && c == 2) ? mag * 1000.0f : mag * 1e-3f * u) : off + mag * u;
      }
      v[i] = make_float4(x[0], x[1], x[2], x[3]);
      w1[i] = v[i];
    }
    ln_rows<4>(v, gg, bb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 one[1] = {w1[i]};
      ln_rows<1>(one, gg, bb);
      bad0 += sc_differ(one[0].x, v[i].x) + sc_differ(one[0].y, v[i].y) + sc_differ(one[0].z, v[i].z) + sc_differ(one[0].w, v[i].w);
    }
  }
  // [1], [2]: values one by one
  __shared__ float lds[4][64][8];
  for (int it = 0; it < 64; ++it) {
    const unsigned long long h0 = sc_mix(base + 31ull * (unsigned long long)(it * 64 + lane) + 5);
    const unsigned long long h1 = sc_mix(h0 + 1);
    const float t0 = sc_value(h0, (int)(h0 >> 4)), t1 = sc_value(h1, (int)(h1 >> 4));
    unsigned hi, lo;
    split_t2(t0, t1, hi, lo);
    const unsigned rh = pk_h2(t0, t1);
    const f16x2 hv = __builtin_bit_cast(f16x2, rh);
    const f16x2 rl = {(_Float16)((t0 - (float)hv[0]) * H_LO_SCALE), (_Float16)((t1 - (float)hv[1]) * H_LO_SCALE)};
    const f16x2 gh = __builtin_bit_cast(f16x2, hi), gl = __builtin_bit_cast(f16x2, lo);
    bad1 += sc_differ((float)gh[0], (float)hv[0]) + sc_differ((float)gh[1], (float)hv[1]) +
            sc_differ((float)gl[0], (float)rl[0]) + sc_differ((float)gl[1], (float)rl[1]);
    // a plane pair as act_st4 lays it down (four values: 8 bytes of hi, 8 bytes of lo), read back both ways
    const unsigned long long h2 = sc_mix(h0 + 2), h3 = sc_mix(h0 + 3);
    const float4 x4 = make_float4(t0, t1, sc_value(h2, (int)(h2 >> 4)), sc_value(h3, (int)(h3 >> 4)));
    float* row = &lds[wave][lane][0];
    act_st4<true>(row, 0, x4);
    const float4 g4 = act_ld4<true>(row, 0);
    const char* pp = reinterpret_cast<const char*>(row);
    const uint2 ph = *reinterpret_cast<const uint2*>(pp), pl = *reinterpret_cast<const uint2*>(pp + 16);
    const f16x2 a0 = __builtin_bit_cast(f16x2, ph.x), a1 = __builtin_bit_cast(f16x2, ph.y);
    const f16x2 b0 = __builtin_bit_cast(f16x2, pl.x), b1 = __builtin_bit_cast(f16x2, pl.y);
    constexpr float UH = 1.0f / H_ACT_SCALE, UL = 1.0f / (H_ACT_SCALE * H_LO_SCALE);
    bad2 += sc_differ(g4.x, fmaf((float)b0[0], UL, (float)a0[0] * UH)) + sc_differ(g4.y, fmaf((float)b0[1], UL, (float)a0[1] * UH)) +
            sc_differ(g4.z, fmaf((float)b1[0], UL, (float)a1[0] * UH)) + sc_differ(g4.w, fmaf((float)b1[1], UL, (float)a1[1] * UH));
  }
  if (bad0) atomicAdd(mismatches + 0, bad0);
  if (bad1) atomicAdd(mismatches + 1, bad1);
  if (bad2) atomicAdd(mismatches + 2, bad2);
}
}  // namespace

int launch_rowops_selfcheck(int n_blocks, unsigned long long seed, unsigned long long* mismatches, hipStream_t s) {
  TC_REQUIRE(n_blocks > 0 && mismatches != nullptr, "rowops_selfcheck: bad arguments");
  hipLaunchKernelGGL(rowops_selfcheck_kernel, dim3(n_blocks), dim3(256), 0, s, seed, mismatches);
  return check_launch("rowops_selfcheck");
}

}  // namespace tc
