// NMS-free box decoding (CODER:39-90, UTIL:26-52, HEAD:1018): sigmoid, top-300
// of the 9000 (query, class) scores, gather + denormalise, centre-range mask,
// z -= h/2.  One workgroup per sample; a key is 64 bits (score bits << 32 | ~index), so keys are
// unique and an 8-bit radix SELECT finds the max_num-th largest exactly.  Ties resolve to the lower
// flat index (torch.topk leaves tie order unspecified).  <1 % of the frame, but 4 % of the one-frame
// latency path when it took 20 us (round 2: four barriers and a one-wave scan per pass, 300 atomics on
// one LDS word, a 45-stage bitonic sort of 512 keys behind 16 waves' barriers).  Round 3:
//   * a thread's <= 12 keys stay in registers; the eight passes' histograms are separate (zeroed once):
//     ONE barrier per pass, every wave does the 256-bin scan itself (no broadcast through LDS);
//   * the survivors are compacted with one atomic per wave-instruction and ordered by RANK
//     (rank = the number of larger survivors: two threads per survivor, 16-byte LDS reads) -- one barrier.
#include "kernels.hpp"

namespace tc {

constexpr int DEC_THREADS = 1024;
constexpr int DEC_MAXN = 12288;   // Q * num_classes upper bound
constexpr int DEC_NK = DEC_MAXN / DEC_THREADS;
constexpr int DEC_MAXK = 512;
constexpr int DEC_PASSES = 9;     // the bucket pass + up to eight byte passes (bit 63 = the score's sign = 0)

struct DecK {
  const float* cls; const float* box; int Q, ncls, code, K;
  float pcr[6];
  float* boxes; float* scores; int* labels; unsigned char* valid;      // fixed-size rows + mask (may be null)
  // round 4: the rows NMSFreeCoder keeps (inside post_center_range, above the score threshold), compacted in score
  // order, and their number -- what decode_single returns (CODER:66-84) without a mask select on the host side
  float* kboxes; float* kscores; long long* klabels; int* kcount;      // (may be null)
  float thr; int use_thr; int z_shift;
};

#ifdef TC_CHAIN_STAMPS
__device__ long long g_dec_stamps[16];
#define DEC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_dec_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEC_STAMP(i) do {} while (0)
#endif

struct DecLds {
  alignas(16) unsigned int hist[DEC_PASSES][256];
  alignas(16) unsigned long long sel[DEC_MAXK];      // survivors, any order; zero behind them
  unsigned int rank[DEC_MAXK];                       // survivor -> its place in descending order
  unsigned long long keep[DEC_MAXK / 64];            // bit i: output row i is kept
  unsigned int count, ccount;
};

__device__ __forceinline__ unsigned bucket(unsigned long long k) { return (unsigned)min(max((int)(k >> 51) - 1776, 0), 255); }

// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 4, 8; row_bcast 15 / 31), integers
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_iadd(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xF, false); }
__device__ __forceinline__ int wave_scan_incl(int v) {
  v = dpp_iadd<0x111, 0xF>(v); v = dpp_iadd<0x112, 0xF>(v); v = dpp_iadd<0x114, 0xF>(v); v = dpp_iadd<0x118, 0xF>(v);
  v = dpp_iadd<0x142, 0xA>(v); v = dpp_iadd<0x143, 0xC>(v);
  return v;
}

__global__ __launch_bounds__(DEC_THREADS) void box_decode_kernel(DecK p) {
  __shared__ DecLds S;
  extern __shared__ __align__(16) unsigned long long cand[];     // [Q * num_classes] the K-th key's bucket
  DEC_STAMP(0);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = p.Q * p.ncls;
  const float* cls = p.cls + (size_t)b * n;
  float logit[DEC_NK];
#pragma unroll
  for (int j = 0; j < DEC_NK; ++j) {
    const int i = tid + j * DEC_THREADS;
    logit[j] = i < n ? cls[i] : 0.0f;
  }
  // (under the loads)
  for (int i = tid; i < DEC_PASSES * 256; i += DEC_THREADS) (&S.hist[0][0])[i] = 0u;
  for (int i = tid; i < DEC_MAXK; i += DEC_THREADS) { S.sel[i] = 0ull; S.rank[i] = 0u; }
  if (tid == 0) { S.count = 0u; S.ccount = 0u; }
  if (tid < DEC_MAXK / 64) S.keep[tid] = 0ull;
  unsigned long long key[DEC_NK];
#pragma unroll
  for (int j = 0; j < DEC_NK; ++j) {
    const int i = tid + j * DEC_THREADS;
    // sigmoid in [0, 1]: the bit pattern is order preserving.  0 marks "no key" (a real key's low word is ~i != 0)
    key[j] = i < n ? ((unsigned long long)__float_as_uint(sigmoidf_(logit[j])) << 32) | (unsigned int)(0xFFFFFFFFu - (unsigned)i)
                   : 0ull;
  }
  __syncthreads();                                    // histograms zeroed
  DEC_STAMP(1);
#pragma unroll
  for (int j = 0; j < DEC_NK; ++j)
    if (key[j] != 0ull) atomicAdd(&S.hist[0][bucket(key[j])], 1u);
  __syncthreads();
  DEC_STAMP(2);
  const int K = min(p.K, n);
  // radix select: the K-th largest key.
  // Pass 0 does NOT look at a bit field: the exponent byte of a sigmoid output takes a handful of values, the 64
  // lanes of a wave serialise on one LDS word (17 K cycles even with one add per distinct digit of a
  // wave-instruction).  Any MONOTONE bucket function will do: bucket(key) = the score's exponent and top four
  // mantissa bits, v = key >> 51, as v - 1776 clamped to [0, 255] -- sixteen buckets per octave from 2^-16 up to 1;
  // bucket 0 (everything up to 2^-16) and bucket 255 (v >= 2031) are catch-alls.
  // A sweep over a thread's 12 keys costs ~2.5 K cycles however little it does (16 waves x 12 keys on 4 SIMDs), so
  // there are two: the one above (keys + bucket histogram) and ONE that files every key by its bucket -- above the
  // K-th key's bucket: selected; in it: a candidate (LDS list).  The byte passes then run over the candidates
  // only.  The keys of a bucket 1..254 share their top 12 bits: the byte passes continue below them; in a
  // catch-all they start from the top.
  unsigned long long prefix = 0, pmask = 0, floor_ = 0;
  int remaining = K, npass = 0, sh = 0;
  bool done;
  // every wave scans a histogram's 256 bins from the top down (lane l owns bins 255-4l .. 252-4l): the bin that
  // holds the remaining-th key, the keys above that bin, the bin's own count -- the same in every wave
  auto scan = [&](const unsigned int* hist, int& bin, int& cum, int& cnt) {
    const uint4 h4 = *reinterpret_cast<const uint4*>(&hist[252 - 4 * lane]);
    const int c0 = (int)h4.w, c1 = (int)h4.z, c2 = (int)h4.y, c3 = (int)h4.x;
    const int sum = c0 + c1 + c2 + c3;
    const int incl = wave_scan_incl(sum);
    const int excl = incl - sum;
    const bool mine = excl < remaining && remaining <= incl;      // exactly one lane
    cum = excl; bin = 255 - 4 * lane; cnt = c0;
    if (cum + c0 < remaining) { cum += c0; bin -= 1; cnt = c1;
      if (cum + c1 < remaining) { cum += c1; bin -= 1; cnt = c2;
        if (cum + c2 < remaining) { cum += c2; bin -= 1; cnt = c3; } } }
    const int owner = __builtin_ctzll(__ballot(mine));
    bin = __builtin_amdgcn_readlane(bin, owner); cum = __builtin_amdgcn_readlane(cum, owner);
    cnt = __builtin_amdgcn_readlane(cnt, owner);
  };
  const unsigned long long below = (1ull << lane) - 1ull;
  int fstar, cum, cnt;
  scan(S.hist[0], fstar, cum, cnt);
  remaining -= cum;
  // every key of the chosen bin is wanted: the lower digits cannot change the selection, the remaining
  // passes (normally those over the index half of the key, ties between scores being rare) are skipped
  done = remaining == cnt;
  floor_ = fstar == 0 ? 0ull : (unsigned long long)(fstar + 1776) << 51;         // the bucket's lower bound
  DEC_STAMP(3);
  {
    unsigned long long ms[DEC_NK], mc[DEC_NK];
    int ts = 0, tc_ = 0;
#pragma unroll
    for (int j = 0; j < DEC_NK; ++j) {
      const int bj = key[j] != 0ull ? (int)bucket(key[j]) : -1;
      ms[j] = __ballot(bj > fstar || (done && bj == fstar));
      mc[j] = __ballot(!done && bj == fstar);
      ts += __popcll(ms[j]); tc_ += __popcll(mc[j]);
    }
    unsigned bs = 0, bc = 0;
    if (lane == 0) {                                     // one atomic per list and wave
      if (ts) bs = atomicAdd(&S.count, (unsigned)ts);
      if (tc_) bc = atomicAdd(&S.ccount, (unsigned)tc_);
    }
    bs = (unsigned)__builtin_amdgcn_readfirstlane((int)bs); bc = (unsigned)__builtin_amdgcn_readfirstlane((int)bc);
#pragma unroll
    for (int j = 0; j < DEC_NK; ++j) {
      if ((ms[j] >> lane) & 1ull) { const unsigned pos = bs + (unsigned)__popcll(ms[j] & below); if (pos < DEC_MAXK) S.sel[pos] = key[j]; }
      if ((mc[j] >> lane) & 1ull) cand[bc + (unsigned)__popcll(mc[j] & below)] = key[j];
      bs += (unsigned)__popcll(ms[j]); bc += (unsigned)__popcll(mc[j]);
    }
  }
  DEC_STAMP(4);
  if (!done) {
    const bool normal = fstar >= 1 && fstar <= 254;
    prefix = normal ? floor_ : 0ull;
    pmask = normal ? 0xFFFull << 51 : 0ull;
    sh = normal ? 51 : 63;
    __syncthreads();
    const int ncand = (int)S.ccount;
#pragma unroll 1
    for (;;) {
      const int w = min(8, sh);                          // the last field is what is left: 3 or 7 bits
      sh -= w;
      const unsigned dmask = (1u << w) - 1u;
      unsigned int* hist = S.hist[++npass];
      for (int i = tid; i < ncand; i += DEC_THREADS) {
        const unsigned long long k = cand[i];
        if ((k & pmask) == prefix) atomicAdd(&hist[(unsigned)(k >> sh) & dmask], 1u);
      }
      pmask |= (unsigned long long)dmask << sh;
      __syncthreads();
      int bin;
      scan(hist, bin, cum, cnt);
      remaining -= cum;
      prefix |= (unsigned long long)(unsigned)bin << sh;
      DEC_STAMP(4 + npass);
      if (remaining == cnt || sh == 0) break;
    }
    // exactly `remaining` candidates are >= prefix (keys are unique; prefix has zeros below the digits examined;
    // in the upper catch-all an early exit can leave it below the bucket's floor -- the list holds the bucket only)
    for (int i0 = 0; i0 < ncand; i0 += DEC_THREADS) {
      const int i = i0 + tid;
      const unsigned long long k = i < ncand ? cand[i] : 0ull;
      const unsigned long long m = __ballot(k != 0ull && k >= prefix);
      if (m != 0ull) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&S.count, (unsigned)__popcll(m));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned pos = base + (unsigned)__popcll(m & below);
        if (((m >> lane) & 1ull) && pos < DEC_MAXK) S.sel[pos] = k;
      }
    }
  }
  DEC_STAMP(11);
  __syncthreads();
  DEC_STAMP(12);
  // rank, descending: a survivor's place = the number of survivors above it.  A thread compares FOUR survivors
  // (registers) with one slice of the list (16-byte LDS reads: with one survivor per thread the 1024 threads
  // pulled 1.2 MB through the LDS port -- 10 K cycles) and adds its counts to the survivors' rank words.
  {
    const int kpad = (K + 3) & ~3;
    const int groups = kpad >> 2, parts = DEC_THREADS / groups;           // K = 300: 75 groups x 13 slices
    const int g = tid % groups, part = tid / groups;
    int chunk = (kpad + parts - 1) / parts;
    chunk = (chunk + 1) & ~1;
    if (part < parts) {
      const ulonglong2 a01 = *reinterpret_cast<const ulonglong2*>(&S.sel[4 * g]);
      const ulonglong2 a23 = *reinterpret_cast<const ulonglong2*>(&S.sel[4 * g + 2]);
      int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
      const int j1 = min(kpad, (part + 1) * chunk);
      for (int j = part * chunk; j < j1; j += 2) {
        const ulonglong2 o = *reinterpret_cast<const ulonglong2*>(&S.sel[j]);
        r0 += (o.x > a01.x ? 1 : 0) + (o.y > a01.x ? 1 : 0);
        r1 += (o.x > a01.y ? 1 : 0) + (o.y > a01.y ? 1 : 0);
        r2 += (o.x > a23.x ? 1 : 0) + (o.y > a23.x ? 1 : 0);
        r3 += (o.x > a23.y ? 1 : 0) + (o.y > a23.y ? 1 : 0);
      }
      if (r0) atomicAdd(&S.rank[4 * g + 0], (unsigned)r0);
      if (r1) atomicAdd(&S.rank[4 * g + 1], (unsigned)r1);
      if (r2) atomicAdd(&S.rank[4 * g + 2], (unsigned)r2);
      if (r3) atomicAdd(&S.rank[4 * g + 3], (unsigned)r3);
    }
  }
  __syncthreads();
  DEC_STAMP(13);
  // thread t decodes survivor t into output row rank[t]; rows K .. max_num - 1 (fewer candidates than max_num): zeros
  // (max_num <= DEC_MAXK < DEC_THREADS: one survivor per thread at most)
  {
    const int t = tid;
    const bool row = t < p.K, have = t < K;
    const int i = have ? (int)S.rank[t] : t;
    float o[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float score = 0.f;
    int label = -1;
    bool ok = false;
    if (have) {
      const unsigned long long k = S.sel[t];
      const int idx = (int)(0xFFFFFFFFu - (unsigned int)(k & 0xFFFFFFFFull));
      score = __uint_as_float((unsigned int)(k >> 32));
      label = idx % p.ncls;
      const float* nb = p.box + ((size_t)b * p.Q + idx / p.ncls) * p.code;
      // UTIL:26-52
      const float rot = atan2f(nb[6], nb[7]);
      const float cx = nb[0], cy = nb[1], cz = nb[4];
      const float w = expf(nb[2]), l = expf(nb[3]), h = expf(nb[5]);
      ok = cx >= p.pcr[0] && cy >= p.pcr[1] && cz >= p.pcr[2] && cx <= p.pcr[3] && cy <= p.pcr[4] && cz <= p.pcr[5];
      o[0] = cx; o[1] = cy; o[2] = p.z_shift ? cz - h * 0.5f : cz;   // HEAD:1018
      o[3] = w; o[4] = l; o[5] = h; o[6] = rot;
      o[7] = p.code > 8 ? nb[8] : 0.f; o[8] = p.code > 9 ? nb[9] : 0.f;
    }
    if (row && p.boxes != nullptr) {
      float* ob = p.boxes + ((size_t)b * p.K + i) * 9;
#pragma unroll
      for (int j = 0; j < 9; ++j) ob[j] = o[j];
      p.scores[(size_t)b * p.K + i] = score; p.labels[(size_t)b * p.K + i] = label;
      p.valid[(size_t)b * p.K + i] = ok ? 1 : 0;
    }
    if (p.kboxes != nullptr) {                      // (uniform)
      // CODER:62-76: the range mask is taken on the coder's own (un-shifted) centre, the threshold is strict
      const bool kept = have && ok && (!p.use_thr || score > p.thr);
      if (kept) atomicOr(&S.keep[i >> 6], 1ull << (i & 63));
      __syncthreads();
      int below_words = 0, total = 0;
#pragma unroll
      for (int wd = 0; wd < DEC_MAXK / 64; ++wd) {
        const int c = __popcll(S.keep[wd]);
        total += c;
        if (wd < (i >> 6)) below_words += c;
      }
      if (kept) {
        const int pos = below_words + __popcll(S.keep[i >> 6] & ((1ull << (i & 63)) - 1ull));
        float* ob = p.kboxes + ((size_t)b * p.K + pos) * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) ob[j] = o[j];
        p.kscores[(size_t)b * p.K + pos] = score; p.klabels[(size_t)b * p.K + pos] = (long long)label;
      }
      if (tid == 0) p.kcount[b] = total;
    }
  }
  DEC_STAMP(14);
}

#ifdef TC_CHAIN_STAMPS
extern "C" int tc_debug_decode_stamps(long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dec_stamps), sizeof(long long) * 16);
}
#endif

size_t box_decode_ws_bytes(int, int, int) { return 256; }

int launch_box_decode(const float* cls, const float* box, int B, int Q, int ncls, int code,
                      int max_num, const float* pcr6_host, float* boxes, float* scores, int* labels,
                      unsigned char* valid, void*, size_t, hipStream_t s, const BoxDecodeKept* kept) {
  TC_REQUIRE(Q * ncls <= DEC_MAXN, "box_decode: Q*num_classes=%d > %d", Q * ncls, DEC_MAXN);
  TC_REQUIRE(max_num >= 1 && max_num <= DEC_MAXK, "box_decode: max_num=%d (1..%d)", max_num, DEC_MAXK);
  TC_REQUIRE(code >= 8, "box_decode: code_size=%d", code);
  DecK p;
  p.cls = cls; p.box = box; p.Q = Q; p.ncls = ncls; p.code = code; p.K = max_num;
  for (int i = 0; i < 6; ++i) p.pcr[i] = pcr6_host[i];
  p.boxes = boxes; p.scores = scores; p.labels = labels; p.valid = valid;
  TC_REQUIRE(boxes == nullptr || (scores != nullptr && labels != nullptr && valid != nullptr), "box_decode: fixed-size outputs come together");
  p.kboxes = nullptr; p.kscores = nullptr; p.klabels = nullptr; p.kcount = nullptr; p.thr = 0.f; p.use_thr = 0; p.z_shift = 1;
  if (kept != nullptr) {
    TC_REQUIRE(kept->boxes && kept->scores && kept->labels && kept->count, "box_decode: kept outputs come together");
    p.kboxes = kept->boxes; p.kscores = kept->scores; p.klabels = kept->labels; p.kcount = kept->count;
    p.thr = kept->score_threshold; p.use_thr = kept->use_threshold; p.z_shift = kept->z_shift;
  }
  TC_REQUIRE(boxes != nullptr || kept != nullptr, "box_decode: no output");
  const size_t lds = (size_t)((Q * ncls + 1) & ~1) * 8;
  static DeviceOnce once;
  if (const int once_dev = once.need(); once_dev >= 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(box_decode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DEC_MAXN * 8);
    if (e != hipSuccess) { set_error("box_decode: %s", hipGetErrorString(e)); return (int)e; }
    once.done(once_dev);
  }
  hipLaunchKernelGGL(box_decode_kernel, dim3(B), dim3(DEC_THREADS), lds, s, p);
  return check_launch("box_decode");
}

}  // namespace tc
