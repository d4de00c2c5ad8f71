// A2 + A3 + A4  RPN proposal generation for gfx950: anchor enumeration, pre-NMS top-k, box decode, clip, min-size filter
// for ALL images and ALL FPN levels of a batch, device resident (no D2H), in 5 launches (round 1: 6;
// the third histogram pass over the score maps is gone).
//
// Replaces GenerateProposals.forward up to the NMS call (lib/model/generate_proposals.py:31-109), which per level and
// per image does: numpy anchor meshgrid + H2D (:49-56,124-149), permute+contiguous of scores/deltas + 2 D2H (:64-73),
// host argpartition/argsort (:77-86), ~25 tiny torch kernels for decode/clip (:96-100,165-238), D2H + host filter
// (:101-109).
//
//   rpn_hist<0,1>     2-pass radix histogram (12+12 bits of the order-preserving score key) locating the 24-bit bin that holds
//                     the K-th largest score of every (image, level) segment; LDS histograms, one global atomic per non-empty
//                     bin.  Scores are read in their native conv layout [A,H,W] (coalesced); the (A,H,W)->(H,W,A) permute
//                     of :64,72 is only an index formula.
//   rpn_compact       elements above that bin -> 64-bit (score desc, canonical index asc) keys that are certainly selected;
//                     elements inside it (a few dozen at most, unless the map is constant) -> candidate keys.
//   rpn_sort          one workgroup per segment: bitonic sort of selected + candidate keys in LDS (the first K are the
//                     answer, ties broken by the canonical index);
//   rpn_decode        256 ranks per workgroup: anchor = f(index) from the A base anchors
//                     (never materialised, :124-149), decode (:165-214), clip (:216-238), filter (:151-163),
//                     order-preserving compaction.  More candidates than the sort holds (constant / saturated maps): an
//                     in-workgroup radix select over the candidate keys first.
// Output per segment: boxes [K,4] + scores [K] in descending score order + count  == the `dets` handed to NMS at :115.
#include "block_sort.h"
#include "dtc_common.h"
#include "radix_select.h"

namespace dtc {
DTC_PT_TABLE(proposals)

constexpr int kRpnMaxLevels = 8;
constexpr int kRpnMaxAnchors = 16;
constexpr int kHistBins = 4096;   // 12 bits per pass
constexpr int kHistThreads = 256;
constexpr int kChunk = 4096;  // elements per workgroup in the streaming passes

struct RpnLevelDev {
  const float* cls;     // [B, A, H, W]
  const float* bbox;    // [B, 4A, H, W]
  int A, H, W, N;       // N = A*H*W
  int K;                // effective pre-NMS top-n = min(pre_nms_top_n, N) (or N if pre_nms_top_n <= 0)
  int chunk_begin;      // first chunk id of this level inside one image
  int tie_begin;        // offset of this level's tie list inside one image's tie buffer
  float feat_stride;
  int logit;            // scores are pre-sigmoid logits (detector.py:125 folded into this path): see TieBand
  float anchors[kRpnMaxAnchors * 4];  // base anchors, float32 (exact: integers / half-integers)
};

struct RpnParams {
  RpnLevelDev lv[kRpnMaxLevels];
  int n_levels, batch, chunks_per_image, ties_per_image, k_stride;
  float im_h, im_w, min_size;
  uint32_t* hist;       // [S][2][kHistBins]
  uint32_t* counters;   // [S][2] : selected count, candidate count
  uint64_t* gt_keys;    // [S][k_stride]
  uint64_t* cand_keys;  // [B][ties_per_image]
  uint64_t* sorted_keys;  // [S][k_stride]   keys of ranks [0, n_rank) in (score desc, index asc) order: rpn_sort -> rpn_decode
  int32_t* n_rank;        // [S]
  uint32_t* ticket;       // [S]             dynamic block index of rpn_decode (cleared with the histograms)
  uint32_t* blk_cnt;      // [S][decode blocks]  (count << 1) | ready, cleared with the histograms
  int dec_blocks;         // decode workgroups per segment
  int resident;           // rpn_decode: the whole grid is co-resident -> launch-order block numbers instead of tickets
  float* out_boxes;     // [S][k_stride][4]
  float* out_scores;    // [S][k_stride]
  int32_t* out_counts;  // [S]
};

__device__ __forceinline__ int find_level(const RpnParams& p, int chunk) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kRpnMaxLevels; i++)
    if (i < p.n_levels && chunk >= p.lv[i].chunk_begin) l = i;
  return l;
}

// threshold state after `passes` completed passes: prefix (ordered-key bits found so far) and remaining rank
struct SelState { uint32_t p0, p1, krem; };

template <int PASSES>
__device__ __forceinline__ SelState load_state(const RpnParams& p, int seg, uint32_t K, uint32_t* sh) {
  SelState st; st.p0 = st.p1 = 0; st.krem = K;
  const uint32_t* H = p.hist + (size_t)seg * 2 * kHistBins;
  DigitBins b0, b1;                            // both histograms are requested before the first selection consumes its bins
  if (PASSES >= 1) load_digit_bins(H, kHistBins, b0);
  if (PASSES >= 2) load_digit_bins(H + kHistBins, kHistBins, b1);
  if (PASSES >= 1) { select_digit_loaded(b0, kHistBins, st.krem, sh); st.p0 = sh[0]; st.krem = sh[1]; __syncthreads(); }
  if (PASSES >= 2) { select_digit_loaded(b1, kHistBins, st.krem, sh); st.p1 = sh[0]; st.krem = sh[1]; __syncthreads(); }
  return st;
}

// ---- RPN-head epilogue fusion: scores handed over as LOGITS ----------------------------------------------
// The reference ranks sigmoid(logit) (detector.py:125 -> generate_proposals.py:77-86).  sigmoid is monotone, so the radix
// histograms run on the raw logits untouched; but float32 sigmoid is many-to-one (every logit > 16.7 gives 1.0f), and equal
// PROBABILITIES must tie-break by index exactly as if the probabilities had been materialised.  The K-th largest logit lies
// in the key bin [bl, bh] the histograms found; every logit whose probability equals that of some logit of the bin could tie
// with the K-th element, so the candidate band is widened to [lo, hi] with sigmoid(lo) == sigmoid(bl) and
// sigmoid(hi) == sigmoid(bh) (two bisections on the ordered keys; sigmoid evaluated in double and rounded once: the same value
// oracle / numpy produce).  Logits above hi have a strictly larger probability than the K-th element: certainly selected.
// Only selected and candidate elements ever get a sigmoid -- the [B,A,H,W] probability map is never written or read.
__device__ __forceinline__ float sigmoid_cr(float x) {
  return (float)(1.0 / (1.0 + exp(-(double)x)));
}
struct KeyBand { uint32_t lo, hi; };   // ordered-key band of the candidates

// lane 0 bisects downwards from bl, lane 1 upwards from bh; result broadcast through sh2[0..1].  Call from all threads.
__device__ __forceinline__ KeyBand candidate_band(uint32_t bl, uint32_t bh, bool logit, uint32_t* sh2) {
  KeyBand kb; kb.lo = bl; kb.hi = bh;
  if (!logit) return kb;
  const uint32_t kinf_p = float_to_ordered(__uint_as_float(0x7f800000u)), kinf_n = float_to_ordered(__uint_as_float(0xff800000u));
  __syncthreads();
  if (threadIdx.x < 2) {
    const bool up = threadIdx.x == 1;
    // start inside the finite range (keys beyond +-inf are NaN patterns: never produced by a conv)
    uint32_t a = up ? min(max(bh, kinf_n), kinf_p) : max(min(bl, kinf_p), kinf_n);
    const uint32_t pk = float_to_ordered(sigmoid_cr(ordered_to_float(a)));
    // invariant: f(a) == pk, f(b) != pk (b is outside the band or the end of the finite range)
    uint32_t b = up ? kinf_p : kinf_n;
    if (float_to_ordered(sigmoid_cr(ordered_to_float(b))) == pk) a = b;      // band reaches +-inf
    else {
      while ((up ? b - a : a - b) > 1u) {
        const uint32_t m = up ? a + ((b - a) >> 1) : a - ((a - b) >> 1);
        if (float_to_ordered(sigmoid_cr(ordered_to_float(m))) == pk) a = m; else b = m;
      }
    }
    sh2[up ? 1 : 0] = up ? max(a, bh) : min(a, bl);
  }
  __syncthreads();
  kb.lo = sh2[0]; kb.hi = sh2[1];
  __syncthreads();
  return kb;
}

template <int PASS>
__global__ __launch_bounds__(kHistThreads) void rpn_hist_kernel(RpnParams p) {
  __shared__ uint32_t h[kHistBins];
  __shared__ uint32_t sh[2];
  const int b = blockIdx.y;
  const int l = find_level(p, blockIdx.x);
  const RpnLevelDev& L = p.lv[l];
  if (L.K >= L.N) return;  // take everything: no selection needed
  const int seg = b * p.n_levels + l;
  const int chunk = blockIdx.x - L.chunk_begin;
  [[maybe_unused]] const int ptb = blockIdx.y * gridDim.x + blockIdx.x;
  DTC_PT(PASS, ptb, 0);
  // the chunk's 16 scores per thread are requested FIRST (independent loads, back to back) and consumed after the histogram
  // is cleared and the previous pass' digit is known: the load latency hides behind that work.  (A load -> LDS atomic loop
  // paid one L2 round trip per element: 7.4 of the kernel's 9.2 us.)
  const float* sc = L.cls + (size_t)b * L.N;
  const int begin = chunk * kChunk, end = min(begin + kChunk, L.N);
  float v[kChunk / kHistThreads];
#pragma unroll
  for (int u = 0; u < kChunk / kHistThreads; u++) {
    const int i = begin + u * kHistThreads + (int)threadIdx.x;
    v[u] = i < end ? sc[i] : 0.f;
  }
  for (int i = threadIdx.x; i < kHistBins; i += kHistThreads) h[i] = 0;
  const SelState st = load_state<PASS>(p, seg, (uint32_t)L.K, sh);  // ends with a barrier (or needs one for PASS 0)
  if (PASS == 0) __syncthreads();
  DTC_PT(PASS, ptb, 1);
#pragma unroll
  for (int u = 0; u < kChunk / kHistThreads; u++) {
    const int i = begin + u * kHistThreads + (int)threadIdx.x;
    if (i < end) {
      const uint32_t o = float_to_ordered(v[u]);
      if (PASS == 0) atomicAdd(&h[o >> 20], 1u);
      if (PASS == 1) { if ((o >> 20) == st.p0) atomicAdd(&h[(o >> 8) & 4095u], 1u); }
    }
  }
  __syncthreads();
  DTC_PT(PASS, ptb, 2);
  uint32_t* G = p.hist + ((size_t)seg * 2 + PASS) * kHistBins;
  for (int i = threadIdx.x; i < kHistBins; i += kHistThreads) {
    const uint32_t v = h[i];
    if (v) atomicAdd(&G[i], v);
  }
  DTC_PT(PASS, ptb, 3);
}

__global__ __launch_bounds__(kHistThreads) void rpn_compact_kernel(RpnParams p) {
  // Selected elements and candidates are staged in LDS (LDS atomics hand out the slots: selected from the front, candidates
  // from the back of one array -- an element is at most one of the two) and the workgroup claims its output range with ONE
  // global atomic per list: per-wave global atomics on a single counter per segment serialise at the L2 (measured 64 us
  // for 8 images before this change).
  __shared__ uint32_t sh[2];
  __shared__ uint32_t lcnt[2], gbase[2];
  __shared__ uint64_t stage[kChunk];
  const int b = blockIdx.y;
  const int l = find_level(p, blockIdx.x);
  const RpnLevelDev& L = p.lv[l];
  const int seg = b * p.n_levels + l;
  const int chunk = blockIdx.x - L.chunk_begin;
  const bool take_all = L.K >= L.N;
  [[maybe_unused]] const int ptb = blockIdx.y * gridDim.x + blockIdx.x;
  DTC_PT(2, ptb, 0);
  const float* sc = L.cls + (size_t)b * L.N;      // the chunk's scores: requested first, consumed behind the threshold look-up
  const int begin = chunk * kChunk, end = min(begin + kChunk, L.N);
  float v[kChunk / kHistThreads];
#pragma unroll
  for (int u = 0; u < kChunk / kHistThreads; u++) {
    const int i = begin + u * kHistThreads + (int)threadIdx.x;
    v[u] = i < end ? sc[i] : 0.f;
  }
  if (threadIdx.x < 2) lcnt[threadIdx.x] = 0;
  KeyBand kb; kb.lo = kb.hi = 0;
  if (!take_all) {
    const SelState st = load_state<2>(p, seg, (uint32_t)L.K, sh);
    const uint32_t bl = ((st.p0 << 12) | st.p1) << 8;
    kb = candidate_band(bl, bl | 0xffu, L.logit != 0, sh);
  }
  __syncthreads();
  DTC_PT(2, ptb, 1);
  const int HW = L.H * L.W;
#pragma unroll
  for (int u = 0; u < kChunk / kHistThreads; u++) {
    const int i = begin + u * kHistThreads + (int)threadIdx.x;
    if (i >= end) continue;
    const float s = v[u];
    const uint32_t o = float_to_ordered(s);
    const bool is_gt = take_all || o > kb.hi;
    const bool is_cand = !take_all && o >= kb.lo && o <= kb.hi;
    if (is_gt || is_cand) {
      // memory index i = (a*H + h)*W + w  ->  canonical index n = (h*W + w)*A + a   (generate_proposals.py:64,72)
      const int a = i / HW, hw = i - a * HW;
      const uint32_t n = (uint32_t)(hw * L.A + a);
      const uint64_t key = make_desc_key(L.logit ? sigmoid_cr(s) : s, n);
      if (is_gt) stage[atomicAdd(&lcnt[0], 1u)] = key;
      else stage[kChunk - 1 - atomicAdd(&lcnt[1], 1u)] = key;
    }
  }
  __syncthreads();
  DTC_PT(2, ptb, 2);
  uint32_t* cnt = p.counters + (size_t)seg * 2;
  if (threadIdx.x < 2) gbase[threadIdx.x] = lcnt[threadIdx.x] ? atomicAdd(&cnt[threadIdx.x], lcnt[threadIdx.x]) : 0u;
  __syncthreads();
  uint64_t* gt = p.gt_keys + (size_t)seg * p.k_stride;
  uint64_t* cand = p.cand_keys + (size_t)b * p.ties_per_image + L.tie_begin;
  for (uint32_t j = threadIdx.x; j < lcnt[0]; j += kHistThreads)
    if (gbase[0] + j < (uint32_t)p.k_stride) gt[gbase[0] + j] = stage[j];
  for (uint32_t j = threadIdx.x; j < lcnt[1]; j += kHistThreads) cand[gbase[1] + j] = stage[kChunk - 1 - j];
  DTC_PT(2, ptb, 3);
}

// generate_proposals.py:165-214 (weights (1,1,1,1)) + :216-238 + :151-163
__device__ __forceinline__ float clip1(float v, float hi) { v = fminf(v, hi); return fmaxf(v, 0.f); }

__device__ __forceinline__ void decode_box(float ax1, float ay1, float ax2, float ay2, float dx, float dy, float dw,
                                           float dh, float out[4]) {
  const float widths = ax2 - ax1 + 1.0f, heights = ay2 - ay1 + 1.0f;          // :175-176
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;      // :177-178
  const float clipv = 4.135166556742356f;                                     // :165 log(1000/16) as float32
  dw = fminf(dw, clipv); dh = fminf(dh, clipv);                               // :191-192
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;          // :194-195
  const float pw = fexp_cr(dw) * widths, ph = fexp_cr(dh) * heights;          // :196-197
  out[0] = pcx - 0.5f * pw; out[1] = pcy - 0.5f * ph;                         // :201-203
  out[2] = pcx + 0.5f * pw - 1.f; out[3] = pcy + 0.5f * ph - 1.f;             // :205-207
}

constexpr int kSortDecodeThreads = 1024;
constexpr int kRadixKeysPerThread = 8;
// ONE place that says when rpn_sort takes the LSD radix sort and how much dynamic LDS it needs (host launch + kernel branch):
// more than 2048 keys to order and a key capacity the two key buffers + 16 x 256 per-wave digit counters fit LDS with
// (8192 keys: 2 x 64 KB + 16 KB = 144 KB dynamic + ~8.3 KB static of the 160 KB).
__host__ __device__ constexpr bool rpn_sort_radix_capable(int sort_cap) {
  return sort_cap > 2048 && sort_cap <= kSortDecodeThreads * kRadixKeysPerThread;
}
__host__ __device__ constexpr size_t rpn_sort_lds_bytes(int sort_cap) {
  return rpn_sort_radix_capable(sort_cap)
             ? (size_t)sort_cap * sizeof(uint64_t) * 2 + (size_t)(kSortDecodeThreads / 64) * 256 * sizeof(uint32_t)
             : (size_t)sort_cap * sizeof(uint64_t);
}

// Ascending digit select inside the workgroup: smallest digit d with  count(digits <= d) >= rem.  hist[nbins] in LDS,
// nbins <= 2048.  Returns d in sh[0] and the rank left inside digit d in sh[1].  (select_digit picks from the top: feed
// it the mirrored histogram.)
__device__ __forceinline__ void select_digit_asc(uint32_t* hist, int nbins, uint32_t rem, uint32_t* sh) {
  __syncthreads();
  for (int i = threadIdx.x; i < nbins / 2; i += blockDim.x) { const uint32_t t = hist[i]; hist[i] = hist[nbins - 1 - i]; hist[nbins - 1 - i] = t; }
  __syncthreads();
  select_digit(hist, nbins, rem, sh);
  if (threadIdx.x == 0) sh[0] = (uint32_t)(nbins - 1) - sh[0];
  __syncthreads();
}

__global__ __launch_bounds__(kSortDecodeThreads) void rpn_sort_kernel(RpnParams p, int sort_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ uint32_t sh[2];
  __shared__ __attribute__((aligned(16))) uint32_t hsel[2048];
  __shared__ int running;
  const int seg = blockIdx.x;
  const int b = seg / p.n_levels, l = seg - b * p.n_levels;
  const RpnLevelDev& L = p.lv[l];
  const int tid = threadIdx.x;
  DTC_PT(3, seg, 0);
  const uint32_t n_gt = min(p.counters[(size_t)seg * 2], (uint32_t)p.k_stride);
  const uint32_t n_cand = p.counters[(size_t)seg * 2 + 1];
  const uint64_t* gt = p.gt_keys + (size_t)seg * p.k_stride;
  const uint64_t* cand = p.cand_keys + (size_t)b * p.ties_per_image + L.tie_begin;
  const int K = L.K;
  const uint32_t need = (uint32_t)K - n_gt;  // candidates to take (0 when everything is taken)
  // More candidates than the sort holds (a constant map, a saturated sigmoid): the `need` smallest candidate keys are found
  // with a radix select over (a) the 32 score bits, (b) the canonical index among the candidates that share the threshold
  // score.  Not a fast path: it re-reads the candidate list once per digit.
  bool filtered = false;
  uint32_t u_star = 0, idx_limit = 0xffffffffu;
  if (n_gt + n_cand > (uint32_t)sort_cap) {
    filtered = true;
    uint32_t rem = need;
    uint64_t prefix = 0;       // bits of the key above the digit being resolved
    for (int pass = 0; pass < 6; pass++) {
      // digits: key bits [63:53] [52:42] [41:32] | [26:18] [17:9] [8:0]   (bits [31:27] of an index are zero: N < 2^27)
      const int shift = pass == 0 ? 53 : pass == 1 ? 42 : pass == 2 ? 32 : pass == 3 ? 18 : pass == 4 ? 9 : 0;
      const int bits = pass < 2 ? 11 : pass == 2 ? 10 : 9;
      const int hi_shift = pass == 3 ? 32 : shift + bits;        // the first index digit sits right below the score word
      const int nb = 1 << bits;
      for (int i = tid; i < nb; i += kSortDecodeThreads) hsel[i] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n_cand; i += kSortDecodeThreads) {
        const uint64_t v = cand[i];
        if (pass == 0 || (v >> hi_shift) == prefix) atomicAdd(&hsel[(uint32_t)(v >> shift) & (uint32_t)(nb - 1)], 1u);
      }
      select_digit_asc(hsel, nb, rem, sh);
      const uint32_t d = sh[0];
      rem = sh[1];
      prefix = pass == 3 ? (prefix << 14) | d : (prefix << bits) | d;   // pass 3: 5 zero bits + 9 digit bits below the score word
      __syncthreads();
    }
    // prefix is now the complete key of the last candidate taken
    u_star = (uint32_t)(prefix >> 32);
    idx_limit = (uint32_t)prefix;
  }
  // gather into LDS: selected keys, then the candidates (all of them, or the filtered `need`)
  const uint32_t n_take = filtered ? need : n_cand;
  const int total = (int)(n_gt + n_take);
  const int np2 = next_pow2(total);
  for (int i = tid; i < np2; i += kSortDecodeThreads) keys[i] = i < (int)n_gt ? gt[i] : kPadKey;
  if (tid == 0) running = (int)n_gt;
  __syncthreads();
  if (!filtered) {
    for (uint32_t i = tid; i < n_cand; i += kSortDecodeThreads) keys[n_gt + i] = cand[i];
  } else {
    const uint64_t last = ((uint64_t)u_star << 32) | idx_limit;
    for (uint32_t i = tid; i < n_cand; i += kSortDecodeThreads) {
      const uint64_t v = cand[i];
      if (v <= last) { const int slot = atomicAdd(&running, 1); if (slot < np2) keys[slot] = v; }
    }
  }
  __syncthreads();
  DTC_PT(3, seg, 1);
  const uint64_t* sorted = keys;
  if (np2 > 2048 && rpn_sort_radix_capable(sort_cap)) {
    // long segments (C4: 6000 ranks; up to 8192 keys -- beyond that the two key buffers do not fit LDS: bitonic): LSD radix sort over the score word and the index bits in use, 6 passes instead of the 91
    // compare-exchange steps of an 8192-key bitonic network (85 -> ~20 us)
    uint64_t* keys2 = keys + sort_cap;
    uint32_t* rcnt = reinterpret_cast<uint32_t*>(keys2 + sort_cap);
    const int idx_bits = 32 - __builtin_clz((unsigned)max(L.N - 1, 1));
    uint32_t dmask = 0xf0u;                                  // the four score digits
    for (int d = 0; d < 4; d++) if (8 * d < idx_bits) dmask |= 1u << d;
    sorted = block_radix_sort_u64<kSortDecodeThreads, kRadixKeysPerThread>(keys, keys2, rcnt, hsel, total, dmask);
  } else {
    block_bitonic_sort<kSortDecodeThreads>(keys, np2);
  }
  DTC_PT(3, seg, 2);

  // the first K keys are the answer: hand them to rpn_decode (a segment's 6000 ranks are decoded by 24 workgroups, not by this one)
  const int n_rank = min(K, total);
  uint64_t* sk = p.sorted_keys + (size_t)seg * p.k_stride;
  for (int k = tid; k < n_rank; k += kSortDecodeThreads) sk[k] = sorted[k];
  if (tid == 0) p.n_rank[seg] = n_rank;
  DTC_PT(3, seg, 3);
}

// ranks [0, K) in score order: decode, clip, filter, ORDERED compaction -- spread over the chip.  Round 4 measured the decode of a
// C4 segment (6000 ranks, two correctly rounded exp each) at 70 us inside the one-workgroup sort kernel: 16 waves x ~1500
// instructions on ONE CU.  Here a workgroup of 256 threads decodes 256 consecutive ranks; the boxes a filter removes (min-size /
// centre-outside, generate_proposals.py:151-163) shift everything behind them, so the output slot of a rank needs the number of
// surviving ranks before it: every workgroup publishes its survivor count and sums the counts of the blocks before it (at most 63:
// one poll per lane).  Blocks are numbered by a TICKET taken at run time, so the blocks a workgroup waits for are always already
// running, whatever order the hardware dispatches workgroups in.
constexpr int kDecodeThreads = 256;
__global__ __launch_bounds__(kDecodeThreads) void rpn_decode_kernel(RpnParams p) {
  __shared__ int sh_blk, sh_base;
  __shared__ int wave_tot[kDecodeThreads / 64];
  const int seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = seg / p.n_levels, l = seg - b * p.n_levels;
  const RpnLevelDev& L = p.lv[l];
  // Block number: a block waits for the blocks BEFORE it, so those must be running.  When the whole grid is co-resident (p.resident:
  // the launcher checks grid <= 1024 workgroups of 256 threads on 256 CUs) the launch index serves and a dependent global round trip
  // goes away; otherwise a run-time TICKET, so that a block only ever waits for blocks that were dispatched before it, whatever order
  // the hardware dispatches workgroups in.
  int blk = blockIdx.x;
  if (!p.resident) {
    if (tid == 0) sh_blk = (int)atomicAdd(&p.ticket[seg], 1u);
    __syncthreads();
    blk = sh_blk;
    // tickets past the block count: the workspace is shared with another in-flight call (include/detectorch_hip.h forbids it) or was
    // not cleared for this launch -- leave instead of writing block counts out of bounds (uniform: after the barrier)
    if (blk >= p.dec_blocks) return;
  }
  const uint64_t* sk = p.sorted_keys + (size_t)seg * p.k_stride;
  // the rank's key is requested together with the segment's rank count (k < k_stride: a valid address whatever the count is)
  const int k_spec = blk * kDecodeThreads + tid;
  const uint64_t key_spec = k_spec < p.k_stride ? sk[k_spec] : 0ull;
  const int n_rank = p.n_rank[seg];
  const float* sc = L.cls + (size_t)b * L.N;
  const float* dl = L.bbox + (size_t)b * L.N * 4;
  const int HW = L.H * L.W;
  float* ob = p.out_boxes + (size_t)seg * p.k_stride * 4;
  float* os = p.out_scores + (size_t)seg * p.k_stride;
  const int k = blk * kDecodeThreads + tid;
  bool ok = false;
  float box[4] = {0.f, 0.f, 0.f, 0.f};
  float s = 0.f;
  if (k < n_rank) {
    const uint64_t key = key_spec;
    const uint32_t n = desc_key_index(key);
    const int a = n % L.A, hw = n / L.A;
    const int h = hw / L.W, w = hw - h * L.W;
    s = L.logit ? desc_key_score(key) : sc[(size_t)a * HW + hw];   // the key carries the probability
    // :124-149 shifted anchor: float64 add of exactly representable values, rounded to float32 (:54) -> exact
    const float sx = (float)w * L.feat_stride, sy = (float)h * L.feat_stride;
    const float ax1 = L.anchors[a * 4 + 0] + sx, ay1 = L.anchors[a * 4 + 1] + sy;
    const float ax2 = L.anchors[a * 4 + 2] + sx, ay2 = L.anchors[a * 4 + 3] + sy;
    const float* d = dl + (size_t)(a * 4) * HW + hw;
    decode_box(ax1, ay1, ax2, ay2, d[0], d[HW], d[2 * HW], d[3 * HW], box);
    box[0] = clip1(box[0], p.im_w - 1.f); box[1] = clip1(box[1], p.im_h - 1.f);   // :230-236
    box[2] = clip1(box[2], p.im_w - 1.f); box[3] = clip1(box[3], p.im_h - 1.f);
    const float ws = box[2] - box[0] + 1.f, hs = box[3] - box[1] + 1.f;           // :155-156
    const float xc = box[0] + fdiv(ws, 2.f), yc = box[1] + fdiv(hs, 2.f);         // :157-158
    ok = (ws >= p.min_size) && (hs >= p.min_size) && (xc < p.im_w) && (yc < p.im_h);  // :159-162
  }
  const uint64_t m = __ballot(ok);
  if (lane == 0) wave_tot[wv] = __builtin_popcountll(m);
  __syncthreads();
  uint32_t* bc = p.blk_cnt + (size_t)seg * p.dec_blocks;
  if (wv == 0) {
    int mine = 0;
#pragma unroll
    for (int q = 0; q < kDecodeThreads / 64; q++) mine += wave_tot[q];
    if (lane == 0) __hip_atomic_store(&bc[blk], ((uint32_t)mine << 1) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // survivors of the blocks before this one: lane t polls block t (their tickets are lower: those workgroups are running)
    int before = 0;
    for (int t0 = 0; t0 < blk; t0 += 64) {
      const int t = t0 + lane;
      uint32_t v = 1u;
      if (t < blk) {
        do { v = __hip_atomic_load(&bc[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); if (!(v & 1u)) __builtin_amdgcn_s_sleep(1); } while (!(v & 1u));
      }
      int c = t < blk ? (int)(v >> 1) : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
      before += c;
    }
    if (lane == 0) {
      sh_base = before;
      if (blk == p.dec_blocks - 1) p.out_counts[seg] = before + mine;       // the last block closes the segment
    }
  }
  __syncthreads();
  int base = sh_base;
  for (int q = 0; q < wv; q++) base += wave_tot[q];
  if (ok) {
    const int slot = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    reinterpret_cast<float4*>(ob)[slot] = make_float4(box[0], box[1], box[2], box[3]);
    os[slot] = s;
  }
}

// After NMS: gather the kept proposals of every segment.  keep [S, keep_stride] positions into the sorted boxes.
__global__ void rpn_gather_kept_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores, int k_stride,
                                       const int32_t* __restrict__ keep, const int32_t* __restrict__ keep_count,
                                       int keep_stride, float4* __restrict__ out_boxes, float* __restrict__ out_scores) {
  const int seg = blockIdx.y;
  const int n = keep_count[seg];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int src = keep[(size_t)seg * keep_stride + i];
  out_boxes[(size_t)seg * keep_stride + i] = boxes[(size_t)seg * k_stride + src];
  out_scores[(size_t)seg * keep_stride + i] = scores[(size_t)seg * k_stride + src];
}

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

struct RpnPlan {
  int chunks_per_image, ties_per_image, k_stride, n_seg, dec_blocks;
  size_t off_hist, off_counters, off_ticket, off_blk, off_gt, off_tie, off_sorted, off_nrank, total;
};

static int make_plan(const dtc_rpn_level* levels, int n_levels, int batch, int k_stride, RpnParams* p, RpnPlan* plan) {
  if (n_levels < 1 || n_levels > kRpnMaxLevels || batch < 1) return DTC_EINVAL;
  int chunks = 0, ties = 0, kmax = 0;
  for (int l = 0; l < n_levels; l++) {
    const dtc_rpn_level& s = levels[l];
    if (s.num_anchors < 1 || s.num_anchors > kRpnMaxAnchors || s.height < 1 || s.width < 1) return DTC_EINVAL;
    const long long N = (long long)s.num_anchors * s.height * s.width;
    if (N >= (1ll << 27)) return DTC_EUNSUPPORTED;
    const int K = (s.pre_nms_top_n <= 0 || s.pre_nms_top_n >= N) ? (int)N : s.pre_nms_top_n;
    if (K > 16384) return DTC_EUNSUPPORTED;
    if (p) {
      RpnLevelDev& d = p->lv[l];
      d.cls = s.cls_prob; d.bbox = s.bbox_pred; d.A = s.num_anchors; d.H = s.height; d.W = s.width; d.N = (int)N; d.K = K;
      d.chunk_begin = chunks; d.tie_begin = ties; d.feat_stride = s.feat_stride; d.logit = s.score_is_logit != 0;
      for (int i = 0; i < s.num_anchors * 4; i++) d.anchors[i] = s.anchors[i];
    }
    chunks += (int)((N + kChunk - 1) / kChunk);
    ties += (int)N;
    if (K > kmax) kmax = K;
  }
  if (k_stride <= 0) k_stride = kmax;
  if (k_stride < kmax) return DTC_EINVAL;
  const int S = batch * n_levels;
  plan->chunks_per_image = chunks; plan->ties_per_image = ties; plan->k_stride = k_stride; plan->n_seg = S;
  size_t o = 0;
  plan->dec_blocks = (k_stride + kDecodeThreads - 1) / kDecodeThreads;
  // [histograms | counters | tickets | block counts]: ONE clearing launch covers everything up to off_gt
  plan->off_hist = o; o += al((size_t)S * 2 * kHistBins * sizeof(uint32_t));
  plan->off_counters = o; o += al((size_t)S * 2 * sizeof(uint32_t));
  plan->off_ticket = o; o += al((size_t)S * sizeof(uint32_t));
  plan->off_blk = o; o += al((size_t)S * plan->dec_blocks * sizeof(uint32_t));
  plan->off_gt = o; o += al((size_t)S * k_stride * sizeof(uint64_t));
  plan->off_tie = o; o += al((size_t)batch * ties * sizeof(uint64_t));
  plan->off_sorted = o; o += al((size_t)S * k_stride * sizeof(uint64_t));
  plan->off_nrank = o; o += al((size_t)S * sizeof(int32_t));
  plan->total = o;
  return DTC_OK;
}

}  // namespace dtc

DTC_API size_t dtc_rpn_topk_decode_workspace_bytes(const dtc_rpn_level* levels, int n_levels, int batch, int k_stride) {
  dtc::RpnPlan plan;
  if (dtc::make_plan(levels, n_levels, batch, k_stride, nullptr, &plan) != DTC_OK) return 0;
  return plan.total;
}

DTC_API int dtc_rpn_topk_decode(const dtc_rpn_level* levels, int n_levels, int batch, float im_h, float im_w,
                                float min_size_scaled, void* workspace, size_t workspace_bytes, float* out_boxes,
                                float* out_scores, int32_t* out_counts, int k_stride, dtc_stream_t stream) {
  if (!levels || !workspace || !out_boxes || !out_scores || !out_counts) return DTC_EINVAL;
  dtc::RpnParams p;
  dtc::RpnPlan plan;
  int rc = dtc::make_plan(levels, n_levels, batch, k_stride, &p, &plan);
  if (rc != DTC_OK) return rc;
  for (int l = 0; l < n_levels; l++) if (!levels[l].cls_prob || !levels[l].bbox_pred) return DTC_EINVAL;
  if (workspace_bytes < plan.total) return DTC_EWORKSPACE;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  p.n_levels = n_levels; p.batch = batch; p.chunks_per_image = plan.chunks_per_image;
  p.ties_per_image = plan.ties_per_image; p.k_stride = plan.k_stride;
  p.im_h = im_h; p.im_w = im_w; p.min_size = min_size_scaled;
  p.hist = reinterpret_cast<uint32_t*>(w + plan.off_hist);
  p.counters = reinterpret_cast<uint32_t*>(w + plan.off_counters);
  p.gt_keys = reinterpret_cast<uint64_t*>(w + plan.off_gt);
  p.cand_keys = reinterpret_cast<uint64_t*>(w + plan.off_tie);
  p.sorted_keys = reinterpret_cast<uint64_t*>(w + plan.off_sorted);
  p.n_rank = reinterpret_cast<int32_t*>(w + plan.off_nrank);
  p.ticket = reinterpret_cast<uint32_t*>(w + plan.off_ticket);
  p.blk_cnt = reinterpret_cast<uint32_t*>(w + plan.off_blk);
  p.dec_blocks = plan.dec_blocks;
  p.resident = (long long)plan.dec_blocks * plan.n_seg <= 1024 ? 1 : 0;     // 256 CUs hold >= 4 workgroups of 256 threads each
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_counts = out_counts;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // histograms + counters + tickets + block counts are contiguous at the start of the workspace
  if (dtc::zero_async(w, plan.off_gt, s) != DTC_OK) return DTC_ELAUNCH;     // a kernel node, never hipMemsetAsync (dtc_common.h)
  const dim3 grid(plan.chunks_per_image, batch), blk(dtc::kHistThreads);
  hipLaunchKernelGGL(dtc::rpn_hist_kernel<0>, grid, blk, 0, s, p);
  hipLaunchKernelGGL(dtc::rpn_hist_kernel<1>, grid, blk, 0, s, p);
  hipLaunchKernelGGL(dtc::rpn_compact_kernel, grid, blk, 0, s, p);
  DTC_CHECK_LAUNCH();
  const int sort_cap = dtc::next_pow2(plan.k_stride) <= 1024 ? 2048 : dtc::next_pow2(plan.k_stride);
  // sort_cap keys; long segments (radix sort): a second key buffer + 16 x 256 per-wave digit counters
  const size_t smem = dtc::rpn_sort_lds_bytes(sort_cap);
  if (smem > 32 * 1024) {   // static __shared__ of the kernel comes on top: raise the limit well before dynamic + static reaches 64 KB
    DTC_RAISE_LDS_ONCE(dtc::rpn_sort_kernel, 150 * 1024);
  }
  hipLaunchKernelGGL(dtc::rpn_sort_kernel, dim3(plan.n_seg), dim3(dtc::kSortDecodeThreads), smem, s, p, sort_cap);
  DTC_CHECK_LAUNCH();        // (a failed sort launch -- LDS over the limit -- must not be masked by the decode launch that follows)
  hipLaunchKernelGGL(dtc::rpn_decode_kernel, dim3(plan.dec_blocks, plan.n_seg), dim3(dtc::kDecodeThreads), 0, s, p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

DTC_API int dtc_gather_kept(const float* sorted_boxes, const float* sorted_scores, int n_seg, int k_stride,
                            const int32_t* keep, const int32_t* keep_count, int keep_stride, float* out_boxes,
                            float* out_scores, dtc_stream_t stream) {
  if (n_seg <= 0 || keep_stride <= 0) return DTC_OK;
  if (!sorted_boxes || !sorted_scores || !keep || !keep_count || !out_boxes || !out_scores) return DTC_EINVAL;
  hipLaunchKernelGGL(dtc::rpn_gather_kept_kernel, dim3((keep_stride + 255) / 256, n_seg), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(sorted_boxes), sorted_scores,
                     k_stride, keep, keep_count, keep_stride, reinterpret_cast<float4*>(out_boxes), out_scores);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
