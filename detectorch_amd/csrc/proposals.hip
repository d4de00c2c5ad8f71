// A2 + A3 + A4  RPN proposal generation for gfx950: anchor enumeration, pre-NMS top-k, box decode, clip, min-size filter
// for ALL images and ALL FPN levels of a batch, device resident (no D2H), in 5 launches.
//
// Replaces GenerateProposals.forward up to the NMS call (lib/model/generate_proposals.py:31-109), which per level and
// per image does: numpy anchor meshgrid + H2D (:49-56,124-149), permute+contiguous of scores/deltas + 2 D2H (:64-73),
// host argpartition/argsort (:77-86), ~25 tiny torch kernels for decode/clip (:96-100,165-238), D2H + host filter
// (:101-109).
//
//   rpn_hist<0,1,2>   3-pass radix select (11+11+10 bits) of the K-th largest score per (image, level) segment; LDS
//                     histograms, one global atomic per non-empty bin.  Scores are read once per pass in their native
//                     conv layout [A,H,W] (coalesced); the (A,H,W)->(H,W,A) permute of :64,72 is only an index formula.
//   rpn_compact       elements above the threshold key -> 64-bit (score desc, canonical index asc) keys; elements
//                     EQUAL to it -> tie list (the canonical tie rule picks the lowest indices among them).
//   rpn_sort_decode   one workgroup per segment: bitonic sort of the <= K candidates in LDS, then per rank: anchor =
//                     f(index) from the A base anchors (never materialised, :124-149), decode (:165-214), clip
//                     (:216-238), filter (:151-163), order-preserving compaction.
// Output per segment: boxes [K,4] + scores [K] in descending score order + count  == the `dets` handed to NMS at :115.
#include "block_sort.h"
#include "dtc_common.h"
#include "radix_select.h"

namespace dtc {

constexpr int kRpnMaxLevels = 8;
constexpr int kRpnMaxAnchors = 16;
constexpr int kHistBins = 2048;
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
  uint32_t* hist;       // [S][3][kHistBins]
  uint32_t* counters;   // [S][2] : gt count, tie count
  uint64_t* gt_keys;    // [S][k_stride]
  uint32_t* tie_idx;    // [B][ties_per_image]
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
struct SelState { uint32_t p0, p1, p2, krem; };

template <int PASSES>
__device__ __forceinline__ SelState load_state(const RpnParams& p, int seg, uint32_t K, uint32_t* sh) {
  SelState st; st.p0 = st.p1 = st.p2 = 0; st.krem = K;
  const uint32_t* H = p.hist + (size_t)seg * 3 * kHistBins;
  if (PASSES >= 1) { select_digit(H, kHistBins, st.krem, sh); st.p0 = sh[0]; st.krem = sh[1]; __syncthreads(); }
  if (PASSES >= 2) { select_digit(H + kHistBins, kHistBins, st.krem, sh); st.p1 = sh[0]; st.krem = sh[1]; __syncthreads(); }
  if (PASSES >= 3) { select_digit(H + 2 * kHistBins, 1024, st.krem, sh); st.p2 = sh[0]; st.krem = sh[1]; __syncthreads(); }
  return st;
}

// ---- RPN-head epilogue fusion: scores handed over as LOGITS ----------------------------------------------
// The reference ranks sigmoid(logit) (detector.py:125 -> generate_proposals.py:77-86).  sigmoid is monotone, so the radix
// select runs on the raw logits untouched; but float32 sigmoid is many-to-one (every logit > 16.7 gives 1.0f), and equal
// PROBABILITIES must tie-break by index exactly as if the probabilities had been materialised.  After the select has found
// the K-th largest logit T, the band [lo, hi] of logits whose probability equals P* = sigmoid(T) is located by bisection on
// the ordered keys (sigmoid evaluated in double and rounded once: the same value oracle/numpy produce): logits above the
// band are certainly selected (their sort key carries their probability), logits inside it are the ties, the rest is out.
// Only the K selected elements ever get a sigmoid evaluated -- the [B,A,H,W] probability map is never written or read.
__device__ __forceinline__ float sigmoid_cr(float x) {
  return (float)(1.0 / (1.0 + exp(-(double)x)));
}
struct TieBand { uint32_t lo, hi; float p; };   // ordered-key band of the threshold score and the score itself

// one wave: lane 0 bisects upwards, lane 1 downwards; result broadcast through sh3[3].  Call from all threads of the block.
__device__ __forceinline__ TieBand tie_band(uint32_t T, bool logit, uint32_t* sh3) {
  TieBand tb;
  if (!logit) { tb.lo = tb.hi = T; tb.p = ordered_to_float(T); return tb; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float pt = sigmoid_cr(ordered_to_float(T));
    const uint32_t pk = float_to_ordered(pt);
    const bool up = threadIdx.x == 0;
    // invariant: f(a) == pk, f(b) != pk (b is outside the band or the end of the finite range)
    uint32_t a = T;
    uint32_t b = up ? float_to_ordered(__uint_as_float(0x7f800000u)) : float_to_ordered(__uint_as_float(0xff800000u));
    if (float_to_ordered(sigmoid_cr(ordered_to_float(b))) == pk) a = b;      // band reaches +-inf
    else {
      while ((up ? b - a : a - b) > 1u) {
        const uint32_t m = up ? a + ((b - a) >> 1) : a - ((a - b) >> 1);
        if (float_to_ordered(sigmoid_cr(ordered_to_float(m))) == pk) a = m; else b = m;
      }
    }
    sh3[up ? 1 : 0] = a;
    if (up) sh3[2] = __float_as_uint(pt);
  }
  __syncthreads();
  tb.lo = sh3[0]; tb.hi = sh3[1]; tb.p = __uint_as_float(sh3[2]);
  __syncthreads();
  return tb;
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
  for (int i = threadIdx.x; i < kHistBins; i += kHistThreads) h[i] = 0;
  const SelState st = load_state<PASS>(p, seg, (uint32_t)L.K, sh);  // ends with a barrier (or needs one for PASS 0)
  if (PASS == 0) __syncthreads();
  const float* sc = L.cls + (size_t)b * L.N;
  const int begin = chunk * kChunk, end = min(begin + kChunk, L.N);
  for (int i = begin + threadIdx.x; i < end; i += kHistThreads) {
    const uint32_t o = float_to_ordered(sc[i]);
    if (PASS == 0) atomicAdd(&h[o >> 21], 1u);
    if (PASS == 1) { if ((o >> 21) == st.p0) atomicAdd(&h[(o >> 10) & 2047u], 1u); }
    if (PASS == 2) { if ((o >> 10) == ((st.p0 << 11) | st.p1)) atomicAdd(&h[o & 1023u], 1u); }
  }
  __syncthreads();
  uint32_t* G = p.hist + ((size_t)seg * 3 + PASS) * kHistBins;
  for (int i = threadIdx.x; i < kHistBins; i += kHistThreads) {
    const uint32_t v = h[i];
    if (v) atomicAdd(&G[i], v);
  }
}

__global__ __launch_bounds__(kHistThreads) void rpn_compact_kernel(RpnParams p) {
  // Selected elements are staged in LDS (LDS atomics hand out the slots) and the workgroup claims its output range with
  // ONE global atomic per list: per-wave global atomics on a single counter per segment serialise at the L2
  // (measured 64 us for 8 images before this change).
  __shared__ uint32_t sh[2];
  __shared__ uint32_t lcnt[2], gbase[2];
  __shared__ uint64_t gt_s[kChunk];
  __shared__ uint32_t tie_s[kChunk];
  const int b = blockIdx.y;
  const int l = find_level(p, blockIdx.x);
  const RpnLevelDev& L = p.lv[l];
  const int seg = b * p.n_levels + l;
  const int chunk = blockIdx.x - L.chunk_begin;
  const bool take_all = L.K >= L.N;
  if (threadIdx.x < 2) lcnt[threadIdx.x] = 0;
  __shared__ uint32_t sh3[3];
  TieBand tb; tb.lo = tb.hi = 0; tb.p = 0.f;
  if (!take_all) {
    const SelState st = load_state<3>(p, seg, (uint32_t)L.K, sh);
    tb = tie_band((st.p0 << 21) | (st.p1 << 10) | st.p2, L.logit != 0, sh3);
  }
  __syncthreads();
  const float* sc = L.cls + (size_t)b * L.N;
  const int begin = chunk * kChunk, end = min(begin + kChunk, L.N);
  const int HW = L.H * L.W;
  for (int i = begin + threadIdx.x; i < end; i += kHistThreads) {
    const float s = sc[i];
    const uint32_t o = float_to_ordered(s);
    const bool is_gt = take_all || o > tb.hi;
    const bool is_tie = !take_all && o >= tb.lo && o <= tb.hi;
    if (is_gt || is_tie) {
      // memory index i = (a*H + h)*W + w  ->  canonical index n = (h*W + w)*A + a   (generate_proposals.py:64,72)
      const int a = i / HW, hw = i - a * HW;
      const uint32_t n = (uint32_t)(hw * L.A + a);
      if (is_gt) gt_s[atomicAdd(&lcnt[0], 1u)] = make_desc_key(L.logit ? sigmoid_cr(s) : s, n);
      else tie_s[atomicAdd(&lcnt[1], 1u)] = n;
    }
  }
  __syncthreads();
  uint32_t* cnt = p.counters + (size_t)seg * 2;
  if (threadIdx.x < 2) gbase[threadIdx.x] = lcnt[threadIdx.x] ? atomicAdd(&cnt[threadIdx.x], lcnt[threadIdx.x]) : 0u;
  __syncthreads();
  uint64_t* gt = p.gt_keys + (size_t)seg * p.k_stride;
  uint32_t* tie = p.tie_idx + (size_t)b * p.ties_per_image + L.tie_begin;
  for (uint32_t j = threadIdx.x; j < lcnt[0]; j += kHistThreads)
    if (gbase[0] + j < (uint32_t)p.k_stride) gt[gbase[0] + j] = gt_s[j];
  for (uint32_t j = threadIdx.x; j < lcnt[1]; j += kHistThreads) tie[gbase[1] + j] = tie_s[j];
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

__global__ __launch_bounds__(kSortDecodeThreads) void rpn_sort_decode_kernel(RpnParams p, int sort_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ uint32_t sh[2];
  __shared__ uint32_t hidx[512];
  __shared__ int wave_tot[kSortDecodeThreads / 64];
  __shared__ int running;
  const int seg = blockIdx.x;
  const int b = seg / p.n_levels, l = seg - b * p.n_levels;
  const RpnLevelDev& L = p.lv[l];
  const int tid = threadIdx.x;
  const uint32_t n_gt = min(p.counters[(size_t)seg * 2], (uint32_t)p.k_stride);
  const uint32_t n_tie = p.counters[(size_t)seg * 2 + 1];
  const uint64_t* gt = p.gt_keys + (size_t)seg * p.k_stride;
  const uint32_t* tie = p.tie_idx + (size_t)b * p.ties_per_image + L.tie_begin;
  const int K = L.K;
  const uint32_t need = (uint32_t)K - n_gt;  // ties to take (0 when take_all)
  float tie_score = 0.f;
  if (n_tie) {  // all ties share the threshold score: recover it from the selection state
    const SelState st = load_state<3>(p, seg, (uint32_t)K, sh);
    tie_score = ordered_to_float((st.p0 << 21) | (st.p1 << 10) | st.p2);
    if (L.logit) tie_score = sigmoid_cr(tie_score);     // == TieBand::p of the compaction pass
  }
  uint32_t idx_limit = 0xffffffffu;  // ties with canonical index <= idx_limit are taken
  uint32_t n_take = n_tie;
  if (n_gt + n_tie > (uint32_t)sort_cap) {
    // Massive tie at the threshold (e.g. a constant score map): pick the `need` lowest canonical indices with a
    // 2 x 9-bit radix select over the tie list (indices < 2^18 would suffice for FPN; 3 passes cover 2^27).
    uint32_t prefix = 0, rem = need;
    for (int pass = 0; pass < 3; pass++) {
      const int shift = 18 - 9 * pass;
      for (int i = tid; i < 512; i += kSortDecodeThreads) hidx[i] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n_tie; i += kSortDecodeThreads) {
        const uint32_t v = tie[i];
        if (pass == 0 || (v >> (shift + 9)) == prefix) atomicAdd(&hidx[(v >> shift) & 511u], 1u);
      }
      __syncthreads();
      if (tid == 0) {  // ascending select: smallest d with cumulative >= rem
        uint32_t acc = 0; int d = 0;
        for (d = 0; d < 512; d++) { if (acc + hidx[d] >= rem) break; acc += hidx[d]; }
        sh[0] = (uint32_t)d; sh[1] = rem - acc;
      }
      __syncthreads();
      prefix = (prefix << 9) | sh[0]; rem = sh[1];
      __syncthreads();
    }
    idx_limit = prefix;
    n_take = need;
  }
  // gather candidates into LDS.  Ties are appended through an LDS cursor when filtered by idx_limit.
  const int total = (int)(n_gt + n_take);
  const int np2 = next_pow2(total);
  for (int i = tid; i < np2; i += kSortDecodeThreads) keys[i] = i < (int)n_gt ? gt[i] : kPadKey;
  if (tid == 0) running = (int)n_gt;
  __syncthreads();
  if (idx_limit == 0xffffffffu) {
    for (uint32_t i = tid; i < n_tie; i += kSortDecodeThreads) keys[n_gt + i] = make_desc_key(tie_score, tie[i]);
  } else {
    for (uint32_t i = tid; i < n_tie; i += kSortDecodeThreads) {
      const uint32_t v = tie[i];
      if (v <= idx_limit) { const int slot = atomicAdd(&running, 1); if (slot < np2) keys[slot] = make_desc_key(tie_score, v); }
    }
  }
  __syncthreads();
  block_bitonic_sort<kSortDecodeThreads>(keys, np2);

  // ranks [0, K) in score order: decode, clip, filter, ordered compaction
  const float* sc = L.cls + (size_t)b * L.N;
  const float* dl = L.bbox + (size_t)b * L.N * 4;
  const int HW = L.H * L.W;
  float* ob = p.out_boxes + (size_t)seg * p.k_stride * 4;
  float* os = p.out_scores + (size_t)seg * p.k_stride;
  if (tid == 0) running = 0;
  __syncthreads();
  const int n_rank = min(K, total);
  for (int k0 = 0; k0 < n_rank; k0 += kSortDecodeThreads) {
    const int k = k0 + tid;
    bool ok = false;
    float box[4] = {0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
    if (k < n_rank) {
      const uint32_t n = desc_key_index(keys[k]);
      const int a = n % L.A, hw = n / L.A;
      const int h = hw / L.W, w = hw - h * L.W;
      s = L.logit ? desc_key_score(keys[k]) : sc[(size_t)a * HW + hw];
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
    // ordered compaction: wave ballot + cross-wave prefix
    const uint64_t m = __ballot(ok);
    const int wv = tid >> 6, lane = tid & 63;
    if (lane == 0) wave_tot[wv] = __builtin_popcountll(m);
    __syncthreads();
    int base = running;
    for (int q = 0; q < wv; q++) base += wave_tot[q];
    if (ok) {
      const int slot = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      reinterpret_cast<float4*>(ob)[slot] = make_float4(box[0], box[1], box[2], box[3]);
      os[slot] = s;
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int q = 0; q < kSortDecodeThreads / 64; q++) t += wave_tot[q]; running += t; }
    __syncthreads();
  }
  if (tid == 0) p.out_counts[seg] = running;
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
  int chunks_per_image, ties_per_image, k_stride, n_seg;
  size_t off_hist, off_counters, off_gt, off_tie, total;
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
  plan->off_hist = o; o += al((size_t)S * 3 * kHistBins * sizeof(uint32_t));
  plan->off_counters = o; o += al((size_t)S * 2 * sizeof(uint32_t));
  plan->off_gt = o; o += al((size_t)S * k_stride * sizeof(uint64_t));
  plan->off_tie = o; o += al((size_t)batch * ties * sizeof(uint32_t));
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
  p.tie_idx = reinterpret_cast<uint32_t*>(w + plan.off_tie);
  p.out_boxes = out_boxes; p.out_scores = out_scores; p.out_counts = out_counts;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // histograms + counters are contiguous at the start of the workspace
  if (dtc::zero_async(w, plan.off_gt, s) != DTC_OK) return DTC_ELAUNCH;     // a kernel node, never hipMemsetAsync (dtc_common.h)
  const dim3 grid(plan.chunks_per_image, batch), blk(dtc::kHistThreads);
  hipLaunchKernelGGL(dtc::rpn_hist_kernel<0>, grid, blk, 0, s, p);
  hipLaunchKernelGGL(dtc::rpn_hist_kernel<1>, grid, blk, 0, s, p);
  hipLaunchKernelGGL(dtc::rpn_hist_kernel<2>, grid, blk, 0, s, p);
  hipLaunchKernelGGL(dtc::rpn_compact_kernel, grid, blk, 0, s, p);
  DTC_CHECK_LAUNCH();
  const int sort_cap = dtc::next_pow2(plan.k_stride) <= 1024 ? 2048 : dtc::next_pow2(plan.k_stride);
  const size_t smem = (size_t)sort_cap * sizeof(uint64_t);
  if (smem > 32 * 1024) {   // static __shared__ of the kernel comes on top: raise the limit well before dynamic + static reaches 64 KB
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(dtc::rpn_sort_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) return DTC_ELAUNCH;
      raised = true;
    }
  }
  hipLaunchKernelGGL(dtc::rpn_sort_decode_kernel, dim3(plan.n_seg), dim3(dtc::kSortDecodeThreads), smem, s, p, sort_cap);
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
