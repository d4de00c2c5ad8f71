// A8  Detection post-processing for gfx950 -- replaces postprocess_output (lib/utils/result_utils.py:76-94) and
// box_results_with_nms_and_limit (:96-168, hard-NMS branch), which run on the host: 3 D2H copies, numpy decode of all
// R x 81 boxes, an 80-iteration Python loop calling Cython NMS, np.sort for the per-image limit.
//
//   det_candidates   grid (class, image): score > thresh (:127) -> ordered compaction (dets_j order), class-specific box
//                    decode with weights (10,10,5,5) (lib/utils/boxes.py:168-208) ONLY for the candidates, clip to the
//                    original image (:150-165), (score desc, index asc) sort in LDS -> NMS input
//                    and the segment's hard NMS (cython_nms.pyx:37-87) in the same workgroup: blocks of 64 rows, diagonal tile by
//                    four waves, greedy walk by one, later columns marked by all (round 4; until then a separate launch pair)
//   det_finalize     grid (image): max_detections_per_img limit (:154-163) = radix select of the 100-th largest score,
//                    `>=` filter (ties kept, like the reference), class-major / roi-ascending output (:165).
#include "block_sort.h"
#include "dtc_common.h"
#include "fpn_map.h"
#include "radix_select.h"

namespace dtc {
DTC_PT_TABLE(detections)

constexpr int kDetThreads = 256;
constexpr int kNmsLdsCap = 512;       // candidates of a (class, image) segment whose boxes the in-workgroup NMS keeps in LDS

struct DetParams {
  const float* rois5;        // [B, R, 5]
  const int32_t* n_rois;     // [B] or NULL (all R valid)
  const float* cls_score;    // [B, R, n_cls]  probabilities, or LOGITS when sm_stats != NULL
  const double* sm_stats;    // [B, R, 2] = (row max, sum_j exp(l_j - max)) of the class logits, or NULL
  const float* bbox_pred;    // [B, R, 4*n_cls]
  const float* decoded;      // [B, R, 4*n_cls] already decoded + clipped boxes (box_results_with_nms_and_limit's input), or NULL
  const float* scale;        // [B] scaling factor per image
  const float* im_size;      // [B, 2] original (h, w)
  int R, n_cls;
  float wx, wy, ww, wh, score_thresh;
  // per (image, class) segment s = b*(n_cls-1) + (j-1), stride R
  float* sorted_boxes;       // [S, R, 4]   score order: scratch of the NMS for segments of more than kNmsLdsCap candidates
  float* q_boxes;            // [S, R, 4]   candidate order
  float* q_scores;           // [S, R]
  int32_t* q_roi;            // [S, R]
  int32_t* cand_count;       // [S]
  // per-class NMS, in this kernel: the sort keys (~ordered score << 32 | candidate index q) of the kept boxes, in score order,
  // and their number -- everything det_finalize needs about a kept box in ONE load
  uint64_t* kept_key;        // [S, R]
  int32_t* keep_count;       // [S]
  float nms_thresh;
  int np2_max;               // next_pow2(R): the sort keys occupy the first np2_max * 8 bytes of the dynamic LDS
};

// lib/utils/boxes.py:168-208 for one (roi, class)
__device__ __forceinline__ void decode_det(const float roi[4], float sf, const float* d, float wx, float wy, float ww,
                                           float wh, float im_h, float im_w, float out[4]) {
  const float x1 = fdiv(roi[0], sf), y1 = fdiv(roi[1], sf), x2 = fdiv(roi[2], sf), y2 = fdiv(roi[3], sf);  // result_utils.py:77
  const float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;           // boxes.py:178-179
  const float ctr_x = x1 + 0.5f * widths, ctr_y = y1 + 0.5f * heights;     // :180-181
  const float dx = fdiv(d[0], wx), dy = fdiv(d[1], wy);                    // :184-185
  float dw = fdiv(d[2], ww), dh = fdiv(d[3], wh);                          // :186-187
  const float clipv = 4.135166556742356f;                                  // :73
  dw = fminf(dw, clipv); dh = fminf(dh, clipv);                            // :190-191
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;       // :193-194
  const float pw = fexp_cr(dw) * widths, ph = fexp_cr(dh) * heights;       // :195-196
  float b0 = pcx - 0.5f * pw, b1 = pcy - 0.5f * ph;                        // :200-202
  float b2 = pcx + 0.5f * pw - 1.f, b3 = pcy + 0.5f * ph - 1.f;            // :204-206
  out[0] = fmaxf(fminf(b0, im_w - 1.f), 0.f); out[1] = fmaxf(fminf(b1, im_h - 1.f), 0.f);  // :158-164
  out[2] = fmaxf(fminf(b2, im_w - 1.f), 0.f); out[3] = fmaxf(fminf(b3, im_h - 1.f), 0.f);
}

// ---- box-head epilogue fusion: class scores handed over as LOGITS -----------------------------------------
// The reference applies F.softmax to the cls_score layer's output (lib/model/detector.py:281) and hands the [R,81]
// probability map to postprocess_output.  Here the probability is formed where it is consumed: one wave per roi row
// reduces (max, sum of exp) once -- 16 bytes per roi instead of a second [R,81] map -- and det_candidates evaluates
//     prob[r,j] = float( exp(double(l[r,j]) - max_r) / sum_r )
// for the one class column it scans.  Evaluated in double and rounded once, the sum taken in a FIXED order (element j goes
// to slot j % 64 in increasing j, then a 64 -> 1 halving tree), so that oracle/oracle.py:softmax_rows gives the same bits;
// torch's float32 softmax agrees with it to rel 1e-6 (tests/test_oracle_golden.py).
__global__ __launch_bounds__(kDetThreads) void det_softmax_stats_kernel(const float* logits, int n_rows, int n_cls, double* stats) {
  const int row = blockIdx.x * (kDetThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const float* l = logits + (size_t)row * n_cls;
  float m = -INFINITY;
  for (int j = lane; j < n_cls; j += 64) m = fmaxf(m, l[j]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  double e = 0.0;
  for (int j = lane; j < n_cls; j += 64) e += exp((double)l[j] - (double)m);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off, 64);       // lane 0: ((s0+s32)+(s16+s48))+... halving tree
  if (lane == 0) { stats[(size_t)row * 2] = (double)m; stats[(size_t)row * 2 + 1] = e; }
}

__global__ __launch_bounds__(kDetThreads) void det_candidates_kernel(DetParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ int running;
  const int j = blockIdx.x + 1, b = blockIdx.y;
  const int seg = b * (p.n_cls - 1) + (j - 1);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nr = p.n_rois ? min(p.n_rois[b], p.R) : p.R;
  const float* sc = p.cls_score + (size_t)b * p.R * p.n_cls + j;
  float* qs = p.q_scores + (size_t)seg * p.R;
  [[maybe_unused]] const int ptb = blockIdx.y * gridDim.x + blockIdx.x;
  DTC_PT(0, ptb, 0);
  if (tid == 0) running = 0;
  __syncthreads();
  // compaction of {r : scores[r, j] > thresh}  (np.where, result_utils.py:127).  Round 6: UNORDERED -- a wave claims its slots with
  // one LDS atomic, no workgroup barrier per chunk.  Nothing downstream depends on the slot order: the sort key carries the ROI INDEX r
  // ((score desc, r asc) is the order (score desc, candidate index asc) was, the candidate index being monotone in r), and the
  // per-candidate scratch (q_boxes / q_scores) is indexed by r.  The scores of FOUR chunks of 256 rois are requested before the first
  // is consumed, and before the image's roi count is known (the array is [B, R, n_cls]: every address is valid; rows past the count
  // are masked afterwards) -- one global round trip in front of the first compare instead of two dependent ones.
  int R0 = 0;
  do {
    float sv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = R0 + u * kDetThreads + tid;
      sv[u] = 0.f;
      if (r < p.R) {
        sv[u] = sc[(size_t)r * p.n_cls];
        if (p.sm_stats) {      // logits in: softmax column formed here (detector.py:281)
          const double* st = p.sm_stats + ((size_t)b * p.R + r) * 2;
          sv[u] = (float)(exp((double)sv[u] - st[0]) / st[1]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = R0 + u * kDetThreads + tid;
      const float s = sv[u];
      const bool ok = r < nr && s > p.score_thresh;
      const uint64_t m = __ballot(ok);
      if (m == 0ull) continue;                                 // uniform per wave
      int base = 0;
      if (lane == 0) base = atomicAdd(&running, __builtin_popcountll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (ok) {
        qs[r] = s;
        keys[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_desc_key(s, (uint32_t)r);
      }
    }
    R0 += 4 * kDetThreads;
  } while (R0 < nr);
  __syncthreads();
  const int n = running;
  if (tid == 0) {
    p.cand_count[seg] = n;
    if (n == 0) p.keep_count[seg] = 0;
  }
  DTC_PT(0, ptb, 1);
  if (n == 0) return;
  const int np2 = next_pow2(n);
  for (int i = n + tid; i < np2; i += kDetThreads) keys[i] = kPadKey;
  block_bitonic_sort<kDetThreads>(keys, np2);
  DTC_PT(0, ptb, 2);
  // the box of rank k (decoded once): to q_boxes[r] for det_finalize, and in SCORE order for the NMS below -- in LDS when the segment
  // has at most kNmsLdsCap candidates (the usual tens to a few hundred), else in the global scratch p.sorted_boxes (LDS sized for
  // every possible segment would leave 2 workgroups per CU)
  float4* qb = reinterpret_cast<float4*>(p.q_boxes) + (size_t)seg * p.R;
  float4* sbox_l = reinterpret_cast<float4*>(smem + (size_t)p.np2_max * sizeof(uint64_t));     // [kNmsLdsCap]
  float4* sorted_g = reinterpret_cast<float4*>(p.sorted_boxes) + (size_t)seg * p.R;
  const bool in_lds = n <= kNmsLdsCap;
  const float sf = p.decoded ? 1.f : p.scale[b];
  const float im_h = p.decoded ? 0.f : p.im_size[b * 2 + 0], im_w = p.decoded ? 0.f : p.im_size[b * 2 + 1];
  for (int k = tid; k < n; k += kDetThreads) {
    const int r = (int)desc_key_index(keys[k]);
    float4 v;
    if (p.decoded) {        // lib/utils/result_utils.py:128: boxes[inds, j * 4:(j + 1) * 4] taken as they are
      const float* d = p.decoded + ((size_t)b * p.R + r) * 4 * p.n_cls + 4 * j;
      v = make_float4(d[0], d[1], d[2], d[3]);
    } else {
      const float* roi = p.rois5 + ((size_t)b * p.R + r) * 5 + 1;
      const float* d = p.bbox_pred + ((size_t)b * p.R + r) * 4 * p.n_cls + 4 * j;
      const float rr[4] = {roi[0], roi[1], roi[2], roi[3]};
      float o[4];
      decode_det(rr, sf, d, p.wx, p.wy, p.ww, p.wh, im_h, im_w, o);
      v = make_float4(o[0], o[1], o[2], o[3]);
    }
    qb[r] = v;                                               // by roi index (global, det_finalize)
    if (in_lds) sbox_l[k] = v; else sorted_g[k] = v;         // score order (the NMS)
  }
  __syncthreads();
  DTC_PT(0, ptb, 3);
  DTC_PT(0, ptb, 4);
  // ---- the segment's hard NMS, here (cython_nms.pyx:37-87: greedy over the score order; the kept box of rank i suppresses every
  // later box j with inter / (area_i + area_j - inter) >= thresh, IEEE division).  The class segments of a detection batch hold tens
  // of candidates (one 64-row block); as a separate launch pair (mask tiles + reduce over 640 segments) they cost 25 us of which
  // 20 were launch latency.  Blocks of 64 rows: (1) the four waves form the block's diagonal tile (16 rows each), (2) wave 0 walks
  // the block greedily with the bits removed by earlier blocks, (3) the block's kept rows mark the later columns they suppress
  // (one 64-column word per wave and step).  n^2 / 2 pair tests by ONE workgroup: 50 candidates 3 us, 1000 ~100 us, 4096 would be
  // ~1.5 ms -- a deliberate trade for the 81-class detection heads this path serves (segments of tens of candidates); a model with a
  // handful of classes, a very low score threshold or collect_top_n >> 2000 makes a class segment the critical path of the launch
  // (timing guard: tests/test_hip_fpn_det_mask.py::test_postprocess_crowded_classes_vs_oracle, R = 1500 x 3 classes).
  {
    uint64_t* removed = reinterpret_cast<uint64_t*>(sbox_l + kNmsLdsCap);                            // [(R + 63) / 64]
    __shared__ uint32_t diag_s[kDetThreads / 64][64];
    __shared__ uint64_t keptm_s;
    __shared__ int kept_s;
    const int ncb = (n + 63) >> 6;
    DTC_PT(0, ptb, 5);
    for (int wd = tid; wd < ncb; wd += kDetThreads) removed[wd] = 0ull;
    if (tid == 0) kept_s = 0;
    __syncthreads();
    const float thr = p.nms_thresh;
    const bool thr_pos = thr > 0.f;
    auto area_of = [](const float4& b) { return (b.z - b.x + 1.f) * (b.w - b.y + 1.f); };           // :44
    // `inter / (area_r + area_c - inter) >= thresh` with an IEEE division -- decided WITHOUT dividing whenever the sign of
    // d = fl(inter - fl(thresh * u)) is reliable (|d| > 2^-21 thresh u, u > 0, thresh > 0: rounding is monotone, nms.hip explains);
    // the division only when some lane of the wavefront is inside that band (about one pair in 10^6)
    auto iou_ge = [&](const float4& r, float rarea, const float4& c, float carea) {
      const float xx1 = fmaxf(r.x, c.x), yy1 = fmaxf(r.y, c.y), xx2 = fminf(r.z, c.z), yy2 = fminf(r.w, c.w);   // :76-79
      const float w = fmaxf(0.0f, xx2 - xx1 + 1.f), h = fmaxf(0.0f, yy2 - yy1 + 1.f);                         // :80-81
      const float inter = w * h;                                                                            // :82
      const float u = rarea + carea - inter;
      const float pu = thr * u;
      const float d = inter - pu;
      const float t = __builtin_fabsf(d) - pu * 4.76837158203125e-07f;                                       // 2^-21
      const bool unsure = !(fminf(t, u) > 0.f);
      if (__builtin_amdgcn_ballot_w64(unsure) == 0ull && thr_pos) return d > 0.f;
      return (unsure || !thr_pos) ? fdiv(inter, u) >= thr : d > 0.f;                                         // :83-84
    };
    uint64_t* K = p.kept_key + (size_t)seg * p.R;
    const float4 pad = make_float4(0.f, 0.f, -1.f, -1.f);
    auto run_blocks = [&](auto lds_tag) {
    const float4* sbox = decltype(lds_tag)::value ? static_cast<const float4*>(sbox_l) : static_cast<const float4*>(sorted_g);
    for (int rb = 0; rb < ncb; rb++) {
      const int i0 = rb * 64, nrow = min(64, n - i0);
      // (1) diagonal tile: which rows r < lane of this block suppress column i0 + lane; wave wv tests rows [16 wv, 16 wv + 16)
      const float4 cbx = lane < nrow ? sbox[i0 + lane] : pad;
      const float carea = area_of(cbx);
      uint32_t part = 0;
      for (int r = 16 * wv; r < min(16 * wv + 16, nrow); r++) {
        const float4 rbx = sbox[i0 + r];                       // uniform address: LDS broadcast
        const bool sup = lane > r && lane < nrow && iou_ge(rbx, area_of(rbx), cbx, carea);
        part |= sup ? (1u << (r - 16 * wv)) : 0u;
      }
      diag_s[wv][lane] = part;
      __syncthreads();
      DTC_PT(0, ptb, 6 + 3 * min(rb, 2));
      // (2) greedy walk of the block (wave 0): a row is kept iff no earlier kept row (earlier blocks: `removed`) suppresses it
      if (wv == 0) {
        const uint64_t colword = (uint64_t)diag_s[0][lane] | ((uint64_t)diag_s[1][lane] << 16) | ((uint64_t)diag_s[2][lane] << 32) |
                                 ((uint64_t)diag_s[3][lane] << 48);
        // the greedy answer as a fixed point: row j is kept iff it is a candidate and no KEPT earlier row of the block suppresses it;
        // K <- {j : cand_j and colword_j & K == 0} from K = cand settles rows in increasing order (row 0 after one step, row j once
        // the rows before it have settled) -- a few wave-wide steps instead of one dependent step per row
        const uint64_t cand = __ballot(lane < nrow) & ~removed[rb];
        const bool mine = (cand >> lane) & 1ull;
        uint64_t keptm = cand;
        for (int it = 0; it < 64; it++) {
          const uint64_t nk = __ballot(mine && (colword & keptm) == 0ull);
          if (nk == keptm) break;
          keptm = nk;
        }
        const int base = kept_s;
        if ((keptm >> lane) & 1ull) K[base + __builtin_popcountll(keptm & ((1ull << lane) - 1ull))] = keys[i0 + lane];
        if (lane == 0) { keptm_s = keptm; kept_s = base + __builtin_popcountll(keptm); }
      }
      __syncthreads();
      DTC_PT(0, ptb, 7 + 3 * min(rb, 2));
      // (3) the kept rows of this block against every later column: a wave takes one 64-column word per step
      const uint64_t keptm = keptm_s;
      // wave wv takes the block's rows [16 wv, 16 wv + 16) against EVERY later 64-column word (lane <-> column): a segment of ~100
      // candidates has one such word, and its ~50 kept rows are four waves' 13 instead of one wave's 50
      const uint64_t mine_rows = keptm & (0xffffull << (16 * wv));
      if (mine_rows) {
        for (int c0 = i0 + 64; c0 < n; c0 += 64) {
          const int j = c0 + lane;
          const float4 cj = j < n ? sbox[j] : pad;
          const float aj = area_of(cj);
          bool sup = false;
          for (uint64_t km = mine_rows; km; km &= km - 1ull) {   // uniform
            const float4 rbx = sbox[i0 + __builtin_ctzll(km)];
            sup = sup || iou_ge(rbx, area_of(rbx), cj, aj);
          }
          const uint64_t m = __ballot(j < n && sup);
          if (lane == 0 && m) atomicOr(reinterpret_cast<unsigned long long*>(&removed[c0 >> 6]), (unsigned long long)m);
        }
      }
      __syncthreads();
    }
    };
    if (in_lds) run_blocks(std::true_type{}); else run_blocks(std::false_type{});
    if (tid == 0) p.keep_count[seg] = kept_s;
    DTC_PT(0, ptb, 15);
  }
}

constexpr int kFinThreads = 1024;
constexpr int kFinMaxCls = 256;

struct FinParams {
  const uint64_t* kept_key;   // [S, R] sort keys of the kept boxes (det_candidates' NMS): ~ordered score << 32 | candidate index q
  const int32_t* keep_count;  // [S]
  const float* q_boxes;       // [S, R, 4]
  const float* q_scores;      // [S, R]
  const int32_t* q_roi;       // [S, R]
  const float* scale;         // [B]
  int R, n_cls, max_det, max_out;
  float* dets;                // [B, max_out, 6]
  int32_t* det_roi;           // [B, max_out]
  float* det_rois_scaled;     // [B, max_out, 4]  boxes * scaling_factor (eval_mask_FPN.ipynb:249), may be NULL
  int32_t* det_count;         // [B]
  FpnMapOut fm;               // fm.on: also emit the FPN level mapping of the detection rows (the mask branch's rois), fpn_map.h
};

// off[c] = sum of cnt(c') for c' < c, c = 0 .. nseg (off[nseg] = total); one wavefront, nseg <= 256.
template <typename F> __device__ __forceinline__ void fin_prefix(int* off, F cnt, int nseg, int lane) {
  int v[4], t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { v[k] = cnt(lane * 4 + k); t += v[k]; }
  int incl = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  int e = incl - t;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int c = lane * 4 + k; if (c <= nseg) off[c] = e; e += v[k]; }
  if (lane == 63 && nseg >= 256) off[nseg] = e;
}

// kept entries staged in LDS (ordered score + candidate id, 8 bytes each): dynamic, sized by the launcher for the worst case the
// arguments allow up to kFinStageMax (112 KB); more -> every pass re-fetches from global
constexpr int kFinStageMax = 14336;     // 112 KB dynamic + ~37 KB static

constexpr int kFinSurvMax = 1024;    // survivors of the per-image limit the fast output path ranks by scanning (usual: ~100)

__global__ __launch_bounds__(kFinThreads) void det_finalize_kernel(FinParams p, int stage_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
  uint32_t* st_key = reinterpret_cast<uint32_t*>(fin_smem);             // [stage_cap] ordered score of kept entry f
  uint32_t* st_cq = st_key + stage_cap;                                   // [stage_cap] (class c << 12) | candidate index q of kept entry f
  // one histogram per radix pass (11 + 11 + 10 bits), cleared once under the first global round trip: a pass is its atomics, one
  // barrier and the digit selection -- not clear / barrier / atomics / barrier / select / barrier on ONE array (round 6: 15 -> 8 barriers)
  __shared__ __attribute__((aligned(16))) uint32_t h0[2048], h1[2048], h2[1024];
  __shared__ uint32_t sh[2];
  __shared__ int kcnt[kFinMaxCls];
  __shared__ int koff[kFinMaxCls + 1];
  __shared__ int ccnt[kFinMaxCls];
  __shared__ int coff[kFinMaxCls + 1];
  __shared__ uint64_t bitmap[kFinThreads / 64][64];  // per wave: up to 4096 candidates per class
  __shared__ __attribute__((aligned(16))) uint32_t surv[kFinSurvMax + 4];
  __shared__ int nsurv;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NWV = kFinThreads / 64, kClsPerWave = kFinMaxCls / NWV;
  const int nseg = p.n_cls - 1;
  const int seg0 = b * nseg;
  DTC_PT(1, b, 0);
  // Round 6: ONE global round trip in front of the LDS phases instead of three dependent ones (counts -> prefix -> binary search ->
  // keys: 6 us of this kernel's 20).  Wave w owns classes w, w + 16, ...: it requests the class's kept count AND, speculatively, the
  // first 64 kept keys of the class (the array is [S, R]: the addresses are valid whatever the count is; a class keeps a few dozen
  // boxes) in the same breath; entries past 64 are fetched behind the prefix, the rare case.
  int cnt_r[kClsPerWave];
  uint64_t key_r[kClsPerWave];
#pragma unroll
  for (int k = 0; k < kClsPerWave; k++) {
    const int c = wv + k * NWV;
    cnt_r[k] = 0; key_r[k] = 0;
    if (c < nseg) {
      cnt_r[k] = p.keep_count[seg0 + c];
      if (lane < p.R) key_r[k] = p.kept_key[(size_t)(seg0 + c) * p.R + lane];
    }
  }
  if (tid == 0) nsurv = 0;
  for (int i = tid; i < 2048; i += kFinThreads) { h0[i] = 0; h1[i] = 0; if (i < 1024) h2[i] = 0; }
#pragma unroll
  for (int k = 0; k < kClsPerWave; k++) { const int c = wv + k * NWV; if (c < kFinMaxCls && lane == 0) kcnt[c] = cnt_r[k]; }
  __syncthreads();
  // exclusive prefix of the per-class kept counts: wavefront 0, four classes per lane (nseg <= kFinMaxCls = 256), shuffle scan
  if (wv == 0) fin_prefix(koff, [&](int c) { return c < nseg ? kcnt[c] : 0; }, nseg, lane);
  __syncthreads();
  const int total = koff[nseg];
  const bool staged = total <= stage_cap;
  // kept entry e of class c of this image -> (ordered score, candidate index q)
  auto fetch = [&](int c, int e, uint32_t& o, int& q) {
    const uint64_t key = p.kept_key[(size_t)(seg0 + c) * p.R + e];
    q = (int)desc_key_index(key);
    o = ~(uint32_t)(key >> 32);                             // == float_to_ordered(q_scores[q]) (block_sort.h: make_desc_key)
  };
  auto class_of = [&](int f) { int lo = 0, hi = nseg; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (koff[mid] <= f) lo = mid; else hi = mid; } return lo; };
  if (staged) {   // every later phase runs from LDS
#pragma unroll
    for (int k = 0; k < kClsPerWave; k++) {
      const int c = wv + k * NWV;
      if (c >= nseg) continue;
      const int n_c = cnt_r[k], base = koff[c];
      if (lane < n_c) {
        const uint32_t o = ~(uint32_t)(key_r[k] >> 32);
        st_key[base + lane] = o; st_cq[base + lane] = ((uint32_t)c << 12) | desc_key_index(key_r[k]);
        atomicAdd(&h0[o >> 21], 1u);                          // the first pass of the limit's radix select, while the entry is in hand
      }
      for (int e = 64 + lane; e < n_c; e += 64) {             // a crowded class: the rest of its entries, one more round trip
        uint32_t o; int q;
        fetch(c, e, o, q);
        st_key[base + e] = o; st_cq[base + e] = ((uint32_t)c << 12) | (uint32_t)q;
        atomicAdd(&h0[o >> 21], 1u);
      }
    }
    __syncthreads();
  }
  DTC_PT(1, b, 1);
  // ---- per-image limit (result_utils.py:154-163): threshold = max_det-th largest kept score ----
  uint32_t T = 0;  // ordered-key threshold; 0 == keep everything
  if (p.max_det > 0 && total > p.max_det) {
    uint32_t prefix = 0, krem = (uint32_t)p.max_det;
    for (int pass = 0; pass < 3; pass++) {
      uint32_t* h = pass == 0 ? h0 : pass == 1 ? h1 : h2;
      if (pass > 0 || !staged) {
        for (int f = tid; f < total; f += kFinThreads) {
          uint32_t o;
          if (staged) o = st_key[f];
          else { const int c = class_of(f); int q; fetch(c, f - koff[c], o, q); }
          if (pass == 0) atomicAdd(&h[o >> 21], 1u);
          else if (pass == 1) { if ((o >> 21) == prefix) atomicAdd(&h[(o >> 10) & 2047u], 1u); }
          else { if ((o >> 10) == prefix) atomicAdd(&h[o & 1023u], 1u); }
        }
        __syncthreads();
      }
      select_digit(h, pass == 2 ? 1024 : 2048, krem, sh);      // ends with a barrier; sh is rewritten behind the next pass's barrier
      prefix = (prefix << (pass == 2 ? 10 : 11)) | sh[0];
      krem = sh[1];
    }
    T = prefix;
  }
  __syncthreads();
  DTC_PT(1, b, 2);
  const float sf = p.scale ? p.scale[b] : 1.f;
  float4* sbox_fm = reinterpret_cast<float4*>(&bitmap[0][0]);     // fast path + fm.on: the scaled boxes by output row (max_out <= 512)
  auto emit = [&](int slot, int c, int q, bool to_lds) {      // one detection row (:143 dets_j[keep], :165 vstack)
    const int seg = seg0 + c;
    const float4 bx = reinterpret_cast<const float4*>(p.q_boxes)[(size_t)seg * p.R + q];
    if (to_lds) sbox_fm[slot] = make_float4(bx.x * sf, bx.y * sf, bx.z * sf, bx.w * sf);
    float* o = p.dets + ((size_t)b * p.max_out + slot) * 6;
    o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w;
    o[4] = p.q_scores[(size_t)seg * p.R + q];
    o[5] = (float)(c + 1);
    p.det_roi[(size_t)b * p.max_out + slot] = q;                               // the candidate index IS the roi index (round 6)
    if (p.det_rois_scaled)
      reinterpret_cast<float4*>(p.det_rois_scaled)[(size_t)b * p.max_out + slot] = make_float4(bx.x * sf, bx.y * sf, bx.z * sf, bx.w * sf);
  };
  // ---- fast output path (round 6): the survivors of the limit are ~100 rows.  They are appended to an LDS list in any order; the
  // output row of a survivor is the number of survivors with a smaller (class, candidate) code -- class-major, candidate (= roi)
  // ascending, exactly the reference's vstack order -- found by scanning the list with broadcast 16-byte reads; then ONE round of
  // gathers.  (The per-class bitmap walk below paid five rounds of dependent loads per wave: 7 us.)
  bool fast_done = false;
  if (staged) {
    for (int f = tid; f < total; f += kFinThreads) {
      if (st_key[f] >= T) {                                                    // :161 `>=`
        const int i = atomicAdd(&nsurv, 1);
        if (i < kFinSurvMax) surv[i] = st_cq[f];
      }
    }
    __syncthreads();
    const int ns = nsurv;
    if (ns <= kFinSurvMax) {
      fast_done = true;
      if (tid < 4) surv[ns + tid] = 0xffffffffu;                              // pad to a multiple of four: never smaller
      if (tid == 0) p.det_count[b] = ns;
      __syncthreads();
      const int n4 = (ns + 3) >> 2;
      for (int i = tid; i < ns; i += kFinThreads) {
        const uint32_t me = surv[i];
        int rank = 0;
        for (int j = 0; j < n4; j++) {
          const uint4 v = reinterpret_cast<const uint4*>(surv)[j];
          rank += (v.x < me ? 1 : 0) + (v.y < me ? 1 : 0) + (v.z < me ? 1 : 0) + (v.w < me ? 1 : 0);
        }
        if (rank < p.max_out) emit(rank, (int)(me >> 12), (int)(me & 4095u), p.fm.on != 0);
      }
    }
  }
  if (fast_done) {
    if (p.fm.on) {       // the mask branch's level mapping of these rows, here instead of in a launch of its own (round 6)
      __syncthreads();
      fpn_map_rows(p.fm, b, p.max_out, min(nsurv, p.max_out), [&](int t) { return sbox_fm[t]; }, h0, h1);
    }
    DTC_PT(1, b, 5);
    return;
  }
  // ---- general path: more survivors than the list holds (max_det <= 0 on a large head, mass ties) or entries not staged ----
  // pass A: survivors per class
  for (int c = wv; c < nseg; c += kFinThreads / 64) {
    const int nk = koff[c + 1] - koff[c];
    int cnt = 0;
    for (int e0 = 0; e0 < nk; e0 += 64) {
      const int e = e0 + lane;
      bool ok = false;
      if (e < nk) {
        uint32_t o; int q;
        if (staged) o = st_key[koff[c] + e]; else fetch(c, e, o, q);
        ok = o >= T;                                                         // :161 `>=`
      }
      cnt += __builtin_popcountll(__ballot(ok));
    }
    if (lane == 0) ccnt[c] = cnt;
  }
  __syncthreads();
  DTC_PT(1, b, 3);
  if (wv == 0) {
    fin_prefix(coff, [&](int c) { return c < nseg ? ccnt[c] : 0; }, nseg, lane);
    if (lane == 0) p.det_count[b] = coff[nseg];
  }
  __syncthreads();
  DTC_PT(1, b, 4);
  // ---- pass B: class-major, candidate(roi)-ascending output (:143 dets_j[keep], :165 vstack) ----
  for (int c = wv; c < nseg; c += kFinThreads / 64) {
    const int nk = koff[c + 1] - koff[c];
    if (ccnt[c] == 0) continue;
    uint64_t* bm = bitmap[wv];
    bm[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < nk; e0 += 64) {
      const int e = e0 + lane;
      if (e < nk) {
        uint32_t o; int q;
        if (staged) { o = st_key[koff[c] + e]; q = (int)(st_cq[koff[c] + e] & 4095u); } else fetch(c, e, o, q);
        if (o >= T && q < 4096) atomicOr(reinterpret_cast<unsigned long long*>(&bm[q >> 6]), 1ull << (q & 63));
      }
    }
    __builtin_amdgcn_wave_barrier();
    uint64_t w = bm[lane];
    // exclusive prefix of popcounts across lanes
    int pc = __builtin_popcountll(w), incl = pc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    int slot = coff[c] + incl - pc;
    while (w) {
      const int bit = __builtin_ctzll(w);
      w &= w - 1;
      const int q = lane * 64 + bit;
      if (slot < p.max_out) emit(slot, c, q, false);
      slot++;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (p.fm.on) {         // general path: the rows come back from global memory (written by this workgroup: fence + barrier, uncached loads)
    __threadfence();
    __syncthreads();
    const float* ds = p.det_rois_scaled + (size_t)b * p.max_out * 4;
    auto ld = [&](const float* a) { return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    fpn_map_rows(p.fm, b, p.max_out, min(coff[nseg], p.max_out),
                 [&](int t) { return make_float4(ld(ds + 4 * t), ld(ds + 4 * t + 1), ld(ds + 4 * t + 2), ld(ds + 4 * t + 3)); }, h0, h1);
  }
  DTC_PT(1, b, 5);
}

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace dtc

namespace dtc {
struct DetPlan { size_t sorted_boxes, q_boxes, q_scores, q_roi, cand_count, kept_key, keep_count, sm_stats, total; };
static DetPlan det_plan(int batch, int R, int n_cls) {
  DetPlan d;
  const size_t S = (size_t)batch * (n_cls - 1);
  size_t o = 0;
  d.sorted_boxes = o; o += al256(S * R * 4 * sizeof(float));
  d.q_boxes = o; o += al256(S * R * 4 * sizeof(float));
  d.q_scores = o; o += al256(S * R * sizeof(float));
  d.q_roi = o; o += al256(S * R * sizeof(int32_t));
  d.cand_count = o; o += al256(S * sizeof(int32_t));
  d.kept_key = o; o += al256(S * R * sizeof(uint64_t));
  d.keep_count = o; o += al256(S * sizeof(int32_t));
  d.sm_stats = o; o += al256((size_t)batch * R * 2 * sizeof(double));
  d.total = o;
  return d;
}
}  // namespace dtc

DTC_API size_t dtc_postprocess_detections_workspace_bytes(int batch, int max_rois, int n_cls) {
  if (batch < 1 || max_rois < 1 || n_cls < 2) return 0;
  return dtc::det_plan(batch, max_rois, n_cls).total;
}

static int postprocess_detections_impl(const float* rois5, const int32_t* n_rois, const float* cls_score, int scores_are_logits,
                                       const float* bbox_pred, const float* decoded_boxes, const float* scaling_factor, const float* im_size,
                                       int batch, int max_rois, int n_cls, float wx, float wy, float ww, float wh,
                                       float score_thresh, float nms_thresh, int max_det, void* workspace,
                                       size_t workspace_bytes, float* dets, int32_t* det_roi, float* det_rois_scaled,
                                       int32_t* det_count, int max_out, dtc_stream_t stream, const dtc_fpn_map_out* fpn = nullptr) {
  if (batch < 0 || max_rois < 1 || n_cls < 2 || n_cls - 1 > dtc::kFinMaxCls || max_out < 1) return DTC_EINVAL;
  if (fpn) {
    if (!det_rois_scaled || !fpn->rois5 || !fpn->roi_levels || !fpn->n_out || !fpn->rois_by_level || !fpn->level_counts || !fpn->idx_restore ||
        fpn->k_max < fpn->k_min || fpn->k_max - fpn->k_min + 1 > 8)
      return DTC_EINVAL;
    if (max_out > dtc::kFpnMapMaxRows) return DTC_EUNSUPPORTED;      // longer lists: dtc_fpn_collect_distribute on det_rois_scaled
  }
  if (batch == 0) return DTC_OK;
  if (max_rois > 4096) return DTC_EUNSUPPORTED;
  if (!cls_score || !workspace || !dets || !det_roi || !det_count) return DTC_EINVAL;
  if (!decoded_boxes && (!rois5 || !bbox_pred || !scaling_factor || !im_size)) return DTC_EINVAL;
  if (decoded_boxes && det_rois_scaled) return DTC_EINVAL;
  const dtc::DetPlan pl = dtc::det_plan(batch, max_rois, n_cls);
  if (workspace_bytes < pl.total) return DTC_EWORKSPACE;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dtc::DetParams p;
  p.rois5 = rois5; p.n_rois = n_rois; p.cls_score = cls_score; p.bbox_pred = bbox_pred; p.scale = scaling_factor;
  p.decoded = decoded_boxes;
  p.sm_stats = nullptr;
  if (scores_are_logits) {
    double* st = reinterpret_cast<double*>(w + pl.sm_stats);
    const int rows = batch * max_rois;
    hipLaunchKernelGGL(dtc::det_softmax_stats_kernel, dim3((rows + dtc::kDetThreads / 64 - 1) / (dtc::kDetThreads / 64)),
                       dim3(dtc::kDetThreads), 0, s, cls_score, rows, n_cls, st);
    DTC_CHECK_LAUNCH();
    p.sm_stats = st;
  }
  p.im_size = im_size; p.R = max_rois; p.n_cls = n_cls; p.wx = wx; p.wy = wy; p.ww = ww; p.wh = wh;
  p.score_thresh = score_thresh;
  p.sorted_boxes = reinterpret_cast<float*>(w + pl.sorted_boxes);
  p.q_boxes = reinterpret_cast<float*>(w + pl.q_boxes); p.q_scores = reinterpret_cast<float*>(w + pl.q_scores);
  p.q_roi = reinterpret_cast<int32_t*>(w + pl.q_roi); p.cand_count = reinterpret_cast<int32_t*>(w + pl.cand_count);
  // dynamic LDS: sort keys [next_pow2(R)] x 8 B, then up to kNmsLdsCap boxes in score order (16 B) and one removed-bit per candidate
  const int np2 = dtc::next_pow2(max_rois);
  const size_t smem = (size_t)np2 * sizeof(uint64_t) + (size_t)dtc::kNmsLdsCap * sizeof(float4) +
                      (size_t)((max_rois + 63) / 64) * sizeof(uint64_t);
  if (smem > 48 * 1024) { DTC_RAISE_LDS_ONCE(dtc::det_candidates_kernel, 152 * 1024); }
  uint64_t* kept_key = reinterpret_cast<uint64_t*>(w + pl.kept_key);
  int32_t* keep_count = reinterpret_cast<int32_t*>(w + pl.keep_count);
  p.kept_key = kept_key; p.keep_count = keep_count; p.nms_thresh = nms_thresh; p.np2_max = np2;
  hipLaunchKernelGGL(dtc::det_candidates_kernel, dim3(n_cls - 1, batch), dim3(dtc::kDetThreads), smem, s, p);
  DTC_CHECK_LAUNCH();
  dtc::FinParams f;
  f.kept_key = kept_key; f.keep_count = keep_count; f.q_boxes = p.q_boxes; f.q_scores = p.q_scores;
  f.q_roi = p.q_roi; f.scale = scaling_factor; f.R = max_rois; f.n_cls = n_cls; f.max_det = max_det; f.max_out = max_out;
  f.dets = dets; f.det_roi = det_roi; f.det_rois_scaled = det_rois_scaled; f.det_count = det_count;
  f.fm.on = fpn ? 1 : 0;
  if (fpn) {
    f.fm.rois5 = fpn->rois5; f.fm.roi_levels = fpn->roi_levels; f.fm.n_out = fpn->n_out; f.fm.rois_by_level = fpn->rois_by_level;
    f.fm.level_counts = fpn->level_counts; f.fm.idx_restore = fpn->idx_restore; f.fm.roi_order = fpn->roi_order;
    f.fm.roi_desc = fpn->roi_order ? fpn->roi_desc : nullptr; f.fm.k_min = fpn->k_min; f.fm.k_max = fpn->k_max;
    f.fm.band_log2 = 4;                                               // as dtc_fpn_collect_distribute (fpn.hip)
  }
  long long cap = (long long)max_rois * (n_cls - 1);            // every candidate of every class kept
  if (cap > dtc::kFinStageMax) cap = dtc::kFinStageMax;
  const int stage_cap = (int)((cap + 3) & ~3ll);
  const size_t fsm = (size_t)stage_cap * 8;
  if (fsm > 16 * 1024) { DTC_RAISE_LDS_ONCE(dtc::det_finalize_kernel, 116 * 1024); }      // + ~37 KB static: under the 160 KB of a CU
  hipLaunchKernelGGL(dtc::det_finalize_kernel, dim3(batch), dim3(dtc::kFinThreads), fsm, s, f, stage_cap);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

DTC_API int dtc_postprocess_detections(const float* rois5, const int32_t* n_rois, const float* cls_score,
                                       const float* bbox_pred, const float* scaling_factor, const float* im_size,
                                       int batch, int max_rois, int n_cls, float wx, float wy, float ww, float wh,
                                       float score_thresh, float nms_thresh, int max_det, void* workspace,
                                       size_t workspace_bytes, float* dets, int32_t* det_roi, float* det_rois_scaled,
                                       int32_t* det_count, int max_out, dtc_stream_t stream) {
  return postprocess_detections_impl(rois5, n_rois, cls_score, 0, bbox_pred, nullptr, scaling_factor, im_size, batch, max_rois, n_cls, wx,
                                     wy, ww, wh, score_thresh, nms_thresh, max_det, workspace, workspace_bytes, dets, det_roi,
                                     det_rois_scaled, det_count, max_out, stream);
}

DTC_API int dtc_postprocess_detections_logits(const float* rois5, const int32_t* n_rois, const float* cls_logits,
                                              const float* bbox_pred, const float* scaling_factor, const float* im_size,
                                              int batch, int max_rois, int n_cls, float wx, float wy, float ww, float wh,
                                              float score_thresh, float nms_thresh, int max_det, void* workspace,
                                              size_t workspace_bytes, float* dets, int32_t* det_roi, float* det_rois_scaled,
                                              int32_t* det_count, int max_out, dtc_stream_t stream) {
  return postprocess_detections_impl(rois5, n_rois, cls_logits, 1, bbox_pred, nullptr, scaling_factor, im_size, batch, max_rois, n_cls, wx,
                                     wy, ww, wh, score_thresh, nms_thresh, max_det, workspace, workspace_bytes, dets, det_roi,
                                     det_rois_scaled, det_count, max_out, stream);
}

DTC_API int dtc_postprocess_detections_fpn(const float* rois5, const int32_t* n_rois, const float* cls_score, int scores_are_logits,
                                           const float* bbox_pred, const float* scaling_factor, const float* im_size,
                                           int batch, int max_rois, int n_cls, float wx, float wy, float ww, float wh,
                                           float score_thresh, float nms_thresh, int max_det, void* workspace,
                                           size_t workspace_bytes, float* dets, int32_t* det_roi, float* det_rois_scaled,
                                           int32_t* det_count, int max_out, const dtc_fpn_map_out* fpn, dtc_stream_t stream) {
  if (!fpn) return DTC_EINVAL;
  return postprocess_detections_impl(rois5, n_rois, cls_score, scores_are_logits ? 1 : 0, bbox_pred, nullptr, scaling_factor, im_size, batch,
                                     max_rois, n_cls, wx, wy, ww, wh, score_thresh, nms_thresh, max_det, workspace, workspace_bytes, dets,
                                     det_roi, det_rois_scaled, det_count, max_out, stream, fpn);
}

DTC_API int dtc_box_results_nms_limit(const float* scores, const float* boxes, const int32_t* n_rois, int batch, int max_rois,
                                      int n_cls, float score_thresh, float nms_thresh, int max_det, void* workspace,
                                      size_t workspace_bytes, float* dets, int32_t* det_roi, int32_t* det_count, int max_out,
                                      dtc_stream_t stream) {
  if (!boxes) return DTC_EINVAL;
  return postprocess_detections_impl(nullptr, n_rois, scores, 0, nullptr, boxes, nullptr, nullptr, batch, max_rois, n_cls, 1.f, 1.f,
                                     1.f, 1.f, score_thresh, nms_thresh, max_det, workspace, workspace_bytes, dets, det_roi,
                                     nullptr, det_count, max_out, stream);
}
