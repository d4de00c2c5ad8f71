// A7  FPN collect + distribute for gfx950 -- replaces
//   collect      lib/model/collect_and_distribute_fpn_rpn_proposals.py:84-105  (cat + torch.sort + top post_nms_topN)
//   distribute   :108-128 (D2H + numpy level mapping + np.where per level + argsort)
//   map_rois_to_fpn_levels   lib/utils/multilevel_rois.py:41-53 ; add_multilevel_rois_for_test :19-39 (mask branch)
// One workgroup per image, everything stays on the device.  Besides the reference's outputs (per-level roi lists +
// idx_restore) it emits what the multi-level RoIAlign kernel consumes directly: rois5 in collected order + a level id per
// RoI, so the pooled features come out already "restored" (no cat / index_select, lib/model/detector.py:266-270).
#include "block_sort.h"
#include <stdlib.h>

#include <mutex>

#include "dtc_common.h"
#include "fpn_map.h"

namespace dtc {
DTC_PT_TABLE(fpn)

constexpr int kFpnThreads = 1024;
constexpr int kFpnMaxLevels = 8;

struct FpnParams {
  const float* in_boxes;     // [B, L_in, P, 4]
  const float* in_scores;    // [B, L_in, P]     (NULL: no sort, take the first counts[b*L_in] rows of level 0 as they are)
  const int32_t* in_counts;  // [B, L_in]
  int L_in, P, top_n, k_min, k_max;
  int band_log2;             // visiting-order band height in feature rows (log2)
  int inputs_sorted;         // every input list is already in (score desc) order (NMS output): merge by rank, no sort
  // dtc_fpn_collect_distribute_kept (fast kernel only): in_boxes / in_scores are the SORTED pre-NMS arrays [B * L_in, k_stride, ...] and
  // list l of image b is their rows keep[(b * L_in + l) * P + j], j < in_counts -- proposals[keep] of generate_proposals.py:119-120
  // read in place instead of through a gather launch
  const int32_t* keep;
  int k_stride;
  float* rois5;              // [B, top_n, 5]   (b, x1, y1, x2, y2) in collected (score) order
  float* roi_scores;         // [B, top_n]      (may be NULL)
  int32_t* roi_levels;       // [B, top_n]      level - k_min, or -1 for rows >= n_out[b]
  int32_t* n_out;            // [B]
  float* rois_by_level;      // [B, top_n, 4]   rows grouped by level (the reference's distr_rois, concatenated)
  int32_t* level_counts;     // [B, k_max-k_min+1]
  int32_t* idx_restore;      // [B, top_n]      rois_by_level[idx_restore[r]] == roi r
  int32_t* roi_order;        // [B, top_n]      (nullable) global row ids b*top_n + r sorted by (level, y centre): the order in
                             //                 which RoIAlign should VISIT the rois (L2 locality); padding rows last
  float* roi_desc;           // [B, top_n, 8]   (nullable, needs roi_order) packed (b,x1,y1,x2,y2,level,global_row,0) in visiting order
};

__global__ __launch_bounds__(kFpnThreads) void fpn_collect_distribute_kernel(FpnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  __shared__ int lvl_off[kFpnMaxLevels + 1];
  __shared__ int in_off[kFpnMaxLevels + 1];
  __shared__ int wave_cnt[kFpnMaxLevels][kFpnThreads / 64];
  __shared__ int lvl_run[kFpnMaxLevels];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nl_out = p.k_max - p.k_min + 1;
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < p.L_in; l++) { in_off[l] = acc; acc += min(p.in_counts[b * p.L_in + l], p.P); }
    in_off[p.L_in] = acc;
  }
  if (tid < kFpnMaxLevels) lvl_run[tid] = 0;
  __syncthreads();
  const int n = in_off[p.L_in];
  const int m = min(n, p.top_n);                                   // :104
  const float* boxes = p.in_boxes + (size_t)b * p.L_in * p.P * 4;
  int np2 = 2;
  if (p.in_scores && p.inputs_sorted) {
    // L_in sorted lists -> rank of every element in the merged (score desc, concat index asc) order by binary search:
    // rank(l, j) = j + sum over the other lists of #elements that precede it.  No barriers, no sort.
    const float* scores = p.in_scores + (size_t)b * p.L_in * p.P;
    float* sc = reinterpret_cast<float*>(keys + p.top_n);            // staged scores, concat layout
    for (int i = tid; i < n; i += kFpnThreads) {
      int l = 0;
      for (int q = 1; q < p.L_in; q++) if (i >= in_off[q]) l = q;
      sc[i] = scores[(size_t)l * p.P + (i - in_off[l])];
    }
    __syncthreads();
    for (int i = tid; i < n; i += kFpnThreads) {
      int l = 0;
      for (int q = 1; q < p.L_in; q++) if (i >= in_off[q]) l = q;
      const float s = sc[i];
      int rank = i - in_off[l];
      for (int q = 0; q < p.L_in; q++) {
        if (q == l) continue;
        // list q is non-increasing: count elements with score > s (q after l) or >= s (q before l: ties go to the
        // smaller concat index)
        int lo = in_off[q], hi = in_off[q + 1];
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          const float v = sc[mid];
          const bool before = (q < l) ? (v >= s) : (v > s);
          if (before) lo = mid + 1; else hi = mid;
        }
        rank += lo - in_off[q];
      }
      if (rank < p.top_n) keys[rank] = make_desc_key(s, (uint32_t)i);
    }
    __syncthreads();
  } else if (p.in_scores) {
    const float* scores = p.in_scores + (size_t)b * p.L_in * p.P;
    np2 = next_pow2(n);
    for (int i = tid; i < np2; i += kFpnThreads) {
      uint64_t k = kPadKey;
      if (i < n) {
        int l = 0;
        for (int q = 1; q < p.L_in; q++) if (i >= in_off[q]) l = q;
        const int j = i - in_off[l];
        // key index = position in the concatenation (:95-97): ties resolve to the earlier level / earlier row.
        k = make_desc_key(scores[(size_t)l * p.P + j], (uint32_t)i);
      }
      keys[i] = k;
    }
    block_bitonic_sort<kFpnThreads>(keys, np2);                     // :102
  }
  // ranks [0, m): roi, level, distribution
  for (int r0 = 0; r0 < p.top_n; r0 += kFpnThreads) {
    const int r = r0 + tid;
    int lvl = -1;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.f;
    if (r < m) {
      int src = r;
      if (p.in_scores) src = (int)desc_key_index(keys[r]);
      int l = 0;
      for (int q = 1; q < p.L_in; q++) if (src >= in_off[q]) l = q;
      const int j = src - in_off[l];
      bx = reinterpret_cast<const float4*>(boxes)[(size_t)l * p.P + j];
      if (p.in_scores) sc = p.in_scores[((size_t)b * p.L_in + l) * p.P + j];
      lvl = fpn_level(bx.x, bx.y, bx.z, bx.w, p.k_min, p.k_max) - p.k_min;
    }
    if (r < p.top_n) {
      float* o = p.rois5 + ((size_t)b * p.top_n + r) * 5;
      o[0] = (float)b; o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
      p.roi_levels[(size_t)b * p.top_n + r] = lvl;
      if (p.roi_scores) p.roi_scores[(size_t)b * p.top_n + r] = sc;
    }
    // position of r inside its level, in rank order (np.where(lvls == lvl)[0] is ascending, :123)
    int my_before = 0;
    for (int l = 0; l < nl_out; l++) {
      const uint64_t mk = __ballot(lvl == l);
      if (lane == 0) wave_cnt[l][wv] = __builtin_popcountll(mk);
      if (lvl == l) my_before = __builtin_popcountll(mk & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    int pos_in_level = -1;
    if (lvl >= 0) {
      int base = lvl_run[lvl];
      for (int q = 0; q < wv; q++) base += wave_cnt[lvl][q];
      pos_in_level = base + my_before;
    }
    __syncthreads();
    if (tid < nl_out) { int t = 0; for (int q = 0; q < kFpnThreads / 64; q++) t += wave_cnt[tid][q]; lvl_run[tid] += t; }
    // stash (level, pos) for the second sweep in the key array's upper part is not possible (keys still needed), so
    // write pos_in_level to idx_restore now and add the level offset below.
    if (r < p.top_n) p.idx_restore[(size_t)b * p.top_n + r] = pos_in_level;
    __syncthreads();
  }
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < nl_out; l++) { lvl_off[l] = acc; acc += lvl_run[l]; p.level_counts[b * nl_out + l] = lvl_run[l]; }
    lvl_off[nl_out] = acc;
    p.n_out[b] = m;
  }
  __syncthreads();
  for (int r = tid; r < m; r += kFpnThreads) {
    const int lvl = p.roi_levels[(size_t)b * p.top_n + r];
    const int dst = lvl_off[lvl] + p.idx_restore[(size_t)b * p.top_n + r];
    p.idx_restore[(size_t)b * p.top_n + r] = dst;                  // :127 argsort(concat(idx_lvl)) == inverse permutation
    const float* o = p.rois5 + ((size_t)b * p.top_n + r) * 5;
    reinterpret_cast<float4*>(p.rois_by_level)[(size_t)b * p.top_n + dst] = make_float4(o[1], o[2], o[3], o[4]);
  }
  if (p.roi_order) {
    // visiting order for RoIAlign.  Purely a performance hint; any permutation is correct.
    __syncthreads();
    const int np2o = max(4, next_pow2(p.top_n));     // the counting rank below reads the key table four entries at a time
    for (int r = tid; r < np2o; r += kFpnThreads) {
      uint64_t k = kPadKey;
      if (r < p.top_n) {
        const int lvl = p.roi_levels[(size_t)b * p.top_n + r];
        const float* o = p.rois5 + ((size_t)b * p.top_n + r) * 5;
        // Locality code of a RoI: (level | band of 2^band_log2 feature rows | x centre in feature pixels).  Neighbours in this
        // order overlap in BOTH directions: K consecutive RoIs form a compact patch -- what the cluster-stationary RoIAlign
        // kernel merges into one staged window -- and patches that follow each other share columns that are still in the
        // XCD's L2.  (Interleaving two or four vertically adjacent bands per 32..256-pixel x cell, so that the rows bands share
        // are re-read sooner, was measured in round 2: 0.404-0.441 ms per box-head launch against 0.391-0.396 for the plain
        // order -- the patches merge less well -- and was dropped.)
        const uint32_t yc = (uint32_t)fminf(fmaxf((o[2] + o[4]) * 0.5f, 0.f), 65535.f);
        const uint32_t xc = (uint32_t)fminf(fmaxf((o[1] + o[3]) * 0.5f, 0.f), 65535.f);
        const uint32_t lv4 = lvl < 0 ? 15u : (uint32_t)lvl;
        const uint32_t fs = min((uint32_t)p.k_min + lv4, 15u);                  // log2 feature stride
        const uint32_t band = min((yc >> fs) >> p.band_log2, 63u), xf = min(xc >> fs, 4095u);
        const uint32_t loc = (band << 12) | xf;                                   // 18 bits
        k = ((uint64_t)lv4 << 52) | ((uint64_t)loc << 20) | (uint32_t)r;
      }
      keys[r] = k;
    }
    if (p.top_n <= 2048) {
      // rank by counting (keys are unique; n^2 compares, no barriers: beats 66 bitonic stages).  The order is only a
      // locality hint, so the 64-bit key is squeezed into 32 bits -- level:3 | locality code:18 | rank:11 -- and each
      // thread walks the table with 16-byte broadcast LDS reads (4 keys per ds_read_b128): 4x fewer LDS instructions than
      // one 8-byte read per compare, which is what bounded this phase (16 waves x 1000 reads on one CU = ~50 us).
      __syncthreads();
      uint32_t* k32 = reinterpret_cast<uint32_t*>(keys + np2o);
      uint32_t* sorted32 = k32 + np2o;
      for (int r = tid; r < np2o; r += kFpnThreads) {
        const uint64_t k = keys[r];
        uint32_t c = 0xffffffffu;
        if (r < p.top_n) {
          const uint32_t lv4 = (uint32_t)(k >> 52) & 0xfu, loc = (uint32_t)(k >> 20) & 0x3ffffu;
          c = (min(lv4, 7u) << 29) | (loc << 11) | (uint32_t)r;
        }
        k32[r] = c;
      }
      __syncthreads();
      const int n4 = (p.top_n + 3) >> 2;                       // entries past top_n are 0xffffffff: never smaller
      for (int r = tid; r < p.top_n; r += kFpnThreads) {
        const uint32_t k = k32[r];
        int rank = 0;
        for (int j = 0; j < n4; j++) {
          const uint4 q = reinterpret_cast<const uint4*>(k32)[j];
          rank += (q.x < k ? 1 : 0) + (q.y < k ? 1 : 0) + (q.z < k ? 1 : 0) + (q.w < k ? 1 : 0);
        }
        sorted32[rank] = k;
      }
      __syncthreads();
      for (int r = tid; r < p.top_n; r += kFpnThreads) keys[r] = (uint64_t)(sorted32[r] & 0x7ffu);
      __syncthreads();
    } else {
      block_bitonic_sort<kFpnThreads>(keys, np2o);
    }
    for (int i = tid; i < p.top_n; i += kFpnThreads) {
      const int r = (int)(keys[i] & 0xfffffu);
      p.roi_order[(size_t)b * p.top_n + i] = b * p.top_n + r;
      if (p.roi_desc) {
        const float* o = p.rois5 + ((size_t)b * p.top_n + r) * 5;
        float4* d = reinterpret_cast<float4*>(p.roi_desc + ((size_t)b * p.top_n + i) * 8);
        d[0] = make_float4(o[0], o[1], o[2], o[3]);
        d[1] = make_float4(o[4], (float)p.roi_levels[(size_t)b * p.top_n + r], (float)(b * p.top_n + r), 0.f);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast path: top_n <= 2048, every input list already in score order (the NMS output) or no scores at all (mask branch).
// Thread t OWNS output ranks t (and t + 1024 when top_n > 1024): box, score, level and position stay in its registers from the merge to the last store -- the
// general kernel above round-trips them through global memory five times (~3 us each, one workgroup per image, nothing
// else to hide it behind) -- the list merge runs its binary searches in lock-step (all (element, other list) pairs of a
// thread advance together: ten dependent LDS reads in total instead of two hundred), and the visiting order is written by
// the owner straight to its sorted slot.  Measured on MI355X (batch 8, 5 x 1000 -> 1000): 51 -> see profiles/r02_*.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFastMaxTop = 2048;      // R = 1: top_n <= 1024 (one output rank per thread), R = 2: <= 2048 (BASELINE cfg5's 2000)
constexpr int kOrderBuckets = 512;     // visiting-order key >> 23 = (level:3 | band:6)
constexpr int fast_hdr_bytes(int R) { return 2 * R * kFpnThreads * 4 + 2 * kOrderBuckets * 4; }

template <int R>   // output ranks per thread: thread t owns ranks rr * kFpnThreads + t, rr < R
__global__ __launch_bounds__(kFpnThreads) void fpn_collect_fast_kernel(FpnParams p, int n_max) {
  constexpr int kTop = R * kFpnThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* k32 = reinterpret_cast<uint32_t*>(smem);                       // [kTop]  order keys, bucket by bucket (later)
  int* src_of_rank = reinterpret_cast<int*>(smem) + kTop;                    // [kTop]
  uint32_t* bstart = reinterpret_cast<uint32_t*>(smem) + 2 * kTop;           // [512]   visiting order: bucket histogram -> start
  uint32_t* bcur = bstart + kOrderBuckets;                                   // [512]   members placed so far
  uint64_t* kbuf = reinterpret_cast<uint64_t*>(smem + fast_hdr_bytes(R));    // [n_max + (L_in - 1) * top_n] input keys, then merge outputs
  int* kk_s = reinterpret_cast<int*>(kbuf + n_max + (p.L_in - 1) * p.top_n);   // [n_max] (keep form only) row of concat element i in its sorted segment
  __shared__ uint32_t bwsum[kOrderBuckets / 64];
  __shared__ int cnt_s[kFpnMaxLevels];
  __shared__ int l_off[kFpnMaxLevels], l_len[kFpnMaxLevels];
  __shared__ int wave_cnt[kFpnMaxLevels][R * kFpnThreads / 64];              // per (level, rank-wave): rank-wave = rr * 16 + wave
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nl_out = p.k_max - p.k_min + 1;
  DTC_PT(p.L_in > 1 ? 0 : 1, b, 0);
  if (tid < p.L_in) cnt_s[tid] = min(p.in_counts[b * p.L_in + tid], p.P);
  // Round 6: the scores of EVERY row of every input list (n_max = L_in * P <= 8192: up to eight per thread) are requested before the
  // lists' counts are known -- one global round trip in front of the merge instead of two dependent ones; rows past a count hold
  // whatever the producer left there and are dropped when the keys are filed.  With `keep` a row is keep -> score: the index is
  // clamped into the segment, so a stale index past the count cannot leave the array.
  constexpr int kSpec = 8;
  float sp_s[kSpec];
  int sp_k[kSpec];
  const bool spec = p.in_scores != nullptr && p.L_in > 1;
  if (spec) {
    // straight-line code (indices clamped, no branch per element): all index loads leave together, then all score loads
    size_t row[kSpec];
    const float rP = 1.0f / (float)p.P;
#pragma unroll
    for (int u = 0; u < kSpec; u++) {
      const int idx = min(tid + u * kFpnThreads, n_max - 1);
      const int l = (int)(((float)idx + 0.5f) * rP);                          // idx / P, exact: idx < 2^13, >= 0.5 / P from an integer
      row[u] = ((size_t)b * p.L_in + l) * p.P + (idx - l * p.P);
      sp_k[u] = 0;
      if (p.keep) sp_k[u] = p.keep[row[u]];                                   // (uniform condition)
    }
#pragma unroll
    for (int u = 0; u < kSpec; u++) {
      if (p.keep) {
        const int idx = min(tid + u * kFpnThreads, n_max - 1);
        const int l = (int)(((float)idx + 0.5f) * rP);
        sp_k[u] = min(max(sp_k[u], 0), p.k_stride - 1);
        row[u] = ((size_t)b * p.L_in + l) * p.k_stride + sp_k[u];
      }
      sp_s[u] = p.in_scores[row[u]];
    }
  }
  __syncthreads();
  int in_off[kFpnMaxLevels + 1];
  in_off[0] = 0;
#pragma unroll
  for (int l = 0; l < kFpnMaxLevels; l++) in_off[l + 1] = in_off[l] + (l < p.L_in ? cnt_s[l] : 0);
  const int n = in_off[kFpnMaxLevels];
  const int m = min(n, p.top_n);                                             // :104
  const float* boxes = p.in_boxes + (size_t)b * p.L_in * p.P * 4;
  const bool merge = p.in_scores != nullptr && p.L_in > 1;
  if (merge) {
    // The input lists are sorted (NMS output): collect's cat + torch.sort + [:post_nms_topN] (:95-104) is a k-way MERGE
    // truncated to top_n.  Pairwise merge tree, every output found independently by a merge-path search on its diagonal
    // (<= 10 steps of two 8-byte LDS reads): round 1 merges (L0,L1) (L2,L3) ..., round 2 the results, ...  Keys are
    // (score desc, concat index asc) -- unique, so ties need no special case: the earlier level / earlier row wins (:95-97).
#pragma unroll
    for (int u = 0; u < kSpec; u++) {                                          // the speculative loads above, filed by concat index
      const int idx = tid + u * kFpnThreads;
      if (idx < n_max) {
        const int l = (int)(((float)idx + 0.5f) * (1.0f / (float)p.P)), j = idx - l * p.P;
        int off = 0, cnt = 0;
#pragma unroll
        for (int q = 0; q < kFpnMaxLevels; q++) if (q == l) { off = in_off[q]; cnt = in_off[q + 1] - in_off[q]; }
        if (j < cnt) {
          kbuf[off + j] = make_desc_key(sp_s[u], (uint32_t)(off + j));
          if (p.keep) kk_s[off + j] = sp_k[u];
        }
      }
    }
    if (tid < p.L_in) { l_off[tid] = in_off[tid]; l_len[tid] = cnt_s[tid]; }
    __syncthreads();
    int scratch = n_max;                                                      // next free slot of kbuf for merge outputs
    for (int nl = p.L_in; nl > 1; nl = (nl + 1) >> 1) {
      const int npairs = nl >> 1;
      for (int w = tid; w < npairs * p.top_n; w += kFpnThreads) {
        const int pr = w / p.top_n, k = w - pr * p.top_n;
        const int oa = l_off[2 * pr], na = l_len[2 * pr], ob = l_off[2 * pr + 1], nb = l_len[2 * pr + 1];
        if (k < min(na + nb, p.top_n)) {
          const uint64_t* A = kbuf + oa;
          const uint64_t* B = kbuf + ob;
          int lo = max(0, k - nb), hi = min(k, na);                           // i = #elements taken from A among the first k
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (A[mid] < B[k - mid - 1]) lo = mid + 1; else hi = mid;
          }
          const int i = lo, j = k - lo;
          uint64_t out;
          if (j >= nb) out = A[i];
          else if (i >= na) out = B[j];
          else { const uint64_t x = A[i], y = B[j]; out = x < y ? x : y; }
          kbuf[scratch + pr * p.top_n + k] = out;
        }
      }
      __syncthreads();
      // next round's descriptors: read the old ones into registers first, publish after a barrier
      int noff = 0, nlen = 0, coff = 0, clen = 0;
      if (tid < npairs) { nlen = min(l_len[2 * tid] + l_len[2 * tid + 1], p.top_n); noff = scratch + tid * p.top_n; }
      if ((nl & 1) && tid == 0) { coff = l_off[nl - 1]; clen = l_len[nl - 1]; }              // odd list carried over
      __syncthreads();
      if (tid < npairs) { l_off[tid] = noff; l_len[tid] = nlen; }
      if ((nl & 1) && tid == 0) { l_off[npairs] = coff; l_len[npairs] = clen; }
      __syncthreads();
      scratch += npairs * p.top_n;
    }
    for (int r = tid; r < m; r += kFpnThreads) src_of_rank[r] = (int)desc_key_index(kbuf[l_off[0] + r]);
    __syncthreads();
  }
  DTC_PT(p.L_in > 1 ? 0 : 1, b, 1);
  // ---- rank r: roi, level, position inside its level -- all in registers from here on ---------------------------------
  constexpr int kRW = kFpnThreads / 64;                                        // waves per round of ranks
  int lvl[R], my_before[R];
  float4 bx[R];
  float score[R];
  uint32_t key[R];
#pragma unroll
  for (int rr = 0; rr < R; rr++) {
    const int r = rr * kFpnThreads + tid;
    lvl[rr] = -1; my_before[rr] = 0;
    bx[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    score[rr] = 0.f;
    if (r < m) {
      const int src = merge ? src_of_rank[r] : r;
      int l = 0;
#pragma unroll
      for (int q = 1; q < kFpnMaxLevels; q++) if (q < p.L_in && src >= in_off[q]) l = q;
      if (p.keep) {
        const size_t row = ((size_t)b * p.L_in + l) * p.k_stride + kk_s[src];
        bx[rr] = reinterpret_cast<const float4*>(p.in_boxes)[row];
        score[rr] = p.in_scores[row];
      } else {
        bx[rr] = reinterpret_cast<const float4*>(boxes)[(size_t)l * p.P + (src - in_off[l])];
        if (p.in_scores) score[rr] = p.in_scores[((size_t)b * p.L_in + l) * p.P + (src - in_off[l])];
      }
      lvl[rr] = fpn_level(bx[rr].x, bx[rr].y, bx[rr].z, bx[rr].w, p.k_min, p.k_max) - p.k_min;
    }
  }
#pragma unroll
  for (int rr = 0; rr < R; rr++) {
    for (int l = 0; l < nl_out; l++) {                                         // np.where(lvls == lvl)[0] is ascending (:123)
      const uint64_t mk = __ballot(lvl[rr] == l);
      if (lane == 0) wave_cnt[l][rr * kRW + wv] = __builtin_popcountll(mk);
      if (lvl[rr] == l) my_before[rr] = __builtin_popcountll(mk & ((1ull << lane) - 1ull));
    }
    // order key for the RoIAlign visiting order (see the general kernel): level | band | x, all in feature pixels
    const int r = rr * kFpnThreads + tid;
    key[rr] = 0xffffffffu;
    if (r < p.top_n) key[rr] = fpn_order_key(bx[rr], lvl[rr], p.k_min, p.band_log2, r);
  }
  __syncthreads();                                                             // src_of_rank / sc reads done; wave_cnt complete
  if (tid < kOrderBuckets) { bstart[tid] = 0u; bcur[tid] = 0u; }
  int lvl_tot[kFpnMaxLevels], dst[R];
#pragma unroll
  for (int rr = 0; rr < R; rr++) dst[rr] = -1;
  {
    // ranks ascend with the rank-wave index g = rr * kRW + wave, then with the lane: members of level l in front of mine =
    // all of the rank-waves before mine + the lanes before me; levels are laid out one after the other (:123-127)
    int acc = 0;
    for (int l = 0; l < nl_out; l++) {
      int t = 0, before[R];
#pragma unroll
      for (int rr = 0; rr < R; rr++) before[rr] = 0;
      for (int g = 0; g < R * kRW; g++) {
        const int c = wave_cnt[l][g];
#pragma unroll
        for (int rr = 0; rr < R; rr++) if (g < rr * kRW + wv) before[rr] += c;
        t += c;
      }
      lvl_tot[l] = t;
#pragma unroll
      for (int rr = 0; rr < R; rr++) if (l == lvl[rr]) dst[rr] = acc + before[rr] + my_before[rr];   // argsort(concat(idx_lvl)) == inverse permutation
      acc += t;
    }
  }
#pragma unroll
  for (int rr = 0; rr < R; rr++) {
    const int r = rr * kFpnThreads + tid;
    if (r < p.top_n) {
      const size_t g = (size_t)b * p.top_n + r;
      float* o = p.rois5 + g * 5;
      o[0] = (float)b; o[1] = bx[rr].x; o[2] = bx[rr].y; o[3] = bx[rr].z; o[4] = bx[rr].w;
      p.roi_levels[g] = lvl[rr];
      if (p.roi_scores) p.roi_scores[g] = score[rr];
      p.idx_restore[g] = dst[rr];
      if (dst[rr] >= 0) reinterpret_cast<float4*>(p.rois_by_level)[(size_t)b * p.top_n + dst[rr]] = bx[rr];
    }
  }
  if (tid < nl_out) p.level_counts[b * nl_out + tid] = lvl_tot[tid];
  if (tid == 0) p.n_out[b] = m;
  if (!p.roi_order) return;
  __syncthreads();
  DTC_PT(p.L_in > 1 ? 0 : 1, b, 2);
  // Rank of this RoI's key among all keys (they are unique).  Round 2 counted `key' < key` over ALL top_n keys per thread
  // (250 broadcast ds_read_b128 + 1000 compares each: the longest phase of the kernel); the key's top 9 bits (level | band)
  // name one of a few dozen populated buckets of ~75 RoIs, so: histogram -> exclusive scan -> members placed bucket by
  // bucket (any order) -> each thread counts the smaller keys of ITS bucket only.  rank = bucket start + that count.
#pragma unroll
  for (int rr = 0; rr < R; rr++)
    if (rr * kFpnThreads + tid < p.top_n) atomicAdd(&bstart[key[rr] >> 23], 1u);
  __syncthreads();
  uint32_t hv = 0, incl = 0;
  if (tid < kOrderBuckets) {
    hv = bstart[tid]; incl = hv;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) bwsum[wv] = incl;
  }
  __syncthreads();
  if (tid < kOrderBuckets) {
    uint32_t base = 0;
    for (int q = 0; q < wv; q++) base += bwsum[q];
    bstart[tid] = base + incl - hv;
  }
  __syncthreads();
  uint32_t start[R];
#pragma unroll
  for (int rr = 0; rr < R; rr++) {
    start[rr] = 0;
    if (rr * kFpnThreads + tid < p.top_n) { const uint32_t bk = key[rr] >> 23; start[rr] = bstart[bk]; k32[start[rr] + atomicAdd(&bcur[bk], 1u)] = key[rr]; }
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < R; rr++) {
    const int r = rr * kFpnThreads + tid;
    if (r < p.top_n) {
      const int cnt = (int)bcur[key[rr] >> 23];
      int rank = (int)start[rr];
      for (int j = 0; j < cnt; j++) rank += k32[start[rr] + j] < key[rr] ? 1 : 0;
      const size_t g = (size_t)b * p.top_n + rank;
      p.roi_order[g] = b * p.top_n + r;
      if (p.roi_desc) {
        float4* d = reinterpret_cast<float4*>(p.roi_desc + g * 8);
        d[0] = make_float4((float)b, bx[rr].x, bx[rr].y, bx[rr].z);
        d[1] = make_float4(bx[rr].w, (float)lvl[rr], (float)(b * p.top_n + r), 0.f);
      }
    }
  }
  DTC_PT(p.L_in > 1 ? 0 : 1, b, 3);
}

}  // namespace dtc

static int fpn_collect_launch(dtc::FpnParams p, int batch, long long n_max, dtc_stream_t stream) {
  const int n_in_levels = p.L_in, post_nms_top_n = p.top_n, in_stride = p.P;
  const float* in_scores = p.in_scores;
  const int inputs_sorted = p.inputs_sorted;
  size_t smem = in_scores ? (size_t)dtc::next_pow2((int)n_max) * sizeof(uint64_t) : 16;
  if (in_scores && inputs_sorted) smem = (size_t)post_nms_top_n * sizeof(uint64_t) + (size_t)n_max * sizeof(float) + 16;
  if (p.roi_order) { const size_t so = (size_t)(dtc::next_pow2(post_nms_top_n) < 4 ? 4 : dtc::next_pow2(post_nms_top_n)) * sizeof(uint64_t) * 2; if (so > smem) smem = so; }
  if (post_nms_top_n > 16384) return DTC_EUNSUPPORTED;
  static const bool no_fast = getenv("DTC_FPN_NO_FAST") != nullptr;     // A/B knob (general kernel), resolved once
  const bool fast = post_nms_top_n <= dtc::kFastMaxTop && in_stride <= 1024 && n_max <= 8192 &&
                    (!in_scores || inputs_sorted) && (!no_fast || p.keep);
  if (fast) {
    // k32 + src_of_rank + order buckets + input keys (n_max) + merge outputs (one top_n-strided list per pairwise merge:
    // L - 1 of them): 8 B each; the keep form adds one int per input row
    const int R = post_nms_top_n <= dtc::kFpnThreads ? 1 : 2;
    const size_t fsm = (size_t)dtc::fast_hdr_bytes(R) +
                       (size_t)(in_scores && n_in_levels > 1 ? n_max + (long long)(n_in_levels - 1) * post_nms_top_n : 0) * 8 +
                       (p.keep ? (size_t)n_max * 4 : 0) + 16;
    if (fsm <= 150 * 1024) {
      if (R == 1) {
        DTC_RAISE_LDS_ONCE(dtc::fpn_collect_fast_kernel<1>, 152 * 1024);
        hipLaunchKernelGGL(dtc::fpn_collect_fast_kernel<1>, dim3(batch), dim3(dtc::kFpnThreads), fsm, reinterpret_cast<hipStream_t>(stream), p, (int)n_max);
      } else {
        DTC_RAISE_LDS_ONCE(dtc::fpn_collect_fast_kernel<2>, 152 * 1024);
        hipLaunchKernelGGL(dtc::fpn_collect_fast_kernel<2>, dim3(batch), dim3(dtc::kFpnThreads), fsm, reinterpret_cast<hipStream_t>(stream), p, (int)n_max);
      }
      DTC_CHECK_LAUNCH();
      return DTC_OK;
    }
  }
  if (p.keep) return DTC_EUNSUPPORTED;         // the keep form exists in the fast kernel only: dtc_gather_kept + dtc_fpn_collect_distribute
  if (smem > 32 * 1024) {   // static __shared__ of the kernel comes on top: raise the limit well before dynamic + static reaches 64 KB
    DTC_RAISE_LDS_ONCE(dtc::fpn_collect_distribute_kernel, 144 * 1024);
  }
  hipLaunchKernelGGL(dtc::fpn_collect_distribute_kernel, dim3(batch), dim3(dtc::kFpnThreads), smem,
                     reinterpret_cast<hipStream_t>(stream), p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

DTC_API int dtc_fpn_collect_distribute(const float* in_boxes, const float* in_scores, const int32_t* in_counts, int batch,
                                       int n_in_levels, int in_stride, int post_nms_top_n, int k_min, int k_max,
                                       float* rois5, float* roi_scores, int32_t* roi_levels, int32_t* n_out,
                                       float* rois_by_level, int32_t* level_counts, int32_t* idx_restore,
                                       int32_t* roi_order, float* roi_desc, int inputs_sorted, dtc_stream_t stream) {
  if (batch < 0 || n_in_levels < 1 || n_in_levels > dtc::kFpnMaxLevels || in_stride < 1 || post_nms_top_n < 1 ||
      k_max < k_min || k_max - k_min + 1 > dtc::kFpnMaxLevels)
    return DTC_EINVAL;
  if (batch == 0) return DTC_OK;
  if (!in_boxes || !in_counts || !rois5 || !roi_levels || !n_out || !rois_by_level || !level_counts || !idx_restore)
    return DTC_EINVAL;
  const long long n_max = (long long)n_in_levels * in_stride;
  if (in_scores && n_max > 16384) return DTC_EUNSUPPORTED;
  dtc::FpnParams p;
  p.in_boxes = in_boxes; p.in_scores = in_scores; p.in_counts = in_counts; p.L_in = n_in_levels; p.P = in_stride;
  p.top_n = post_nms_top_n; p.k_min = k_min; p.k_max = k_max; p.inputs_sorted = inputs_sorted; p.rois5 = rois5; p.roi_scores = roi_scores;
  p.roi_levels = roi_levels; p.n_out = n_out; p.rois_by_level = rois_by_level; p.level_counts = level_counts;
  p.keep = nullptr; p.k_stride = 0;
  // visiting-order band height (log2 feature rows): 16 rows suits the cluster-stationary RoIAlign kernel (clusters of ~5
  // neighbours stay ~28 rows x 32 pixels; measured 8 rows 0.48, 16 rows 0.41, 32 rows 0.43 ms per 8000-RoI box-head launch)
  p.band_log2 = 4;
  p.idx_restore = idx_restore; p.roi_order = roi_order; p.roi_desc = roi_order ? roi_desc : nullptr;
  return fpn_collect_launch(p, batch, n_max, stream);
}

DTC_API int dtc_fpn_collect_distribute_kept(const float* sorted_boxes, const float* sorted_scores, int k_stride, const int32_t* keep,
                                            const int32_t* keep_count, int keep_stride, int batch, int n_in_levels,
                                            int post_nms_top_n, int k_min, int k_max, float* rois5, float* roi_scores,
                                            int32_t* roi_levels, int32_t* n_out, float* rois_by_level, int32_t* level_counts,
                                            int32_t* idx_restore, int32_t* roi_order, float* roi_desc, dtc_stream_t stream) {
  if (batch < 0 || n_in_levels < 2 || n_in_levels > dtc::kFpnMaxLevels || keep_stride < 1 || k_stride < 1 || post_nms_top_n < 1 ||
      k_max < k_min || k_max - k_min + 1 > dtc::kFpnMaxLevels)
    return n_in_levels == 1 ? DTC_EUNSUPPORTED : DTC_EINVAL;
  if (batch == 0) return DTC_OK;
  if (!sorted_boxes || !sorted_scores || !keep || !keep_count || !rois5 || !roi_levels || !n_out || !rois_by_level || !level_counts ||
      !idx_restore)
    return DTC_EINVAL;
  const long long n_max = (long long)n_in_levels * keep_stride;
  dtc::FpnParams p;
  p.in_boxes = sorted_boxes; p.in_scores = sorted_scores; p.in_counts = keep_count; p.L_in = n_in_levels; p.P = keep_stride;
  p.top_n = post_nms_top_n; p.k_min = k_min; p.k_max = k_max; p.inputs_sorted = 1; p.rois5 = rois5; p.roi_scores = roi_scores;
  p.roi_levels = roi_levels; p.n_out = n_out; p.rois_by_level = rois_by_level; p.level_counts = level_counts;
  p.keep = keep; p.k_stride = k_stride; p.band_log2 = 4;
  p.idx_restore = idx_restore; p.roi_order = roi_order; p.roi_desc = roi_order ? roi_desc : nullptr;
  return fpn_collect_launch(p, batch, n_max, stream);
}
