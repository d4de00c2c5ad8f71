// A5  Hard NMS for gfx950 -- replaces cython_nms.nms (lib/utils_cython/cython_nms.pyx:37-87; entry lib/utils/boxes.py:332).
//
// Segmented: one launch handles every (image, FPN level) or (image, class) segment of a batch.
//   1. segment_sort_desc   : per segment, (score desc, index asc) order through the shared 64-bit key (block_sort.h)  [:45]
//   2. nms_mask            : 64x64 tiles of the suppression matrix (only those on or above the diagonal are enumerated);
//                            lane <-> column box, the row boxes are broadcast from LDS, the 64-lane compare is collected
//                            with a ballot into one 64-bit word per row.  IoU in the reference's float32 operation order
//                            [:76-84]: inter / (iarea + areas[j] - inter) >= thresh -- decided from the sign of
//                            inter - thresh * union, with the IEEE division only inside a 2^-21 band (see the kernel).
//   3. nms_reduce          : one wavefront per segment walks the row blocks in score order; the in-block dependency chain
//                            is resolved as a wave-wide fixed point on the transposed diagonal tile, rows of kept boxes are
//                            OR-ed into an LDS-resident `removed` bit-vector.  nms_reduce_lds: the same walk from an LDS
//                            copy of the whole matrix, for few long segments (the RPN call).
// Output order: positions in score order (what `keep[:post_nms_top_n]` needs, generate_proposals.py:117); dtc_nms()
// additionally maps back to ascending original indices like np.where(suppressed == 0)[0]  [:87].
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "block_sort.h"
#include "dtc_common.h"

namespace dtc {

DTC_PT_TABLE(nms)

// ---------------------------------------------------------------------------------------------------------------------
// 1. segment sort.  scores [S, n_stride] (+counts) -> order [S, n_stride] int32 (original index of the k-th best),
//    optional gathered boxes [S, n_stride, 4] / scores.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSortThreads = 1024;

__global__ __launch_bounds__(kSortThreads) void segment_sort_desc_kernel(
    const float* __restrict__ scores, int score_stride_elems, const float* __restrict__ boxes, int box_stride_elems,
    const int32_t* __restrict__ counts, int n_stride, int32_t* __restrict__ order, float* __restrict__ sorted_boxes,
    float* __restrict__ sorted_scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  const int s = blockIdx.x;
  const int n = counts ? min(counts[s], n_stride) : n_stride;
  const int np2 = next_pow2(n);
  const float* sc = scores + (size_t)s * n_stride * score_stride_elems;
  for (int i = threadIdx.x; i < np2; i += kSortThreads)
    keys[i] = i < n ? make_desc_key(sc[(size_t)i * score_stride_elems], (uint32_t)i) : kPadKey;
  block_bitonic_sort<kSortThreads>(keys, np2);
  const float* bx = boxes ? boxes + (size_t)s * n_stride * box_stride_elems : nullptr;
  for (int i = threadIdx.x; i < n; i += kSortThreads) {
    const uint64_t k = keys[i];
    const uint32_t src = desc_key_index(k);
    if (order) order[(size_t)s * n_stride + i] = (int32_t)src;
    if (sorted_scores) sorted_scores[(size_t)s * n_stride + i] = sc[(size_t)src * score_stride_elems];
    if (sorted_boxes) {
      const float* b = bx + (size_t)src * box_stride_elems;
      float4 v = make_float4(b[0], b[1], b[2], b[3]);
      reinterpret_cast<float4*>(sorted_boxes)[(size_t)s * n_stride + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2. suppression-matrix tiles.  boxes [S, n_stride, 4] score-sorted; mask [S, n_stride, ncb_stride] u64, only tiles with
//    cb >= rb are written (and only those are read).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_area(float4 b) { return (b.z - b.x + 1.f) * (b.w - b.y + 1.f); }  // cython_nms.pyx:44

// v_max_f32 / v_min_f32 on operands that come straight from memory.  fmaxf() makes the compiler canonicalise such operands
// first (v_max x, x, x: signalling NaNs), one extra instruction per LDS-broadcast row coordinate and pair; the hardware
// instruction in IEEE mode already quiets them, and for everything that is not a signalling NaN the result is the same.
__device__ __forceinline__ float vmax(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmin(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

constexpr int kMaskWaves = 4;  // a workgroup = 4 wavefronts = 4 consecutive column blocks of one row block

__global__ __launch_bounds__(64 * kMaskWaves) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                                   const int32_t* __restrict__ counts, int n_stride, int n_cap,
                                                                   int ncb_stride, float thresh,
                                                                   uint64_t* __restrict__ mask,
                                                                   uint64_t* __restrict__ diag_t) {
  // grid = (GX, 1, n_seg): a workgroup walks the (row block, column-block group) tiles of ITS segment with stride GX.
  // The iteration space is derived from the segment's actual count, so the 600+ class segments of a detection batch
  // (mostly <= 64 candidates = one tile) cost one short workgroup each instead of a worst-case 16x4 tile grid whose
  // empty workgroups dominated the launch (measured 90 us -> see profiles/).
  __shared__ float4 rbox_s[64];
  __shared__ float rarea_s[64];
  const int s = blockIdx.z;
  // n_cap: only the leading n_cap rows / columns of the segment (the first phase of a keep[:max_keep] call, see dtc_nms_sorted);
  // a NEGATIVE count = a segment that needs nothing (n <= 0: no tile is enumerated)
  const int n = min(counts ? min(counts[s], n_stride) : n_stride, n_cap);
  const int ncb = (n + 63) >> 6;
  const int ncg = (ncb + kMaskWaves - 1) / kMaskWaves;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4* B = boxes + (size_t)s * n_stride;
  // only the tile groups that reach the diagonal or lie above it are enumerated: row block rb has ncg - rb / kMaskWaves of them
  // (a 16 x 16-block segment: 40 of 64 -- the other 24 used to be workgroups that were launched to return at once)
  const int full = ncb / kMaskWaves;                          // row-block bands of kMaskWaves with the same group count
  int n_upper = 0;
  for (int q = 0; q <= full; q++) n_upper += (ncg - q) * min(kMaskWaves, ncb - q * kMaskWaves);
  for (int t = blockIdx.x; t < n_upper; t += gridDim.x) {
    int rb = 0, rem = t;
    for (int q = 0; q <= full; q++) {                         // uniform: at most ncb / 4 + 1 steps
      const int per = ncg - q, rows = min(kMaskWaves, ncb - q * kMaskWaves);
      if (rem < per * rows) { rb = q * kMaskWaves + rem / per; rem -= (rem / per) * per; break; }
      rem -= per * rows;
    }
    const int cg = rb / kMaskWaves + rem;
    const int cb0 = cg * kMaskWaves;
    __syncthreads();                                          // previous tile's readers are done with rbox_s
    if (wv == 0) {
      const int row = rb * 64 + lane;
      // padding rows: a box that intersects nothing (inter = 0 against any column) with a stand-in area of 1, so that
      // u = 1 + areas[j] stays positive and the row loop's fast path holds for them too (their words are never stored)
      const float4 rbx = row < n ? B[row] : make_float4(0.f, 0.f, -1.f, -1.f);
      rbox_s[lane] = rbx;
      rarea_s[lane] = row < n ? box_area(rbx) : 1.f;
    }
    __syncthreads();
    const int cb = cb0 + wv;
    if (cb < rb || cb >= ncb) continue;                       // per-wave skip; barriers above are reached by all waves
    const int col = cb * 64 + lane;
    const bool col_ok = col < n;
    const float4 cbox = col_ok ? B[col] : make_float4(0.f, 0.f, -1.f, -1.f);
    const float carea = box_area(cbox);
    // Two instances of the row loop, chosen per wavefront (cb, rb are wave-uniform): only the DIAGONAL tile needs the
    // (_j > _i) test of cython_nms.pyx:72 -- above the diagonal every column index exceeds every row index -- and only it
    // produces the transposed word the reduce kernel resolves the in-block chain on.  The kernel is VALU-bound (30 instructions
    // per pair in round 1), so the loop is written for instruction count: see the notes at vmax and inside.
    uint64_t myword = 0, mycol = 0;
    // LDS row tables behind a base the optimiser must keep in a VGPR: with a uniform (SGPR) base every ds_read of the unrolled
    // loop was preceded by a v_mov of its address
    int lds_base = 0;
    asm volatile("" : "+v"(lds_base));
    const float4* rb_v = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(rbox_s) + lds_base);
    const float* ra_v = reinterpret_cast<const float*>(reinterpret_cast<const char*>(rarea_s) + lds_base);
    const int nrow = min(64, n - rb * 64);                    // rows past the segment's count are never stored: not computed
    const bool thr_pos = thresh > 0.f;
    auto rows = [&](auto diag_tag) {
      constexpr bool kDiag = decltype(diag_tag)::value;
      // chunks of 8 rows, the chunk unrolled by hand (a run-time trip count keeps the compiler from unrolling, and the loop
      // overhead is a quarter of the body); rows of the last chunk past nrow are padding rows: computed, never stored
      for (int i0 = 0; i0 < nrow; i0 += 8) {
        uint32_t cbits = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
        const int i = i0 + k;
        const float4 r = rb_v[i];                                  // uniform address: LDS broadcast
        const float iarea = ra_v[i];
        const float xx1 = vmax(r.x, cbox.x), yy1 = vmax(r.y, cbox.y);            // cython_nms.pyx:76-77
        const float xx2 = vmin(r.z, cbox.z), yy2 = vmin(r.w, cbox.w);            // :78-79
        const float w = fmaxf(0.0f, xx2 - xx1 + 1.f), h = fmaxf(0.0f, yy2 - yy1 + 1.f);  // :80-81
        const float inter = w * h;                                               // :82
        // :83-84  `inter / (iarea + areas[j] - inter) >= thresh` with an IEEE float division.  Rounding is monotone, so the
        // rounded quotient is >= thresh whenever the exact one is, and < thresh whenever the exact one is below
        // thresh * (1 - 2^-24): the division itself is needed only if inter - thresh * u cannot be signed reliably.
        // d = fl(inter - fl(thresh * u)) carries <= 2^-23 * pu of error, so outside the band |d| <= 2^-21 * pu its sign IS the
        // answer: five instructions and ONE compare per pair, and that compare is the ballot mask itself.  (The previous form
        // -- two scaled products, three compares -- made the compiler rebuild the mask with a 0/1 select + re-compare.)
        // u <= 0 (degenerate boxes), thresh <= 0 and the band take the division, exactly like the reference; they are decided
        // per wavefront (a uniform branch: about one pair in 10^6 is that close to the threshold).
        const float u = iarea + carea - inter;
        const float pu = thresh * u;
        const float d = inter - pu;
        const float t = __builtin_fabsf(d) - pu * 4.76837158203125e-07f;         // 2^-21 (exact scaling)
        const float m = vmin(t, u);                          // <= 0 (or NaN-free t with u NaN: not suppressed either way)
        uint64_t word;
        bool sup;
        if (__builtin_amdgcn_ballot_w64(!(m > 0.f)) == 0ull && thr_pos) {
          // padding columns need no test here: their box (0, 0, -1, -1) intersects nothing (inter = 0, u = iarea > 0 -> d < 0)
          sup = kDiag ? (d > 0.f) && (lane > i) : (d > 0.f);                     // :72 (_j > _i) on the diagonal tile only
          word = __builtin_amdgcn_ballot_w64(sup);
        } else {
          const bool need = !(m > 0.f) || !thr_pos;
          const bool gd = fdiv(inter, u) >= thresh;
          sup = col_ok && (need ? gd : d > 0.f);
          if (kDiag) sup = sup && (lane > i);
          word = __builtin_amdgcn_ballot_w64(sup);
        }
        if (lane == i) myword = word;
        if (kDiag) cbits |= sup ? (1u << k) : 0u;          // column view of the same tile: rows that suppress MY column
        }
        if (kDiag) mycol |= (uint64_t)cbits << i0;
      }
    };
    if (cb == rb) rows(std::true_type{}); else rows(std::false_type{});
    const int row = rb * 64 + lane;
    if (row < n) mask[((size_t)s * n_stride + row) * ncb_stride + cb] = myword;
    if (cb == rb && col_ok) diag_t[(size_t)s * n_stride + col] = mycol;   // transposed diagonal tile for the reduce
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. greedy reduce, one wavefront per segment.  keep [S, keep_stride] = kept positions (score order), keep_count [S].
//    WPL = words of `removed` per lane (supports n <= 64*64*WPL boxes).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// One wavefront per segment.  lane = (rg, cbl): row group rg = lane >> 4 owns rows r = 4k + rg of the current 64-row
// block, cbl = lane & 15 one column word of a 16-word chunk.  All 16 loads of a (row block, chunk) are independent of the
// resolve and are issued together (the diagonal chunk and diag words of the NEXT row block are prefetched while the current
// one is resolved), so a row block costs one overlapped L2 round trip instead of two dependent ones.
struct ReduceRegs { uint64_t w[16]; uint64_t diag; };   // diag: TRANSPOSED diagonal tile word (suppressors of my column)

__device__ __forceinline__ void reduce_load(ReduceRegs& R, const uint64_t* __restrict__ M,
                                            const uint64_t* __restrict__ DT, int ncb_stride, int n, int ncb, int rb,
                                            int cbase, bool with_diag) {
  const int lane = threadIdx.x, rg = lane >> 4, cbl = lane & 15;
  const int c = cbase + cbl;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int row = rb * 64 + 4 * k + rg;
    R.w[k] = (row < n && c < ncb && c > rb) ? M[(size_t)row * ncb_stride + c] : 0ull;
  }
  if (with_diag) {
    const int row = rb * 64 + lane;
    R.diag = row < n ? DT[row] : 0ull;
  }
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(const uint64_t* __restrict__ mask,
                                                        const uint64_t* __restrict__ diag_t,
                                                        const int32_t* __restrict__ counts, int n_stride, int n_cap,
                                                        int ncb_stride, int max_keep, int32_t* __restrict__ keep,
                                                        int keep_stride, int32_t* __restrict__ keep_count,
                                                        int32_t* __restrict__ next_counts) {
  __shared__ uint64_t removed[256];            // one bit per box, up to 16384 boxes
  const int s = blockIdx.x, lane = threadIdx.x, rg = lane >> 4, cbl = lane & 15;
  if (counts && counts[s] < 0) {               // a segment that is already reduced (the first phase below finished it, or the CALLER says so): keep / keep_count stay
    if (next_counts && lane == 0) next_counts[s] = -1;   // ... in the second phase of a keep[:max_keep] call too (it runs on next_counts)
    return;
  }
  const int n_full = counts ? min(counts[s], n_stride) : n_stride;
  const int n = min(n_full, n_cap);
  const int ncb = (n + 63) >> 6;
  const uint64_t* M = mask + (size_t)s * n_stride * ncb_stride;
  const uint64_t* DT = diag_t + (size_t)s * n_stride;
  int32_t* K = keep + (size_t)s * keep_stride;
  const int cap = max_keep > 0 ? min(max_keep, keep_stride) : keep_stride;
  for (int i = lane; i < 256; i += 64) removed[i] = 0;
  __syncthreads();
  int kept = 0;
  ReduceRegs cur, nxt;
  if (ncb > 0) reduce_load(cur, M, DT, ncb_stride, n, ncb, 0, 0, true);
  for (int rb = 0; rb < ncb && kept < cap; rb++) {
    const int cbase0 = rb & ~15;
    if (rb + 1 < ncb) reduce_load(nxt, M, DT, ncb_stride, n, ncb, rb + 1, (rb + 1) & ~15, true);   // prefetch
    const int left = n - rb * 64;
    const uint64_t valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    uint64_t cand = valid & ~removed[rb];
    cand = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cand >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cand);   // uniform by construction: keep it in SGPRs
    // In-block greedy resolve as a fixed point: row i is kept iff it is a candidate and no KEPT earlier row of the block
    // suppresses it.  F(K) = {i in cand : (suppressors(i) & K) == 0} is antitone and the greedy answer is its unique fixed
    // point; iterating from K = cand fixes the i-th candidate after at most i steps (typically 2-4 wave-wide steps in all,
    // instead of one dependent scalar step per kept row).  suppressors(i) is this lane's word of the transposed diagonal
    // tile, which only has bits of rows < i.
    uint64_t keepw = cand;
    if (cand != 0) {
      for (int it = 0; it < 64; it++) {
        const bool k_i = ((cand >> lane) & 1ull) && ((cur.diag & keepw) == 0ull);
        const uint64_t nk = __ballot(k_i);
        if (nk == keepw) break;
        keepw = nk;
      }
      const int room = cap - kept;               // keep[:max_keep]: only the first `room` survivors of this block count
      if (__builtin_popcountll(keepw) > room) {
        uint64_t t = keepw, kk = 0;
        for (int q = 0; q < room; q++) { kk |= t & (~t + 1ull); t &= t - 1ull; }
        keepw = kk;
      }
      kept += __builtin_popcountll(keepw);
    }
    if ((keepw >> lane) & 1ull) {
      const int before = __builtin_popcountll(keepw & ((1ull << lane) - 1ull));
      K[kept - __builtin_popcountll(keepw) + before] = rb * 64 + lane;
    }
    if (rb + 1 < ncb && kept < cap) {
      for (int cbase = cbase0; cbase < ncb; cbase += 16) {
        if (cbase != cbase0) reduce_load(cur, M, DT, ncb_stride, n, ncb, rb, cbase, false);
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++)
          if ((keepw >> (4 * k + rg)) & 1ull) acc |= cur.w[k];
        uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
        lo |= __shfl_xor(lo, 16, 64); hi |= __shfl_xor(hi, 16, 64);
        lo |= __shfl_xor(lo, 32, 64); hi |= __shfl_xor(hi, 32, 64);
        if (rg == 0 && cbase + cbl < ncb) removed[cbase + cbl] |= ((uint64_t)hi << 32) | lo;
      }
      __syncthreads();
    }
    cur = nxt;
  }
  if (lane == 0) {
    keep_count[s] = kept;
    // first phase of a keep[:max_keep] call: the segment is finished when max_keep boxes are kept (rows past the last kept one can
    // not change the first max_keep survivors) or when it had no more than n_cap rows; otherwise the second phase redoes it in full
    if (next_counts) next_counts[s] = (kept >= cap || n_full <= n_cap) ? -1 : n_full;
  }
}

// The same walk with the segment's whole suppression matrix in LDS: for few, long segments (the RPN call: 40 segments of 1000
// boxes, 16 row blocks each).  The one-wave kernel above pays one L2 round trip per row block that its one-block prefetch
// cannot hide, and ~700 instructions per block in a single wavefront (guarded 64-bit global addressing of 16 row words, register
// ping-pong, shuffle merge): 1.5 us per block, 25 us for the RPN call.  Here 256 threads copy the matrix (n rows x ncb words,
// + the transposed diagonal words) into LDS with all loads in flight, then wavefront 0 walks the blocks from LDS:
//   * row words are read unguarded: rows past n and words left of the diagonal (never written by nms_mask) are zero in LDS;
//   * lane (rg, cbl) ORs the words of the KEPT rows 4k + rg with a sign-extended bit field as the mask (3 instructions per
//     row), the four row groups meet in LDS with one ds_or_b64;
//   * no prefetch registers: the 17 LDS reads of a block are issued at the top of its iteration, under the resolve.
constexpr int kReduceLdsThreads = 1024;     // the copy wants loads in flight (round 6: 256 -> 1024 threads, 8 instead of 32 16-byte loads each); the walk is wave 0's

__global__ __launch_bounds__(kReduceLdsThreads) void nms_reduce_lds_kernel(const uint64_t* __restrict__ mask,
                                                                           const uint64_t* __restrict__ diag_t,
                                                                           const int32_t* __restrict__ counts, int n_stride,
                                                                           int ncb_stride, int max_keep, int32_t* __restrict__ keep,
                                                                           int keep_stride, int32_t* __restrict__ keep_count) {
  __shared__ uint64_t removed[256];            // one bit per box
  extern __shared__ __attribute__((aligned(16))) unsigned char reduce_smem[];   // [nrow_pad * ncb_stride] + [nrow_pad] words
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, rg = lane >> 4, cbl = lane & 15;
  const int nrow_pad = ncb_stride * 64;        // LDS image: ncb_stride x 64 rows; rows past the segment's count are zero-filled
  const uint64_t* Mg = mask + (size_t)s * n_stride * ncb_stride;
  const uint64_t* DTg = diag_t + (size_t)s * n_stride;
  int32_t* K = keep + (size_t)s * keep_stride;
  const int cap = max_keep > 0 ? min(max_keep, keep_stride) : keep_stride;
  uint64_t* Ml = reinterpret_cast<uint64_t*>(reduce_smem);
  uint64_t* DTl = Ml + (size_t)nrow_pad * ncb_stride;
  DTC_PT(0, s, 0);
  if (tid < 256) removed[tid] = 0;
  int n;
  {
    // 16-byte copies (ncb_stride is even and <= 16: the launcher checks) of the words on and right of the diagonal, every load
    // of a thread in flight at once -- and issued BEFORE the segment's count is known (the count is one more dependent global
    // round trip): rows of the workspace past the count hold stale words, they are replaced by zeros on the way into LDS, as
    // are the words left of the diagonal (never written by nms_mask)
    constexpr int kPerThread = 128 * 1024 / 16 / kReduceLdsThreads;      // x threads x 16 B = 128 KB
    const int ppr = ncb_stride >> 1;           // 16-byte pairs per row
    const int npair = nrow_pad * ppr;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(Mg);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(Ml);
    constexpr int kDv = 1024 / kReduceLdsThreads;
    uint64_t dv[kDv];
#pragma unroll
    for (int k = 0; k < kDv; k++) { const int i = tid + k * kReduceLdsThreads; dv[k] = i < n_stride ? DTg[i] : 0ull; }
    ulonglong2 v[kPerThread];
    // (row, pair) of element tid + k * 256, stepped without a division per element
    const int row0 = tid / ppr, pr0 = tid - row0 * ppr, drow = kReduceLdsThreads / ppr, dpr = kReduceLdsThreads - drow * ppr;
    int row = row0, pr = pr0;
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
      const int i = tid + k * kReduceLdsThreads;
      v[k] = make_ulonglong2(0, 0);
      if (i < npair && row < n_stride && 2 * pr + 1 >= (row >> 6)) v[k] = src[i];
      row += drow; pr += dpr;
      if (pr >= ppr) { pr -= ppr; row++; }
    }
    n = counts ? min(counts[s], n_stride) : n_stride;
    if (n < 0) return;                           // a segment somebody else has already reduced (uniform: the whole workgroup leaves)
    row = row0; pr = pr0;
#pragma unroll
    for (int k = 0; k < kPerThread; k++) {
      const int i = tid + k * kReduceLdsThreads;
      if (i < npair) dst[i] = row < n ? v[k] : make_ulonglong2(0, 0);
      row += drow; pr += dpr;
      if (pr >= ppr) { pr -= ppr; row++; }
    }
#pragma unroll
    for (int k = 0; k < kDv; k++) { const int i = tid + k * kReduceLdsThreads; if (i < nrow_pad) DTl[i] = i < n ? dv[k] : 0ull; }
  }
  const int ncb = (n + 63) >> 6;
  __syncthreads();
  DTC_PT(0, s, 1);
  if (tid >= 64) return;                       // the serial part belongs to wavefront 0: no workgroup barrier below
  int kept = 0;
  // byte address of this lane's word of row rg in LDS; row block rb, row 4k + rg, column chunk cbase: + the offsets below
  const uint32_t row_bytes = (uint32_t)ncb_stride * 8u;
  const char* lane_base = reinterpret_cast<const char*>(Ml) + (uint32_t)rg * row_bytes + (uint32_t)cbl * 8u;
  for (int rb = 0; rb < ncb && kept < cap; rb++) {
    const uint64_t diag = DTl[rb * 64 + lane];
    // the 16 row words of this block (rows 4k + rg, column word cbl of the chunk that holds the diagonal) are read now, under
    // the resolve: they do not depend on it
    const char* blk = lane_base + (uint32_t)rb * 64u * row_bytes;
    uint64_t w[16];
    {
      const char* src = blk + (uint32_t)(rb & ~15) * 8u;
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = *reinterpret_cast<const uint64_t*>(src + (uint32_t)(4 * k) * row_bytes);
    }
    const int left = n - rb * 64;
    const uint64_t valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    uint64_t cand = valid & ~removed[rb];
    cand = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cand >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cand);   // uniform by construction: keep it in SGPRs
    // in-block greedy resolve as a fixed point (see nms_reduce_kernel)
    uint64_t keepw = cand;
    if (cand != 0) {
      for (int it = 0; it < 64; it++) {
        const bool k_i = ((cand >> lane) & 1ull) && ((diag & keepw) == 0ull);
        const uint64_t nk = __ballot(k_i);
        if (nk == keepw) break;
        keepw = nk;
      }
      const int room = cap - kept;               // keep[:max_keep]: only the first `room` survivors of this block count
      if (__builtin_popcountll(keepw) > room) {
        uint64_t t = keepw, kk = 0;
        for (int q = 0; q < room; q++) { kk |= t & (~t + 1ull); t &= t - 1ull; }
        keepw = kk;
      }
      kept += __builtin_popcountll(keepw);
    }
    if (rb + 1 < ncb && kept < cap && keepw != 0ull) {
      // bit 4k of (keepw >> rg) says whether row 4k + rg is kept
      const uint64_t km = keepw >> rg;
      const uint32_t km_lo = (uint32_t)km, km_hi = (uint32_t)(km >> 32);
      for (int cbase = rb & ~15; cbase < ncb; cbase += 16) {
        if (cbase != (rb & ~15)) {
          const char* src = blk + (uint32_t)cbase * 8u;
#pragma unroll
          for (int k = 0; k < 16; k++) w[k] = *reinterpret_cast<const uint64_t*>(src + (uint32_t)(4 * k) * row_bytes);
        }
        uint32_t acc_lo = 0, acc_hi = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int m = k < 8 ? __builtin_amdgcn_sbfe((int)km_lo, 4 * k, 1) : __builtin_amdgcn_sbfe((int)km_hi, 4 * (k - 8), 1);   // 0 / -1
          acc_lo |= (uint32_t)w[k] & (uint32_t)m;
          acc_hi |= (uint32_t)(w[k] >> 32) & (uint32_t)m;
        }
        const uint64_t acc = ((uint64_t)acc_hi << 32) | acc_lo;
        if (acc != 0ull && cbase + cbl < ncb) atomicOr(reinterpret_cast<unsigned long long*>(&removed[cbase + cbl]), (unsigned long long)acc);
      }
    }
    if ((keepw >> lane) & 1ull) {               // off the chain: the kept positions of this block leave after the OR is issued
      const int before = __builtin_popcountll(keepw & ((1ull << lane) - 1ull));
      K[kept - __builtin_popcountll(keepw) + before] = rb * 64 + lane;
    }
    // LDS operations of one wavefront retire in order; the compiler must not carry `removed` across in registers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (lane == 0) keep_count[s] = kept;
  DTC_PT(0, s, 2);
}

// ---------------------------------------------------------------------------------------------------------------------
// finalize for the single-segment drop-in: kept positions (score order) -> ascending ORIGINAL indices, int64.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void nms_finalize_kernel(const int32_t* __restrict__ keep,
                                                                    const int32_t* __restrict__ keep_count,
                                                                    const int32_t* __restrict__ order,
                                                                    int64_t* __restrict__ out, int32_t* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
  const int n = keep_count[0];
  const int np2 = next_pow2(n);
  for (int i = threadIdx.x; i < np2; i += kSortThreads) keys[i] = i < n ? (uint64_t)(uint32_t)order[keep[i]] : kPadKey;
  block_bitonic_sort<kSortThreads>(keys, np2);
  for (int i = threadIdx.x; i < n; i += kSortThreads) out[i] = (int64_t)keys[i];
  if (threadIdx.x == 0) *out_count = n;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace dtc

// ---- C ABI ----------------------------------------------------------------------------------------------------------

DTC_API size_t dtc_nms_sorted_workspace_bytes(int n_seg, int n_stride) {
  const size_t ncb = (size_t)(n_stride + 63) / 64;
  // suppression matrix [n_seg][n_stride][ncb] + transposed diagonal tiles [n_seg][n_stride]
  // + the per-segment counts the second phase of a keep[:max_keep] call runs on
  return dtc::align_up((size_t)n_seg * n_stride * ncb * sizeof(uint64_t), 256) +
         dtc::align_up((size_t)n_seg * n_stride * sizeof(uint64_t), 256) + dtc::align_up((size_t)n_seg * sizeof(int32_t), 256);
}

DTC_API int dtc_nms_sorted(const float* boxes, const int32_t* counts, int n_seg, int n_stride, float thresh,
                           int max_keep, void* workspace, size_t workspace_bytes, int32_t* keep, int keep_stride,
                           int32_t* keep_count, dtc_stream_t stream) {
  if (n_seg < 0 || n_stride < 0 || keep_stride < 0) return DTC_EINVAL;
  if (n_seg == 0) return DTC_OK;
  if (!keep_count || (n_stride > 0 && (!boxes || !keep || !workspace))) return DTC_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n_stride == 0) { return dtc::zero_async(keep_count, sizeof(int32_t) * n_seg, s); }
  if (workspace_bytes < dtc_nms_sorted_workspace_bytes(n_seg, n_stride)) return DTC_EWORKSPACE;
  const int ncb = (n_stride + 63) / 64;
  if (ncb > 64 * 4) return DTC_EUNSUPPORTED;  // > 16384 boxes per segment
  uint64_t* mask = reinterpret_cast<uint64_t*>(workspace);
  uint64_t* diag_t = reinterpret_cast<uint64_t*>(reinterpret_cast<unsigned char*>(workspace) +
                                                 dtc::align_up((size_t)n_seg * n_stride * ((n_stride + 63) / 64) * sizeof(uint64_t), 256));
  int32_t* next_counts = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(diag_t) + dtc::align_up((size_t)n_seg * n_stride * sizeof(uint64_t), 256));
  // tile groups on or above the diagonal of a segment of ncb_ column blocks (see the kernel)
  auto upper_groups = [](int ncb_) {
    const int ncg = (ncb_ + dtc::kMaskWaves - 1) / dtc::kMaskWaves;
    int groups = 0;
    for (int q = 0; q <= ncb_ / dtc::kMaskWaves; q++) {
      const int rows = ncb_ - q * dtc::kMaskWaves < dtc::kMaskWaves ? ncb_ - q * dtc::kMaskWaves : dtc::kMaskWaves;
      groups += (ncg - q) * (rows > 0 ? rows : 0);
    }
    return groups;
  };
  auto launch_mask = [&](const int32_t* cnts, int n_cap) {
    // about 2000 workgroups per launch: every one of a few long segments' groups, ONE workgroup for each of many short segments
    // (the 640 class segments of a detection batch hold ~10 candidates = one tile each)
    const int groups = upper_groups(((n_cap < n_stride ? n_cap : n_stride) + 63) / 64);
    int gx = 2048 / n_seg;
    if (gx < 1) gx = 1;
    if (gx > groups) gx = groups;
    hipLaunchKernelGGL(dtc::nms_mask_kernel, dim3(gx, 1, n_seg), dim3(64 * dtc::kMaskWaves), 0, s,
                       reinterpret_cast<const float4*>(boxes), cnts, n_stride, n_cap, ncb, thresh, mask, diag_t);
  };
  // keep[:max_keep] of a LONG segment (the C4 RPN call: 6000 sorted boxes, 1000 kept -- generate_proposals.py:114-117): the first
  // max_keep survivors are decided by the leading rows alone (the 1000th kept box of the bench's C4 batches is row ~1100), so the
  // first phase builds and walks only the leading n1 x n1 corner of the matrix (n1 = 2048: 528 of 4465 tiles, 2 of 6 column chunks
  // per row block); a segment that has not reached max_keep by row n1 is redone in full by the second phase, every other segment
  // is marked finished (count -1) and costs the second phase two empty workgroups.  Same keep lists either way.
  const int n1 = ((2 * max_keep + 63) / 64 < 16 ? 16 : (2 * max_keep + 63) / 64) * 64;
  const bool two_phase = max_keep > 0 && n_stride >= 2 * n1;
  if (two_phase) {
    launch_mask(counts, n1);
    DTC_CHECK_LAUNCH();
    hipLaunchKernelGGL(dtc::nms_reduce_kernel, dim3(n_seg), dim3(64), 0, s, mask, diag_t, counts, n_stride, n1, ncb, max_keep, keep,
                       keep_stride, keep_count, next_counts);
    DTC_CHECK_LAUNCH();
    counts = next_counts;
  }
  launch_mask(counts, n_stride);
  DTC_CHECK_LAUNCH();
  // few long segments whose matrix fits LDS (the RPN call): the LDS walk (see the kernel); else the one-wave walk
  const size_t lds_need = ((size_t)ncb * 64 * ncb + (size_t)ncb * 64) * sizeof(uint64_t);
  // (each workgroup of the LDS walk holds ~136 KB: one per CU.  Up to 160 segments -- RPN calls up to batch 32 -- are resident
  // together; the hundreds of ~10-candidate class segments of a detection batch are better off in the one-wave kernel.)
  if (n_seg <= 160 && ncb >= 4 && ncb <= 16 && (ncb & 1) == 0 && lds_need <= 150 * 1024) {
    if (lds_need > 48 * 1024) DTC_RAISE_LDS_ONCE(dtc::nms_reduce_lds_kernel, 152 * 1024);   // + 2 KB static
    hipLaunchKernelGGL(dtc::nms_reduce_lds_kernel, dim3(n_seg), dim3(dtc::kReduceLdsThreads), lds_need, s, mask, diag_t, counts,
                       n_stride, ncb, max_keep, keep, keep_stride, keep_count);
  } else {
    hipLaunchKernelGGL(dtc::nms_reduce_kernel, dim3(n_seg), dim3(64), 0, s, mask, diag_t, counts, n_stride, n_stride, ncb, max_keep, keep,
                       keep_stride, keep_count, static_cast<int32_t*>(nullptr));
  }
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

DTC_API int dtc_segment_sort_desc(const float* scores, int score_stride_elems, const float* boxes, int box_stride_elems,
                                  const int32_t* counts, int n_seg, int n_stride, int32_t* order, float* sorted_boxes,
                                  float* sorted_scores, dtc_stream_t stream) {
  if (n_seg < 0 || n_stride < 0 || n_stride > 16384 || score_stride_elems < 1) return n_stride > 16384 ? DTC_EUNSUPPORTED : DTC_EINVAL;
  if (n_seg == 0 || n_stride == 0) return DTC_OK;
  if (!scores || (sorted_boxes && !boxes)) return DTC_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = (size_t)dtc::next_pow2(n_stride) * sizeof(uint64_t);
  if (smem > 32 * 1024) {   // static __shared__ of the kernel comes on top: raise the limit well before dynamic + static reaches 64 KB
    DTC_RAISE_LDS_ONCE(dtc::segment_sort_desc_kernel, 160 * 1024);
  }
  hipLaunchKernelGGL(dtc::segment_sort_desc_kernel, dim3(n_seg), dim3(dtc::kSortThreads), smem, s, scores,
                     score_stride_elems, boxes, box_stride_elems, counts, n_stride, order, sorted_boxes, sorted_scores);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

// Drop-in for cython_nms.nms(dets[N,5] float32, thresh) (lib/utils_cython/cython_nms.pyx:37): device in, device out.
// keep_out int64 [n] receives ascending original indices, keep_count int32 [1] their number.
DTC_API size_t dtc_nms_workspace_bytes(int n) {
  const size_t a = dtc::align_up((size_t)n * 4 * sizeof(float), 256);    // sorted boxes
  const size_t b = dtc::align_up((size_t)n * sizeof(int32_t), 256);      // order
  const size_t c = dtc::align_up((size_t)n * sizeof(int32_t), 256);      // keep positions
  const size_t d = 256;                                                  // count
  return a + b + c + d + dtc_nms_sorted_workspace_bytes(1, n);
}

DTC_API int dtc_nms(const float* dets, int n, float thresh, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                    int32_t* keep_count, dtc_stream_t stream) {
  if (n < 0 || !keep_count) return DTC_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) return dtc::zero_async(keep_count, sizeof(int32_t), s);  // boxes.py:334-335
  if (n > 16384) return DTC_EUNSUPPORTED;
  if (!dets || !keep_out || !workspace) return DTC_EINVAL;
  if (workspace_bytes < dtc_nms_workspace_bytes(n)) return DTC_EWORKSPACE;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  float* sboxes = reinterpret_cast<float*>(w); w += dtc::align_up((size_t)n * 4 * sizeof(float), 256);
  int32_t* order = reinterpret_cast<int32_t*>(w); w += dtc::align_up((size_t)n * sizeof(int32_t), 256);
  int32_t* keep = reinterpret_cast<int32_t*>(w); w += dtc::align_up((size_t)n * sizeof(int32_t), 256);
  int32_t* cnt = reinterpret_cast<int32_t*>(w); w += 256;
  int rc = dtc_segment_sort_desc(dets + 4, 5, dets, 5, nullptr, 1, n, order, sboxes, nullptr, stream);
  if (rc != DTC_OK) return rc;
  rc = dtc_nms_sorted(sboxes, nullptr, 1, n, thresh, 0, w, dtc_nms_sorted_workspace_bytes(1, n), keep, n, cnt, stream);
  if (rc != DTC_OK) return rc;
  const size_t smem = (size_t)dtc::next_pow2(n) * sizeof(uint64_t);
  if (smem > 32 * 1024) {   // static __shared__ of the kernel comes on top: raise the limit well before dynamic + static reaches 64 KB
    DTC_RAISE_LDS_ONCE(dtc::nms_finalize_kernel, 160 * 1024);
  }
  hipLaunchKernelGGL(dtc::nms_finalize_kernel, dim3(1), dim3(dtc::kSortThreads), smem, s, keep, cnt, order, keep_out, keep_count);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// A6  Soft-NMS -- replaces cython_nms.soft_nms (lib/utils_cython/cython_nms.pyx:98-203; entry lib/utils/boxes.py:339-356).
// Off by default in the reference (do_soft_nms=False, lib/utils/result_utils.py:100) and inherently sequential over the
// picks, so this is a correctness-first kernel: ONE wavefront walks the reference's in-place array algorithm on an LDS
// copy -- per pick a wave-wide argmax (first maximum, :128-132), the swap (:135-148), a lane-parallel decay of the rest
// (:159-187) and, only when some score fell below the threshold, lane 0 replays the reference's swap-with-last loop
// (:191-199) on the precomputed flags.  Mixed precision exactly as the Cython compiles: `x2 - x1 + 1` etc. are float
// differences promoted to DOUBLE by the literal 1.0 (see oracle/oracle.c orc_soft_nms).
// ---------------------------------------------------------------------------------------------------------------------
namespace dtc {

__global__ __launch_bounds__(64) void soft_nms_kernel(const float* __restrict__ dets_in, int n, float sigma, float Nt,
                                                      float threshold, int method, float* __restrict__ dets_out,
                                                      int64_t* __restrict__ inds_out, int32_t* __restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* X1 = reinterpret_cast<float*>(smem);
  float* Y1 = X1 + n; float* X2 = Y1 + n; float* Y2 = X2 + n; float* S = Y2 + n;
  int32_t* I = reinterpret_cast<int32_t*>(S + n);
  unsigned char* dead = reinterpret_cast<unsigned char*>(I + n);
  const int lane = threadIdx.x;
  for (int k = lane; k < n; k += 64) {
    X1[k] = dets_in[k * 5 + 0]; Y1[k] = dets_in[k * 5 + 1]; X2[k] = dets_in[k * 5 + 2]; Y2[k] = dets_in[k * 5 + 3];
    S[k] = dets_in[k * 5 + 4]; I[k] = k; dead[k] = 0;
  }
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  int N = n;
  for (int i = 0; i < N; i++) {
    // ---- argmax over [i, N): first maximum in scan order (strict <)  :128-132
    float bs = -INFINITY; int bp = 0x7fffffff;
    for (int pos = i + lane; pos < N; pos += 64) {
      const float s = S[pos];
      if (bp == 0x7fffffff || s > bs) { bs = s; bp = pos; }   // within a lane positions ascend: keep the first max
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off, 64);
      const int op = __shfl_xor(bp, off, 64);
      const bool take = (op != 0x7fffffff) && (bp == 0x7fffffff || os > bs || (os == bs && op < bp));
      if (take) { bs = os; bp = op; }
    }
    const int maxpos = bp;
    // ---- swap rows i and maxpos  :135-148
    if (lane == 0 && maxpos != i) {
      float t;
      t = X1[i]; X1[i] = X1[maxpos]; X1[maxpos] = t;
      t = Y1[i]; Y1[i] = Y1[maxpos]; Y1[maxpos] = t;
      t = X2[i]; X2[i] = X2[maxpos]; X2[maxpos] = t;
      t = Y2[i]; Y2[i] = Y2[maxpos]; Y2[maxpos] = t;
      t = S[i]; S[i] = S[maxpos]; S[maxpos] = t;
      const int ti = I[i]; I[i] = I[maxpos]; I[maxpos] = ti;
    }
    __syncthreads();
    const float tx1 = X1[i], ty1 = Y1[i], tx2 = X2[i], ty2 = Y2[i];
    // ---- decay the rest  :159-187
    bool any_dead = false;
    for (int pos = i + 1 + lane; pos < N; pos += 64) {
      const float x1 = X1[pos], y1 = Y1[pos], x2 = X2[pos], y2 = Y2[pos];
      const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));      // :166
      const float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);              // :167
      bool d = false;
      if (iw > 0.f) {
        const float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);            // :169
        if (ih > 0.f) {
          const float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                                   (double)(iw * ih));                                       // :171
          const float ov = fdiv(iw * ih, ua);                                                 // :172
          float weight;
          if (method == 1) weight = ov > Nt ? (float)(1.0 - (double)ov) : 1.f;                // :174-178
          else if (method == 2) weight = (float)exp((double)fdiv(-(ov * ov), sigma));         // :180
          else weight = ov > Nt ? 0.f : 1.f;                                                  // :182-185
          const float ns = weight * S[pos];                                                   // :187
          S[pos] = ns;
          d = ns < threshold;                                                                 // :191
        }
      }
      dead[pos] = d ? 1 : 0;
      any_dead |= d;
    }
    any_dead = __any(any_dead);
    __syncthreads();
    // ---- discard by swap-with-last, replayed sequentially on the flags  :191-199
    if (any_dead) {
      if (lane == 0) {
        int pos = i + 1;
        while (pos < N) {
          if (dead[pos]) {
            X1[pos] = X1[N - 1]; Y1[pos] = Y1[N - 1]; X2[pos] = X2[N - 1]; Y2[pos] = Y2[N - 1]; S[pos] = S[N - 1];
            I[pos] = I[N - 1]; dead[pos] = dead[N - 1];
            N = N - 1; pos = pos - 1;
          }
          pos = pos + 1;
        }
      }
      N = __shfl(N, 0, 64);
      __syncthreads();
    }
  }
  for (int k = lane; k < N; k += 64) {
    dets_out[k * 5 + 0] = X1[k]; dets_out[k * 5 + 1] = Y1[k]; dets_out[k * 5 + 2] = X2[k]; dets_out[k * 5 + 3] = Y2[k];
    dets_out[k * 5 + 4] = S[k]; inds_out[k] = I[k];
  }
  if (lane == 0) *n_out = N;
}

}  // namespace dtc

// Drop-in for cython_nms.soft_nms(boxes_in, sigma, Nt, threshold, method) (lib/utils_cython/cython_nms.pyx:98): device in /
// device out.  method 0 hard, 1 linear, 2 gaussian (lib/utils/boxes.py:346).  dets_out [n,5] / inds_out int64 [n] receive
// the N' surviving rows in selection order, n_out int32 [1] = N'.  n <= 6000.
DTC_API int dtc_soft_nms(const float* dets, int n, float sigma, float overlap_thresh, float score_thresh, int method,
                         float* dets_out, int64_t* inds_out, int32_t* n_out, dtc_stream_t stream) {
  if (n < 0 || !n_out || method < 0 || method > 2) return DTC_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) return dtc::zero_async(n_out, sizeof(int32_t), s);
  if (n > 6000) return DTC_EUNSUPPORTED;
  if (!dets || !dets_out || !inds_out) return DTC_EINVAL;
  const size_t smem = (size_t)n * (6 * 4 + 1) + 16;
  DTC_RAISE_LDS_ONCE(dtc::soft_nms_kernel, 160 * 1024);
  hipLaunchKernelGGL(dtc::soft_nms_kernel, dim3(1), dim3(64), smem, s, dets, n, sigma, overlap_thresh, score_thresh, method,
                     dets_out, inds_out, n_out);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// A4 (numpy flavour)  bbox_transform + clip_tiled_boxes for ALL classes -- lib/utils/boxes.py:168-208 and :150-165.
// The fused detection kernel decodes only the (roi, class) pairs that pass the score threshold; this entry exists for
// callers of the stand-alone box_utils functions.
// ---------------------------------------------------------------------------------------------------------------------
namespace dtc {
__global__ void bbox_transform_kernel(const float* __restrict__ boxes, const float* __restrict__ deltas, int n, int n_cls,
                                      float wx, float wy, float ww, float wh, int do_clip, float im_h, float im_w,
                                      float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * n_cls) return;
  const int i = t / n_cls;
  const float* b = boxes + (size_t)i * 4;
  const float* d = deltas + (size_t)t * 4;
  const float widths = b[2] - b[0] + 1.0f, heights = b[3] - b[1] + 1.0f;          // boxes.py:178-179
  const float ctr_x = b[0] + 0.5f * widths, ctr_y = b[1] + 0.5f * heights;        // :180-181
  const float dx = fdiv(d[0], wx), dy = fdiv(d[1], wy);                           // :184-185
  float dw = fdiv(d[2], ww), dh = fdiv(d[3], wh);                                 // :186-187
  dw = fminf(dw, 4.135166556742356f); dh = fminf(dh, 4.135166556742356f);         // :190-191
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;              // :193-194
  const float pw = fexp_cr(dw) * widths, ph = fexp_cr(dh) * heights;              // :195-196
  float o0 = pcx - 0.5f * pw, o1 = pcy - 0.5f * ph, o2 = pcx + 0.5f * pw - 1.f, o3 = pcy + 0.5f * ph - 1.f;  // :200-206
  if (do_clip) {                                                                  // :158-164
    o0 = fmaxf(fminf(o0, im_w - 1.f), 0.f); o1 = fmaxf(fminf(o1, im_h - 1.f), 0.f);
    o2 = fmaxf(fminf(o2, im_w - 1.f), 0.f); o3 = fmaxf(fminf(o3, im_h - 1.f), 0.f);
  }
  reinterpret_cast<float4*>(out)[t] = make_float4(o0, o1, o2, o3);
}
}  // namespace dtc

DTC_API int dtc_bbox_transform(const float* boxes, const float* deltas, int n, int n_cls, float wx, float wy, float ww,
                               float wh, int do_clip, float im_h, float im_w, float* out, dtc_stream_t stream) {
  if (n < 0 || n_cls < 1) return DTC_EINVAL;
  if (n == 0) return DTC_OK;
  if (!boxes || !deltas || !out) return DTC_EINVAL;
  const long long total = (long long)n * n_cls;
  hipLaunchKernelGGL(dtc::bbox_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), boxes, deltas, n, n_cls, wx, wy, ww, wh, do_clip, im_h, im_w, out);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
