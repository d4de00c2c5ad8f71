// A1  RoIAlign forward for gfx950 -- BAND-SWEEP kernel (sampling_ratio == 2, 7x7-class bins: the FPN box head).
//
// Replaces roi_align_forward_kernel (lib/cppcuda/roi_align_forward_cuda.cu:82-159) and is bit-compatible with the CPU path
// roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:118-219): same float32 operations, same order.
//
// Why.  The cluster kernel (roi_align_tile.hip) stages the union patch of ~5 neighbouring RoIs per workgroup: 9.9 line fills per
// (RoI, channel) where 2.7 are compulsory, 1.8 GB through the fabric per 8000-RoI launch where 0.73 GB are compulsory, and it
// sits at the fabric bandwidth for the bytes it requests.  Here a workgroup owns a BAND of 32 feature rows of one
// level of one image x 8 channels, and sweeps it once in x with a sliding LDS window (a ring of 64 columns): every row the
// band's windows cover is staged once per channel, 4.2 fills per (RoI, channel) on the bench distribution
// (tools/r03/band_model.py), and vertically adjacent bands of the same channels run side by side on one XCD, so the rows they
// share come out of that XCD's L2.
//
// Structure.
//   band_tab_kernel    one thread per RoI: the 2 x 14 axis samples of pre_calc_for_bilinear_interpolate (:36-95) as a 288-byte
//                      record (so that the set-up a pooling lane repeats for each of the 32 channel groups is two 16-byte loads),
//                      the RoI's window and its (image, level, band) key -- the key of dtc_fpn_collect_distribute's visiting order.
//   band_items_kernel  one workgroup: runs of equal key in the given order = band items [first, count), their row range,
//                      which RoIs do not fit the ring (they take a per-output gather), and eight contiguous item slices, one
//                      per XCD.  Any RoI order is correct; the (level, band, x) order makes the items long and the sweep monotone.
//   roi_align_fwd_band persistent workgroups (one per CU, 1024 threads) take (item, channel group) units from their XCD's
//                      queue (then steal).  Per unit:  plan the next batch (the longest run of RoIs whose windows fit the 64
//                      ring columns, <= 1024 / bins RoIs) | issue the 16-byte row pieces of the columns the ring does not hold
//                      (raw buffer loads into registers) | pool the current batch, lane <-> (RoI, bin), one ds_read_b128 per tap
//                      and channel quad, into an LDS slab | barrier | commit the pieces (transposing ds_write_b32 into
//                      [row][quad][column][4 channels], one pad slot per 8 columns) + slab -> [R,C,PH,PW] as 16-byte stores |
//                      barrier.
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "roi_align_common.h"

namespace dtc {

constexpr int kBandThreads = 1024, kBandWaves = kBandThreads / 64;
constexpr int kRingCols = 64;                       // columns of the sliding window (power of two)
constexpr int kRingSlots = kRingCols + kRingCols / 8;   // 16-byte slots per (row, quad): one pad slot every 8 columns
constexpr int kRowQuadBytes = kRingSlots * 16;      // 1152
constexpr int kBandUnits = 8;                       // 16-byte row pieces a thread carries per batch
constexpr int kBandMaxK = 32;                       // RoIs per batch, upper bound
constexpr int kBandMaxPooled = 8;                   // axis samples per record: 2 x 8

enum { kBandFlagGather = 1, kBandFlagZero = 2 };
enum { kItemPool = 0, kItemZero = 1, kItemGather = 2 };

struct BandHdr { uint32_t xw, yw; int32_t r; uint32_t flags; uint32_t key; int32_t item; int32_t pad[2]; };   // 32 B
struct BandEntry { uint32_t lohi; float l; };       // lo | hi << 16 ; l < 0: sample outside the map (weights 0, :49-63)
struct BandRoiTab { BandHdr h; BandEntry y[2 * kBandMaxPooled]; BandEntry x[2 * kBandMaxPooled]; };   // 288 B
static_assert(sizeof(BandRoiTab) == 288, "BandRoiTab layout");
struct BandItem { int first, count, b, lvl, rbase, rows, kind, pad; };   // 32 B
constexpr int kSlices = kXcds + 1;             // one item slice per XCD + the RoIs that take the gather path
struct BandCtl { int n_items, n_gather, pad0[6]; int slice_first[kSlices + 7]; int slice_count[kSlices + 7]; int ctr[kSlices + 7]; };
struct BandWs { BandCtl* ctl; BandItem* items; int* imin; int* imax; BandRoiTab* tab; };

typedef uint32_t bu32x4 __attribute__((ext_vector_type(4)));
typedef float bf32x2 __attribute__((ext_vector_type(2)));
typedef float bf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int buni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int ring_phys(int col) { const int c = col & (kRingCols - 1); return c + (c >> 3); }

// ---- prep 1: per-RoI records -------------------------------------------------------------------------------------------------
struct BandPrepParams { int fs[DTC_MAX_LEVELS]; int band_log2; };

__global__ __launch_bounds__(64) void band_tab_kernel(RoiAlignParams p, BandPrepParams pp, BandRoiTab* __restrict__ tab) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= p.n_rois) return;
  const RoiRaw raw = load_roi_raw(p, i);
  const RoiHead hd = roi_head_from_raw(p, raw);
  BandRoiTab* T = tab + i;
  BandHdr h;
  h.xw = h.yw = 0; h.r = hd.r; h.flags = 0; h.key = 0xffffffffu; h.item = 0; h.pad[0] = h.pad[1] = 0;
  if (hd.lvl < 0 || hd.lvl >= p.n_levels) {
    h.flags = kBandFlagZero;
  } else {
    const int H = p.lv[hd.lvl].height, W = p.lv[hd.lvl].width;
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    for (int s = 0; s < 2 * p.pooled_h; s++) {
      const AxisEntry e = make_axis(hd.sh, hd.bin_h, s >> 1, s & 1, 2, H);
      BandEntry be; be.lohi = (uint32_t)e.lo | ((uint32_t)e.hi << 16); be.l = e.h == 0.f ? -1.f : e.l;   // valid: h = 1 - l > 0
      T->y[s] = be;
      if (s == 0) y0 = e.lo;
      y1 = e.hi;
    }
    for (int s = 0; s < 2 * p.pooled_w; s++) {
      const AxisEntry e = make_axis(hd.sw, hd.bin_w, s >> 1, s & 1, 2, W);
      BandEntry be; be.lohi = (uint32_t)e.lo | ((uint32_t)e.hi << 16); be.l = e.h == 0.f ? -1.f : e.l;
      T->x[s] = be;
      if (s == 0) x0 = e.lo;
      x1 = e.hi;
    }
    h.xw = (uint32_t)x0 | ((uint32_t)x1 << 16);
    h.yw = (uint32_t)y0 | ((uint32_t)y1 << 16);
    if ((x1 | 3) - (x0 & ~3) + 1 > kRingCols) h.flags = kBandFlagGather;      // a window wider than the ring
    // the band of fpn.hip's visiting order: centre row of the box in feature rows of its level >> band_log2
    const uint32_t yc = (uint32_t)fminf(fmaxf((raw.d0.z + raw.d1.x) * 0.5f, 0.f), 65535.f);
    const uint32_t band = min((yc >> pp.fs[hd.lvl]) >> pp.band_log2, 63u);
    h.key = ((uint32_t)min(hd.b, 0xfffe) << 16) | ((uint32_t)hd.lvl << 8) | band;
  }
  T->h = h;
}

// ---- prep 2: band items ------------------------------------------------------------------------------------------------------
constexpr int kItemsPer = 16;       // RoIs per thread: n_rois <= 16384

__global__ __launch_bounds__(kBandThreads) void band_items_kernel(int n_rois, BandWs ws, int rows_cap) {
  __shared__ int s_cnt[kBandThreads];
  __shared__ int s_total;
  const int tid = threadIdx.x;
  BandRoiTab* tab = ws.tab;
  const int per = (n_rois + kBandThreads - 1) / kBandThreads;
  const int i0 = min(tid * per, n_rois), i1 = min(i0 + per, n_rois);
  // pass 1: run starts of my chunk (bit j of `starts`)
  uint32_t starts = 0;
  uint32_t kprev = i0 > 0 && i0 < n_rois ? tab[i0 - 1].h.key : 0;
  for (int i = i0; i < i1; i++) {
    const uint32_t k = tab[i].h.key;
    if (i == 0 || k != kprev) starts |= 1u << (i - i0);
    kprev = k;
  }
  const int mine = __popc(starts);
  s_cnt[tid] = mine;
  __syncthreads();
  // exclusive scan over 1024 counts (Hillis-Steele in LDS; this kernel is latency-, not throughput-bound)
  for (int o = 1; o < kBandThreads; o <<= 1) {
    const int v = tid >= o ? s_cnt[tid - o] : 0;
    __syncthreads();
    s_cnt[tid] += v;
    __syncthreads();
  }
  const int base = s_cnt[tid] - mine;
  if (tid == kBandThreads - 1) s_total = s_cnt[tid];
  // pass 2: item heads
  {
    int j = base;
    for (int i = i0; i < i1; i++) {
      if (starts & (1u << (i - i0))) {
        const BandHdr h = tab[i].h;
        BandItem it;
        it.first = i; it.count = 0; it.b = (int)(h.key >> 16); it.lvl = (int)((h.key >> 8) & 0xff); it.rbase = 0; it.rows = 0;
        it.kind = (h.flags & kBandFlagZero) ? kItemZero : kItemPool; it.pad = 0;
        ws.items[j] = it;
        ws.imin[j] = 0x7fffffff; ws.imax[j] = -1;
        j++;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  const int n_items = s_total;
  // pass 3: row range of every item (RoIs in front of my first start continue the previous thread's last item)
  {
    int j = base - 1;
    for (int i = i0; i < i1; i++) {
      if (starts & (1u << (i - i0))) j++;
      const BandHdr h = tab[i].h;
      tab[i].h.item = j;
      // (a window that cannot fit the LDS image anyway must not drag the item's first row away from the others)
      const int wy0 = (int)(h.yw & 0xffff), wy1 = (int)(h.yw >> 16);
      if (h.flags == 0 && wy1 - wy0 + 1 <= rows_cap) {
        atomicMin(&ws.imin[j], wy0);
        atomicMax(&ws.imax[j], wy1);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int j = tid; j < n_items; j += kBandThreads) {
    const int first = ws.items[j].first;
    const int next = j + 1 < n_items ? ws.items[j + 1].first : n_rois;
    ws.items[j].count = next - first;
    // (atomic loads: the minima / maxima were formed by read-modify-writes in the L2)
    const int lo = __hip_atomic_load(&ws.imin[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int hi = __hip_atomic_load(&ws.imax[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ws.items[j].rbase = lo == 0x7fffffff ? 0 : lo;
    ws.items[j].rows = lo == 0x7fffffff ? 1 : min(hi - lo + 1, rows_cap);
  }
  __threadfence_block();
  __syncthreads();
  // pass 4: RoIs whose window reaches below the rows the ring holds (or is wider than it) take the gather path: each
  // becomes an item of its own behind the band items (the sweep skips it)
  if (tid == 0) s_total = 0;
  __syncthreads();
  for (int i = i0; i < i1; i++) {
    const BandHdr h = tab[i].h;
    if (h.flags & kBandFlagZero) continue;
    const BandItem bi = ws.items[h.item];
    uint32_t fl = h.flags;
    if ((int)(h.yw >> 16) - bi.rbase >= rows_cap || (int)(h.yw & 0xffff) < bi.rbase) fl |= kBandFlagGather;
    if (fl & kBandFlagGather) {
      tab[i].h.flags = fl;
      BandItem g = bi;
      g.first = i; g.count = 1; g.kind = kItemGather;
      ws.items[n_items + atomicAdd(&s_total, 1)] = g;
    }
  }
  __syncthreads();
  const int n_gather = s_total;
  // pass 5: eight contiguous item slices of (about) equal RoI count, one per XCD; reset the work counters
  if (tid <= 8) {
    const long long target = ((long long)n_rois * tid) / 8;
    int lo = 0, hi = n_items;                       // first item whose first RoI >= target
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ws.items[mid].first >= target) hi = mid; else lo = mid + 1; }
    s_cnt[tid] = tid == 8 ? n_items : lo;
  }
  __syncthreads();
  if (tid < 8) {
    ws.ctl->slice_first[tid] = s_cnt[tid];
    ws.ctl->slice_count[tid] = s_cnt[tid + 1] - s_cnt[tid];
    ws.ctl->ctr[tid] = 0;
  }
  if (tid == 0) {
    ws.ctl->slice_first[kXcds] = n_items; ws.ctl->slice_count[kXcds] = n_gather; ws.ctl->ctr[kXcds] = 0;
    ws.ctl->n_items = n_items; ws.ctl->n_gather = n_gather;
  }
}

// ---- the sweep -----------------------------------------------------------------------------------------------------------------
template <typename TOut> __device__ __forceinline__ void band_store4(TOut* d, float4 v);
template <> __device__ __forceinline__ void band_store4<float>(float* d, float4 v) { *reinterpret_cast<float4*>(d) = v; }
template <> __device__ __forceinline__ void band_store4<__half>(__half* d, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  uint2 r; r.x = *reinterpret_cast<const uint32_t*>(&a); r.y = *reinterpret_cast<const uint32_t*>(&b);
  *reinterpret_cast<uint2*>(d) = r;
}
template <> __device__ __forceinline__ void band_store4<bf16_t>(bf16_t* d, float4 v) {
  uint2 r;
  r.x = (uint32_t)from_f32<bf16_t>(v.x).bits | ((uint32_t)from_f32<bf16_t>(v.y).bits << 16);
  r.y = (uint32_t)from_f32<bf16_t>(v.z).bits | ((uint32_t)from_f32<bf16_t>(v.w).bits << 16);
  *reinterpret_cast<uint2*>(d) = r;
}

struct BandPlan { int i0, n, kind, xa, xb; };        // kind: 0 pool from the ring, 1 skip one RoI (it is an item of its own), -1 none
struct BandPre { uint32_t xw, fl; };

__device__ __forceinline__ BandPre band_prefetch(const BandRoiTab* tab, int i0, int i_end, int kmax, int lane) {
  BandPre pr; pr.xw = 0; pr.fl = 0;
  const int ci = i0 + lane;
  if (lane < kmax && ci < i_end) {
    const uint4 h = *reinterpret_cast<const uint4*>(&tab[ci].h);
    pr.xw = h.x; pr.fl = h.w;
  }
  return pr;
}

// The next batch: the longest run of RoIs from i0 whose windows fit the ring together.  Every wavefront computes it for
// itself from the same records (uniform result, no LDS round trip, no barrier).
__device__ __forceinline__ BandPlan band_plan(const BandPre& pr, int i0, int i_end, int kmax, int lane) {
  BandPlan pl; pl.i0 = i0; pl.n = 0; pl.kind = -1; pl.xa = 0; pl.xb = -1;
  if (i0 >= i_end) return pl;
  const bool valid = lane < kmax && i0 + lane < i_end;
  int mn = valid ? (int)(pr.xw & 0xffff) & ~3 : 0x7fffffff;
  int mx = valid ? (int)(pr.xw >> 16) | 3 : -1;
#pragma unroll
  for (int o = 1; o < kBandMaxK; o <<= 1) {          // inclusive prefix min / max over the lanes
    const int a = __shfl_up(mn, o, 64), b = __shfl_up(mx, o, 64);
    if (lane >= o) { mn = min(mn, a); mx = max(mx, b); }
  }
  const bool ok = valid && pr.fl == 0 && mx - mn + 1 <= kRingCols;
  const uint64_t m = __ballot(ok);
  const int n = m == ~0ull ? 64 : __builtin_ctzll(~m);
  if (n == 0) {
    pl.n = 1; pl.kind = 1;
    return pl;
  }
  pl.n = n; pl.kind = 0;
  pl.xa = __builtin_amdgcn_readlane(mn, n - 1);
  pl.xb = __builtin_amdgcn_readlane(mx, n - 1);
  return pl;
}

template <typename TOut, int NQ>
__global__ __launch_bounds__(kBandThreads, 1) void roi_align_fwd_band(RoiAlignParams p, BandWs ws, int rows_cap, int kmax, int max_units, int persist_from) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CG = 4 * NQ;
  constexpr int kRowBytes = NQ * kRowQuadBytes;
  __shared__ int s_unit;
  __shared__ int s_r[kBandMaxK];
  const int tid = threadIdx.x, lane = tid & 63, wv = buni(tid >> 6);     // wave index: uniform, so the per-unit row / quad arithmetic is scalar
  const int bins = p.pooled_h * p.pooled_w;
  char* ring = reinterpret_cast<char*>(smem);                                  // [rows_cap][NQ][kRingSlots][4 channels] float32
  float* slab = reinterpret_cast<float*>(smem + (size_t)rows_cap * kRowBytes);   // [kmax][CG][bins]
  const int ncg = p.channels / CG;
  const int xcd = blockIdx.x & (kXcds - 1);
  const BandRoiTab* __restrict__ tab = ws.tab;
  TOut* out = reinterpret_cast<TOut*>(p.out);
  // lane <-> (RoI of the batch, bin)
  const float rbinv = 1.0f / (float)bins;
  const int rl = (int)(((float)tid + 0.5f) * rbinv);                           // exact: tid < 2^10
  const int bin = tid - rl * bins;
  const float rpw = 1.0f / (float)p.pooled_w;
  const int ph = (int)(((float)bin + 0.5f) * rpw), pw = bin - ph * p.pooled_w;
  const int g16 = lane & 15, cl = lane >> 4;

  // (workgroups >= persist_from -- the last ones dispatched -- stay until the queues are empty whatever max_units says)
  for (int done = 0; max_units <= 0 || done < max_units || (int)blockIdx.x >= persist_from; done++) {
    // ---- next unit: own XCD's queue first, then the others' ----------------------------------------------------------------
    if (tid == 0) {
      int found = -1;
      for (int k = 0; k < kSlices && found < 0; k++) {
        const int x = k < kXcds ? (xcd + k) & (kXcds - 1) : kXcds;
        const int n = ws.ctl->slice_count[x] * ncg;
        if (n <= 0) continue;
        if (__hip_atomic_load(&ws.ctl->ctr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) continue;
        const int u = atomicAdd(&ws.ctl->ctr[x], 1);
        if (u < n) found = (x << 24) | u;
      }
      s_unit = found;
    }
    __syncthreads();
    const int unit = buni(s_unit);
    __syncthreads();
    if (unit < 0) break;
    const int ux = unit >> 24, uu = unit & 0xffffff;
    const int scount = buni(ws.ctl->slice_count[ux]);
    const int cg = uu / scount;
    const BandItem it = ws.items[buni(ws.ctl->slice_first[ux]) + (uu - cg * scount)];
    const int c0 = cg * CG;
    const int first = buni(it.first), i_end = first + buni(it.count);
    if (buni(it.kind) == kItemZero) {        // padding rows of a fixed-shape batch (level -1): defined output
      const int per = CG * bins;
      for (int o = tid; o < buni(it.count) * per; o += kBandThreads) {
        const int k = o / per, e = o - k * per;
        out[((size_t)tab[first + k].h.r * p.channels + c0) * bins + e] = from_f32<TOut>(0.f);
      }
      continue;
    }
    const dtc_feat_level L = p.lv[buni(it.lvl)];
    const int H = L.height, W = L.width;
    const int rbase = buni(it.rbase), rows = buni(it.rows);
    const float* fbase = reinterpret_cast<const float*>(L.data) + (int64_t)buni(it.b) * L.stride_n + (int64_t)c0 * L.stride_c;
    if (buni(it.kind) == kItemGather) {
      // a window the ring cannot hold (wider than 64 columns, or below the rows of its band): per-output gather from global
      // memory, geometry on the fly -- the reference's loop for one (RoI, channel, bin) per thread
      const RoiHead hd = load_roi_head(p, first);
      TOut* og = out + ((size_t)hd.r * p.channels + c0) * bins;
      for (int o = tid; o < CG * bins; o += kBandThreads) {
        const int c = o / bins, gb = o - c * bins;
        const int gph = gb / p.pooled_w, gpw = gb - gph * p.pooled_w;
        const float* d = fbase + (int64_t)c * L.stride_c;
        float acc = 0.f;
        for (int iy = 0; iy < 2; iy++) {
          const AxisEntry y = make_axis(hd.sh, hd.bin_h, gph, iy, 2, H);
          const int64_t yl0 = (int64_t)y.lo * L.stride_h, yh0 = (int64_t)y.hi * L.stride_h;
          for (int ix = 0; ix < 2; ix++) {
            const AxisEntry x = make_axis(hd.sw, hd.bin_w, gpw, ix, 2, W);
            const int64_t xl0 = (int64_t)x.lo * L.stride_w, xh0 = (int64_t)x.hi * L.stride_w;
            const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
            acc += w1 * d[yl0 + xl0] + w2 * d[yl0 + xh0] + w3 * d[yh0 + xl0] + w4 * d[yh0 + xh0];
          }
        }
        og[o] = from_f32<TOut>(acc * 0.25f);
      }
      continue;
    }
    const bool vec = L.stride_w == 1 && (W & 3) == 0 && ((L.stride_h | L.stride_c | L.stride_n) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(L.data) & 15) == 0 && L.stride_h > 0 && L.stride_c > 0 &&
                     L.stride_h * (int64_t)H + L.stride_c * 4 * NQ < (1ll << 28);
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fbase), 0, 0xffffffff, 0x00020000);
    const int sh32 = (int)L.stride_h, sc32 = (int)L.stride_c;
    const int units = NQ * rows;             // wave-level load units: (quad, row) x 16 column groups x 4 channels

    int res_a = 0, res_b = 0;                // columns [res_a, res_b) are resident in the ring
    float4 v[kBandUnits];
    uint32_t voff = 0;                       // per-lane byte offset of the batch's piece (channel cl, column group)
    int pcol = 0;                            // first column of that piece
    int ldl = 0;                             // per-lane LDS byte offset inside a (row, quad) line
    bool live = false;                       // this lane's column group is a new one (not a duplicate)
    int new_groups = 0;
    uint4 ty = make_uint4(0, 0, 0, 0), tx = make_uint4(0, 0, 0, 0);
    int tr = 0;

    // columns of [xa, xb] the ring does not hold: at most two runs, left (gA groups) and right (gB groups) of what stays.
    // ALWAYS defines every piece register (zeros when there is nothing to load): a piece register that is written under a
    // condition becomes a loop-carried value and the register allocator spills all 32 of them.
    auto issue = [&](const BandPlan& pl) {
      const int keep_a = max(pl.xa, res_a), keep_b = min(pl.xb + 1, res_b);
      const bool ov = keep_b > keep_a;
      const int ngx = (pl.xb + 1 - pl.xa) >> 2;
      const int gA = ov ? (keep_a - pl.xa) >> 2 : ngx, gB = ov ? (pl.xb + 1 - keep_b) >> 2 : 0;
      new_groups = pl.kind == 0 ? gA + gB : 0;
      const int ge = min(g16, max(new_groups, 1) - 1);
      live = g16 < new_groups;
      const int col = ge < gA ? pl.xa + 4 * ge : keep_b + 4 * (ge - gA);
      voff = (uint32_t)(cl * sc32 + col) * 4u;
      pcol = col;
      ldl = ring_phys(col) * 16 + cl * 4;
      if (vec && new_groups > 0) {
        // straight-line: units past the end repeat the last one (same bytes, an L1 hit; never committed).  `rows` is made opaque
        // so that the eight (quad, row) offsets are scalar arithmetic HERE instead of loop-invariant values kept (spilled)
        // across the sweep.
        int rows_o = rows;
        asm volatile("" : "+s"(rows_o));
#pragma unroll
        for (int k = 0; k < kBandUnits; k++) {
          const int u = min(wv + kBandWaves * k, NQ * rows_o - 1);
          const int q = NQ == 1 ? 0 : (NQ == 2 ? (u >= rows_o ? 1 : 0) : u / rows_o);
          const int row = u - q * rows_o;
          const int frow = min(rbase + row, H - 1);
          const uint32_t soff = (uint32_t)(4 * q * sc32 + frow * sh32) * 4u;
          const bu32x4 w = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0);
          v[k] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
        }
      } else {
#pragma unroll
        for (int k = 0; k < kBandUnits; k++) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto commit = [&](const BandPlan& pl) {
      if (new_groups == 0) { res_a = pl.xa; res_b = pl.xb + 1; return; }
      if (vec) {
        int ldc = ldl, rows_o = rows;
        asm volatile("" : "+v"(ldc), "+s"(rows_o));   // the store addresses are formed HERE, not hoisted above the pooling loop
#pragma unroll
        for (int k = 0; k < kBandUnits; k++) {
          const int u = wv + kBandWaves * k;
          if (u < NQ * rows_o) {
            const int q = NQ == 1 ? 0 : (NQ == 2 ? (u >= rows_o ? 1 : 0) : u / rows_o);
            const int row = u - q * rows_o;
            if (live) {
              float* d = reinterpret_cast<float*>(ring + (row * NQ + q) * kRowQuadBytes + ldc);
              d[0] = v[k].x; d[4] = v[k].y; d[8] = v[k].z; d[12] = v[k].w;
            }
          }
        }
      } else {
        // strided columns / unaligned rows (channels_last maps, widths that are not a multiple of 4): clamped scalar loads
        // straight into LDS.  Correct for any strides; not a fast path.
        const int col = pcol;
        for (int u = wv; u < units; u += kBandWaves) {
          const int q = u / rows, row = u - q * rows;
          const int frow = min(rbase + row, H - 1);
          if (live) {
            const float* s = fbase + (int64_t)(4 * q + cl) * L.stride_c + (int64_t)frow * L.stride_h;
            float* d = reinterpret_cast<float*>(ring + (row * NQ + q) * kRowQuadBytes + ldl);
            d[0] = s[(int64_t)min(col, W - 1) * L.stride_w]; d[4] = s[(int64_t)min(col + 1, W - 1) * L.stride_w];
            d[8] = s[(int64_t)min(col + 2, W - 1) * L.stride_w]; d[12] = s[(int64_t)min(col + 3, W - 1) * L.stride_w];
          }
        }
      }
      res_a = pl.xa; res_b = pl.xb + 1;
    };
    auto load_tables = [&](const BandPlan& pl) {
      if (rl < pl.n) {
        const BandRoiTab* T = tab + pl.i0 + rl;
        ty = *reinterpret_cast<const uint4*>(&T->y[2 * ph]);
        tx = *reinterpret_cast<const uint4*>(&T->x[2 * pw]);
        tr = T->h.r;
      }
    };

    BandPre pre = band_prefetch(tab, first, i_end, kmax, lane);
    BandPlan cur = band_plan(pre, first, i_end, kmax, lane);
    pre = band_prefetch(tab, cur.i0 + cur.n, i_end, kmax, lane);
    issue(cur);
    if (cur.kind == 0) { load_tables(cur); commit(cur); }
    __syncthreads();

    while (cur.kind >= 0) {
      const BandPlan nxt = band_plan(pre, cur.i0 + cur.n, i_end, kmax, lane);
      pre = band_prefetch(tab, nxt.i0 + nxt.n, i_end, kmax, lane);
      // ---- this lane's (RoI, bin) of the current batch: tap offsets and weights from the record ----------------------------
      const bool on = cur.kind == 0 && rl < cur.n;
      int a[2][2][4];
      float yl[2], yh[2], xl[2], xh[2];
      {
        const uint32_t ye[2] = {ty.x, ty.z}, xe[2] = {tx.x, tx.z};
        const float yf[2] = {__uint_as_float(ty.y), __uint_as_float(ty.w)}, xf[2] = {__uint_as_float(tx.y), __uint_as_float(tx.w)};
        int ylo[2], yhi[2], xlo[2], xhi[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const bool yv = yf[i] >= 0.f, xv = xf[i] >= 0.f;
          yl[i] = yv ? yf[i] : 0.f; yh[i] = yv ? (float)(1.0 - (double)yf[i]) : 0.f;      // roi_align_cpu_loop.cpp:92-94
          xl[i] = xv ? xf[i] : 0.f; xh[i] = xv ? (float)(1.0 - (double)xf[i]) : 0.f;
          ylo[i] = min(max((int)(ye[i] & 0xffff) - rbase, 0), rows - 1) * kRowBytes;
          yhi[i] = min(max((int)(ye[i] >> 16) - rbase, 0), rows - 1) * kRowBytes;
          xlo[i] = ring_phys((int)(xe[i] & 0xffff)) << 4; xhi[i] = ring_phys((int)(xe[i] >> 16)) << 4;
        }
#pragma unroll
        for (int iy = 0; iy < 2; iy++)
#pragma unroll
          for (int ix = 0; ix < 2; ix++) {
            int t0 = ylo[iy] + xlo[ix], t1 = ylo[iy] + xhi[ix], t2 = yhi[iy] + xlo[ix], t3 = yhi[iy] + xhi[ix];
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));   // one finished VGPR per tap
            a[iy][ix][0] = t0; a[iy][ix][1] = t1; a[iy][ix][2] = t2; a[iy][ix][3] = t3;
          }
        if (on && bin == 0) s_r[rl] = tr;
      }
      issue(nxt);                             // in flight (registers) while this batch is pooled

      if (on) {
        float* so = slab + rl * (CG * bins) + bin;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const char* wq = ring + q * kRowQuadBytes;
          bf32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
          // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
          for (int iy = 0; iy < 2; iy++)
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
              bf32x4 t[4];
#pragma unroll
              for (int k = 0; k < 4; k++)
                t[k] = *reinterpret_cast<const bf32x4*>(__builtin_assume_aligned(wq + a[iy][ix][k], 16));
              const float w1 = yh[iy] * xh[ix], w2 = yh[iy] * xl[ix];               // roi_align_cpu_loop.cpp:95
              const float w3 = yl[iy] * xh[ix], w4 = yl[iy] * xl[ix];
              a01 += w1 * t[0].lo + w2 * t[1].lo + w3 * t[2].lo + w4 * t[3].lo;    // :208-211
              a23 += w1 * t[0].hi + w2 * t[1].hi + w3 * t[2].hi + w4 * t[3].hi;
              __builtin_amdgcn_sched_barrier(0);      // one sample's taps live at a time
            }
          // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32
          float* o = so + 4 * q * bins;
          o[0] = a01.x * 0.25f; o[bins] = a01.y * 0.25f; o[2 * bins] = a23.x * 0.25f; o[3 * bins] = a23.y * 0.25f;
        }
      }
      if (nxt.kind == 0) load_tables(nxt);    // in flight across the barrier, the commit and the slab stores
      __syncthreads();
      if (nxt.kind == 0) commit(nxt);
      if (cur.kind == 0) {
        // slab [RoI][CG][bins] is contiguous per RoI exactly like the [R, C, PH, PW] output: 16-byte stores
        const int n4 = (CG * bins) >> 2, total = cur.n * n4;
        const float r4 = 1.0f / (float)n4;
        for (int idx = tid; idx < total; idx += kBandThreads) {
          const int k = (int)(((float)idx + 0.5f) * r4);            // exact for idx < 2^13
          const int e = idx - k * n4;
          const float4 val = reinterpret_cast<const float4*>(slab)[idx];
          band_store4<TOut>(out + ((size_t)s_r[k] * p.channels + c0) * bins + 4 * e, val);
        }
      }
      __syncthreads();
      cur = nxt;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct BandConfig {
  int enabled = 1;
  int rows_cap = 0;    // 0: what the LDS holds next to the slab
  int kmax = 0;        // RoIs per batch (0: threads / bins, at most 20)
  int grid = 0;        // workgroups (0: one per CU)
  int max_units = 0;   // units a workgroup takes before it leaves (0: until the queues are empty)
};
static BandConfig& band_config() {
  static BandConfig cfg = [] {
    BandConfig c;
    if (const char* e = getenv("DTC_RA_BAND")) c.enabled = atoi(e) != 0;
    if (const char* e = getenv("DTC_RA_BAND_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 128) c.rows_cap = v; }
    if (const char* e = getenv("DTC_RA_BAND_K")) { const int v = atoi(e); if (v >= 1 && v <= kBandMaxK) c.kmax = v; }
    if (const char* e = getenv("DTC_RA_BAND_GRID")) { const int v = atoi(e); if (v >= 1) c.grid = v; }
    if (const char* e = getenv("DTC_RA_BAND_MAXUNITS")) { const int v = atoi(e); if (v >= 0) c.max_units = v; }
    return c;
  }();
  return cfg;
}

static size_t band_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t band_ws_bytes(int n_rois) {
  const size_t n = (size_t)(n_rois > 0 ? n_rois : 0) + 1;
  return band_align(sizeof(BandCtl)) + band_align(2 * n * sizeof(BandItem)) + 2 * band_align(n * sizeof(int)) + band_align(n * sizeof(BandRoiTab));
}

bool roi_align_band_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  if (!band_config().enabled) return false;
  if (p.sampling_ratio != 2 || !p.roi_desc) return false;
  if (p.pooled_h > kBandMaxPooled || p.pooled_w > kBandMaxPooled || p.pooled_h * p.pooled_w > 64) return false;
  if ((p.channels & 7) != 0 || p.n_rois > kBandThreads * kItemsPer) return false;
  if (in_dtype != DTC_F32) return false;
  if (out_dtype != DTC_F32 && out_dtype != DTC_F16 && out_dtype != DTC_BF16) return false;
  for (int l = 0; l < p.n_levels; l++) {
    if (p.lv[l].height > 65535 || p.lv[l].width > 65535) return false;
    if (p.lv[l].stride_c == 1 && p.channels > 1) return false;       // channels_last maps: roi_align_fwd_nhwc
  }
  return true;
}

template <typename TOut>
static int launch_band_t(const RoiAlignParams& p, const BandWs& ws, hipStream_t stream) {
  constexpr int NQ = 2;
  const BandConfig& cfg = band_config();
  const int bins = p.pooled_h * p.pooled_w;
  int kmax = cfg.kmax ? cfg.kmax : kBandThreads / bins;
  if (kmax > 20) kmax = 20;
  if (kmax * bins > kBandThreads) kmax = kBandThreads / bins;
  const int lds_total = 160 * 1024 - 1024;          // static __shared__ of the kernel comes on top
  const int slab_b = kmax * 4 * NQ * bins * 4;
  int rows_cap = (lds_total - slab_b) / (NQ * kRowQuadBytes);
  if (cfg.rows_cap && cfg.rows_cap < rows_cap) rows_cap = cfg.rows_cap;
  if (rows_cap * NQ > kBandUnits * kBandWaves) rows_cap = kBandUnits * kBandWaves / NQ;      // what the register pipeline carries
  if (rows_cap < 8) return DTC_EUNSUPPORTED;
  BandPrepParams pp;
  for (int l = 0; l < DTC_MAX_LEVELS; l++) pp.fs[l] = 0;
  for (int l = 0; l < p.n_levels; l++) {
    int fs = 0;
    float s = p.lv[l].spatial_scale;
    while (s > 0.f && s < 0.75f && fs < 15) { s *= 2.f; fs++; }         // log2 of the feature stride
    pp.fs[l] = fs;
  }
  pp.band_log2 = kVisitBandLog2Sweep;
  hipLaunchKernelGGL(band_tab_kernel, dim3((unsigned)((p.n_rois + 63) / 64)), dim3(64), 0, stream, p, pp, ws.tab);
  DTC_CHECK_LAUNCH();
  hipLaunchKernelGGL(band_items_kernel, dim3(1), dim3(kBandThreads), 0, stream, p.n_rois, ws, rows_cap);
  DTC_CHECK_LAUNCH();
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int n_cu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_band<TOut, NQ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cu = n;
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  const int lds_b = rows_cap * NQ * kRowQuadBytes + slab_b;
  const int grid = cfg.grid ? cfg.grid : n_cu;
  hipLaunchKernelGGL((roi_align_fwd_band<TOut, NQ>), dim3((unsigned)grid), dim3(kBandThreads), lds_b, stream, p, ws, rows_cap, kmax, cfg.max_units, grid > n_cu ? grid - n_cu : 0);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

int launch_roi_align_band(const RoiAlignParams& p, int in_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  if (!workspace || workspace_bytes < band_ws_bytes(p.n_rois) || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return DTC_EWORKSPACE;
  const size_t n = (size_t)p.n_rois + 1;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  BandWs ws;
  ws.ctl = reinterpret_cast<BandCtl*>(w); w += band_align(sizeof(BandCtl));
  ws.items = reinterpret_cast<BandItem*>(w); w += band_align(2 * n * sizeof(BandItem));    // band items + gather items
  ws.imin = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.imax = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.tab = reinterpret_cast<BandRoiTab*>(w);
  if (out_dtype == DTC_F32) return launch_band_t<float>(p, ws, stream);
  if (out_dtype == DTC_F16) return launch_band_t<__half>(p, ws, stream);
  if (out_dtype == DTC_BF16) return launch_band_t<bf16_t>(p, ws, stream);
  return DTC_EUNSUPPORTED;
}

size_t roi_align_band_workspace_bytes(int n_rois) { return band_ws_bytes(n_rois); }

}  // namespace dtc

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------
DTC_API size_t dtc_roi_align_band_workspace_bytes(int n_rois) { return dtc::roi_align_band_workspace_bytes(n_rois); }

DTC_API int dtc_roi_align_forward_banded(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                         const float* roi_desc, int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                         void* out, int out_dtype, void* workspace, size_t workspace_bytes, dtc_stream_t stream) {
  if (!levels || n_levels < 1 || n_levels > DTC_MAX_LEVELS || channels < 1 || n_rois < 0 || pooled_h < 1 || pooled_w < 1 ||
      (n_rois > 0 && (!roi_desc || !out)))
    return DTC_EINVAL;
  dtc::RoiAlignParams p;
  memset(&p, 0, sizeof(p));
  for (int i = 0; i < n_levels; i++) {
    if (!levels[i].data || levels[i].height < 1 || levels[i].width < 1) return DTC_EINVAL;
    p.lv[i] = levels[i];
  }
  p.roi_desc = roi_desc; p.out = out; p.n_levels = n_levels; p.channels = channels; p.roi_cols = 5; p.n_rois = n_rois;
  p.pooled_h = pooled_h; p.pooled_w = pooled_w; p.sampling_ratio = sampling_ratio;
  if (workspace && dtc::roi_align_band_supported(p, in_dtype, out_dtype))
    return dtc::launch_roi_align_band(p, in_dtype, out_dtype, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  // anything the sweep does not cover (other sampling ratios / bin counts / dtypes, channels_last maps): the packed entry
  return dtc_roi_align_forward_packed(levels, n_levels, channels, in_dtype, roi_desc, n_rois, pooled_h, pooled_w, sampling_ratio,
                                      out, out_dtype, stream);
}
