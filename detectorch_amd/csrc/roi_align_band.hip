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
// every row of the ring starts 5 slots further round the 16 bank-slots of a ds_read_b128: without the skew the slot of a tap
// depends on its column only and the lanes of one bin COLUMN always collide (12.6 LDS cycles per read on the bench
// distribution, 9.4 with it; 4 are conflict-free -- tools/r03/lds_taps.py)
constexpr int kRowSkewBytes = 5 * 16;
constexpr int kBandUnits = 8;                       // 16-byte row pieces a thread carries per batch
constexpr int kBandMaxK = 32;                       // RoIs per batch, upper bound (the batching scan covers 32 lanes)
constexpr int kBandMaxPooled = 8;                   // axis samples per record: 2 x 8
constexpr int kBandMaxRois = 16384;
constexpr int kLdsItems = 2048;                     // band items whose bookkeeping fits the prep kernel's LDS (more: global arrays)

enum { kBandFlagGather = 1, kBandFlagZero = 2 };
enum { kItemPool = 0, kItemZero = 1, kItemGather = 2 };

// One axis sample of pre_calc_for_bilinear_interpolate (roi_align_cpu_loop.cpp:36-95), ready for the pooling lane: for x the
// LDS byte offsets of the two columns inside a (row, quad) line of the ring, for y the two feature rows; weights l, h = 1 - l
// (both 0 for a sample outside the map, :49-63).
struct BandAxis { uint32_t lo, hi; float l, h; };   // 16 B
struct BandRoiTab { int32_t r, pad[3]; BandAxis y[2 * kBandMaxPooled]; BandAxis x[2 * kBandMaxPooled]; };
static_assert(sizeof(BandRoiTab) == 528, "BandRoiTab layout");
struct BandItem { int first, count, b, lvl, rbase, rows, kind, nbatch; };   // 32 B
struct BandBatch { int i0, n, xa, xb; };            // RoIs [i0, i0 + n) of the table; their windows' columns [xa, xb] (4-aligned)
constexpr int kSlices = kXcds + 1;                  // one item slice per XCD + the RoIs that take the gather path
struct BandCtl { int n_items, n_gather, pad0[6]; int slice_first[kSlices + 7]; int slice_count[kSlices + 7]; int ctr[kSlices + 7]; };
struct BandWs {
  BandCtl* ctl; BandItem* items; BandBatch* batches;
  uint32_t *key, *xw, *yw, *fl;                     // per RoI, structure of arrays (the prep kernel reads them coalesced)
  int *imin, *imax, *ifst, *ikey;                   // per item (used when there are more than kLdsItems items)
  BandRoiTab* tab;
};

typedef uint32_t bu32x4 __attribute__((ext_vector_type(4)));
typedef float bf32x2 __attribute__((ext_vector_type(2)));
typedef float bf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int buni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int ring_phys(int col) { const int c = col & (kRingCols - 1); return c + (c >> 3); }

// Development aid (-DDTC_BAND_TRACE, tools/r03/band_trace.py): thread 0 of every workgroup accumulates the shader clock per
// phase of the sweep in LDS and adds the sums to a global table at the end.  Compiled out of the product.
#ifdef DTC_BAND_TRACE
constexpr int kBtSlots = 16;
__device__ unsigned long long g_band_trace[kBtSlots];
#define BT_DECL __shared__ unsigned int s_bt[dtc::kBtSlots]; unsigned long long bt_last = 0;
#define BT_INIT do { if (threadIdx.x < dtc::kBtSlots) s_bt[threadIdx.x] = 0; bt_last = __builtin_readcyclecounter(); } while (0)
#define BT(i) do { if (threadIdx.x == 0) { const unsigned long long n__ = __builtin_readcyclecounter(); s_bt[i] += (unsigned int)(n__ - bt_last); bt_last = n__; } } while (0)
#define BT_FLUSH do { __syncthreads(); if (threadIdx.x < dtc::kBtSlots) atomicAdd(&dtc::g_band_trace[threadIdx.x], (unsigned long long)s_bt[threadIdx.x]); } while (0)
#else
#define BT_DECL
#define BT_INIT ((void)0)
#define BT(i) ((void)0)
#define BT_FLUSH ((void)0)
#endif

// inclusive prefix min / max over lanes 0..31 with DPP row shifts (ALU speed; ds_bpermute shuffles cost an LDS round trip each)
template <int CTRL> __device__ __forceinline__ int band_dpp(int v, int identity) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ void band_scan32(int& mn, int& mx) {
  constexpr int kMaxI = 0x7fffffff, kMinI = -0x7fffffff - 1;
  mn = min(mn, band_dpp<0x111>(mn, kMaxI)); mx = max(mx, band_dpp<0x111>(mx, kMinI));     // row_shr:1
  mn = min(mn, band_dpp<0x112>(mn, kMaxI)); mx = max(mx, band_dpp<0x112>(mx, kMinI));     // row_shr:2
  mn = min(mn, band_dpp<0x114>(mn, kMaxI)); mx = max(mx, band_dpp<0x114>(mx, kMinI));     // row_shr:4
  mn = min(mn, band_dpp<0x118>(mn, kMaxI)); mx = max(mx, band_dpp<0x118>(mx, kMinI));     // row_shr:8
  // lanes 16..31 take lane 15's prefix on top (row_bcast:15 reaches every following row; only row 1 matters here)
  mn = min(mn, __builtin_amdgcn_update_dpp(kMaxI, mn, 0x142, 0xa, 0xf, false));
  mx = max(mx, __builtin_amdgcn_update_dpp(kMinI, mx, 0x142, 0xa, 0xf, false));
}

// ---- prep 1: per-RoI records -------------------------------------------------------------------------------------------------
struct BandPrepParams { int fs[DTC_MAX_LEVELS]; int band_log2; };

__global__ __launch_bounds__(64) void band_tab_kernel(RoiAlignParams p, BandPrepParams pp, BandWs ws) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= p.n_rois) return;
  const RoiRaw raw = load_roi_raw(p, i);
  const RoiHead hd = roi_head_from_raw(p, raw);
  BandRoiTab* T = ws.tab + i;
  T->r = hd.r;
  uint32_t key = 0xffffffffu, xw = 0, yw = 0, fl = 0;
  if (hd.lvl < 0 || hd.lvl >= p.n_levels) {
    fl = kBandFlagZero;
  } else {
    const int H = p.lv[hd.lvl].height, W = p.lv[hd.lvl].width;
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    for (int s = 0; s < 2 * p.pooled_h; s++) {
      const AxisEntry e = make_axis(hd.sh, hd.bin_h, s >> 1, s & 1, 2, H);
      BandAxis a; a.lo = (uint32_t)e.lo; a.hi = (uint32_t)e.hi; a.l = e.l; a.h = e.h;
      T->y[s] = a;
      if (s == 0) y0 = e.lo;
      y1 = e.hi;
    }
    for (int s = 0; s < 2 * p.pooled_w; s++) {
      const AxisEntry e = make_axis(hd.sw, hd.bin_w, s >> 1, s & 1, 2, W);
      BandAxis a; a.lo = (uint32_t)(ring_phys(e.lo) << 4); a.hi = (uint32_t)(ring_phys(e.hi) << 4); a.l = e.l; a.h = e.h;
      T->x[s] = a;
      if (s == 0) x0 = e.lo;
      x1 = e.hi;
    }
    xw = (uint32_t)x0 | ((uint32_t)x1 << 16);
    yw = (uint32_t)y0 | ((uint32_t)y1 << 16);
    if ((x1 | 3) - (x0 & ~3) + 1 > kRingCols) fl = kBandFlagGather;      // a window wider than the ring
    // the band of fpn.hip's visiting order: centre row of the box in feature rows of its level >> band_log2
    const uint32_t yc = (uint32_t)fminf(fmaxf((raw.d0.z + raw.d1.x) * 0.5f, 0.f), 65535.f);
    const uint32_t band = min((yc >> pp.fs[hd.lvl]) >> pp.band_log2, 63u);
    key = ((uint32_t)min(hd.b, 0xfffe) << 16) | ((uint32_t)hd.lvl << 8) | band;
  }
  ws.key[i] = key; ws.xw[i] = xw; ws.yw[i] = yw; ws.fl[i] = fl;
}

// ---- prep 2: band items and their batches --------------------------------------------------------------------------------------
// One workgroup.  Everything per RoI sits in LDS (x window 4 B, item id 2 B, class 1 B), everything per item too while there
// are at most kLdsItems of them (IN_LDS; otherwise the item arrays of the workspace -- the two flavours are separate
// instantiations so that the LDS one uses DS atomics, not flat ones); global memory is read coalesced (structure of arrays
// written by band_tab_kernel).
constexpr int kPrepE = kBandMaxRois / kBandThreads;      // RoIs per thread of the prep kernel (strided ownership: coalesced)

template <bool IN_LDS>
__device__ __forceinline__ void band_items_body(int n, const BandWs& ws, int rows_cap, int kmax, int n_items, int my_first_id,
                                                uint32_t* sxw, uint16_t* sitem, uint8_t* sf, int* imin, int* imax, int* ifst,
                                                uint32_t* ikey, int* s_wsum, int* s_ngather, const uint32_t (&ryw)[kPrepE],
                                                const uint32_t (&rfl)[kPrepE]) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + kBandThreads - 1) / kBandThreads;
  const int c0 = min(tid * per, n), c1 = min(c0 + per, n);
  auto ld = [](const int* p) { return IN_LDS ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  {
    int id = my_first_id - 1;
    for (int i = c0; i < c1; i++) {
      if (sf[i]) { id++; ifst[id] = i; imin[id] = 0x7fffffff; imax[id] = -1; ikey[id] = ws.key[i]; }
      sitem[i] = (uint16_t)id;
    }
  }
  __threadfence_block();
  __syncthreads();
  // pass 3: row range of every item (a window that cannot fit the LDS image anyway must not drag the item's first row away)
#pragma unroll
  for (int e = 0; e < kPrepE; e++) {
    const int i = tid + e * kBandThreads;
    if (i < n) {
      const int wy0 = (int)(ryw[e] & 0xffff), wy1 = (int)(ryw[e] >> 16);
      if (rfl[e] == 0 && wy1 - wy0 + 1 <= rows_cap) {
        atomicMin(&imin[sitem[i]], wy0);
        atomicMax(&imax[sitem[i]], wy1);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // pass 4: class of every RoI: 0 pooled from the ring, 2 gather (window wider than the ring, or outside the rows it holds),
  // 3 padding row.  A gather RoI becomes an item of its own behind the band items.
#pragma unroll
  for (int e = 0; e < kPrepE; e++) {
    const int i = tid + e * kBandThreads;
    if (i < n) {
      const uint32_t yw = ryw[e], fl = rfl[e];
      const int id = sitem[i];
      const int lo = ld(&imin[id]);
      const int rbase = lo == 0x7fffffff ? 0 : lo;
      uint8_t cls = 0;
      if (fl & kBandFlagZero) cls = 3;
      else if ((fl & kBandFlagGather) || (int)(yw >> 16) - rbase >= rows_cap || (int)(yw & 0xffff) < rbase) cls = 2;
      sf[i] = cls;
      if (cls == 2) {
        const uint32_t k = IN_LDS ? ikey[id] : __hip_atomic_load(&ikey[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        BandItem g;
        g.first = i; g.count = 1; g.b = (int)(k >> 16); g.lvl = (int)((k >> 8) & 0xff); g.rbase = 0; g.rows = 1; g.kind = kItemGather; g.nbatch = 0;
        ws.items[n_items + atomicAdd(s_ngather, 1)] = g;
      }
    }
  }
  __syncthreads();
  // pass 5: one wavefront per item: its record, and its batches -- the longest runs of poolable RoIs whose windows fit the ring
  // columns together (<= kmax RoIs).  Batch k of an item lives at batches[first + k] (an item has at most `count` batches).
  for (int j = wv; j < n_items; j += kBandWaves) {
    const int first = ld(&ifst[j]), end = j + 1 < n_items ? ld(&ifst[j + 1]) : n;
    const uint32_t k = IN_LDS ? ikey[j] : __hip_atomic_load(&ikey[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lo = ld(&imin[j]), hi = ld(&imax[j]);
    BandItem it;
    it.first = first; it.count = end - first; it.b = (int)(k >> 16); it.lvl = (int)((k >> 8) & 0xff);
    it.rbase = lo == 0x7fffffff ? 0 : lo; it.rows = lo == 0x7fffffff ? 1 : min(hi - lo + 1, rows_cap);
    it.kind = sf[first] == 3 ? kItemZero : kItemPool; it.nbatch = 0;
    if (it.kind == kItemPool) {
      int i = first, nb = 0;
      while (i < end) {
        const int ci = i + lane;
        const bool valid = lane < kmax && ci < end;
        const int cls = valid ? (int)sf[ci] : 1;
        if (buni(cls) != 0) { i++; continue; }           // gather RoI: not part of any batch
        const uint32_t xw = valid ? sxw[ci] : 0;
        int mn = valid ? (int)(xw & 0xffff) & ~3 : 0x7fffffff;
        int mx = valid ? (int)(xw >> 16) | 3 : -0x7fffffff - 1;
        band_scan32(mn, mx);
        const bool ok = valid && cls == 0 && mx - mn + 1 <= kRingCols;
        const uint64_t m = __ballot(ok);
        const int nn = max(1, __builtin_ctzll(~m));        // >= 1 anyway: lane 0 is a poolable RoI whose window fits
        if (lane == nn - 1) { BandBatch bb; bb.i0 = i; bb.n = nn; bb.xa = mn; bb.xb = mx; ws.batches[first + nb] = bb; }
        nb++;
        i += nn;
      }
      it.nbatch = nb;
    }
    if (lane == 0) ws.items[j] = it;
  }
  // eight contiguous item slices of (about) equal RoI count, one per XCD; reset the work counters
  __syncthreads();
  if (tid <= 8) {
    const long long target = ((long long)n * tid) / 8;
    int lo = 0, hi = n_items;                       // first item whose first RoI >= target
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ld(&ifst[mid]) >= target) hi = mid; else lo = mid + 1; }
    s_wsum[tid] = tid == 8 ? n_items : lo;
  }
  __syncthreads();
  if (tid < 8) {
    ws.ctl->slice_first[tid] = s_wsum[tid];
    ws.ctl->slice_count[tid] = s_wsum[tid + 1] - s_wsum[tid];
    ws.ctl->ctr[tid] = 0;
  }
  if (tid == 0) {
    const int n_gather = *s_ngather;
    ws.ctl->slice_first[kXcds] = n_items; ws.ctl->slice_count[kXcds] = n_gather; ws.ctl->ctr[kXcds] = 0;
    ws.ctl->n_items = n_items; ws.ctl->n_gather = n_gather;
  }
}

__global__ __launch_bounds__(kBandThreads) void band_items_kernel(int n, BandWs ws, int rows_cap, int kmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* sxw = reinterpret_cast<uint32_t*>(smem);                        // [n]
  uint16_t* sitem = reinterpret_cast<uint16_t*>(smem + (size_t)n * 4);       // [n]
  uint8_t* sf = smem + (size_t)n * 6;                                       // [n]  pass 1-2: run start; later: class of the RoI
  __shared__ int s_imin[kLdsItems], s_imax[kLdsItems], s_ifst[kLdsItems + 1];
  __shared__ uint32_t s_ikey[kLdsItems];
  __shared__ int s_wsum[kBandWaves];
  __shared__ int s_ngather;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // pass 1: everything this thread needs from global memory, all loads in flight at once (strided ownership: coalesced)
  uint32_t rk[kPrepE], rkp[kPrepE], rxw[kPrepE], ryw[kPrepE], rfl[kPrepE];
#pragma unroll
  for (int e = 0; e < kPrepE; e++) {
    const int i = tid + e * kBandThreads;
    rk[e] = rkp[e] = rxw[e] = ryw[e] = rfl[e] = 0;
    if (i < n) { rk[e] = ws.key[i]; rkp[e] = i > 0 ? ws.key[i - 1] : ~rk[e]; rxw[e] = ws.xw[i]; ryw[e] = ws.yw[i]; rfl[e] = ws.fl[i]; }
  }
#pragma unroll
  for (int e = 0; e < kPrepE; e++) {
    const int i = tid + e * kBandThreads;
    if (i < n) { sf[i] = (i == 0 || rk[e] != rkp[e]) ? 1 : 0; sxw[i] = rxw[e]; }
  }
  if (tid == 0) s_ngather = 0;
  __syncthreads();
  // pass 2: item id of every RoI = (number of run starts up to it) - 1: chunked counts + block scan
  const int per = (n + kBandThreads - 1) / kBandThreads;
  const int c0 = min(tid * per, n), c1 = min(c0 + per, n);
  int mine = 0;
  for (int i = c0; i < c1; i++) mine += sf[i];
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) s_wsum[wv] = incl;
  __syncthreads();
  int wbase = 0, n_items = 0;
  for (int w = 0; w < kBandWaves; w++) { const int t = s_wsum[w]; if (w < wv) wbase += t; n_items += t; }
  __syncthreads();                                   // s_wsum is reused for the slices
  const int my_first_id = wbase + incl - mine;
  if (n_items <= kLdsItems)
    band_items_body<true>(n, ws, rows_cap, kmax, n_items, my_first_id, sxw, sitem, sf, s_imin, s_imax, s_ifst, s_ikey, s_wsum, &s_ngather, ryw, rfl);
  else
    band_items_body<false>(n, ws, rows_cap, kmax, n_items, my_first_id, sxw, sitem, sf, ws.imin, ws.imax, ws.ifst,
                           reinterpret_cast<uint32_t*>(ws.ikey), s_wsum, &s_ngather, ryw, rfl);
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

__device__ __forceinline__ BandBatch band_batch(const BandBatch* __restrict__ batches, int idx, bool have) {
  BandBatch b; b.i0 = 0; b.n = 0; b.xa = 0; b.xb = -1;
  if (have) {          // uniform index: a scalar load
    const int4 r = *reinterpret_cast<const int4*>(batches + idx);
    b.i0 = buni(r.x); b.n = buni(r.y); b.xa = buni(r.z); b.xb = buni(r.w);
  }
  return b;
}

typedef __attribute__((address_space(3))) const bf32x4 band_lds_cf4;
typedef __attribute__((address_space(3))) const float4 band_lds_cfl4;

// Workgroup shapes: NT = 1024 threads x NQ = 2 channel quads, one workgroup per CU (batches of 20 RoIs); or NT = 512 x NQ = 1,
// two workgroups per CU (batches of 10 RoIs) whose phases -- LDS tap gather | loads, commit, stores -- overlap each other.
// A unit of work is (band item, `gpasses` consecutive channel groups): the sweep of the next channel group starts inside the
// pipeline of the current one (its first batch is loaded while the last batch of the current group is pooled), so the
// latency of a unit's first loads and of fetching the unit is paid once per `gpasses` sweeps.
template <typename TOut, int NQ, int NT>
__global__ __launch_bounds__(NT, 4) void roi_align_fwd_band(RoiAlignParams p, BandWs ws, int rows_cap, int kmax, int max_units, int persist_from, int gpasses) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CG = 4 * NQ;
  constexpr int NW = NT / 64;
  constexpr int kRowBytes = NQ * kRowQuadBytes + kRowSkewBytes;
  __shared__ int s_unit[2];
  __shared__ int s_slice[2 * kSlices];
  __shared__ int s_r[kBandMaxK];
  BT_DECL
  const int tid = threadIdx.x, lane = tid & 63, wv = buni(tid >> 6);     // wave index: uniform, so the per-unit row / quad arithmetic is scalar
  const int bins = p.pooled_h * p.pooled_w;
  char* ring = reinterpret_cast<char*>(smem);                                  // [rows_cap][NQ][kRingSlots][4 channels] float32 (+ row skew)
  float* slab = reinterpret_cast<float*>(smem + (size_t)rows_cap * kRowBytes);   // [kmax][CG][bins]
  const uint32_t ring32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;    // LDS address of the ring
  const int ncg = p.channels / CG;
  const int ncgp = (ncg + gpasses - 1) / gpasses;                              // units per item
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
  BT_INIT;
  if (tid < kSlices) { s_slice[tid] = ws.ctl->slice_first[tid]; s_slice[kSlices + tid] = ws.ctl->slice_count[tid]; }
  __syncthreads();

  // the next unit of this workgroup: own XCD's queue first, then the others', then the gather items.  One atomic per queue
  // tried; a counter that runs past its queue's length is harmless.
  auto grab = [&]() {
    int found = -1;
    for (int k = 0; k < kSlices && found < 0; k++) {
      const int x = k < kXcds ? (xcd + k) & (kXcds - 1) : kXcds;
      const int n = s_slice[kSlices + x] * (x == kXcds ? ncg : ncgp);
      if (n <= 0) continue;
      const int u = atomicAdd(&ws.ctl->ctr[x], 1);
      if (u < n) found = (x << 24) | u;
    }
    return found;
  };
  if (tid == 0) s_unit[0] = grab();
  __syncthreads();

  // (workgroups >= persist_from -- the last ones dispatched -- stay until the queues are empty whatever max_units says)
  for (int done = 0; max_units <= 0 || done < max_units || (int)blockIdx.x >= persist_from; done++) {
    const int unit = buni(s_unit[done & 1]);
    if (unit < 0) break;
    int next_unit = -1;
    if (tid == 0) next_unit = grab();        // the atomic's round trip hides behind this unit; published at its end
    BT(0);
    const int ux = unit >> 24, uu = unit & 0xffffff;
    const int scount = buni(s_slice[kSlices + ux]);
    const int cgp = uu / scount;                                    // gather items: channel group; band items: group of `gpasses`
    const BandItem it = ws.items[buni(s_slice[ux]) + (uu - cgp * scount)];
    const int first = buni(it.first), kind = buni(it.kind);
    const int cg0 = kind == kItemGather ? cgp : cgp * gpasses;
    const int npass = kind == kItemGather ? 1 : min(gpasses, ncg - cg0);
    if (kind == kItemZero) {        // padding rows of a fixed-shape batch (level -1): defined output
      const int per = npass * CG * bins;
      for (int o = tid; o < buni(it.count) * per; o += NT) {
        const int k = o / per, e = o - k * per;
        out[((size_t)tab[first + k].r * p.channels + cg0 * CG) * bins + e] = from_f32<TOut>(0.f);
      }
    } else if (kind == kItemGather) {
      // a window the ring cannot hold (wider than 64 columns, or outside the rows of its band): per-output gather from global
      // memory, geometry on the fly -- the reference's loop for one (RoI, channel, bin) per thread
      const dtc_feat_level L = p.lv[buni(it.lvl)];
      const int c0 = cg0 * CG;
      const float* fbase = reinterpret_cast<const float*>(L.data) + (int64_t)buni(it.b) * L.stride_n + (int64_t)c0 * L.stride_c;
      const RoiHead hd = load_roi_head(p, first);
      TOut* og = out + ((size_t)hd.r * p.channels + c0) * bins;
      for (int o = tid; o < CG * bins; o += NT) {
        const int c = o / bins, gb = o - c * bins;
        const int gph = gb / p.pooled_w, gpw = gb - gph * p.pooled_w;
        const float* d = fbase + (int64_t)c * L.stride_c;
        float acc = 0.f;
        for (int iy = 0; iy < 2; iy++) {
          const AxisEntry y = make_axis(hd.sh, hd.bin_h, gph, iy, 2, L.height);
          const int64_t yl0 = (int64_t)y.lo * L.stride_h, yh0 = (int64_t)y.hi * L.stride_h;
          for (int ix = 0; ix < 2; ix++) {
            const AxisEntry x = make_axis(hd.sw, hd.bin_w, gpw, ix, 2, L.width);
            const int64_t xl0 = (int64_t)x.lo * L.stride_w, xh0 = (int64_t)x.hi * L.stride_w;
            const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
            acc += w1 * d[yl0 + xl0] + w2 * d[yl0 + xh0] + w3 * d[yh0 + xl0] + w4 * d[yh0 + xh0];
          }
        }
        og[o] = from_f32<TOut>(acc * 0.25f);
      }
    } else if (buni(it.nbatch) > 0) {        // (nbatch == 0: every RoI of the band takes the gather path)
      const int nbatch = buni(it.nbatch);
      const dtc_feat_level L = p.lv[buni(it.lvl)];
      const int H = L.height, W = L.width;
      const int rbase = buni(it.rbase), rows = buni(it.rows);
      const float* fbase = reinterpret_cast<const float*>(L.data) + (int64_t)buni(it.b) * L.stride_n + (int64_t)(cg0 * CG) * L.stride_c;
      const bool vec = L.stride_w == 1 && (W & 3) == 0 && ((L.stride_h | L.stride_c | L.stride_n) & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(L.data) & 15) == 0 && L.stride_h > 0 && L.stride_c > 0 &&
                       L.stride_h * (int64_t)H + L.stride_c * (int64_t)(CG * npass) < (1ll << 28);
      const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fbase), 0, 0xffffffff, 0x00020000);
      const int sh32 = (int)L.stride_h, sc32 = (int)L.stride_c;
      const int units = NQ * rows;             // wave-level load units: (quad, row) x 16 column groups x 4 channels
      const int rb_bytes = rbase * kRowBytes - (int)ring32;      // tap offsets become complete LDS addresses
      const BandBatch* __restrict__ batches = ws.batches + first;

      int res_a = 0, res_b = 0;                // columns [res_a, res_b) are resident in the ring
      float4 v[kBandUnits];
      uint32_t voff = 0;                       // per-lane byte offset of the batch's piece (channel cl, column group)
      int pcol = 0;                            // first column of that piece
      int ldl = 0;                             // per-lane LDS byte offset inside a (row, quad) line
      bool live = false;                       // this lane's column group is a new one (not a duplicate)
      int new_groups = 0;
      uint4 ty0 = make_uint4(0, 0, 0, 0), ty1 = ty0, tx0 = ty0, tx1 = ty0;
      int tr = 0;

      // columns of [xa, xb] the ring does not hold: at most two runs, left (gA groups) and right (gB groups) of what stays.
      // ALWAYS defines every piece register (zeros when there is nothing to load): a piece register that is written under a
      // condition becomes a loop-carried value and the register allocator spills all 32 of them.  `pass`: channel group of the
      // unit the batch belongs to (its channel offset goes into the scalar offset of the loads).
      auto issue = [&](const BandBatch& pl, int pass) {
        const int keep_a = max(pl.xa, res_a), keep_b = min(pl.xb + 1, res_b);
        const bool ov = keep_b > keep_a;
        const int ngx = (pl.xb + 1 - pl.xa) >> 2;
        const int gA = ov ? (keep_a - pl.xa) >> 2 : ngx, gB = ov ? (pl.xb + 1 - keep_b) >> 2 : 0;
        new_groups = pl.n > 0 ? gA + gB : 0;
        const int ge = min(g16, max(new_groups, 1) - 1);
        live = g16 < new_groups;
        const int col = ge < gA ? pl.xa + 4 * ge : keep_b + 4 * (ge - gA);
        voff = (uint32_t)(cl * sc32 + col) * 4u;
        pcol = col;
        ldl = ring_phys(col) * 16 + cl * 4;
        if (vec && new_groups > 0) {
          // straight-line: units past the end repeat the last one (same bytes, an L1 hit; never committed).  `rows` is made
          // opaque so that the eight (quad, row) offsets are scalar arithmetic HERE instead of loop-invariant values kept
          // (spilled) across the sweep.
          int rows_o = rows;
          asm volatile("" : "+s"(rows_o));
          const int pass_off = pass * CG * sc32;
#pragma unroll
          for (int k = 0; k < kBandUnits; k++) {
            const int u = min(wv + NW * k, NQ * rows_o - 1);
            const int q = NQ == 1 ? 0 : (NQ == 2 ? (u >= rows_o ? 1 : 0) : u / rows_o);
            const int row = u - q * rows_o;
            const int frow = min(rbase + row, H - 1);
            const uint32_t soff = (uint32_t)(pass_off + 4 * q * sc32 + frow * sh32) * 4u;
            const bu32x4 w = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, 0);
            v[k] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
          }
        } else {
#pragma unroll
          for (int k = 0; k < kBandUnits; k++) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto commit = [&](const BandBatch& pl, int pass) {
        if (new_groups == 0) { res_a = pl.xa; res_b = pl.xb + 1; return; }
        if (vec) {
          int ldc = ldl, rows_o = rows;
          asm volatile("" : "+v"(ldc), "+s"(rows_o));   // the store addresses are formed HERE, not hoisted above the pooling loop
#pragma unroll
          for (int k = 0; k < kBandUnits; k++) {
            const int u = wv + NW * k;
            if (u < NQ * rows_o) {
              const int q = NQ == 1 ? 0 : (NQ == 2 ? (u >= rows_o ? 1 : 0) : u / rows_o);
              const int row = u - q * rows_o;
              if (live) {
                float* d = reinterpret_cast<float*>(ring + row * kRowBytes + q * kRowQuadBytes + ldc);
                d[0] = v[k].x; d[4] = v[k].y; d[8] = v[k].z; d[12] = v[k].w;
              }
            }
          }
        } else {
          // strided columns / unaligned rows (channels_last maps, widths that are not a multiple of 4): clamped scalar loads
          // straight into LDS.  Correct for any strides; not a fast path.
          const int col = pcol;
          for (int u = wv; u < units; u += NW) {
            const int q = u / rows, row = u - q * rows;
            const int frow = min(rbase + row, H - 1);
            if (live) {
              const float* s = fbase + (int64_t)(pass * CG + 4 * q + cl) * L.stride_c + (int64_t)frow * L.stride_h;
              float* d = reinterpret_cast<float*>(ring + row * kRowBytes + q * kRowQuadBytes + ldl);
              d[0] = s[(int64_t)min(col, W - 1) * L.stride_w]; d[4] = s[(int64_t)min(col + 1, W - 1) * L.stride_w];
              d[8] = s[(int64_t)min(col + 2, W - 1) * L.stride_w]; d[12] = s[(int64_t)min(col + 3, W - 1) * L.stride_w];
            }
          }
        }
        res_a = pl.xa; res_b = pl.xb + 1;
      };
      auto load_tables = [&](const BandBatch& pl) {
        if (rl < pl.n) {
          const BandRoiTab* T = tab + pl.i0 + rl;
          const uint4* ya = reinterpret_cast<const uint4*>(&T->y[2 * ph]);
          const uint4* xa = reinterpret_cast<const uint4*>(&T->x[2 * pw]);
          ty0 = ya[0]; ty1 = ya[1]; tx0 = xa[0]; tx1 = xa[1];
          tr = T->r;
        }
      };

      const int steps = npass * nbatch;          // the batches of `npass` consecutive sweeps, one pipeline
      BandBatch cur = band_batch(batches, 0, true);
      BandBatch pre = band_batch(batches, nbatch > 1 ? 1 : 0, steps > 1);     // the record after next, fetched a whole batch ahead
      issue(cur, 0);
      load_tables(cur);
      commit(cur, 0);
      __syncthreads();
      BT(1);

      int pass = 0, bi = 0;                      // (channel group, batch) of `cur`
      for (int st = 0; st < steps; st++) {
        const BandBatch nxt = pre;
        int nb = bi + 1, npx = pass;             // (batch, channel group) of `nxt`
        if (nb == nbatch) { nb = 0; npx = pass + 1; }
        {
          int b2 = nb + 1;
          if (b2 >= nbatch) b2 = 0;
          pre = band_batch(batches, b2, st + 2 < steps);
        }
        // ---- this lane's (RoI, bin) of the current batch: tap addresses and weights from the record ---------------------------
        const bool on = rl < cur.n;
        uint32_t a[2][2][4];
        float yl[2], yh[2], xl[2], xh[2];
        {
          const int ylo[2] = {(int)ty0.x * kRowBytes - rb_bytes, (int)ty1.x * kRowBytes - rb_bytes};
          const int yhi[2] = {(int)ty0.y * kRowBytes - rb_bytes, (int)ty1.y * kRowBytes - rb_bytes};
          const int xlo[2] = {(int)tx0.x, (int)tx1.x}, xhi[2] = {(int)tx0.y, (int)tx1.y};
          yl[0] = __uint_as_float(ty0.z); yh[0] = __uint_as_float(ty0.w); yl[1] = __uint_as_float(ty1.z); yh[1] = __uint_as_float(ty1.w);
          xl[0] = __uint_as_float(tx0.z); xh[0] = __uint_as_float(tx0.w); xl[1] = __uint_as_float(tx1.z); xh[1] = __uint_as_float(tx1.w);
#pragma unroll
          for (int iy = 0; iy < 2; iy++)
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
              uint32_t t0 = (uint32_t)(ylo[iy] + xlo[ix]), t1 = (uint32_t)(ylo[iy] + xhi[ix]);
              uint32_t t2 = (uint32_t)(yhi[iy] + xlo[ix]), t3 = (uint32_t)(yhi[iy] + xhi[ix]);
              asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));   // one finished VGPR per tap
              a[iy][ix][0] = t0; a[iy][ix][1] = t1; a[iy][ix][2] = t2; a[iy][ix][3] = t3;
            }
          if (on && bin == 0) s_r[rl] = tr;
        }
        BT(2);
        if (nb == 0) res_a = res_b = 0;         // the next batch opens the sweep of the next channel group: nothing of it is resident
        issue(nxt, npx);                        // in flight (registers) while this batch is pooled
        BT(3);

        if (on) {
          float* so = slab + rl * (CG * bins) + bin;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            bf32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
            // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
              for (int ix = 0; ix < 2; ix++) {
                // two taps in flight at a time (8 registers): w1*v1 + w2*v2, then + w3*v3, + w4*v4 -- the reference's order
                const bf32x4 t0 = *reinterpret_cast<band_lds_cf4*>(a[iy][ix][0] + q * kRowQuadBytes);
                const bf32x4 t1 = *reinterpret_cast<band_lds_cf4*>(a[iy][ix][1] + q * kRowQuadBytes);
                const float w1 = yh[iy] * xh[ix], w2 = yh[iy] * xl[ix];               // roi_align_cpu_loop.cpp:95
                bf32x2 s01 = w1 * t0.lo + w2 * t1.lo, s23 = w1 * t0.hi + w2 * t1.hi;  // :208-209
                __builtin_amdgcn_sched_barrier(0);
                const bf32x4 t2 = *reinterpret_cast<band_lds_cf4*>(a[iy][ix][2] + q * kRowQuadBytes);
                const bf32x4 t3 = *reinterpret_cast<band_lds_cf4*>(a[iy][ix][3] + q * kRowQuadBytes);
                const float w3 = yl[iy] * xh[ix], w4 = yl[iy] * xl[ix];
                s01 = s01 + w3 * t2.lo + w4 * t3.lo; s23 = s23 + w3 * t2.hi + w4 * t3.hi;        // :210-211
                a01 += s01; a23 += s23;
                __builtin_amdgcn_sched_barrier(0);
              }
            // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32
            float* o = so + 4 * q * bins;
            o[0] = a01.x * 0.25f; o[bins] = a01.y * 0.25f; o[2 * bins] = a23.x * 0.25f; o[3 * bins] = a23.y * 0.25f;
          }
        }
        BT(4);
        if (nxt.n > 0) load_tables(nxt);        // in flight across the barrier, the commit and the slab stores
        __syncthreads();
        BT(5);
        if (nxt.n > 0) commit(nxt, npx);
        BT(6);
        {
          // slab [RoI][CG][bins] is contiguous per RoI exactly like the [R, C, PH, PW] output: a wavefront stores whole RoIs
          // (at most two), 16 bytes per lane, every LDS read issued before the first store
          const int n4 = (CG * bins) >> 2;       // <= 128
          const int co = (cg0 + pass) * CG;
          const bool e0 = lane < n4, e1 = lane + 64 < n4;
#pragma unroll
          for (int kk = 0; kk < 2; kk++) {
            const int k = wv + kk * NW;
            if (k < cur.n) {
              const float4* sk = reinterpret_cast<const float4*>(slab) + k * n4;
              float4 d0, d1;
              if (e0) d0 = sk[lane];
              if (e1) d1 = sk[lane + 64];
              TOut* ok = out + ((size_t)buni(s_r[k]) * p.channels + co) * bins;
              if (e0) band_store4<TOut>(ok + 4 * lane, d0);
              if (e1) band_store4<TOut>(ok + 4 * (lane + 64), d1);
            }
          }
        }
        BT(7);
        __syncthreads();
        BT(8);
        cur = nxt; pass = npx; bi = nb;
      }
    }
    if (tid == 0) s_unit[(done + 1) & 1] = next_unit;
    __syncthreads();
  }
  BT_FLUSH;
}

#ifdef DTC_BAND_TRACE
}  // namespace dtc
extern "C" __attribute__((visibility("default"))) int dtc_debug_band_trace(void* host_dst, int reset) {
  unsigned long long z[dtc::kBtSlots] = {0};
  if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(dtc::g_band_trace), sizeof(z), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(dtc::g_band_trace), z, sizeof(z), 0, hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}
namespace dtc {
#endif

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct BandConfig {
  int enabled = 0;     // DTC_RA_BAND=1: the workspace entry takes the band sweep (experimental; default: prepared cluster kernel)
  int rows_cap = 0;    // 0: what the LDS holds next to the slab
  int kmax = 0;        // RoIs per batch (0: threads / bins, at most 20)
  int grid = 0;        // workgroups (0: one per CU)
  int max_units = 0;   // units a workgroup takes before it leaves (0: until the queues are empty)
  int stop = 0;        // development: 1 = only the record kernel, 2 = records + items, 0 = everything
  int shape = 1;       // 1: 1024 threads x 2 quads, one workgroup per CU; 2: 512 threads x 1 quad, two per CU
  int gpasses = 2;     // consecutive channel groups a unit sweeps (one pipeline)
};
static BandConfig& band_config() {
  static BandConfig cfg = [] {
    BandConfig c;
    if (const char* e = getenv("DTC_RA_BAND")) c.enabled = atoi(e) != 0;
    if (const char* e = getenv("DTC_RA_BAND_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 128) c.rows_cap = v; }
    if (const char* e = getenv("DTC_RA_BAND_K")) { const int v = atoi(e); if (v >= 1 && v <= kBandMaxK) c.kmax = v; }
    if (const char* e = getenv("DTC_RA_BAND_GRID")) { const int v = atoi(e); if (v >= 1) c.grid = v; }
    if (const char* e = getenv("DTC_RA_BAND_MAXUNITS")) { const int v = atoi(e); if (v >= 0) c.max_units = v; }
    if (const char* e = getenv("DTC_RA_BAND_STOP")) c.stop = atoi(e);
    if (const char* e = getenv("DTC_RA_BAND_SHAPE")) { const int v = atoi(e); if (v == 1 || v == 2) c.shape = v; }
    if (const char* e = getenv("DTC_RA_BAND_GPASSES")) { const int v = atoi(e); if (v >= 1 && v <= 64) c.gpasses = v; }
    return c;
  }();
  return cfg;
}

static size_t band_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t band_ws_bytes(int n_rois) {
  const size_t n = (size_t)(n_rois > 0 ? n_rois : 0) + 1;
  return band_align(sizeof(BandCtl)) + band_align(2 * n * sizeof(BandItem)) + band_align(n * sizeof(BandBatch)) +
         8 * band_align(n * sizeof(int)) + band_align(n * sizeof(BandRoiTab));
}

bool roi_align_band_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  if (!band_config().enabled) return false;
  if (p.sampling_ratio != 2 || !p.roi_desc) return false;
  if (p.pooled_h > kBandMaxPooled || p.pooled_w > kBandMaxPooled || p.pooled_h * p.pooled_w > 64) return false;
  if ((p.channels & 7) != 0 || p.n_rois > kBandMaxRois) return false;
  if (in_dtype != DTC_F32) return false;
  if (out_dtype != DTC_F32 && out_dtype != DTC_F16 && out_dtype != DTC_BF16) return false;
  for (int l = 0; l < p.n_levels; l++) {
    if (p.lv[l].height > 65535 || p.lv[l].width > 65535) return false;
    if (p.lv[l].stride_c == 1 && p.channels > 1) return false;       // channels_last maps: roi_align_fwd_nhwc
  }
  return true;
}

template <typename TOut, int NQ, int NT>
static int launch_band_t(const RoiAlignParams& p, const BandWs& ws, hipStream_t stream) {
  const BandConfig& cfg = band_config();
  const int bins = p.pooled_h * p.pooled_w;
  int kmax = cfg.kmax ? cfg.kmax : NT / bins;
  if (kmax > 20) kmax = 20;
  if (kmax * bins > NT) kmax = NT / bins;
  const int lds_total = (160 * 1024) / (1024 / NT) - 1024;          // static __shared__ of the kernel comes on top
  const int slab_b = kmax * 4 * NQ * bins * 4;
  int rows_cap = (lds_total - slab_b) / (NQ * kRowQuadBytes + kRowSkewBytes);
  if (cfg.rows_cap && cfg.rows_cap < rows_cap) rows_cap = cfg.rows_cap;
  if (rows_cap * NQ > kBandUnits * (NT / 64)) rows_cap = kBandUnits * (NT / 64) / NQ;      // what the register pipeline carries
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
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  static int n_cu = 256;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_band<TOut, NQ, NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (160 * 1024) / (1024 / NT) - 256);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(band_items_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kBandMaxRois * 7 + 64);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cu = n;
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  hipLaunchKernelGGL(band_tab_kernel, dim3((unsigned)((p.n_rois + 63) / 64)), dim3(64), 0, stream, p, pp, ws);
  DTC_CHECK_LAUNCH();
  if (cfg.stop == 1) return DTC_OK;
  hipLaunchKernelGGL(band_items_kernel, dim3(1), dim3(kBandThreads), (size_t)p.n_rois * 7 + 16, stream, p.n_rois, ws, rows_cap, kmax);
  DTC_CHECK_LAUNCH();
  if (cfg.stop == 2) return DTC_OK;
  const int lds_b = rows_cap * (NQ * kRowQuadBytes + kRowSkewBytes) + slab_b;
  const int resident = n_cu * (1024 / NT);
  const int grid = cfg.grid ? cfg.grid : resident;
  hipLaunchKernelGGL((roi_align_fwd_band<TOut, NQ, NT>), dim3((unsigned)grid), dim3(NT), lds_b, stream, p, ws, rows_cap, kmax, cfg.max_units, grid > resident ? grid - resident : 0, cfg.gpasses);
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
  ws.batches = reinterpret_cast<BandBatch*>(w); w += band_align(n * sizeof(BandBatch));
  ws.key = reinterpret_cast<uint32_t*>(w); w += band_align(n * sizeof(int));
  ws.xw = reinterpret_cast<uint32_t*>(w); w += band_align(n * sizeof(int));
  ws.yw = reinterpret_cast<uint32_t*>(w); w += band_align(n * sizeof(int));
  ws.fl = reinterpret_cast<uint32_t*>(w); w += band_align(n * sizeof(int));
  ws.imin = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.imax = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.ifst = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.ikey = reinterpret_cast<int*>(w); w += band_align(n * sizeof(int));
  ws.tab = reinterpret_cast<BandRoiTab*>(w);
  const bool two = band_config().shape == 2;        // two 512-thread workgroups per CU, one channel quad each
  if (out_dtype == DTC_F32) return two ? launch_band_t<float, 1, 512>(p, ws, stream) : launch_band_t<float, 2, 1024>(p, ws, stream);
  if (out_dtype == DTC_F16) return two ? launch_band_t<__half, 1, 512>(p, ws, stream) : launch_band_t<__half, 2, 1024>(p, ws, stream);
  if (out_dtype == DTC_BF16) return two ? launch_band_t<bf16_t, 1, 512>(p, ws, stream) : launch_band_t<bf16_t, 2, 1024>(p, ws, stream);
  return DTC_EUNSUPPORTED;
}

size_t roi_align_band_workspace_bytes(int n_rois) { return band_ws_bytes(n_rois); }

}  // namespace dtc

// ---- C ABI ----------------------------------------------------------------------------------------------------------------------
DTC_API size_t dtc_roi_align_band_workspace_bytes(int n_rois) {
  // enough for either kernel the entry may take, at the two pooled sizes of the FPN heads
  const size_t a = dtc::roi_align_band_workspace_bytes(n_rois);
  const size_t b = dtc::roi_align_tile_workspace_bytes(n_rois, 7, 7), c = dtc::roi_align_tile_workspace_bytes(n_rois, 14, 14);
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

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
  // the cluster-stationary kernel with its per-launch preparation pass: small channel blocks in channel-major order
  static const bool no_prep = [] { const char* e = getenv("DTC_RA_TILE_PREP"); return e && atoi(e) == 0; }();
  bool all_nhwc = channels > 1;
  for (int i = 0; i < n_levels; i++) all_nhwc = all_nhwc && levels[i].stride_c == 1;
  if (workspace && !no_prep && !all_nhwc && (channels & 3) == 0 && pooled_h <= 16 && pooled_w <= 16 &&
      workspace_bytes >= dtc::roi_align_tile_workspace_bytes(n_rois, pooled_h, pooled_w) && dtc::roi_align_tile_supported(p, in_dtype, out_dtype)) {
    p.xcd_remap = 1;
    return dtc::launch_roi_align_tile_prepared(p, in_dtype, out_dtype, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  }
  // anything the sweep does not cover (other sampling ratios / bin counts / dtypes, channels_last maps): the packed entry
  return dtc_roi_align_forward_packed(levels, n_levels, channels, in_dtype, roi_desc, n_rois, pooled_h, pooled_w, sampling_ratio,
                                      out, out_dtype, stream);
}
