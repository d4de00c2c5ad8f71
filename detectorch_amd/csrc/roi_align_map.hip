// A1  RoIAlign forward, MAP-stationary kernel: the single-level (C4) configurations with adaptive sampling.
//
// Replaces roi_align_forward_kernel (lib/cppcuda_cffi/src/cuda/roi_align_forward_cuda.cu:82-159) for the R-50-C4 heads
// (detector.py:240-248: RoIAlign on res4, [B,1024,50,84], sampling_ratio 0 -> ceil(roi / pooled) samples per bin and axis).
//
// Why a third formulation: on one stride-16 map every RoI is large relative to the map (a 512-pixel proposal covers a quarter
// of it), RoIs overlap many-fold and the adaptive grid makes 4..400 taps per bin.  A RoI-stationary workgroup stages its window
// once per 128 channels: 8000 RoIs x 1024 channels re-read 6.4 GB of features that hold 0.14 GB (measured: the same launch is
// 3.5 x slower when the RoI order destroys L2 locality), and a large window leaves room for only 8 channels per pass.  But the
// whole map of a channel quad is 50 x 84 x 16 B = 67 KB: it FITS the 160 KB LDS of a CU.  So here a workgroup owns
// (a run of RoIs, 8 channels), stages the ENTIRE map of those channels once (every feature byte is read once per run), and its
// 16 wavefronts pool one RoI each, lane <-> bin, straight from LDS:
//   * no window, no axis tables, no per-RoI barriers: a wave needs a workgroup barrier only when the image changes;
//   * sampling positions and weights (make_axis: the reference's own operations, roi_align_cpu_loop.cpp:36-95) are formed once per
//     RoI, one axis entry per lane, and fetched in the sample loops with wave shuffles; waves take RoIs from a shared counter
//     (the cost of a RoI varies 1 : 100 with the adaptive grid);
//   * a tap is one ds_read_b128 per channel quad, accumulated with packed fp32 multiplies / adds in the reference's order
//     (for iy, for ix: acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 ; /= count) -> bit-identical to the CPU reference;
//   * the [8 channels][bins] results of a RoI are contiguous in the [R,C,PH,PW] output: they go through a per-wave LDS slab
//     and leave as 16-byte stores.
// RoIs must arrive image-major (the packed descriptors of dtc_fpn_collect_distribute are); any order is CORRECT, but every
// change of image re-stages the map.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "roi_align_common.h"

namespace dtc {

void roi_align_set_exact(int exact);
int roi_align_get_exact();

constexpr int kMapThreads = 1024;
constexpr int kMapWaves = kMapThreads / 64;
constexpr int kMapLdsBytes = 160 * 1024 - 512;      // dynamic LDS budget (static: the rendezvous arrays)

typedef float mf32x2 __attribute__((ext_vector_type(2)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));

template <typename TOut> __device__ __forceinline__ void map_store4(TOut* d, float4 v);
template <> __device__ __forceinline__ void map_store4<float>(float* d, float4 v) { store_stream16(d, v); }     // streaming stores: dtc_common.h
template <> __device__ __forceinline__ void map_store4<__half>(__half* d, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  store_stream8(d, *reinterpret_cast<const uint32_t*>(&a), *reinterpret_cast<const uint32_t*>(&b));
}
template <> __device__ __forceinline__ void map_store4<bf16_t>(bf16_t* d, float4 v) {
  store_stream8(d, (uint32_t)from_f32<bf16_t>(v.x).bits | ((uint32_t)from_f32<bf16_t>(v.y).bits << 16),
                (uint32_t)from_f32<bf16_t>(v.z).bits | ((uint32_t)from_f32<bf16_t>(v.w).bits << 16));
}

template <typename TIn> struct MapLoad4;
template <> struct MapLoad4<float> { static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); } };
template <> struct MapLoad4<__half> {
  static __device__ __forceinline__ float4 ld(const __half* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&r.x), b = *reinterpret_cast<const __half2*>(&r.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
  }
};
template <> struct MapLoad4<bf16_t> {
  static __device__ __forceinline__ float4 ld(const bf16_t* p) { return bf16x4_to_f32(*reinterpret_cast<const uint2*>(p)); }
};

// One sample of one bin: 4 taps x NQ channel quads.  ylo / yhi: byte offsets of the two rows, xlo / xhi of the two columns.
template <int NQ>
__device__ __forceinline__ void map_sample(const char* map, int plane_bytes, int ylo, int yhi, int xlo, int xhi, float yl,
                                           float yh, float xl, float xh, mf32x2 (&acc)[NQ][2]) {
  const float w1 = yh * xh, w2 = yh * xl, w3 = yl * xh, w4 = yl * xl;                          // roi_align_cpu_loop.cpp:95
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const char* m = map + q * plane_bytes;
    const mf32x4 v1 = *reinterpret_cast<const mf32x4*>(__builtin_assume_aligned(m + ylo + xlo, 16));
    const mf32x4 v2 = *reinterpret_cast<const mf32x4*>(__builtin_assume_aligned(m + ylo + xhi, 16));
    const mf32x4 v3 = *reinterpret_cast<const mf32x4*>(__builtin_assume_aligned(m + yhi + xlo, 16));
    const mf32x4 v4 = *reinterpret_cast<const mf32x4*>(__builtin_assume_aligned(m + yhi + xhi, 16));
    acc[q][0] += w1 * v1.lo + w2 * v2.lo + w3 * v3.lo + w4 * v4.lo;                               // :208-211
    acc[q][1] += w1 * v1.hi + w2 * v2.hi + w3 * v3.hi + w4 * v4.hi;
  }
}

__device__ __forceinline__ int map_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Per-RoI record of the per-launch preparation pass (map_prep_kernel): everything about a RoI that does not depend on the
// channels -- which each of the C / 8 channel-group workgroups used to re-derive (two IEEE divisions for the bin sizes, two
// more for the adaptive grid, one make_axis per lane and axis: ~170 of the ~350 instructions a wavefront spent per RoI
// outside the sample loops).  hdr[]: word k is fetched by lane k and read with readlane.
enum { kMhR = 0, kMhB, kMhFlags, kMhGh, kMhGw, kMhInv, kMhCount, kMhRcpLo, kMhRcpHi, kMhSh, kMhSw, kMhBinH, kMhBinW, kMhWords = 64 };
enum { kMfPad = 1, kMfYtab = 2, kMfXtab = 4, kMfMerged = 8 };
struct MapAxis { int32_t lo, hi; float l, h; };            // lo / hi: LDS byte offsets (row * pitch * 16, column * 16)
struct MapPrepRoi { uint32_t hdr[kMhWords]; MapAxis y[64]; MapAxis x[64]; };    // 256 + 2 x 1024 B
static_assert(sizeof(MapPrepRoi) == 2304, "MapPrepRoi layout");

// FAST mode (dtc_roi_align_set_exact(0), not the default): bilinear weights factor into a row and a column term, so the gh x gw
// samples x 4 taps of a bin collapse to (gh + 1) x (gw + 1) taps -- pixel (row r, column c) of the bin's footprint times
// Wy[r] * Wx[c], Wy[r] = sum of the y weights of the samples that touch row r (and Wx likewise).  9 taps instead of 16 on a 2 x 2
// grid, 16 instead of 36 on 3 x 3, 169 instead of 576 on 12 x 12.  The SUM is the reference's in exact arithmetic; in float32 the
// different association moves the result by a few 1e-7 relative (tested: <= 1e-5 absolute on O(1) features, inside the 1e-4
// contract of BASELINE.json) -- not bit-identical, hence opt-in.  The merged axis tables are formed here, per RoI: entry
// (bin row ph, k) by lane ph * (gh + 1) + k = {byte offset of row lo(ph, 0) + k, Wy}; RoIs whose tables do not fit 64 lanes per
// axis stay on the exact path.
template <bool IS_Y>
__device__ __forceinline__ MapAxis map_merged_entry(const RoiHead& hd, int lane, int g, int pooled, int extent, int unit) {
  const int K = g + 1;
  const int ph = min((int)(((float)lane + 0.5f) * __frcp_rn((float)K)), pooled - 1), k = lane - ph * K;
  const float start = IS_Y ? hd.sh : hd.sw, bin = IS_Y ? hd.bin_h : hd.bin_w;
  const int base = make_axis(start, bin, ph, 0, g, extent).lo;
  const int row = min(base + k, extent - 1);            // past the footprint (or the map): weight 0 on a valid pixel
  float wsum = 0.f;
  for (int i = 0; i < g; i++) {
    const AxisEntry e = make_axis(start, bin, ph, i, g, extent);
    if (e.lo == base + k) wsum += e.h;
    if (e.hi == base + k) wsum += e.l;                      // (clamped at the last row / column: both taps are that pixel)
  }
  MapAxis a; a.lo = row * unit; a.hi = a.lo; a.l = 0.f; a.h = k < K ? wsum : 0.f;
  return a;
}

__global__ __launch_bounds__(256) void map_prep_kernel(RoiAlignParams p, MapPrepRoi* __restrict__ prep, int fast, int pitch) {
  const int ri = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ri >= p.n_rois) return;
  const RoiHead hd = roi_head_from_raw(p, load_roi_raw(p, ri));
  MapPrepRoi* T = prep + ri;
  const bool padrow = hd.lvl < 0 || hd.lvl >= p.n_levels;
  const int H = p.lv[0].height, W = p.lv[0].width;
  const int gh = hd.gh, gw = hd.gw;
  const bool ytab = p.pooled_h * gh <= 64, xtab = p.pooled_w * gw <= 64;
  // merged tables hold g + 1 rows (columns) per bin row (column): enough whenever the samples are at most a pixel apart -- always
  // with adaptive sampling (g = ceil(bin size)); a FIXED sampling ratio on a large RoI can spread them further: checked per RoI
  bool merged = fast && !padrow && p.pooled_h * (gh + 1) <= 64 && p.pooled_w * (gw + 1) <= 64;
  if (merged) {
    bool fits = true;
    if (lane < p.pooled_h) fits = make_axis(hd.sh, hd.bin_h, lane, gh - 1, gh, H).hi - make_axis(hd.sh, hd.bin_h, lane, 0, gh, H).lo <= gh;
    if (lane < p.pooled_w) fits = fits && make_axis(hd.sw, hd.bin_w, lane, gw - 1, gw, W).hi - make_axis(hd.sw, hd.bin_w, lane, 0, gw, W).lo <= gw;
    merged = __ballot(!fits) == 0ull;
  }
  if (merged) {
    MapAxis my = map_merged_entry<true>(hd, lane, gh, p.pooled_h, H, pitch * 16);
    my.h = my.h * fdiv(1.0f, hd.count);                  // the mean over the gh x gw samples, folded into the row weights
    T->y[lane] = my;
    T->x[lane] = map_merged_entry<false>(hd, lane, gw, p.pooled_w, W, 16);
  } else if (!padrow) {
    // entry e = (bin e / g, sample e % g) by lane e: (lane + .5) * (1 / g) truncated is exact (>= .5 / g away from an integer)
    const int qy = (int)(((float)lane + 0.5f) * __frcp_rn((float)gh)), qx = (int)(((float)lane + 0.5f) * __frcp_rn((float)gw));
    const AxisEntry ey = make_axis(hd.sh, hd.bin_h, min(qy, p.pooled_h - 1), lane - qy * gh, gh, H);
    const AxisEntry ex = make_axis(hd.sw, hd.bin_w, min(qx, p.pooled_w - 1), lane - qx * gw, gw, W);
    MapAxis ay; ay.lo = ey.lo * pitch * 16; ay.hi = ey.hi * pitch * 16; ay.l = ey.l; ay.h = ey.h;
    MapAxis ax; ax.lo = ex.lo << 4; ax.hi = ex.hi << 4; ax.l = ex.l; ax.h = ex.h;
    T->y[lane] = ay; T->x[lane] = ax;
  }
  const double rc = 1.0 / (double)hd.count;
  uint32_t w = 0;
  switch (lane) {
    case kMhR: w = (uint32_t)hd.r; break;
    case kMhB: w = (uint32_t)hd.b; break;
    case kMhFlags: w = (padrow ? kMfPad : 0) | (ytab ? kMfYtab : 0) | (xtab ? kMfXtab : 0) | (merged ? kMfMerged : 0); break;
    case kMhGh: w = (uint32_t)gh; break;
    case kMhGw: w = (uint32_t)gw; break;
    case kMhInv: w = __float_as_uint(hd.inv_count); break;
    case kMhCount: w = __float_as_uint(hd.count); break;
    case kMhRcpLo: w = (uint32_t)(__double_as_longlong(rc) & 0xffffffffll); break;
    case kMhRcpHi: w = (uint32_t)((unsigned long long)__double_as_longlong(rc) >> 32); break;
    case kMhSh: w = __float_as_uint(hd.sh); break;
    case kMhSw: w = __float_as_uint(hd.sw); break;
    case kMhBinH: w = __float_as_uint(hd.bin_h); break;
    case kMhBinW: w = __float_as_uint(hd.bin_w); break;
    default: break;
  }
  T->hdr[lane] = w;
}

struct MapRec { RoiRaw raw; uint32_t hw; MapAxis ay, ax; };       // what a wave holds of a RoI: its descriptor, or its prepared record

template <typename TIn, typename TOut, int NQ, bool PREP>
__global__ __launch_bounds__(kMapThreads) void roi_align_fwd_map(RoiAlignParams p, int seg_len, int use_slab, int pitch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int pend_idx[kMapWaves];
  __shared__ int pend_img[kMapWaves];
  __shared__ int s_next;                                          // next RoI of the run nobody has taken yet
  const dtc_feat_level L = p.lv[0];
  const int H = L.height, W = L.width, HW = H * W;
  // rows of the LDS image are `pitch` slots apart: W, or W + 1 when W is even (map_pitch) -- an odd pitch spreads the rows of a
  // RoI's bins over the 16 slots a ds_read_b128 lane group can read at once (simulated on RPN-like boxes, tools/r04/lds_map_sim.py:
  // 9.8 -> 8.7 LDS cycles per tap read on the 84-column res4 map)
  const int plane_bytes = H * pitch * 16;
  const char* map = reinterpret_cast<const char*>(smem);          // [NQ][H*W][4 channels] float32
  float* mapw = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bins = p.pooled_h * p.pooled_w;
  float* slab = reinterpret_cast<float*>(smem + NQ * plane_bytes) + (size_t)wv * (4 * NQ) * bins;   // [4 NQ][bins], use_slab only
  constexpr int CG = 4 * NQ;
  const int ncg = ceil_div(p.channels, CG);
  const int cg = blockIdx.x % ncg, seg = blockIdx.x / ncg;
  const int c0 = cg * CG, nc = min(CG, p.channels - c0);
  const int r_end = min(p.n_rois, (seg + 1) * seg_len);
  TOut* out = reinterpret_cast<TOut*>(p.out);
  const float rpw = __frcp_rn((float)p.pooled_w);
  // lane <-> bin.  A ds_read_b128 is served in four FIXED groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
  // + 32 -- each conflict-free when its lanes hit 16 distinct 16-byte slots modulo 256 B.  bin = lane puts three or four bin rows into
  // a group; when the bins fit (pooled_w <= 8, pooled_h <= 8: the 7 x 7 box head) every group takes TWO whole bin rows instead --
  // fewer distinct map rows per group: simulated on the cfg2 proposals 8.25 -> 7.13 LDS cycles per tap read (4 = conflict-free),
  // tools/r06/lds_lane_maps.py.  The output slab is indexed by bin, so nothing else changes.
  const bool rows2 = p.pooled_w <= 8 && p.pooled_h <= 8;
  int pm_bin = 0;
  bool pm_on = false;
  {
    const int h = lane & 31;
    const bool t0 = h < 4 || (h >= 12 && h < 16) || (h >= 20 && h < 28);
    const int pos = t0 ? (h < 4 ? h : h < 16 ? h - 8 : h - 12) : (h < 12 ? h - 4 : h < 20 ? h - 8 : h - 16);
    const int grp = (lane >> 5) * 2 + (t0 ? 0 : 1);
    const int hi = pos >= p.pooled_w ? 1 : 0;
    const int prow = 2 * grp + hi, pcol = pos - hi * p.pooled_w;
    pm_on = prow < p.pooled_h && pcol < p.pooled_w;
    pm_bin = pm_on ? prow * p.pooled_w + pcol : min(2 * grp, p.pooled_h - 1) * p.pooled_w;      // idle lanes repeat the first bin of their OWN group (a broadcast)
  }
  if (tid == 0) s_next = seg * seg_len;
  __syncthreads();
  int cur = -1;                               // image whose map is staged (uniform)
  int ri = -1;                                // the RoI this wave holds (taken from s_next; kept across a change of image)
  MapRec rec;                                 // ... and its record
  rec.raw.d0 = rec.raw.d1 = make_float4(0.f, 0.f, 0.f, 0.f);
  rec.hw = 0; rec.ay.lo = rec.ay.hi = rec.ax.lo = rec.ax.hi = 0; rec.ay.l = rec.ay.h = rec.ax.l = rec.ax.h = 0.f;
  const MapPrepRoi* __restrict__ prep = reinterpret_cast<const MapPrepRoi*>(p.prep);
  auto fetch = [&](int i, MapRec& r) {         // three coalesced loads (prepared) or the 32-byte descriptor
    if (PREP) { const MapPrepRoi* T = prep + i; r.hw = T->hdr[lane]; r.ay = T->y[lane]; r.ax = T->x[lane]; }
    else r.raw = load_roi_raw(p, i);
  };
  for (;;) {
    int want = -1;
    // ---- pool RoIs of the staged image; waves take them one by one (their cost varies 1 : 100 with the adaptive grid) -----
    for (;;) {
      if (ri < 0) {                            // nothing held: take a RoI and fetch its record
        int t = 0;
        if (lane == 0) t = atomicAdd(&s_next, 1);
        ri = map_uni(t);
        if (ri < r_end) fetch(ri, rec);
      }
      if (ri >= r_end) break;
      // everything about the RoI is uniform across the wave: scalar registers, scalar loops
      int hr, hb, gh, gw;
      bool padrow, ytab, xtab, merged = false;
      float sh, sw, bin_h, bin_w, inv_count, count;
      double rcp_count;
      if (PREP) {
        auto hw = [&](int k) { return (uint32_t)__builtin_amdgcn_readlane((int)rec.hw, k); };
        const uint32_t fl = hw(kMhFlags);
        hr = (int)hw(kMhR); hb = (int)hw(kMhB); gh = (int)hw(kMhGh); gw = (int)hw(kMhGw);
        padrow = (fl & kMfPad) != 0; ytab = (fl & kMfYtab) != 0; xtab = (fl & kMfXtab) != 0; merged = (fl & kMfMerged) != 0;
        sh = __uint_as_float(hw(kMhSh)); sw = __uint_as_float(hw(kMhSw)); bin_h = __uint_as_float(hw(kMhBinH)); bin_w = __uint_as_float(hw(kMhBinW));
        inv_count = __uint_as_float(hw(kMhInv)); count = __uint_as_float(hw(kMhCount));
        rcp_count = __longlong_as_double((long long)(((unsigned long long)hw(kMhRcpHi) << 32) | hw(kMhRcpLo)));
      } else {
        const RoiHead hd = roi_head_from_raw(p, rec.raw);
        hr = hd.r; hb = hd.b; padrow = hd.lvl < 0 || hd.lvl >= p.n_levels;
        gh = map_uni(hd.gh); gw = map_uni(hd.gw);
        sh = __uint_as_float(map_uni(__float_as_uint(hd.sh))); sw = __uint_as_float(map_uni(__float_as_uint(hd.sw)));
        bin_h = __uint_as_float(map_uni(__float_as_uint(hd.bin_h))); bin_w = __uint_as_float(map_uni(__float_as_uint(hd.bin_w)));
        inv_count = hd.inv_count; count = hd.count;
        rcp_count = 1.0 / (double)count;
        // more than 64 entries per axis (a RoI of >= 10 x the pooled size): formed on the fly instead of once per RoI
        ytab = p.pooled_h * gh <= 64; xtab = p.pooled_w * gw <= 64;
      }
      if (!padrow && hb != cur) { want = hb; break; }           // needs another image: keep the RoI, go to the rendezvous
      // take the NEXT RoI now: its record is in flight while this one is pooled
      int rn;
      { int t = 0; if (lane == 0) t = atomicAdd(&s_next, 1); rn = map_uni(t); }
      MapRec recn = rec;
      if (rn < r_end) fetch(rn, recn);
      TOut* orow = out + ((size_t)hr * p.channels + c0) * bins;
      if (padrow) {                            // padding row of a fixed-shape batch: defined output
        for (int o = lane; o < nc * bins; o += 64) orow[o] = from_f32<TOut>(0.f);
        ri = rn; rec = recn;
        continue;
      }
      // Axis entries (roi_align_cpu_loop.cpp:36-95) are formed ONCE per RoI, entry e = (bin row e / gh, sample e % gh) by lane e
      // -- here, or by map_prep_kernel for all channel groups at once -- and fetched inside the sample loops with wave shuffles
      // (ds_bpermute): no divisions, no float -> int in the loops.
      int ey_lo, ey_hi, ex_lo, ex_hi;
      AxisEntry ey, ex;
      if (PREP) {
        ey_lo = rec.ay.lo; ey_hi = rec.ay.hi; ey.l = rec.ay.l; ey.h = rec.ay.h;
        ex_lo = rec.ax.lo; ex_hi = rec.ax.hi; ex.l = rec.ax.l; ex.h = rec.ax.h;
      } else {
        // lane / g for the uniform small g: (lane + .5) * (1 / g) truncated is exact (the product is >= .5 / g away from an integer)
        const int qy = (int)(((float)lane + 0.5f) * __frcp_rn((float)gh)), qx = (int)(((float)lane + 0.5f) * __frcp_rn((float)gw));
        ey = make_axis(sh, bin_h, min(qy, p.pooled_h - 1), lane - qy * gh, gh, H);
        ex = make_axis(sw, bin_w, min(qx, p.pooled_w - 1), lane - qx * gw, gw, W);
        ey_lo = ey.lo * pitch * 16; ey_hi = ey.hi * pitch * 16; ex_lo = ex.lo << 4; ex_hi = ex.hi << 4;
      }
#pragma unroll 1
      for (int b0 = 0; b0 < bins; b0 += 64) {
        const int bin = rows2 ? pm_bin : min(b0 + lane, bins - 1);        // lanes past the last bin repeat one (uniform control flow), never stored
        const bool on = rows2 ? pm_on : b0 + lane < bins;
        const int ph = (int)(((float)bin + 0.5f) * rpw), pw = bin - ph * p.pooled_w;     // bin / pooled_w, exact (bins < 2^16)
        mf32x2 acc[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; q++) { acc[q][0] = mf32x2{0.f, 0.f}; acc[q][1] = mf32x2{0.f, 0.f}; }
        if (merged) {
          // fast mode: one tap per (row, column) of the bin's footprint, weight Wy * Wx (see map_prep_kernel).  The column entries of
          // FOUR taps are fetched once (registers) and reused by every footprint row -- two wave shuffles per four taps instead of per
          // tap -- and a tap is one fused multiply-add per channel pair (contract mode has no operation order to keep): per tap and
          // channel-quad pair 2 ds_read_b128 + 4 v_pk_fma_f32 where round 5 issued 2 + 2 shuffles and 4 v_pk_mul + 4 v_pk_add.
          const int KY = gh + 1, KX = gw + 1;
          auto tap = [&](const char* m, int xo, float w) {
            const mf32x2 wv = {w, w};
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              const mf32x4 v = *reinterpret_cast<const mf32x4*>(__builtin_assume_aligned(m + q * plane_bytes + xo, 16));
              acc[q][0] = __builtin_elementwise_fma(wv, v.lo, acc[q][0]); acc[q][1] = __builtin_elementwise_fma(wv, v.hi, acc[q][1]);
            }
          };
#pragma unroll 1
          for (int x0 = 0; x0 < KX; x0 += 4) {
            const int nk = min(4, KX - x0);                       // uniform
            int xo[4]; float wx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int sx = pw * KX + min(x0 + k, KX - 1);
              xo[k] = __shfl(ex_lo, sx, 64); wx[k] = __shfl(ex.h, sx, 64);
            }
#pragma unroll 1
            for (int iy = 0; iy < KY; iy++) {
              const int sy = ph * KY + iy;
              const int yo = __shfl(ey_lo, sy, 64); const float wy = __shfl(ey.h, sy, 64);
              const char* m = map + yo;
              tap(m, xo[0], wy * wx[0]);
              if (nk > 1) tap(m, xo[1], wy * wx[1]);
              if (nk > 2) tap(m, xo[2], wy * wx[2]);
              if (nk > 3) tap(m, xo[3], wy * wx[3]);
            }
          }
        } else
        // reference order: for iy { for ix { acc += ... } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll 1
        for (int iy = 0; iy < gh; iy++) {
          int ylo, yhi; float yl, yh;
          if (ytab) {
            const int src = ph * gh + iy;
            ylo = __shfl(ey_lo, src, 64); yhi = __shfl(ey_hi, src, 64); yl = __shfl(ey.l, src, 64); yh = __shfl(ey.h, src, 64);
          } else {
            const AxisEntry y = make_axis(sh, bin_h, ph, iy, gh, H);
            ylo = y.lo * pitch * 16; yhi = y.hi * pitch * 16; yl = y.l; yh = y.h;
          }
#pragma unroll 1
          for (int ix = 0; ix < gw; ix++) {
            int xlo, xhi; float xl, xh;
            if (xtab) {
              const int src = pw * gw + ix;
              xlo = __shfl(ex_lo, src, 64); xhi = __shfl(ex_hi, src, 64); xl = __shfl(ex.l, src, 64); xh = __shfl(ex.h, src, 64);
            } else {
              const AxisEntry x = make_axis(sw, bin_w, pw, ix, gw, W);
              xlo = x.lo << 4; xhi = x.hi << 4; xl = x.l; xh = x.h;
            }
            map_sample<NQ>(map, plane_bytes, ylo, yhi, xlo, xhi, yl, yh, xl, xh, acc);
          }
        }
        float res[CG];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          res[4 * q + 0] = acc[q][0].x; res[4 * q + 1] = acc[q][0].y; res[4 * q + 2] = acc[q][1].x; res[4 * q + 3] = acc[q][1].y;
        }
        // :216  output_val /= count.  count = gh * gw is a small integer: a power of two -> the product with its reciprocal is
        // the quotient; otherwise float(double(x) * double(1 / count)) IS the correctly rounded float32 quotient -- x / count
        // can never sit within 2^-33 (relative) of a rounding boundary of float32 (a 25-bit midpoint times an integer < 2^8
        // is not a 24-bit number), while the double product is within 2^-52 of it -- at 3 instructions instead of the ~12 of
        // an IEEE division (eight of them per RoI and channel group).  The argument needs count < 2^8: a very large RoI -- gh * gw >=
        // 256 samples per bin -- takes the IEEE division.  ONE uniform branch around the channel loop (written per channel the
        // compiler emitted a chain of branches per channel); fast mode: 1 / count is folded into the row weights by map_prep_kernel.
        if (merged) {
        } else if (inv_count != 0.f) {
#pragma unroll
          for (int c = 0; c < CG; c++) res[c] = res[c] * inv_count;
        } else if (count < 256.f) {
#pragma unroll
          for (int c = 0; c < CG; c++) res[c] = (float)((double)res[c] * rcp_count);
        } else {
#pragma unroll
          for (int c = 0; c < CG; c++) res[c] = fdiv(res[c], count);
        }
        if (use_slab) {
          // bins <= 64, whole channel quads: [CG][bins] is ONE contiguous run of the output -> wave-private slab, 16-byte stores
          if (on) {
#pragma unroll
            for (int c = 0; c < CG; c++) slab[c * bins + bin] = res[c];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int n4 = (CG * bins) >> 2;
          for (int i = lane; i < n4; i += 64) map_store4<TOut>(orow + 4 * i, reinterpret_cast<const float4*>(slab)[i]);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        } else if (on) {
#pragma unroll
          for (int c = 0; c < CG; c++)
            if (c < nc) orow[(size_t)c * bins + bin] = from_f32<TOut>(res[c]);
        }
      }
      ri = rn; rec = recn;
    }
    // ---- rendezvous: every wave has finished the staged image (or the run); the lowest waiting RoI names the next image ----
    if (lane == 0) { pend_idx[wv] = want >= 0 ? ri : 0x7fffffff; pend_img[wv] = want; }
    __syncthreads();
    int best = 0x7fffffff, img = -1;
#pragma unroll
    for (int w = 0; w < kMapWaves; w++) {
      const int pi = pend_idx[w];
      if (pi < best) { best = pi; img = pend_img[w]; }
    }
    if (best == 0x7fffffff) break;            // uniform: nothing left
    // ---- stage the whole map of image `img`, channels [c0, c0 + nc): [quad][pixel][4 channels] float32 ------------------
    const TIn* src = reinterpret_cast<const TIn*>(L.data) + (int64_t)img * L.stride_n + (int64_t)c0 * L.stride_c;
    const bool rows_contig = L.stride_w == 1 && L.stride_h == W;
    const bool vec4 = rows_contig && (HW & 3) == 0 && ((L.stride_c | L.stride_n) & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(L.data) & (4 * sizeof(TIn) - 1)) == 0;
    constexpr int SU = 4;                                             // loads in flight per thread
    const int pad = pitch - W;                                         // LDS slot of map pixel px = px + (px / W) * pad
    const float rcpw = __frcp_rn((float)W);
    auto slot_of = [&](int px) { return px + (int)(((float)px + 0.5f) * rcpw) * pad; };     // px / W exact: px < 2^16, >= 0.5 / W from an integer
    if (vec4) {                                                       // a plane is one contiguous run: 4 pixels per load
      const int n4 = HW >> 2, total = CG * n4;
      for (int base = tid; base < total; base += SU * kMapThreads) {
        float4 v[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = min(base + u * kMapThreads, total - 1);
          const int c = e / n4, g = e - c * n4;
          v[u] = MapLoad4<TIn>::ld(src + (int64_t)min(c, nc - 1) * L.stride_c + 4 * g);   // channel tail: duplicate, never stored
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = base + u * kMapThreads;
          if (e < total) {
            const int c = e / n4, g = e - c * n4;
            float* d = mapw + (size_t)(c >> 2) * (plane_bytes >> 2) + (c & 3);
            if ((W & 3) == 0) {             // the four pixels of a load lie in one row
              d += slot_of(4 * g) * 4;
              d[0] = v[u].x; d[4] = v[u].y; d[8] = v[u].z; d[12] = v[u].w;
            } else {
              d[slot_of(4 * g) * 4] = v[u].x; d[slot_of(4 * g + 1) * 4] = v[u].y; d[slot_of(4 * g + 2) * 4] = v[u].z; d[slot_of(4 * g + 3) * 4] = v[u].w;
            }
          }
        }
      }
    } else {
      const int total = CG * HW;
      for (int base = tid; base < total; base += SU * kMapThreads) {
        float v[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = min(base + u * kMapThreads, total - 1);
          const int c = e / HW, px = e - c * HW;
          int64_t off = (int64_t)min(c, nc - 1) * L.stride_c;
          if (rows_contig) off += px;
          else { const int row = px / W, col = px - row * W; off += (int64_t)row * L.stride_h + (int64_t)col * L.stride_w; }
          v[u] = to_f32<TIn>(src[off]);
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = base + u * kMapThreads;
          if (e < total) { const int c = e / HW, px = e - c * HW; mapw[(size_t)(c >> 2) * (plane_bytes >> 2) + slot_of(px) * 4 + (c & 3)] = v[u]; }
        }
      }
    }
    __syncthreads();
    cur = img;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// channel quads per workgroup, whether the per-wave output slab fits beside the image, and the row pitch of the image: W + 1 for an
// even W (odd pitch: fewer LDS conflicts in the tap gather) whenever that costs neither the second quad nor the slab
static int map_nq(const RoiAlignParams& p, int* use_slab, int* pitch) {
  const int H = p.lv[0].height, W = p.lv[0].width;
  const int bins = p.pooled_h * p.pooled_w;
  const bool slab_ok = bins <= 64 && (p.channels & 3) == 0;
  static const bool pad_on = [] { const char* e = getenv("DTC_RA_MAP_PITCH"); return !(e && e[0] == '0'); }();     // A/B: DTC_RA_MAP_PITCH=0
  for (int nq = 2; nq >= 1; nq--) {
    if ((p.channels % (4 * nq)) != 0 && nq > 1) continue;            // whole channel groups (a tail only with single quads)
    const long long slab = slab_ok ? (long long)kMapWaves * 4 * nq * bins * 4 : 0;
    for (int with_slab = 1; with_slab >= 0; with_slab--) {
      for (int padded = (pad_on && (W & 1) == 0) ? 1 : 0; padded >= 0; padded--) {
        const long long image = (long long)H * (W + padded) * 16 * nq;
        if (image + (with_slab ? slab : 0) <= kMapLdsBytes) { *use_slab = with_slab && slab_ok ? 1 : 0; *pitch = W + padded; return nq; }
      }
    }
  }
  return 0;
}

bool roi_align_map_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  if (p.n_levels != 1) return false;
  if (!(p.roi_desc || p.roi_cols == 4)) return false;                // image-major order is known only for these callers
  if (p.lv[0].stride_c == 1 && p.channels > 1) return false;         // channels_last: the NHWC / LDS kernels
  if ((long long)p.n_rois * p.channels < 64 * 1024) return false;    // too little work to pay for staging whole maps
  int slab, pitch;
  if (map_nq(p, &slab, &pitch) == 0) return false;
  const bool f = in_dtype == DTC_F32, h = in_dtype == DTC_F16, b = in_dtype == DTC_BF16;
  return (f && (out_dtype == DTC_F32 || out_dtype == DTC_F16 || out_dtype == DTC_BF16)) ||
         (h && (out_dtype == DTC_F32 || out_dtype == DTC_F16)) || (b && (out_dtype == DTC_F32 || out_dtype == DTC_BF16));
}

template <typename TIn, typename TOut, int NQ>
static int launch_map_nq(const RoiAlignParams& p, int use_slab, int pitch, hipStream_t stream) {
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_map<TIn, TOut, NQ, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kMapLdsBytes);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_map<TIn, TOut, NQ, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kMapLdsBytes);
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  const int bins = p.pooled_h * p.pooled_w;
  const int ncg = ceil_div(p.channels, 4 * NQ);
  // runs of RoIs per workgroup: ~8 workgroups per CU over the launch (one is resident per CU), at least 8 RoIs per wave
  int n_seg = (8 * 256 + ncg - 1) / ncg;
  int seg_len = ceil_div(p.n_rois, n_seg < 1 ? 1 : n_seg);
  if (seg_len < 8 * kMapWaves) seg_len = 8 * kMapWaves;
  seg_len = ceil_div(seg_len, kMapWaves) * kMapWaves;
  n_seg = ceil_div(p.n_rois, seg_len);
  const size_t lds = (size_t)p.lv[0].height * pitch * 16 * NQ + (use_slab ? (size_t)kMapWaves * 4 * NQ * bins * 4 : 0);
  if (p.prep) {
    hipLaunchKernelGGL(map_prep_kernel, dim3((unsigned)ceil_div(p.n_rois, 4)), dim3(256), 0, stream, p,
                       reinterpret_cast<MapPrepRoi*>(const_cast<void*>(p.prep)), roi_align_get_exact() ? 0 : 1, pitch);
    DTC_CHECK_LAUNCH();
    hipLaunchKernelGGL((roi_align_fwd_map<TIn, TOut, NQ, true>), dim3((unsigned)(ncg * n_seg)), dim3(kMapThreads), lds, stream, p,
                       seg_len, use_slab, pitch);
  } else {
    hipLaunchKernelGGL((roi_align_fwd_map<TIn, TOut, NQ, false>), dim3((unsigned)(ncg * n_seg)), dim3(kMapThreads), lds, stream, p,
                       seg_len, use_slab, pitch);
  }
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

template <typename TIn, typename TOut>
static int launch_map_t(const RoiAlignParams& p, hipStream_t stream) {
  int use_slab = 0;
  int pitch = 0;
  const int nq = map_nq(p, &use_slab, &pitch);
  if (nq == 2) return launch_map_nq<TIn, TOut, 2>(p, use_slab, pitch, stream);
  if (nq == 1) return launch_map_nq<TIn, TOut, 1>(p, use_slab, pitch, stream);
  return DTC_EUNSUPPORTED;
}

int launch_roi_align_map(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return launch_map_t<float, float>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return launch_map_t<__half, float>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return launch_map_t<__half, __half>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F16) return launch_map_t<float, __half>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_F32) return launch_map_t<bf16_t, float>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_BF16) return launch_map_t<bf16_t, bf16_t>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_BF16) return launch_map_t<float, bf16_t>(p, stream);
  return DTC_EUNSUPPORTED;
}

// dtc_roi_align_set_exact(): process-wide; 1 (default) = the reference's float32 operations in the reference's order, bit-identical;
// 0 = the map-stationary kernel may merge taps (above).  Read at launch time.
static std::atomic<int> g_map_exact{1};
void roi_align_set_exact(int exact) { g_map_exact.store(exact ? 1 : 0); }
int roi_align_get_exact() { return g_map_exact.load(); }

size_t roi_align_map_workspace_bytes(int n_rois) { return ((size_t)(n_rois > 0 ? n_rois : 1) * sizeof(MapPrepRoi) + 255) & ~(size_t)255; }

int launch_roi_align_map_ws(const RoiAlignParams& p0, int in_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  RoiAlignParams p = p0;
  static const bool no_prep = [] { const char* e = getenv("DTC_RA_MAP_PREP"); return e && atoi(e) == 0; }();   // A/B knob
  p.prep = nullptr;
  if (workspace && !no_prep && workspace_bytes >= roi_align_map_workspace_bytes(p.n_rois) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
      p.pooled_h <= 64 && p.pooled_w <= 64 && (long long)p.lv[0].height * p.lv[0].width * 16 < (1ll << 30))
    p.prep = workspace;
  return launch_roi_align_map(p, in_dtype, out_dtype, stream);
}

}  // namespace dtc
