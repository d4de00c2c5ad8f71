// RoIAlign forward for 16-bit channels_last (NHWC) feature maps, sampling_ratio 2: the direct-gather kernel of roi_align.hip
// (lane <-> 8 channels of one bin, all 16 taps of a bin as 16-byte loads straight from L1 / L2, packed fp32 pooling) with the bins
// of G CONSECUTIVE RoIs flattened over the lanes.
//
// Why: with one RoI per workgroup a 7 x 7 RoI's 49 bins x 8 lanes fill 6.125 wavefronts but cost 7 wavefront rounds (and the RoI
// record, the axis tables and two barriers are paid per 64 channels of one RoI); 14 x 14 bins with 32-channel blocks cost 4 rounds of
// 64 bin slots for 196 bins.  Flattened, G x bins x 8 lanes are dealt round-robin: 4 RoIs of 7 x 7 cost 25 rounds instead of 28 and
// one prologue instead of four.  The output slab is kept in the OUTPUT type (rounded once, where the value is final), which halves
// its LDS for 16-bit outputs and turns the store phase into plain 16-byte copies.
//
// Arithmetic: exactly the reference's (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:36-116, 143-216): positions and weights by
// make_axis, per bin  sum over (iy, ix) of (w1*v1 + w2*v2 + w3*v3 + w4*v4)  left to right, unfused, then / count (count = 4: * 0.25
// is the same value).  Bit-identical to the other kernels; replaces lib/cppcuda/roi_align_forward_cuda.cu:82-159 for this layout.
//
// Addressing: the launch has ONE scalar base (the lowest map address); a RoI's image is a 32-bit byte offset from it whenever every
// tap of the image lies below 4 GB from that base ("near": global_load ... saddr with a 32-bit lane offset -- one v_add per tap).
// A workgroup with a RoI whose image lies further takes the same loop with 64-bit lane addresses.
#include "roi_align_common.h"

namespace dtc {

constexpr int kN16Threads = 256;
constexpr int kN16MaxG = 8;
constexpr int kN16SlabBytes = 25 * 1024 + 512;   // output slab budget per workgroup (4 RoIs x 64 channels x 49 bins x 2 B = 25 088)

struct N16Roi {            // per RoI of the group (LDS)
  int r, lvl;              // output row; level (< 0: padding row -> zeros)
  uint32_t rebase, far;    // near: byte offset of (image, channel 0) from the launch base; far != 0: use `ptr`
  const char* ptr;         // (image, channel 0) of its level
};
struct N16Tab { uint32_t lo, hi; float l, h; };     // byte offsets inside the image (y: row, x: pixel) + the bilinear weights

template <typename TOut> __device__ __forceinline__ void n16_put(TOut* d, float v) { *d = from_f32<TOut>(v); }

// channel pair k of a 16-byte tap load times the tap weight: 16-bit maps hold 4 pairs per load (one dword each), float32 maps 2
template <typename TIn> __device__ __forceinline__ f32x2 n16_tap_pair(const uint32_t* u, int k, float w) { return mul_pair16<TIn>(u[k], w); }
template <> __device__ __forceinline__ f32x2 n16_tap_pair<float>(const uint32_t* u, int k, float w) {
  const f32x2 v = {__uint_as_float(u[2 * k]), __uint_as_float(u[2 * k + 1])};
  return v * w;
}

// the 16 taps of one bin (8 channels each) and the pooled 8 channels -> slab.  ADDR: callable (y_off, x_off) -> uint4
template <typename TIn, typename TOut, bool FUSED, typename LD>
__device__ __forceinline__ void n16_pool_bin(const N16Tab& y0e, const N16Tab& y1e, const N16Tab& x0e, const N16Tab& x1e, LD ld,
                                             TOut* so, int bins) {
  uint4 t[16];        // all 16 taps in flight before the first one is consumed
  t[0] = ld(y0e.lo, x0e.lo); t[1] = ld(y0e.lo, x0e.hi); t[2] = ld(y0e.hi, x0e.lo); t[3] = ld(y0e.hi, x0e.hi);
  t[4] = ld(y0e.lo, x1e.lo); t[5] = ld(y0e.lo, x1e.hi); t[6] = ld(y0e.hi, x1e.lo); t[7] = ld(y0e.hi, x1e.hi);
  t[8] = ld(y1e.lo, x0e.lo); t[9] = ld(y1e.lo, x0e.hi); t[10] = ld(y1e.hi, x0e.lo); t[11] = ld(y1e.hi, x0e.hi);
  t[12] = ld(y1e.lo, x1e.lo); t[13] = ld(y1e.lo, x1e.hi); t[14] = ld(y1e.hi, x1e.lo); t[15] = ld(y1e.hi, x1e.hi);
  __builtin_amdgcn_sched_barrier(0);      // keep the 16 loads ahead of the arithmetic (the scheduler otherwise interleaves them to save registers)
  constexpr int PP = 8 / (int)sizeof(TIn);      // channel pairs per lane: 4 (16-bit maps, 8 channels) or 2 (float32 maps, 4 channels)
  f32x2 a[PP];
#pragma unroll
  for (int k = 0; k < PP; k++) a[k] = f32x2{0.f, 0.f};
#pragma unroll
  for (int sidx = 0; sidx < 4; sidx++) {   // (iy, ix) = (0,0) (0,1) (1,0) (1,1): the reference's accumulation order
    const N16Tab& y = (sidx >> 1) ? y1e : y0e;
    const N16Tab& x = (sidx & 1) ? x1e : x0e;
    const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;      // roi_align_cpu_loop.cpp:95
    const uint32_t* u1 = reinterpret_cast<const uint32_t*>(&t[sidx * 4]);
    const uint32_t* u2 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 1]);
    const uint32_t* u3 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 2]);
    const uint32_t* u4 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 3]);
    if constexpr (FUSED && sizeof(TIn) == 2) {
      // contract mode on 16-bit maps (dtc_roi_align_set_exact(0)): one fused convert-multiply-accumulate per element (fma_pair16)
#pragma unroll
      for (int k = 0; k < PP; k++) {
        fma_pair16<TIn>(a[k], u1[k], w1); fma_pair16<TIn>(a[k], u2[k], w2);
        fma_pair16<TIn>(a[k], u3[k], w3); fma_pair16<TIn>(a[k], u4[k], w4);
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < PP; k++) {          // pair k: channels 2k and 2k + 1 of the lane
      f32x2 s = n16_tap_pair<TIn>(u1, k, w1) + n16_tap_pair<TIn>(u2, k, w2);
      s = s + n16_tap_pair<TIn>(u3, k, w3);
      s = s + n16_tap_pair<TIn>(u4, k, w4);
      a[k] = a[k] + s;
    }
  }
#pragma unroll
  for (int k = 0; k < PP; k++) {
    const f32x2 o = a[k] * 0.25f;           // / count, count = 4 (:216)
    n16_put<TOut>(so + (2 * k) * bins, o.x); n16_put<TOut>(so + (2 * k + 1) * bins, o.y);
  }
}

// CONTRACT mode on 16-bit maps (dtc_roi_align_set_exact(0)): the 16 taps of a bin share pixels whenever its two samples per axis are
// less than a pixel apart (bins under 2 px: the small RoIs that fill P2 -- 94 % of the bench's proposals): sample 1's low row IS
// sample 0's low or high row, and so for the columns; 4-9 distinct pixels instead of 16.  The direct-gather kernel is bound by the
// texture data path (16 x 16 B per lane and bin through a 64 B/clk return path, TD 83 % busy), so here every DISTINCT pixel is
// requested once: per axis the four (offset, weight) entries are merged where the offsets coincide (weights summed -- bilinear weights
// are separable, sum of products == product of sums in exact arithmetic, a few float32 ulp apart in float32: inside the 1e-4 contract,
// not bit-identical, hence contract mode only), a pixel (row entry i, column entry j) is loaded by the lanes whose entries i and j
// are both live, and the 8 fused multiply-accumulates of a pixel are skipped by a wavefront in which no lane holds it (entry 2 of an
// axis is never live for bins under 2 px: 7 of the 16 combinations).
struct N16Rec { uint32_t off[4]; float w[4]; };      // one bin row / column, merged: bit 0 of off[1..3] = entry is live (offsets are multiples of 16)
__device__ __forceinline__ N16Rec n16_merge_axis(const N16Tab& e0, const N16Tab& e1) {
  uint32_t off[4] = {e0.lo, e0.hi, e1.lo, e1.hi};
  float w[4] = {e0.h, e0.l, e1.h, e1.l};
  bool live[4];
  live[0] = true;
  live[1] = off[1] != off[0];                           // clamped at the last row / column: lo == hi
  if (!live[1]) w[0] += w[1];
#pragma unroll
  for (int k = 2; k < 4; k++) {
    live[k] = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < k && live[k] && live[j] && off[k] == off[j]) { w[j] += w[k]; live[k] = false; }
    }
  }
  N16Rec r;      // a folded entry keeps its (valid, duplicate) offset and gets weight 0: it may be read and accumulated unpredicated
#pragma unroll
  for (int k = 0; k < 4; k++) { r.off[k] = off[k] | (k && live[k] ? 1u : 0u); r.w[k] = live[k] ? w[k] : 0.f; }
  return r;
}

template <typename TIn, typename TOut, typename LD>
__device__ __forceinline__ void n16_pool_bin_shared(const N16Rec& ya, const N16Rec& xa, LD ld, TOut* so, int bins) {
  bool ly[4], lx[4];
  uint32_t oy[4], ox[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { ly[k] = k == 0 || (ya.off[k] & 1u); lx[k] = k == 0 || (xa.off[k] & 1u); oy[k] = ya.off[k] & ~1u; ox[k] = xa.off[k] & ~1u; }
  f32x2 a[4];
#pragma unroll
  for (int k = 0; k < 4; k++) a[k] = f32x2{0.f, 0.f};
  // Wavefronts whose bins are all under 2 px (no lane holds entry 2 of either axis: the usual case on P2) take STRAIGHT-LINE code over
  // the nine pixels {0, 1, 3} x {0, 1, 3}: a lane whose entry 1 or 3 is folded re-reads a pixel it holds anyway (an L1 hit) with
  // weight 0 -- a few more loads than the masked form below, none of its 32 exec-mask blocks.
  if (__ballot(ly[2] || lx[2]) == 0ull) {
    uint4 t9[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) t9[3 * i + j] = ld(oy[i == 2 ? 3 : i], ox[j == 2 ? 3 : j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const float w = ya.w[i == 2 ? 3 : i] * xa.w[j == 2 ? 3 : j];
        const uint32_t* u = reinterpret_cast<const uint32_t*>(&t9[3 * i + j]);
#pragma unroll
        for (int k = 0; k < 4; k++) fma_pair16<TIn>(a[k], u[k], w);
      }
  } else {
    uint4 t[16];              // a lane's t[4 i + j] is defined (and read) only where its entries i and j are both live
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (ly[i] && lx[j]) t[4 * i + j] = ld(oy[i], ox[j]);                      // only the lanes that hold this pixel request it
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (ly[i] && lx[j]) {                                                     // (a wavefront without such a lane skips the block)
          const float w = ya.w[i] * xa.w[j];
          const uint32_t* u = reinterpret_cast<const uint32_t*>(&t[4 * i + j]);
#pragma unroll
          for (int k = 0; k < 4; k++) fma_pair16<TIn>(a[k], u[k], w);
        }
      }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const f32x2 o = a[k] * 0.25f;
    n16_put<TOut>(so + (2 * k) * bins, o.x); n16_put<TOut>(so + (2 * k + 1) * bins, o.y);
  }
}

// CB: channels per workgroup (64, or 32 when a 64-channel slab of one RoI exceeds the budget: 14 x 14 bins with float32 output)
template <typename TIn, typename TOut, int CB, bool FUSED>
// (98 VGPRs: four workgroups per CU; bounding it to five -- 96 VGPRs -- measured no difference and spilled the bf16 variants)
__global__ __launch_bounds__(kN16Threads) void roi_align_fwd_nhwc16(RoiAlignParams p, const char* base0, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bins = p.pooled_h * p.pooled_w;
  const int ny = 2 * p.pooled_h, ne = 2 * (p.pooled_h + p.pooled_w);
  N16Roi* rinfo = reinterpret_cast<N16Roi*>(smem);
  N16Tab* tabs = reinterpret_cast<N16Tab*>(smem + kN16MaxG * sizeof(N16Roi));
  TOut* slab = reinterpret_cast<TOut*>(smem + kN16MaxG * sizeof(N16Roi) + (size_t)G * ne * sizeof(N16Tab));
  constexpr int ESZ = (int)sizeof(TIn), CPL = 16 / ESZ;      // bytes per element; channels per lane (one 16-byte load per tap)
  const int tid = threadIdx.x;
  const int nct = p.channels / CB;
  const int wi = xcd_work_item(blockIdx.x, gridDim.x, p.xcd_remap);
  const int grp = wi / nct;
  const int c0 = (wi - grp * nct) * CB;
  const int ri0 = grp * G;
  const int ng = min(G, p.n_rois - ri0);

  constexpr bool SHARED = FUSED && sizeof(TIn) == 2;       // contract mode on 16-bit maps: merged bin rows / columns (N16Rec) instead of per-sample entries
  const int nrec = p.pooled_h + p.pooled_w;
  N16Rec* recs = reinterpret_cast<N16Rec*>(tabs);            // the same bytes: ne entries of 16 B == nrec records of 32 B
  // ---- A. axis tables of the group's RoIs: thread <-> (RoI, table entry); entry 0's thread also files the RoI ----------------
  for (int t = tid; t < ng * (SHARED ? nrec : ne); t += kN16Threads) {
    if constexpr (SHARED) {
      const int rl = t / nrec, e = t - rl * nrec;
      const RoiHead hd = load_roi_head(p, ri0 + rl);
      N16Rec rec;
#pragma unroll
      for (int k = 0; k < 4; k++) { rec.off[k] = 0u; rec.w[k] = 0.f; }
      N16Roi info; info.r = hd.r; info.lvl = -1; info.rebase = 0u; info.far = 0u; info.ptr = base0;
      if (hd.lvl >= 0 && hd.lvl < p.n_levels) {
        const dtc_feat_level& L = p.lv[hd.lvl];
        const bool isy = e < p.pooled_h;
        const int u = isy ? e : e - p.pooled_h;
        const uint32_t sb = (uint32_t)ESZ * (uint32_t)(isy ? L.stride_h : L.stride_w);
        N16Tab o2[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const AxisEntry a = isy ? make_axis(hd.sh, hd.bin_h, u, i, 2, L.height) : make_axis(hd.sw, hd.bin_w, u, i, 2, L.width);
          o2[i].lo = (uint32_t)a.lo * sb; o2[i].hi = (uint32_t)a.hi * sb; o2[i].l = a.l; o2[i].h = a.h;
        }
        rec = n16_merge_axis(o2[0], o2[1]);
        if (e == 0) {
          const char* img = reinterpret_cast<const char*>(L.data) + ESZ * (int64_t)hd.b * L.stride_n;
          const uint64_t delta = (uint64_t)(img - base0);
          const uint64_t ext = (uint64_t)ESZ * ((uint64_t)L.stride_h * (uint64_t)L.height + (uint64_t)L.stride_w * (uint64_t)L.width + (uint64_t)p.channels);
          info.lvl = hd.lvl; info.ptr = img;
          info.far = (delta + ext) >= (1ull << 32) ? 1u : 0u;
          info.rebase = (uint32_t)delta;
        }
      }
      recs[t] = rec;
      if (e == 0) rinfo[rl] = info;
      continue;
    }
    const int rl = t / ne, e = t - rl * ne;
    const RoiHead hd = load_roi_head(p, ri0 + rl);
    N16Tab o; o.lo = o.hi = 0u; o.l = o.h = 0.f;
    N16Roi info; info.r = hd.r; info.lvl = -1; info.rebase = 0u; info.far = 0u; info.ptr = base0;
    if (hd.lvl >= 0 && hd.lvl < p.n_levels) {
      const dtc_feat_level& L = p.lv[hd.lvl];
      const bool isy = e < ny;
      const int u = isy ? e : e - ny;
      const AxisEntry a = isy ? make_axis(hd.sh, hd.bin_h, u >> 1, u & 1, 2, L.height) : make_axis(hd.sw, hd.bin_w, u >> 1, u & 1, 2, L.width);
      const uint32_t sb = (uint32_t)ESZ * (uint32_t)(isy ? L.stride_h : L.stride_w);
      o.lo = (uint32_t)a.lo * sb; o.hi = (uint32_t)a.hi * sb; o.l = a.l; o.h = a.h;
      if (e == 0) {
        const char* img = reinterpret_cast<const char*>(L.data) + ESZ * (int64_t)hd.b * L.stride_n;
        const uint64_t delta = (uint64_t)(img - base0);
        const uint64_t ext = (uint64_t)ESZ * ((uint64_t)L.stride_h * (uint64_t)L.height + (uint64_t)L.stride_w * (uint64_t)L.width + (uint64_t)p.channels);
        info.lvl = hd.lvl; info.ptr = img;
        info.far = (delta + ext) >= (1ull << 32) ? 1u : 0u;
        info.rebase = (uint32_t)delta;
      }
    }
    tabs[t] = o;
    if (e == 0) rinfo[rl] = info;
  }
  __syncthreads();
  bool any_far = false;
  for (int k = 0; k < ng; k++) any_far = any_far || rinfo[k].far != 0u;
  any_far = __builtin_amdgcn_readfirstlane((int)any_far) != 0;

  // ---- B. bins of all RoIs of the group, dealt over the lanes: item = slot, slot + kStep, ... ---------------------------------
  constexpr int NQ8 = CB / CPL, kStep = kN16Threads / NQ8;      // lanes per bin; bins per round
  const int q8 = tid % NQ8, slot = tid / NQ8;
  const int items = ng * bins;
  // (RoI, bin row, bin column) of the lane's item as a mixed-radix counter: no division per round
  const int step_r = kStep / bins, step_b = kStep - step_r * bins;
  const int step_h = step_b / p.pooled_w, step_w = step_b - step_h * p.pooled_w;
  int rl = slot / bins;
  int ph, pw;
  { const int bin = slot - rl * bins; ph = bin / p.pooled_w; pw = bin - ph * p.pooled_w; }
  const uint32_t lane_off = 16u * (uint32_t)q8;
  const char* sbase = base0 + ESZ * c0;                                  // scalar: near RoIs
  auto pool_items = [&](auto far_tag) {
    constexpr bool FAR = decltype(far_tag)::value;
    for (int it = slot; it < items; it += kStep) {
      const N16Tab* tab = tabs + rl * ne;
      N16Tab y0e, y1e, x0e, x1e;
      N16Rec yr, xr;
      if constexpr (SHARED) { yr = recs[rl * nrec + ph]; xr = recs[rl * nrec + p.pooled_h + pw]; }
      else { y0e = tab[2 * ph]; y1e = tab[2 * ph + 1]; x0e = tab[ny + 2 * pw]; x1e = tab[ny + 2 * pw + 1]; }
      const N16Roi info = rinfo[rl];
      TOut* so = slab + ((size_t)rl * CB + CPL * q8) * bins + (ph * p.pooled_w + pw);
      if (info.lvl < 0) {            // padding row of a fixed-shape batch: defined output
#pragma unroll
        for (int k = 0; k < CPL; k++) n16_put<TOut>(so + k * bins, 0.f);
      } else if constexpr (FAR) {
        const char* lbase = info.ptr + ESZ * c0 + lane_off;
        if constexpr (SHARED) n16_pool_bin_shared<TIn, TOut>(yr, xr, [&](uint32_t yo, uint32_t xo) { return *reinterpret_cast<const uint4*>(lbase + (yo + xo)); }, so, bins);
        else n16_pool_bin<TIn, TOut, FUSED>(y0e, y1e, x0e, x1e, [&](uint32_t yo, uint32_t xo) { return *reinterpret_cast<const uint4*>(lbase + (yo + xo)); }, so, bins);
      } else {
        const uint32_t lo = info.rebase + lane_off;
        if constexpr (SHARED) n16_pool_bin_shared<TIn, TOut>(yr, xr, [&](uint32_t yo, uint32_t xo) { return *reinterpret_cast<const uint4*>(sbase + (yo + lo + xo)); }, so, bins);
        else n16_pool_bin<TIn, TOut, FUSED>(y0e, y1e, x0e, x1e, [&](uint32_t yo, uint32_t xo) { return *reinterpret_cast<const uint4*>(sbase + (yo + lo + xo)); }, so, bins);
      }
      pw += step_w; if (pw >= p.pooled_w) { pw -= p.pooled_w; ph++; }
      ph += step_h; if (ph >= p.pooled_h) { ph -= p.pooled_h; rl++; }
      rl += step_r;
    }
  };
  if (any_far) pool_items(std::true_type{}); else pool_items(std::false_type{});
  __syncthreads();

  // ---- C. slab -> output: a RoI's CB x bins values are one contiguous run (16-byte aligned: checked by the launcher) -----------
  const int n16 = (int)((size_t)CB * bins * sizeof(TOut) / 16);
  for (int i = tid; i < ng * n16; i += kN16Threads) {
    const int k = i / n16, j = i - k * n16;
    TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)rinfo[k].r * p.channels + c0) * bins;
    const uint4 v = reinterpret_cast<const uint4*>(slab + (size_t)k * CB * bins)[j];
    store_stream16(reinterpret_cast<uint4*>(out) + j, v.x, v.y, v.z, v.w);
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static bool n16_enabled() {      // A/B: DTC_RA_NHWC16=0 sends 16-bit channels_last launches to roi_align_fwd_nhwc (one RoI per workgroup)
  static const bool on = [] { const char* e = getenv("DTC_RA_NHWC16"); return !(e && e[0] == '0'); }();
  return on;
}
static bool n16_f32() {          // DTC_RA_NHWC_DIRECT32=0: float32 channels_last maps with <= 64 bins stay on the LDS-DMA kernels (roi_align_nhwc.hip)
  static const bool on = [] { const char* e = getenv("DTC_RA_NHWC_DIRECT32"); return !(e && e[0] == '0'); }();
  return on;
}
static int n16_out_size(int out_dtype) { return out_dtype == DTC_F32 ? 4 : 2; }
static int n16_cb(const RoiAlignParams& p, int out_dtype) {
  const long long per64 = 64ll * p.pooled_h * p.pooled_w * n16_out_size(out_dtype);
  return per64 <= kN16SlabBytes ? 64 : 32;
}

bool roi_align_nhwc16_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  if (!n16_enabled() || p.sampling_ratio != 2 || p.n_rois < 1) return false;
  // float32 maps: 4-channel lanes.  Measured against the LDS-DMA kernels (tools/r04/README.md section 6): box head 0.375 against 0.393 ms on
  // the bench RoIs and 0.40 against 0.51 on the harder set (no window to stage: the cost does not depend on the RoI size); the
  // 14 x 14 mask head 0.148 against 0.130 -> taken for <= 64 bins only
  if (in_dtype == DTC_F32) { if (!n16_f32() || out_dtype != DTC_F32 || p.pooled_h * p.pooled_w > 64) return false; }
  else if (in_dtype != DTC_F16 && in_dtype != DTC_BF16) return false;
  if (out_dtype != DTC_F32 && out_dtype != in_dtype) return false;
  const int epl = in_dtype == DTC_F32 ? 4 : 8;          // elements per 16-byte load
  const int cb = n16_cb(p, out_dtype), bins = p.pooled_h * p.pooled_w, osz = n16_out_size(out_dtype);
  if (p.channels % cb != 0 || (long long)cb * bins * osz > kN16SlabBytes || ((long long)cb * bins * osz) % 16 != 0) return false;
  if (((long long)p.channels * bins * osz) % 16 != 0 || (reinterpret_cast<uintptr_t>(p.out) & 15) != 0) return false;
  if (2 * (p.pooled_h + p.pooled_w) > 256) return false;
  for (int i = 0; i < p.n_levels; i++) {
    const dtc_feat_level& L = p.lv[i];
    if (L.stride_c != 1 || ((L.stride_h | L.stride_w | L.stride_n) & (epl - 1)) != 0 || (reinterpret_cast<uintptr_t>(L.data) & 15) != 0) return false;
    if (L.stride_h < 0 || L.stride_w < 0 || L.stride_n < 0) return false;
    if ((long long)L.stride_h * L.height + (long long)L.stride_w * L.width + p.channels >= (1ll << 30)) return false;   // 32-bit byte offsets inside an image
  }
  return true;
}

template <typename TIn, typename TOut, int CB, bool FUSED>
static int launch_n16_t(const RoiAlignParams& p, hipStream_t stream) {
  const int bins = p.pooled_h * p.pooled_w, ne = 2 * (p.pooled_h + p.pooled_w);
  int G = (int)(kN16SlabBytes / ((size_t)CB * bins * sizeof(TOut)));
  G = G < 1 ? 1 : G > kN16MaxG ? kN16MaxG : G;
  const char* base0 = reinterpret_cast<const char*>(p.lv[0].data);
  for (int i = 1; i < p.n_levels; i++) if (reinterpret_cast<const char*>(p.lv[i].data) < base0) base0 = reinterpret_cast<const char*>(p.lv[i].data);
  const size_t smem = kN16MaxG * sizeof(N16Roi) + (size_t)G * ne * sizeof(N16Tab) + (size_t)G * CB * bins * sizeof(TOut);
  const int nct = p.channels / CB;
  const int ngrp = (p.n_rois + G - 1) / G;
  hipLaunchKernelGGL((roi_align_fwd_nhwc16<TIn, TOut, CB, FUSED>), dim3((unsigned)ngrp * nct), dim3(kN16Threads), smem, stream, p, base0, G);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

template <typename TIn, typename TOut>
static int launch_n16_cb(const RoiAlignParams& p, int cb, hipStream_t stream) {
  if constexpr (sizeof(TIn) == 2) {       // contract mode, read at launch time (roi_align_common.h: fma_pair16)
    if (!roi_align_get_exact()) return cb == 64 ? launch_n16_t<TIn, TOut, 64, true>(p, stream) : launch_n16_t<TIn, TOut, 32, true>(p, stream);
  }
  return cb == 64 ? launch_n16_t<TIn, TOut, 64, false>(p, stream) : launch_n16_t<TIn, TOut, 32, false>(p, stream);
}

int launch_roi_align_nhwc16(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream) {
  const int cb = n16_cb(p, out_dtype);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return launch_n16_cb<float, float>(p, cb, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return launch_n16_cb<__half, __half>(p, cb, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return launch_n16_cb<__half, float>(p, cb, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_BF16) return launch_n16_cb<bf16_t, bf16_t>(p, cb, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_F32) return launch_n16_cb<bf16_t, float>(p, cb, stream);
  return DTC_EINVAL;
}

}  // namespace dtc
