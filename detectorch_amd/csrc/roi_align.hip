// A1  RoIAlign forward for gfx950.
//
// Replaces roi_align_forward_kernel (lib/cppcuda/roi_align_forward_cuda.cu:82-159, cffi twin
// lib/cppcuda_cffi/src/cuda/roi_align_forward_cuda_kernel.cu:83) and is bit-compatible with the CPU path
// roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:118-219): same float32 operations in the same
// order, so the pooled features equal the reference CPU output exactly (tests require <= 1e-4, observe 0).
//
// What is different from the reference kernel (one thread per output element, every thread re-deriving the RoI
// geometry and issuing 4*g^2 scattered loads):
//   * one workgroup per (RoI, channel tile); the RoI geometry is computed ONCE per workgroup into LDS as two separable
//     per-axis tables (the reference's PreCalc entry (ph,iy,pw,ix) is exactly ytab[ph,iy] x xtab[pw,ix]; w1 = hy*hx etc.
//     are formed with the same multiplies, roi_align_cpu_loop.cpp:95);
//   * all FPN levels in one launch (per-RoI level id), output written directly in RoI order -- no cat / index_select
//     (lib/model/detector.py:263-270);
//   * element strides instead of a fixed NCHW layout, fp32 / fp16 / bf16 features, fp32 accumulation.
//
// Kernels in this file (all produce bit-identical results; tests/test_hip_roi_align.py runs every one against the oracle):
//   roi_align_fwd_lds      RoI-stationary: one workgroup per (RoI, channel block), window staged in LDS through a register
//                          prefetch pipeline; stagers StagerNCHW (dword per lane) and StagerNHWC; windows that do not fit are
//                          pooled in slices of bin rows.  The path for multi-level inputs with sampling ratios other than 2
//                          (incl. adaptive), channels_last 14x14, and the A/B partner of the other two RoIAlign kernels.
//   roi_align_fwd_nhwc     channels_last features with few taps per pixel: taps gathered straight from L1/L2
//   roi_align_fwd_general  per-output gather; oversize pooled sizes, and the plain statement of the arithmetic
// The default for the FPN heads (NCHW features, sampling_ratio 2) is the cluster-stationary kernel in roi_align_tile.hip,
// for single-level inputs with adaptive sampling (the C4 heads) the map-stationary kernel in roi_align_map.hip.
// Knobs (resolved once per process): see RoiAlignConfig below.  Variants measured slower in round 1 (wave-specialised
// loader/compute waves, LDS-DMA staging, 16-byte row pieces, row-slot chunking, quad-aligned windows, pixel-pair loads)
// were removed in round 2; DESIGN.md section 3.1 keeps their numbers.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "roi_align_common.h"

namespace dtc {

constexpr int kRoiAlignThreads = 256;
constexpr int kMaxTableEntries = 2048;  // PH*gh + PW*gw ; larger (adaptive sampling on a huge RoI) -> on-the-fly path

// General kernel: any strides, any pooled size, adaptive sampling.  thread <-> output element (c, ph, pw) of the tile,
// consecutive threads -> consecutive addresses of the [R,C,PH,PW] output (coalesced stores).
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kRoiAlignThreads) void roi_align_fwd_general(RoiAlignParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);

  const int c0 = blockIdx.y * p.ch_tile;
  const RoiHead hd = load_roi_head(p, blockIdx.x);
  const int r = hd.r, lvl = hd.lvl, b = hd.b;
  if (lvl < 0 || lvl >= p.n_levels) {  // padding row of a fixed-shape batch (fpn.hip emits level -1): defined output
    const int bins0 = p.pooled_h * p.pooled_w;
    const int nc0 = min(p.ch_tile, p.channels - c0);
    TOut* o0 = reinterpret_cast<TOut*>(p.out) + ((size_t)r * p.channels + c0) * bins0;
    for (int o = threadIdx.x; o < nc0 * bins0; o += kRoiAlignThreads) o0[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[lvl];
  const float sw = hd.sw, sh = hd.sh, bin_h = hd.bin_h, bin_w = hd.bin_w, count = hd.count;
  const int gh = hd.gh, gw = hd.gw;
  const int ny = p.pooled_h * gh, nx = p.pooled_w * gw;
  const bool use_tab = (ny + nx) <= kMaxTableEntries;
  AxisEntry* xtab = ytab + ny;
  if (use_tab) {
    for (int t = threadIdx.x; t < ny + nx; t += kRoiAlignThreads) {
      if (t < ny) ytab[t] = make_axis(sh, bin_h, t / gh, t % gh, gh, L.height);
      else { int u = t - ny; xtab[u] = make_axis(sw, bin_w, u / gw, u % gw, gw, L.width); }
    }
    __syncthreads();
  }

  const int bins = p.pooled_h * p.pooled_w;
  const int nc = min(p.ch_tile, p.channels - c0);
  const TIn* base = reinterpret_cast<const TIn*>(L.data) + (int64_t)b * L.stride_n;
  TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)r * p.channels + c0) * bins;

  for (int o = threadIdx.x; o < nc * bins; o += kRoiAlignThreads) {
    const int c = o / bins, bin = o - c * bins;
    const int ph = bin / p.pooled_w, pw = bin - ph * p.pooled_w;
    const TIn* d = base + (int64_t)(c0 + c) * L.stride_c;
    float acc = 0.f;
    for (int iy = 0; iy < gh; iy++) {
      const AxisEntry y = use_tab ? ytab[ph * gh + iy] : make_axis(sh, bin_h, ph, iy, gh, L.height);
      const int64_t ylo = (int64_t)y.lo * L.stride_h, yhi = (int64_t)y.hi * L.stride_h;
      for (int ix = 0; ix < gw; ix++) {
        const AxisEntry x = use_tab ? xtab[pw * gw + ix] : make_axis(sw, bin_w, pw, ix, gw, L.width);
        const int64_t xlo = (int64_t)x.lo * L.stride_w, xhi = (int64_t)x.hi * L.stride_w;
        const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;  // :95
        const float v1 = to_f32<TIn>(d[ylo + xlo]), v2 = to_f32<TIn>(d[ylo + xhi]);
        const float v3 = to_f32<TIn>(d[yhi + xlo]), v4 = to_f32<TIn>(d[yhi + xhi]);
        acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                                // :208-211 (no contraction)
      }
    }
    out[o] = from_f32<TOut>(fdiv(acc, count));                                       // :216
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged kernel (fixed sampling_ratio > 0): the fast path for the FPN box / mask heads.
//
// Why: with NCHW features a RoI touches, per channel plane, ~20 row pieces of 64-128 B; the per-output gather above
// re-touches every piece once per tap row and is bound by the texture-address path (~90k clk per RoI measured).  Here a
// workgroup owns (RoI, 64 channels) and, per sub-tile of CTs channels,
//   1. copies the RoI's feature WINDOW (rows ytab[0].lo..ytab[last].hi x cols xtab[0].lo..xtab[last].hi) into LDS ONCE,
//      pixel-major [pixel][CTs + 4] -- each global line piece is touched once, 16 lanes per row piece (coalesced); the
//      +4 pad keeps ds_read_b128 aligned and makes the transposing ds_write_b32 at most 2-way conflicted (free);
//   2. computes with lane <-> 4 consecutive channels: every tap is ONE conflict-free ds_read_b128 shared by the 16 lanes
//      of a bin slot, weights come from the per-axis tables; accumulation order is exactly the reference's;
//   3. transposes the results through LDS and stores the [CTs, PH*PW] output slab with coalesced 16-byte stores.
// CTs in {64,32,16,8} is picked per RoI so that window + output slab fit the LDS budget (big windows -> fewer channels
// per pass); a window that does not fit even with CTs = 8 takes the general path for that workgroup.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLdsTableFloats = 1024;      // 4 KB reserved for the axis tables: PH*gh + PW*gw <= 256 entries
constexpr int kLdsPad = 4;

constexpr int kStageMaxK = 16;              // 16-pixel chunks per thread: windows up to 16*4*16 = 1024 pixels
constexpr int kLdsMaxPix = kStageMaxK * (kRoiAlignThreads / 64) * 16;

// LDS window layout: pixel-major, ctp = cts + 4 floats per pixel (the +4 keeps every ds_read_b128 16-byte aligned and
// spreads the transposing ds_write_b32 of 16 consecutive pixels over 8 banks x 2 channels = 2-way, which is free for
// ds_write_b32), plus ONE dummy pixel slot after the window that absorbs the writes of lanes beyond the window (no
// predication in the staging loops).
// ---- window stagers: global -> registers (issue, asynchronous) -> LDS (commit) ------------------------------------------
// All loads of one pass are issued back to back and stay in flight, in registers, while the previous pass is being
// computed; they are written to LDS only after that compute has finished (issue-early / write-late).  One LDS window
// buffer is enough and the global latency of pass i+1 hides under the LDS/VALU work of pass i.

// NCHW: lane -> (pixel in a 16-pixel chunk, channel in a group of 4).  A wave instruction reads 16 px x 4 planes; each
// 16-lane group reads one contiguous row piece (coalesced).  K chunks x G channel groups per thread, K*G = 32 registers
// (so that 4 workgroups = 16 waves fit per CU).  Per element the loops contain exactly one load (uniform base + per-thread
// 32-bit byte offset) and one ds_write_b32 with an immediate offset.
template <typename TIn, int K, int G>
struct StagerNCHW {
  float v[G][K];
  uint32_t voff[K];   // byte offset of (pixel k, channel cl) from the sub-tile base plane
  int32_t lbase[K];   // LDS word index of pixel k + cl
  int cl, nk;         // nk = chunks that contain window pixels (uniform): chunks beyond are neither loaded nor written
  int64_t stride_c;
  __device__ __forceinline__ void init(const dtc_feat_level& L, int y0, int x0, int ww, int wh, int npix, int cts) {
    const int tid = threadIdx.x;
    const int pl = tid & 15, wv = tid >> 6;
    cl = (tid >> 4) & 3;
    stride_c = L.stride_c;
    const int ctp = cts + kLdsPad;
    nk = ceil_div(ceil_div(npix, 16), kRoiAlignThreads / 64);
    const int q64 = 64 / ww, r64 = 64 - q64 * ww;       // uniform
    int pix = wv * 16 + pl;
    int py = pix / ww, px = pix - py * ww;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const bool ok = pix < npix;
      const int ly = ok ? py : wh - 1, lx = ok ? px : ww - 1;     // lanes beyond the window re-read the last pixel ...
      const int lp = ok ? pix : npix;                              // ... and write to the dummy slot
      voff[k] = (uint32_t)(((int64_t)(y0 + ly) * L.stride_h + (int64_t)(x0 + lx) * L.stride_w + (int64_t)cl * L.stride_c) *
                           (int64_t)sizeof(TIn));
      lbase[k] = lp * ctp + cl;
      pix += 64; px += r64; py += q64;
      if (px >= ww) { px -= ww; py++; }
    }
  }
  __device__ __forceinline__ void init_window(const dtc_feat_level& L, int y0, int x0, int ww, int wh, int cts) { init(L, y0, x0, ww, wh, wh * ww, cts); }
  // full sub-tile (all cts channels valid)
  __device__ __forceinline__ void issue(const TIn* cbase, int cts, int nvalid) {
    const char* cb = reinterpret_cast<const char*>(cbase);
    if (nvalid == cts) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (4 * g < cts) {
          const char* gb = cb + (int64_t)(4 * g) * stride_c * (int64_t)sizeof(TIn);   // uniform
#pragma unroll
          for (int k = 0; k < K; k++)
            if (k < nk) v[g][k] = to_f32<TIn>(*reinterpret_cast<const TIn*>(gb + voff[k]));
        }
      }
    } else {  // channel tail (C % 64 != 0): clamp the plane index, results of the clamped planes are never stored
#pragma unroll
      for (int g = 0; g < G; g++) {
        if (4 * g < cts) {
          const int c = min(4 * g + cl, nvalid - 1) - cl;
          const char* gb = cb + (int64_t)c * stride_c * (int64_t)sizeof(TIn);
#pragma unroll
          for (int k = 0; k < K; k++)
            if (k < nk) v[g][k] = to_f32<TIn>(*reinterpret_cast<const TIn*>(gb + voff[k]));
        }
      }
    }
  }
  __device__ __forceinline__ void commit(float* win, int cts) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      if (4 * g < cts) {
#pragma unroll
        for (int k = 0; k < K; k++)
          if (k < nk) win[lbase[k] + 4 * g] = v[g][k];
      }
    }
  }
};

template <typename TIn>
struct StagerNHWC {
  static constexpr int U = 8;
  float4 v[U];
  uint32_t voff[U];
  int32_t loff[U];
  int cq, nu;         // nu = pixel rounds that contain window pixels (uniform)
  bool vec_ok;
  static __device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float4 load4(const __half* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&r.x), b = *reinterpret_cast<const __half2*>(&r.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
  }
  static __device__ __forceinline__ float4 load4(const bf16_t* p) { return bf16x4_to_f32(*reinterpret_cast<const uint2*>(p)); }
  __device__ __forceinline__ void init_window(const dtc_feat_level& L, int y0, int x0, int ww, int wh, int cts) { init(L, y0, x0, ww, wh, wh * ww, cts); }
  __device__ __forceinline__ void init(const dtc_feat_level& L, int y0, int x0, int ww, int wh, int npix, int cts) {
    const int quads = cts >> 2;
    cq = threadIdx.x % quads;
    vec_ok = ((L.stride_h | L.stride_w | L.stride_n) & 3) == 0 && (reinterpret_cast<uintptr_t>(L.data) & 15) == 0;
    const int pslot = threadIdx.x / quads, nslot = kRoiAlignThreads / quads;
    const int ctp = cts + kLdsPad;
    nu = ceil_div(npix, nslot);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int pix = pslot + u * nslot;
      const bool ok = pix < npix;
      const int py = ok ? pix / ww : wh - 1, px = ok ? pix - py * ww : ww - 1;
      voff[u] = (uint32_t)(((int64_t)(y0 + py) * L.stride_h + (int64_t)(x0 + px) * L.stride_w + cq * 4) * (int64_t)sizeof(TIn));
      loff[u] = (ok ? pix : npix) * ctp + (cq << 2);
    }
  }
  __device__ __forceinline__ void issue(const TIn* cbase, int cts, int nvalid) {
    const char* cb = reinterpret_cast<const char*>(cbase);
    const bool full = (nvalid == cts);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (u >= nu) continue;
      const TIn* src = reinterpret_cast<const TIn*>(cb + voff[u]);
      if (full && vec_ok) {
        v[u] = load4(src);       // one 16-byte (fp32) / 8-byte (fp16) load: cts/4 lanes cover a pixel's tile contiguously
      } else if (full) {
        v[u] = make_float4(to_f32<TIn>(src[0]), to_f32<TIn>(src[1]), to_f32<TIn>(src[2]), to_f32<TIn>(src[3]));
      } else {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cq * 4 + 0 < nvalid) v[u].x = to_f32<TIn>(src[0]);
        if (cq * 4 + 1 < nvalid) v[u].y = to_f32<TIn>(src[1]);
        if (cq * 4 + 2 < nvalid) v[u].z = to_f32<TIn>(src[2]);
        if (cq * 4 + 3 < nvalid) v[u].w = to_f32<TIn>(src[3]);
      }
    }
  }
  __device__ __forceinline__ void commit(float* win, int cts) {
#pragma unroll
    for (int u = 0; u < U; u++)
      if (u < nu) *reinterpret_cast<float4*>(win + loff[u]) = v[u];
  }
};

// axis-table entry of the LDS kernel: window-relative, premultiplied
struct LdsAxis { int lo, hi; float l, h; };   // y: (row - y0) * ww * ctp ; x: (col - x0) * ctp   [LDS words]

struct LdsGeom {
  const LdsAxis* ytab; const LdsAxis* xtab;
  float* slab; float* win;
  int cts, bins, gh, gw, pooled_w, nc;
  int bin0, nb;             // bins [bin0, bin0 + nb) are pooled from the staged window (all of them unless a large RoI is split)
  // bin-row slices of a RoI whose window does not fit LDS at once: `rows` bin rows per slice (== pooled_h: one slice)
  int rows, pooled_h, x0, ww, y0, wh, height;
  float sh, bin_h;
  float count, inv_count;   // inv_count != 0 when count is a power of two (x * inv_count == x / count exactly)
};

template <typename TIn, typename TOut, bool SLICED, typename Stager>
__device__ __forceinline__ void run_passes(Stager& st, LdsGeom& G, const dtc_feat_level& L, const TIn* fbase0, int64_t stride_c, TOut* out) {
  const int tid = threadIdx.x;
  const int quads = G.cts >> 2;
  // ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32): number the lanes group by
  // group so that the lanes of one hardware group read as few different pixels as possible (16 quads: ONE pixel = 16
  // consecutive 16-byte units = conflict-free; 8 quads: two pixels instead of four)
  const int ln = tid & 63, lq = (ln >> 2) & 7;
  const int grp = (lq ^ (lq >> 1) ^ (lq >> 2)) & 1;                        // parity of the quad index
  const int vt = (tid & ~63) | (ln & 32) | (grp << 4) | ((ln >> 3) & 3) << 2 | (ln & 3);
  const int cq = vt % quads, slot = vt / quads, nslot = kRoiAlignThreads / quads;
  const float* wq = G.win + cq * 4;
  // SLICED: the RoI is pooled in slices of G.rows bin rows, each with its own window (the kernel's comment); otherwise one
  // iteration over the whole RoI (compile-time: the common path carries no slice state)
#pragma unroll 1
 for (int ph0 = 0; ph0 < (SLICED ? G.pooled_h : 1); ph0 += (SLICED ? G.rows : 1)) {
  int yg = G.y0, whg = G.wh;
  if constexpr (SLICED) {
    const int ph1 = min(G.pooled_h, ph0 + G.rows);
    // window rows of this slice: first sample of bin row ph0 .. last sample of bin row ph1 - 1 (the table values, recomputed)
    yg = make_axis(G.sh, G.bin_h, ph0, 0, G.gh, G.height).lo;
    whg = make_axis(G.sh, G.bin_h, ph1 - 1, G.gh - 1, G.gh, G.height).hi - yg + 1;
    G.bin0 = ph0 * G.pooled_w; G.nb = (ph1 - ph0) * G.pooled_w;
  }
  st.init_window(L, yg, G.x0, G.ww, whg, G.cts);
  st.issue(fbase0, G.cts, min(G.cts, G.nc));
  for (int cs = 0; cs < G.nc; cs += G.cts) {
    const int nvalid = min(G.cts, G.nc - cs);
    st.commit(G.win, G.cts);
    __syncthreads();
    if (cs + G.cts < G.nc)  // prefetch the next channel sub-tile into registers; lands while this one is computed
      st.issue(fbase0 + (int64_t)(cs + G.cts) * stride_c, G.cts, min(G.cts, G.nc - cs - G.cts));
    for (int lb = slot; lb < G.nb; lb += nslot) {
      const int bin = G.bin0 + lb;
      const int ph = bin / G.pooled_w, pw = bin - ph * G.pooled_w;
      // two channels per instruction (v_pk_mul_f32 / v_pk_add_f32: the IEEE results of the scalar forms at twice the rate)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
      // reference order: for iy { for ix { acc += ... } }   (roi_align_cpu_loop.cpp:203-214)
      for (int iy = 0; iy < G.gh; iy++) {
        const LdsAxis y = G.ytab[ph * G.gh + iy];
        for (int ix = 0; ix < G.gw; ix++) {
          const LdsAxis x = G.xtab[pw * G.gw + ix];
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;       // roi_align_cpu_loop.cpp:95
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(wq + y.lo + x.lo);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(wq + y.lo + x.hi);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(wq + y.hi + x.lo);
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(wq + y.hi + x.hi);
          a01 += w1 * v1.lo + w2 * v2.lo + w3 * v3.lo + w4 * v4.lo;                           // :208-211
          a23 += w1 * v1.hi + w2 * v2.hi + w3 * v3.hi + w4 * v4.hi;
        }
      }
      const float a0 = a01.x, a1 = a01.y, a2 = a23.x, a3 = a23.y;
      float* so = G.slab + (cq * 4) * G.nb + lb;
      if (G.inv_count != 0.f) {                                                               // :216
        so[0] = a0 * G.inv_count; so[G.nb] = a1 * G.inv_count; so[2 * G.nb] = a2 * G.inv_count; so[3 * G.nb] = a3 * G.inv_count;
      } else {
        so[0] = fdiv(a0, G.count); so[G.nb] = fdiv(a1, G.count); so[2 * G.nb] = fdiv(a2, G.count); so[3 * G.nb] = fdiv(a3, G.count);
      }
    }
    __syncthreads();
    // coalesced store of the [nvalid][nb] slab
    TOut* og = out + (size_t)cs * G.bins;
    const int n_out = nvalid * G.nb;
    if (G.nb != G.bins) {            // bin-row group of a split RoI: runs of nb consecutive bins per channel
      for (int i = tid; i < n_out; i += kRoiAlignThreads) {
        const int c = i / G.nb, j = i - c * G.nb;
        og[(size_t)c * G.bins + G.bin0 + j] = from_f32<TOut>(G.slab[i]);
      }
    } else if (sizeof(TOut) == 4 && ((reinterpret_cast<uintptr_t>(og) & 15) == 0)) {
      const int n4 = n_out >> 2;
      for (int i = tid; i < n4; i += kRoiAlignThreads)
        store_stream16(reinterpret_cast<float4*>(og) + i, reinterpret_cast<const float4*>(G.slab)[i]);
      for (int i = (n4 << 2) + tid; i < n_out; i += kRoiAlignThreads) og[i] = from_f32<TOut>(G.slab[i]);
    } else {
      for (int i = tid; i < n_out; i += kRoiAlignThreads) og[i] = from_f32<TOut>(G.slab[i]);
    }
    __syncthreads();
  }
 }
}

#ifndef DTC_RA_WAVES
#define DTC_RA_WAVES 2
#endif
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kRoiAlignThreads, DTC_RA_WAVES) void roi_align_fwd_lds(RoiAlignParams p, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);

  const int nct = ceil_div(p.channels, p.ch_block);
  const int wi = xcd_work_item(blockIdx.x, gridDim.x, p.xcd_remap);
  const int ri = wi / nct;
  const int c0 = (wi - ri * nct) * p.ch_block;
  const int nc = min(p.ch_block, p.channels - c0);
  const int bins = p.pooled_h * p.pooled_w;
  const int tid = threadIdx.x;
  const RoiHead hd = load_roi_head(p, ri);
  const int r = hd.r, lvl = hd.lvl, b = hd.b;
  TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)r * p.channels + c0) * bins;
  if (lvl < 0 || lvl >= p.n_levels) {  // padding row (fpn.hip emits level -1): defined output
    for (int o = tid; o < nc * bins; o += kRoiAlignThreads) out[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[lvl];
  const float sw = hd.sw, sh = hd.sh, bin_h = hd.bin_h, bin_w = hd.bin_w, count = hd.count;
  const int gh = hd.gh, gw = hd.gw;
  const int ny = p.pooled_h * gh, nx = p.pooled_w * gw;
  const bool tab_ok = (ny + nx) * 4 <= kLdsTableFloats;
  AxisEntry* xtab = ytab + ny;
  if (tab_ok) {
    for (int t = tid; t < ny + nx; t += kRoiAlignThreads) {
      if (t < ny) ytab[t] = make_axis(sh, bin_h, t / gh, t % gh, gh, L.height);
      else { const int u = t - ny; xtab[u] = make_axis(sw, bin_w, u / gw, u % gw, gw, L.width); }
    }
  }
  __syncthreads();
  const TIn* fbase = reinterpret_cast<const TIn*>(L.data) + (int64_t)b * L.stride_n;
  if (!tab_ok) {
    // huge adaptive grid (RoI much larger than the pooled size): per-output gather, geometry on the fly
    for (int o = tid; o < nc * bins; o += kRoiAlignThreads) {
      const int c = o / bins, bin = o - c * bins;
      const int ph = bin / p.pooled_w, pw = bin - ph * p.pooled_w;
      const TIn* d = fbase + (int64_t)(c0 + c) * L.stride_c;
      float acc = 0.f;
      for (int iy = 0; iy < gh; iy++) {
        const AxisEntry y = make_axis(sh, bin_h, ph, iy, gh, L.height);
        const int64_t ylo = (int64_t)y.lo * L.stride_h, yhi = (int64_t)y.hi * L.stride_h;
        for (int ix = 0; ix < gw; ix++) {
          const AxisEntry x = make_axis(sw, bin_w, pw, ix, gw, L.width);
          const int64_t xlo = (int64_t)x.lo * L.stride_w, xhi = (int64_t)x.hi * L.stride_w;
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
          acc += w1 * to_f32<TIn>(d[ylo + xlo]) + w2 * to_f32<TIn>(d[ylo + xhi]) + w3 * to_f32<TIn>(d[yhi + xlo]) +
                 w4 * to_f32<TIn>(d[yhi + xhi]);
        }
      }
      out[o] = from_f32<TOut>(fdiv(acc, count));
    }
    return;
  }
  const int y0 = ytab[0].lo, y1 = ytab[ny - 1].hi, x1 = xtab[nx - 1].hi;
  const int x0 = xtab[0].lo;
  const int ww = x1 - x0 + 1;
  const int wh = y1 - y0 + 1;
  // the tables take what this RoI needs (28 entries for 7x7 bins x 2 samples), not the 4 KB worst case: ~3.5 KB more window
  const int tabf = ((ny + nx) * 4 + 15) & ~15;
  const int avail = lds_floats - tabf;
  // sub-tile width: largest CTs whose window of wh_ rows (+1 dummy pixel) + output slab fit; one dword per lane, the
  // per-thread share of the window must fit the 32 prefetch registers (npix * cts <= 8192)
  auto pick_cts = [&](int wh_) -> int {
    const int np = wh_ * ww;
    if (p.cts64 && L.stride_c != 1 && nc >= 64 && np <= 128 &&
        (long long)(np + 1) * (64 + kLdsPad) + 64LL * bins <= avail) return 64;
    for (int c = 32; c >= 8; c >>= 1)
      if (np <= kLdsMaxPix && np * c <= 8192 && (long long)(np + 1) * (c + kLdsPad) + (long long)c * bins <= avail) return c;
    return 0;
  };
  int cts = pick_cts(wh);
  // A window that does not fit (adaptive sampling on a large RoI: the C4 configurations pool RoIs of up to 50 x 84 feature
  // pixels) is pooled in GROUPS OF BIN ROWS: bin rows [ph0, ph1) only need the window rows their own samples touch, so the
  // RoI is staged in 2..PH horizontal slices, each through the same LDS pipeline.  (Before: a per-output gather from global
  // memory; 3.5 % of the C4 bench RoIs took 46 % of the launch.)
  int rows = p.pooled_h, wh_max = wh;
  for (int split = 2; split <= p.pooled_h && cts == 0; split++) {
    rows = ceil_div(p.pooled_h, split);
    wh_max = 0;
    for (int ph0 = 0; ph0 < p.pooled_h; ph0 += rows) {
      const int ph1 = min(p.pooled_h, ph0 + rows);
      wh_max = max(wh_max, ytab[ph1 * gh - 1].hi - ytab[ph0 * gh].lo + 1);
    }
    cts = min(pick_cts(wh_max), 8);      // slices run through the largest register pipeline of the layout: 16 chunks x 8 channels
  }
  if (cts == 0) {
    // not even one bin row fits: per-output gather straight from global (same arithmetic)
    for (int o = tid; o < nc * bins; o += kRoiAlignThreads) {
      const int c = o / bins, bin = o - c * bins;
      const int ph = bin / p.pooled_w, pw = bin - ph * p.pooled_w;
      const TIn* d = fbase + (int64_t)(c0 + c) * L.stride_c;
      float acc = 0.f;
      for (int iy = 0; iy < gh; iy++) {
        const AxisEntry y = ytab[ph * gh + iy];
        const int64_t ylo = (int64_t)y.lo * L.stride_h, yhi = (int64_t)y.hi * L.stride_h;
        for (int ix = 0; ix < gw; ix++) {
          const AxisEntry x = xtab[pw * gw + ix];
          const int64_t xlo = (int64_t)x.lo * L.stride_w, xhi = (int64_t)x.hi * L.stride_w;
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
          acc += w1 * to_f32<TIn>(d[ylo + xlo]) + w2 * to_f32<TIn>(d[ylo + xhi]) + w3 * to_f32<TIn>(d[yhi + xlo]) +
                 w4 * to_f32<TIn>(d[yhi + xhi]);
        }
      }
      out[o] = from_f32<TOut>(fdiv(acc, count));
    }
    return;
  }
  // rewrite the tables window-relative and premultiplied (in place: same entry size); a y entry is relative to the first
  // window row of ITS bin-row group
  LdsAxis* yl = reinterpret_cast<LdsAxis*>(ytab);
  LdsAxis* xl = reinterpret_cast<LdsAxis*>(xtab);
  const int ctp = cts + kLdsPad;
  AxisEntry te; te.lo = te.hi = 0; te.l = te.h = 0.f;
  int y0g = 0;
  __syncthreads();                                   // every thread has read the absolute tables above
  if (tid < ny + nx) {
    te = ytab[tid];
    if (tid < ny) y0g = ytab[((tid / gh) / rows) * rows * gh].lo;
  }
  __syncthreads();
  if (tid < ny) { LdsAxis o; o.lo = (te.lo - y0g) * ww * ctp; o.hi = (te.hi - y0g) * ww * ctp; o.l = te.l; o.h = te.h; yl[tid] = o; }
  else if (tid < ny + nx) { LdsAxis o; o.lo = (te.lo - x0) * ctp; o.hi = (te.hi - x0) * ctp; o.l = te.l; o.h = te.h; yl[tid] = o; }
  LdsGeom G;
  G.ytab = yl; G.xtab = xl;
  G.slab = lds + tabf;                            // [cts][nb] output staging (16-float aligned)
  G.win = G.slab + cts * bins;                    // [npix + 1][cts + 4]  (cts*bins is a multiple of 4 -> 16 B aligned)
  G.cts = cts; G.bins = bins; G.gh = gh; G.gw = gw; G.pooled_w = p.pooled_w; G.nc = nc; G.count = count;
  G.inv_count = hd.inv_count;
  const TIn* cbase = fbase + (int64_t)c0 * L.stride_c;
  const int npix_max = wh_max * ww;               // the register pipeline shape is chosen for the largest slice
  G.rows = rows; G.pooled_h = p.pooled_h; G.x0 = x0; G.ww = ww; G.y0 = y0; G.wh = wh; G.height = L.height;
  G.sh = sh; G.bin_h = bin_h; G.bin0 = 0; G.nb = bins;
  if (rows != p.pooled_h) {        // sliced: the largest register pipeline of the layout, one instantiation
    if (L.stride_c == 1) {
      StagerNHWC<TIn> st;
      run_passes<TIn, TOut, true>(st, G, L, cbase, L.stride_c, out);
    } else {
      StagerNCHW<TIn, 16, 2> st;
      run_passes<TIn, TOut, true>(st, G, L, cbase, L.stride_c, out);
    }
    return;
  }
  if (L.stride_c == 1) {
    StagerNHWC<TIn> st;
    run_passes<TIn, TOut, false>(st, G, L, cbase, L.stride_c, out);
  } else {
    const int nk = ceil_div(ceil_div(npix_max, 16), kRoiAlignThreads / 64);
    if (cts == 64) { StagerNCHW<TIn, 2, 16> st; run_passes<TIn, TOut, false>(st, G, L, cbase, L.stride_c, out); }
    else if (nk <= 4) { StagerNCHW<TIn, 4, 8> st; run_passes<TIn, TOut, false>(st, G, L, cbase, L.stride_c, out); }
    else if (nk <= 8) { StagerNCHW<TIn, 8, 4> st; run_passes<TIn, TOut, false>(st, G, L, cbase, L.stride_c, out); }
    else { StagerNCHW<TIn, 16, 2> st; run_passes<TIn, TOut, false>(st, G, L, cbase, L.stride_c, out); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Channels-last (NHWC) direct kernel.
//
// With stride_c == 1 a tap's channels are contiguous: 16 lanes x 16 bytes read one pixel's 64-channel tile (two full
// 128-byte lines), so the taps can be gathered STRAIGHT from L1/L2 with perfectly coalesced loads -- no LDS staging of the
// window, no transposing LDS writes, no staging barriers, ~60 VGPRs, i.e. 5+ waves per SIMD to hide latency, and the window
// rows a workgroup revisits for neighbouring samples are L1 hits (a pixel is used ~3 times).  The texture path moves
// 784 taps x 1 KB per RoI (7x7, sr 2) -- about a third of the per-RoI CU time of the LDS-staged NCHW kernel.
// workgroup = (RoI, 64 channels); lane -> (channel quad, bin slot); results are transposed through a [64][bins] LDS slab and
// stored as one contiguous run, exactly like the LDS kernel.  Same arithmetic, same order, bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
template <typename TIn> struct Vec4Load;
template <> struct Vec4Load<float> {
  static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
};
template <> struct Vec4Load<__half> {
  static __device__ __forceinline__ float4 ld(const __half* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&r.x), b = *reinterpret_cast<const __half2*>(&r.y);
    return make_float4(__low2float(a), __high2float(a), __low2float(b), __high2float(b));
  }
};

template <> struct Vec4Load<bf16_t> {
  static __device__ __forceinline__ float4 ld(const bf16_t* p) { return bf16x4_to_f32(*reinterpret_cast<const uint2*>(p)); }
};

// low / high 16-bit half of a dword as float32 (fp16 or bf16 element pairs of a 16-byte load)
template <typename T> __device__ __forceinline__ float lo16_to_f32(uint32_t w);
template <typename T> __device__ __forceinline__ float hi16_to_f32(uint32_t w);
template <> __device__ __forceinline__ float lo16_to_f32<__half>(uint32_t w) { return __low2float(*reinterpret_cast<const __half2*>(&w)); }
template <> __device__ __forceinline__ float hi16_to_f32<__half>(uint32_t w) { return __high2float(*reinterpret_cast<const __half2*>(&w)); }
template <> __device__ __forceinline__ float lo16_to_f32<bf16_t>(uint32_t w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float hi16_to_f32<bf16_t>(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ float lo16_to_f32<float>(uint32_t w) { return 0.f; }      // never instantiated for float maps
template <> __device__ __forceinline__ float hi16_to_f32<float>(uint32_t w) { return 0.f; }

// [n_out] float32 slab -> contiguous output of type TOut: 16-byte stores where the alignment allows (4 floats, or 8 16-bit values
// rounded exactly as from_f32 does element by element); a 2-byte store per element cost 12 x the time per byte
template <typename TOut, int THREADS>
__device__ __forceinline__ void store_slab_vec(TOut* out, const float* slab, int n_out, int tid) {
  if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    if constexpr (sizeof(TOut) == 4) {
      const int n4 = n_out >> 2;
      for (int i = tid; i < n4; i += THREADS) store_stream16(reinterpret_cast<float4*>(out) + i, reinterpret_cast<const float4*>(slab)[i]);
      for (int i = (n4 << 2) + tid; i < n_out; i += THREADS) out[i] = from_f32<TOut>(slab[i]);
    } else {
      const int n8 = n_out >> 3;
      for (int i = tid; i < n8; i += THREADS) {
        const float4 a = reinterpret_cast<const float4*>(slab)[2 * i], b = reinterpret_cast<const float4*>(slab)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const TOut lo = from_f32<TOut>(v[2 * k]), hi = from_f32<TOut>(v[2 * k + 1]);
          w[k] = (uint32_t)*reinterpret_cast<const uint16_t*>(&lo) | ((uint32_t)*reinterpret_cast<const uint16_t*>(&hi) << 16);
        }
        store_stream16(reinterpret_cast<uint4*>(out) + i, w[0], w[1], w[2], w[3]);
      }
      for (int i = (n8 << 3) + tid; i < n_out; i += THREADS) out[i] = from_f32<TOut>(slab[i]);
    }
  } else {
    for (int i = tid; i < n_out; i += THREADS) out[i] = from_f32<TOut>(slab[i]);
  }
}

// CB: channels per workgroup (64; 32 for 16-bit maps with more than 64 bins, whose [CB][bins] float32 slab would otherwise take 50 KB)
template <typename TIn, typename TOut, int CB>
__global__ __launch_bounds__(kRoiAlignThreads) void roi_align_fwd_nhwc(RoiAlignParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);
  float* slab = reinterpret_cast<float*>(smem) + kLdsTableFloats;
  const int nct = ceil_div(p.channels, CB);
  const int wi = xcd_work_item(blockIdx.x, gridDim.x, p.xcd_remap);
  const int ri = wi / nct;
  const int c0 = (wi - ri * nct) * CB;
  const int nc = min(CB, p.channels - c0);
  const int bins = p.pooled_h * p.pooled_w;
  const int tid = threadIdx.x;
  const RoiHead hd = load_roi_head(p, ri);
  const int r = hd.r, lvl = hd.lvl, b = hd.b;
  TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)r * p.channels + c0) * bins;
  if (lvl < 0 || lvl >= p.n_levels) {
    for (int o = tid; o < nc * bins; o += kRoiAlignThreads) out[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[lvl];
  const float sw = hd.sw, sh = hd.sh, bin_h = hd.bin_h, bin_w = hd.bin_w, count = hd.count, inv_count = hd.inv_count;
  const int gh = hd.gh, gw = hd.gw;
  const int ny = p.pooled_h * gh, nx = p.pooled_w * gw;
  const bool tab_ok = (ny + nx) * 4 <= kLdsTableFloats;
  AxisEntry* xtab = ytab + ny;
  if (tab_ok) {
    // entries premultiplied by the element strides: lo/hi become element offsets inside the image
    for (int t = tid; t < ny + nx; t += kRoiAlignThreads) {
      AxisEntry e;
      if (t < ny) { e = make_axis(sh, bin_h, t / gh, t % gh, gh, L.height); e.lo *= (int)L.stride_h; e.hi *= (int)L.stride_h; }
      else { const int u = t - ny; e = make_axis(sw, bin_w, u / gw, u % gw, gw, L.width); e.lo *= (int)L.stride_w; e.hi *= (int)L.stride_w; }
      ytab[t] = e;
    }
  }
  __syncthreads();
  if constexpr (sizeof(TIn) == 2) {
    // 16-bit maps, 2 x 2 sampling grid, a full 64-channel block: lane <-> (8 channels = one 16-byte load per tap, bin slot of 32):
    // a 7 x 7 RoI takes TWO rounds of 16 loads in flight per lane where the 4-channel mapping below takes four -- the kernel is
    // bound by those dependent L1 / L2 round trips, not by bytes (0.32 -> 0.2x ms per 8000-RoI launch).  Same arithmetic, same order.
    const bool wide = nc == CB && tab_ok && gh == 2 && gw == 2 && ((L.stride_h | L.stride_w | L.stride_n) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(L.data) & 15) == 0 && inv_count != 0.f &&
                      (int64_t)L.stride_h * L.height + (int64_t)L.stride_w * L.width < (1ll << 30);     // 32-bit byte offsets
    if (wide) {
      constexpr int NQ8 = CB / 8;
      const int q8 = tid % NQ8, slot8 = tid / NQ8;
      // one scalar base for the workgroup, 32-bit byte offsets per lane (global_load ... saddr): a tap costs one v_add
      const int bu = __builtin_amdgcn_readfirstlane(b);
      const char* ubase = reinterpret_cast<const char*>(reinterpret_cast<const TIn*>(L.data) + (int64_t)bu * L.stride_n + c0);
      const uint32_t lane_off = 16u * (uint32_t)q8;
      auto ld = [&](uint32_t yo, uint32_t xo) { return *reinterpret_cast<const uint4*>(ubase + (yo + xo)); };
      // bin -> (ph, pw) without a division per round: the lane's bins are slot8, slot8 + 32, ...
      constexpr int kStep = kRoiAlignThreads / NQ8;
      const int step_h = kStep / p.pooled_w, step_w = kStep - step_h * p.pooled_w;
      int ph = slot8 / p.pooled_w, pw = slot8 - ph * p.pooled_w;
      for (int bin = slot8; bin < bins; bin += kStep) {
        const AxisEntry y0e = ytab[ph * 2], y1e = ytab[ph * 2 + 1], x0e = xtab[pw * 2], x1e = xtab[pw * 2 + 1];
        // byte offsets (tables hold element offsets inside the image; the map is < 2^31 bytes per image -- checked by `wide`)
        const uint32_t ya = 2u * (uint32_t)y0e.lo + lane_off, yb = 2u * (uint32_t)y0e.hi + lane_off;
        const uint32_t yc = 2u * (uint32_t)y1e.lo + lane_off, yd = 2u * (uint32_t)y1e.hi + lane_off;
        const uint32_t xa = 2u * (uint32_t)x0e.lo, xb = 2u * (uint32_t)x0e.hi, xc = 2u * (uint32_t)x1e.lo, xd = 2u * (uint32_t)x1e.hi;
        uint4 t[16];        // all 16 taps in flight before the first one is consumed
        t[0] = ld(ya, xa); t[1] = ld(ya, xb); t[2] = ld(yb, xa); t[3] = ld(yb, xb);
        t[4] = ld(ya, xc); t[5] = ld(ya, xd); t[6] = ld(yb, xc); t[7] = ld(yb, xd);
        t[8] = ld(yc, xa); t[9] = ld(yc, xb); t[10] = ld(yd, xa); t[11] = ld(yd, xb);
        t[12] = ld(yc, xc); t[13] = ld(yc, xd); t[14] = ld(yd, xc); t[15] = ld(yd, xd);
        // two channels per instruction: {w * lo16, w * hi16} in one or two instructions (mul_pair16), sums as v_pk_add_f32 --
        // the IEEE results of the scalar forms, the reference's order (roi_align_cpu_loop.cpp:95-98)
        f32x2 a[4];
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = f32x2{0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {   // (iy, ix) = (0,0) (0,1) (1,0) (1,1): the reference's accumulation order
          const AxisEntry& y = (sidx >> 1) ? y1e : y0e;
          const AxisEntry& x = (sidx & 1) ? x1e : x0e;
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
          const uint32_t* u1 = reinterpret_cast<const uint32_t*>(&t[sidx * 4]);
          const uint32_t* u2 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 1]);
          const uint32_t* u3 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 2]);
          const uint32_t* u4 = reinterpret_cast<const uint32_t*>(&t[sidx * 4 + 3]);
#pragma unroll
          for (int k = 0; k < 4; k++) {           // dword k holds channels 2k (low half) and 2k + 1 (high half)
            f32x2 s = mul_pair16<TIn>(u1[k], w1) + mul_pair16<TIn>(u2[k], w2);
            s = s + mul_pair16<TIn>(u3[k], w3);
            s = s + mul_pair16<TIn>(u4[k], w4);
            a[k] = a[k] + s;
          }
        }
        float* so = slab + (8 * q8) * bins + bin;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const f32x2 o = a[k] * inv_count;
          so[(2 * k) * bins] = o.x; so[(2 * k + 1) * bins] = o.y;
        }
        ph += step_h; pw += step_w;
        if (pw >= p.pooled_w) { pw -= p.pooled_w; ph++; }
      }
      __syncthreads();
      store_slab_vec<TOut, kRoiAlignThreads>(out, slab, nc * bins, tid);
      return;
    }
  }
  constexpr int NQ4 = CB / 4;
  const int q = tid % NQ4, slot = tid / NQ4;
  const TIn* base = reinterpret_cast<const TIn*>(L.data) + (int64_t)b * L.stride_n + c0 + 4 * q;
  const bool full = (4 * q + 3) < nc;                       // this lane's 4 channels all exist
  const bool vec = full && ((L.stride_h | L.stride_w | L.stride_n) & 3) == 0 && ((c0 & 3) == 0) &&
                   (reinterpret_cast<uintptr_t>(L.data) & 15) == 0;
  if (4 * q < nc) {
    for (int bin = slot; bin < bins; bin += kRoiAlignThreads / NQ4) {
      const int ph = bin / p.pooled_w, pw = bin - ph * p.pooled_w;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (vec && tab_ok && gh == 2 && gw == 2) {
        // 2x2 sampling grid: all 16 taps (four 16-byte loads per sample) are issued before the first one is consumed, so a
        // bin costs ONE L2 round trip instead of four dependent ones
        const AxisEntry x0e = xtab[pw * 2], x1e = xtab[pw * 2 + 1];
        const AxisEntry y0e = ytab[ph * 2], y1e = ytab[ph * 2 + 1];
        float4 t[16];
        t[0] = Vec4Load<TIn>::ld(base + y0e.lo + x0e.lo); t[1] = Vec4Load<TIn>::ld(base + y0e.lo + x0e.hi);
        t[2] = Vec4Load<TIn>::ld(base + y0e.hi + x0e.lo); t[3] = Vec4Load<TIn>::ld(base + y0e.hi + x0e.hi);
        t[4] = Vec4Load<TIn>::ld(base + y0e.lo + x1e.lo); t[5] = Vec4Load<TIn>::ld(base + y0e.lo + x1e.hi);
        t[6] = Vec4Load<TIn>::ld(base + y0e.hi + x1e.lo); t[7] = Vec4Load<TIn>::ld(base + y0e.hi + x1e.hi);
        t[8] = Vec4Load<TIn>::ld(base + y1e.lo + x0e.lo); t[9] = Vec4Load<TIn>::ld(base + y1e.lo + x0e.hi);
        t[10] = Vec4Load<TIn>::ld(base + y1e.hi + x0e.lo); t[11] = Vec4Load<TIn>::ld(base + y1e.hi + x0e.hi);
        t[12] = Vec4Load<TIn>::ld(base + y1e.lo + x1e.lo); t[13] = Vec4Load<TIn>::ld(base + y1e.lo + x1e.hi);
        t[14] = Vec4Load<TIn>::ld(base + y1e.hi + x1e.lo); t[15] = Vec4Load<TIn>::ld(base + y1e.hi + x1e.hi);
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {   // (iy, ix) = (0,0) (0,1) (1,0) (1,1): the reference's accumulation order
          const AxisEntry& y = (sidx < 2) ? y0e : y1e;
          const AxisEntry& x = (sidx & 1) ? x1e : x0e;
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
          const float4 v1 = t[sidx * 4], v2 = t[sidx * 4 + 1], v3 = t[sidx * 4 + 2], v4 = t[sidx * 4 + 3];
          a0 += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          a1 += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          a2 += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          a3 += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      } else {
      for (int iy = 0; iy < gh; iy++) {
        const AxisEntry y = tab_ok ? ytab[ph * gh + iy] : [&] { AxisEntry e = make_axis(sh, bin_h, ph, iy, gh, L.height); e.lo *= (int)L.stride_h; e.hi *= (int)L.stride_h; return e; }();
        for (int ix = 0; ix < gw; ix++) {
          const AxisEntry x = tab_ok ? xtab[pw * gw + ix] : [&] { AxisEntry e = make_axis(sw, bin_w, pw, ix, gw, L.width); e.lo *= (int)L.stride_w; e.hi *= (int)L.stride_w; return e; }();
          const float w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;      // roi_align_cpu_loop.cpp:95
          float4 v1, v2, v3, v4;
          if (vec) {
            v1 = Vec4Load<TIn>::ld(base + y.lo + x.lo); v2 = Vec4Load<TIn>::ld(base + y.lo + x.hi);
            v3 = Vec4Load<TIn>::ld(base + y.hi + x.lo); v4 = Vec4Load<TIn>::ld(base + y.hi + x.hi);
          } else {
            const TIn* t1 = base + y.lo + x.lo; const TIn* t2 = base + y.lo + x.hi;
            const TIn* t3 = base + y.hi + x.lo; const TIn* t4 = base + y.hi + x.hi;
            const int nv = nc - 4 * q;   // >= 1
            v1 = make_float4(to_f32<TIn>(t1[0]), nv > 1 ? to_f32<TIn>(t1[1]) : 0.f, nv > 2 ? to_f32<TIn>(t1[2]) : 0.f, nv > 3 ? to_f32<TIn>(t1[3]) : 0.f);
            v2 = make_float4(to_f32<TIn>(t2[0]), nv > 1 ? to_f32<TIn>(t2[1]) : 0.f, nv > 2 ? to_f32<TIn>(t2[2]) : 0.f, nv > 3 ? to_f32<TIn>(t2[3]) : 0.f);
            v3 = make_float4(to_f32<TIn>(t3[0]), nv > 1 ? to_f32<TIn>(t3[1]) : 0.f, nv > 2 ? to_f32<TIn>(t3[2]) : 0.f, nv > 3 ? to_f32<TIn>(t3[3]) : 0.f);
            v4 = make_float4(to_f32<TIn>(t4[0]), nv > 1 ? to_f32<TIn>(t4[1]) : 0.f, nv > 2 ? to_f32<TIn>(t4[2]) : 0.f, nv > 3 ? to_f32<TIn>(t4[3]) : 0.f);
          }
          a0 += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;                              // :208-211
          a1 += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          a2 += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          a3 += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      }
      float* so = slab + (4 * q) * bins + bin;
      if (inv_count != 0.f) { so[0] = a0 * inv_count; so[bins] = a1 * inv_count; so[2 * bins] = a2 * inv_count; so[3 * bins] = a3 * inv_count; }
      else { so[0] = fdiv(a0, count); so[bins] = fdiv(a1, count); so[2 * bins] = fdiv(a2, count); so[3 * bins] = fdiv(a3, count); }   // :216
    }
  }
  __syncthreads();
  store_slab_vec<TOut, kRoiAlignThreads>(out, slab, nc * bins, tid);
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// Development / A-B knobs, resolved ONCE per process (thread-safe static initialisation) -- not per dispatch:
//   DTC_ROIALIGN_TILE=0          use the RoI-stationary kernel of this file instead of the cluster-stationary one (roi_align_tile.hip)
//   DTC_ROIALIGN_MAP=0           single-level inputs (C4) through the RoI-stationary kernel instead of the map-stationary one (roi_align_map.hip)
//   DTC_ROIALIGN_GENERAL=1       force the per-output gather kernel (the plain statement of the arithmetic)
//   DTC_ROIALIGN_NO_NHWC_DIRECT  channels_last features through the LDS-staged kernel
//   DTC_RA_NO_XCD  DTC_RA_NO_CTS64
struct RoiAlignConfig {
  bool tile = true, map = true, general = false, nhwc_direct = true, xcd = true, cts64 = true;
  int lds_bytes = 52 * 1024;
};
static const RoiAlignConfig& roi_align_config() {
  static const RoiAlignConfig cfg = [] {
    RoiAlignConfig c;
    if (const char* e = getenv("DTC_ROIALIGN_TILE")) c.tile = e[0] != '0';
    if (const char* e = getenv("DTC_ROIALIGN_MAP")) c.map = e[0] != '0';
    c.general = getenv("DTC_ROIALIGN_GENERAL") != nullptr;
    c.nhwc_direct = getenv("DTC_ROIALIGN_NO_NHWC_DIRECT") == nullptr;
    c.xcd = getenv("DTC_RA_NO_XCD") == nullptr;
    c.cts64 = getenv("DTC_RA_NO_CTS64") == nullptr;      // 64-channel sub-tiles for windows <= 128 px: +4 % on the bench workload
    return c;
  }();
  return cfg;
}

template <typename TIn, typename TOut, int CB>
static int launch_nhwc_cb(const RoiAlignParams& p, hipStream_t stream) {
  const size_t smem = (size_t)kLdsTableFloats * 4 + (size_t)CB * p.pooled_h * p.pooled_w * 4 + 16;
  static std::once_flag once;
  static hipError_t rc = hipSuccess;
  std::call_once(once, [] { rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_nhwc<TIn, TOut, CB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  if (rc != hipSuccess) return DTC_ELAUNCH;
  const int nct = ceil_div(p.channels, CB);
  hipLaunchKernelGGL((roi_align_fwd_nhwc<TIn, TOut, CB>), dim3((unsigned)p.n_rois * nct), dim3(kRoiAlignThreads), smem, stream, p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

template <typename TIn, typename TOut>
static int launch_nhwc(const RoiAlignParams& p, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  // 16-bit maps with more than 64 bins (the 14 x 14 mask head): 32-channel blocks, so that the [CB][bins] float32 slab stays at 25 KB
  if (sizeof(TIn) == 2 && p.pooled_h * p.pooled_w > 64 && p.channels % 32 == 0) return launch_nhwc_cb<TIn, TOut, 32>(p, stream);
  // (float32 maps with 32-channel blocks -- two rounds of taps instead of four, twice the workgroups -- measured slower: 0.496 ms
  //  against 0.452 per box-head launch; float32 channels_last maps take roi_align_fwd_nhwc_lds first anyway: 0.359 ms)
  return launch_nhwc_cb<TIn, TOut, 64>(p, stream);
}

template <typename TIn, typename TOut>
static int launch_lds(const RoiAlignParams& p, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  static std::once_flag once;
  static hipError_t rc = hipSuccess;
  std::call_once(once, [] { rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_lds<TIn, TOut>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  if (rc != hipSuccess) return DTC_ELAUNCH;
  const int lds_b = roi_align_config().lds_bytes;
  const int nct = ceil_div(p.channels, p.ch_block);
  hipLaunchKernelGGL((roi_align_fwd_lds<TIn, TOut>), dim3((unsigned)p.n_rois * nct), dim3(kRoiAlignThreads), lds_b, stream, p,
                     lds_b / 4);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

template <typename TIn, typename TOut>
static int launch_general(const RoiAlignParams& p, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  // worst-case table bytes: sampling_ratio>0 -> exact; adaptive -> bounded by kMaxTableEntries
  int entries = p.sampling_ratio > 0 ? (p.pooled_h + p.pooled_w) * p.sampling_ratio : kMaxTableEntries;
  if (entries > kMaxTableEntries) entries = kMaxTableEntries;
  size_t smem = (size_t)entries * sizeof(AxisEntry);
  dim3 grid(p.n_rois, ceil_div(p.channels, p.ch_tile));
  hipLaunchKernelGGL((roi_align_fwd_general<TIn, TOut>), grid, dim3(kRoiAlignThreads), smem, stream, p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

enum { kKernNhwc, kKernLds, kKernGeneral };
template <typename TIn, typename TOut>
static int launch_kind(int kind, const RoiAlignParams& p, hipStream_t s) {
  return kind == kKernNhwc ? launch_nhwc<TIn, TOut>(p, s) : kind == kKernLds ? launch_lds<TIn, TOut>(p, s) : launch_general<TIn, TOut>(p, s);
}
// (in, out) dtype pairs: fp32 accumulate always; f16 and bf16 do not mix
static int launch_typed(int kind, const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t s) {
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return launch_kind<float, float>(kind, p, s);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return launch_kind<__half, float>(kind, p, s);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return launch_kind<__half, __half>(kind, p, s);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F16) return launch_kind<float, __half>(kind, p, s);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_F32) return launch_kind<bf16_t, float>(kind, p, s);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_BF16) return launch_kind<bf16_t, bf16_t>(kind, p, s);
  if (in_dtype == DTC_F32 && out_dtype == DTC_BF16) return launch_kind<float, bf16_t>(kind, p, s);
  return DTC_EUNSUPPORTED;
}

}  // namespace dtc

static int roi_align_dispatch(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype, const float* rois,
                             int roi_cols, const int32_t* roi_levels, const int32_t* roi_order, const float* roi_desc,
                             int n_rois, int pooled_h, int pooled_w, int sampling_ratio, void* out, int out_dtype,
                             dtc_stream_t stream, void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (!levels || n_levels < 1 || n_levels > DTC_MAX_LEVELS || channels < 1 || n_rois < 0 || pooled_h < 1 ||
      pooled_w < 1 || (roi_cols != 4 && roi_cols != 5) || (n_rois > 0 && ((!rois && !roi_desc) || !out)))
    return DTC_EINVAL;
  const dtc::RoiAlignConfig& cfg = dtc::roi_align_config();
  dtc::RoiAlignParams p;
  for (int i = 0; i < n_levels; i++) {
    if (!levels[i].data || levels[i].height < 1 || levels[i].width < 1) return DTC_EINVAL;
    p.lv[i] = levels[i];
  }
  p.rois = rois; p.roi_levels = roi_levels; p.roi_order = roi_order; p.roi_desc = roi_desc; p.out = out;
  p.n_levels = n_levels; p.channels = channels; p.roi_cols = roi_cols; p.n_rois = n_rois;
  p.pooled_h = pooled_h; p.pooled_w = pooled_w; p.sampling_ratio = sampling_ratio;
  p.ch_tile = channels < 64 ? channels : 64;
  // channels per workgroup of the RoI-stationary LDS kernel: 128 (the per-RoI setup is paid once per 128 channels and the
  // register prefetch pipeline runs across more passes) unless that leaves too few workgroups to fill 256 CUs x 3.
  // Measured on MI355X, 8000 RoIs x 256 ch: 64 -> 0.710 ms, 128 -> 0.704 ms, 256 -> 0.760 ms.
  p.ch_block = channels > 64 ? 128 : 64;
  if ((long long)n_rois * ((channels + p.ch_block - 1) / p.ch_block) < 3072) p.ch_block = 64;
  p.xcd_remap = cfg.xcd; p.cts64 = cfg.cts64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // LDS-staged kernel for every pooled size whose output slab fits; adaptive sampling (sampling_ratio <= 0) included
  // (tables sized per RoI, oversize grids / windows fall back per workgroup)
  const bool lds_ok = (sampling_ratio <= 0 || (pooled_h + pooled_w) * sampling_ratio * 4 <= dtc::kLdsTableFloats) &&
                      (long long)pooled_h * pooled_w * 8 <= 4096 && !cfg.general;
  bool all_nhwc = channels > 1;
  for (int i = 0; i < n_levels; i++) all_nhwc = all_nhwc && levels[i].stride_c == 1;
  // channels_last: direct gather pays when a window pixel is re-used only a few times (7x7 bins x 2x2 samples over a
  // ~300-pixel window: 2.5 taps per pixel, measured 0.50 vs 0.65 ms); with 14x14 bins (10 taps per pixel) staging the window
  // in LDS wins (0.21 vs 0.37 ms), and the LDS kernel stages channels_last windows with 16-byte loads too.
  const bool few_taps = sampling_ratio > 0 && (long long)pooled_h * pooled_w * sampling_ratio * sampling_ratio <= 256;
  // 16-bit channels_last maps, and float32 ones with <= 64 bins (DTC_RA_NHWC_DIRECT32=0: the LDS-DMA kernels instead), 2 x 2 samples: 16-byte lanes, the bins of several RoIs flattened over a workgroup (roi_align_nhwc16.hip)
  if (all_nhwc && !cfg.general && cfg.nhwc_direct && dtc::roi_align_nhwc16_supported(p, in_dtype, out_dtype))
    return dtc::launch_roi_align_nhwc16(p, in_dtype, out_dtype, s);
  // channels_last, sampling_ratio 2, <= 64 bins: window staged with LDS-DMA, conflict-free tap reads (roi_align_nhwc.hip)
  if (all_nhwc && !cfg.general && cfg.nhwc_direct && dtc::roi_align_nhwc_lds_supported(p, in_dtype, out_dtype))
    return dtc::launch_roi_align_nhwc_lds(p, in_dtype, out_dtype, s);
  // 16-bit maps the grouped kernel's alignment / size conditions exclude: one RoI per workgroup
  const bool wide16 = in_dtype != DTC_F32 && sampling_ratio == 2 && channels % 32 == 0;
  if (lds_ok && all_nhwc && (few_taps || wide16) && cfg.nhwc_direct) return dtc::launch_typed(dtc::kKernNhwc, p, in_dtype, out_dtype, s);
  // one level whose whole map fits LDS (the C4 heads), adaptive sampling: the map-stationary kernel (roi_align_map.hip)
  if (cfg.map && !cfg.general && sampling_ratio != 2 && dtc::roi_align_map_supported(p, in_dtype, out_dtype))
    return dtc::launch_roi_align_map_ws(p, in_dtype, out_dtype, workspace, workspace_bytes, s);
  // NCHW (the reference's layout), sampling_ratio 2: the cluster-stationary kernel (roi_align_tile.hip)
  if (cfg.tile && !all_nhwc && !cfg.general && dtc::roi_align_tile_supported(p, in_dtype, out_dtype))
    return dtc::launch_roi_align_tile(p, in_dtype, out_dtype, s);
  return dtc::launch_typed(lds_ok ? dtc::kKernLds : dtc::kKernGeneral, p, in_dtype, out_dtype, s);
}

DTC_API int dtc_roi_align_forward_ordered(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                             const float* rois, int roi_cols, const int32_t* roi_levels,
                                             const int32_t* roi_order, int n_rois, int pooled_h, int pooled_w,
                                             int sampling_ratio, void* out, int out_dtype, dtc_stream_t stream) {
  return roi_align_dispatch(levels, n_levels, channels, in_dtype, rois, roi_cols, roi_levels, roi_order, nullptr, n_rois,
                            pooled_h, pooled_w, sampling_ratio, out, out_dtype, stream);
}

DTC_API int dtc_roi_align_forward_packed(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                            const float* roi_desc, int n_rois, int pooled_h, int pooled_w,
                                            int sampling_ratio, void* out, int out_dtype, dtc_stream_t stream) {
  return roi_align_dispatch(levels, n_levels, channels, in_dtype, nullptr, 5, nullptr, nullptr, roi_desc, n_rois, pooled_h,
                            pooled_w, sampling_ratio, out, out_dtype, stream);
}

DTC_API size_t dtc_roi_align_workspace_bytes(int n_rois) { return dtc::roi_align_map_workspace_bytes(n_rois); }

DTC_API void dtc_roi_align_set_exact(int exact) { dtc::roi_align_set_exact(exact); }
DTC_API int dtc_roi_align_get_exact(void) { return dtc::roi_align_get_exact(); }

DTC_API int dtc_roi_align_forward_packed_ws(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                            const float* roi_desc, int n_rois, int pooled_h, int pooled_w, int sampling_ratio,
                                            void* out, int out_dtype, void* workspace, size_t workspace_bytes, dtc_stream_t stream) {
  return roi_align_dispatch(levels, n_levels, channels, in_dtype, nullptr, 5, nullptr, nullptr, roi_desc, n_rois, pooled_h,
                            pooled_w, sampling_ratio, out, out_dtype, stream, workspace, workspace_bytes);
}

DTC_API int dtc_roi_align_forward(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                     const float* rois, int roi_cols, const int32_t* roi_levels, int n_rois,
                                     int pooled_h, int pooled_w, int sampling_ratio, void* out, int out_dtype,
                                     dtc_stream_t stream) {
  return dtc_roi_align_forward_ordered(levels, n_levels, channels, in_dtype, rois, roi_cols, roi_levels, nullptr, n_rois,
                                       pooled_h, pooled_w, sampling_ratio, out, out_dtype, stream);
}

DTC_API int launch_roi_align_forward_hip(const int outputElements, const float* bottom_data,
                                            const float* bottom_rois, const float spatial_scale, const int channels,
                                            const int height, const int width, const int pooled_height,
                                            const int pooled_width, const int sampling_ratio, float* top_data,
                                            dtc_stream_t stream) {
  if (channels < 1 || pooled_height < 1 || pooled_width < 1) return 0;
  dtc_feat_level L;
  L.data = bottom_data; L.height = height; L.width = width; L.spatial_scale = spatial_scale; L._pad = 0;
  L.stride_n = (int64_t)channels * height * width; L.stride_c = (int64_t)height * width; L.stride_h = width; L.stride_w = 1;
  const int n_rois = outputElements / channels / pooled_width / pooled_height;  // roi_align_cpu_loop.cpp:131
  int rc = dtc_roi_align_forward(&L, 1, channels, DTC_F32, bottom_rois, 5, nullptr, n_rois, pooled_height, pooled_width,
                                 sampling_ratio, top_data, DTC_F32, stream);
  return rc == DTC_OK ? 1 : 0;
}
