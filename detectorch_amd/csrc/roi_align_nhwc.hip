// A1  RoIAlign forward for gfx950 -- channels_last (NHWC) feature maps, sampling_ratio 2, <= 64 bins (the FPN box head when the
// backbone emits channels_last tensors: MIOpen's preferred layout for 16-bit convolutions on MI355X).
//
// Replaces roi_align_forward_kernel (lib/cppcuda/roi_align_forward_cuda.cu:82-159); bit-compatible with the CPU path
// roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:118-219): same float32 operations, same order.
//
// Why a kernel of its own.  With NCHW maps the two things the cluster-stationary kernel (roi_align_tile.hip) cannot get rid of are
// the transposing LDS commit (316 staged pixels per (RoI, channel), four ds_write_b32 per 16-byte piece) and the bank-conflicted
// tap gather (lane <-> (RoI, bin): 9.5 LDS cycles per ds_read_b128 where 4 are conflict-free, tools/r03/lds_taps.py).  With the
// channels innermost both disappear:
//   * a pixel's 256-byte chunk (64 float32 / 128 16-bit channels) is contiguous in memory AND is what the LDS image wants
//     ([pixel][channels]): the window is staged with LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane land lane-linear, four
//     pixels per wave-instruction) -- no staging registers, no commit instructions, every fetched byte is used: 5.3 line fills
//     per (RoI, channel) where the cluster kernel needs 9.9;
//   * lane <-> (16-byte channel chunk q, bin slot): the 16 lanes of a bin slot read the 16 consecutive 16-byte slots of ONE pixel,
//     and the lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) take complementary chunk ranges from two pixels, so every
//     16-lane group covers the 16 bank-slots exactly once whatever the pixels are: 4 LDS cycles per tap read, by construction;
//   * a bin's 16 tap offsets and 8 axis weights are the same for its 16 lanes: formed once per workgroup into an LDS table
//     (thread <-> bin), fetched with six broadcast reads.
// workgroup = (RoI, 256-byte channel block); a window that does not fit the LDS image is pooled in STRIPS of bin rows; a RoI whose
// single bin row does not fit (or a level whose strides are not the plain channels_last ones) takes the direct-gather kernel
// roi_align_fwd_nhwc's path per output.  Results leave through an LDS slab [channels][bins] as 16-byte stores.
#include <stdlib.h>

#include <mutex>

#include "roi_align_common.h"

namespace dtc {

constexpr int kNlThreads = 256;
constexpr int kNlChunk = 256;                  // bytes of one pixel's channel block in the LDS image
constexpr int kNlMaxBins = 64;
constexpr int kNlBinRec = 96;                  // bytes per bin record: 16 tap offsets (uint32) + y/x weights (8 floats)

typedef float nf32x2 __attribute__((ext_vector_type(2)));
typedef float nf32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t nu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void nl_lds_void;
typedef __attribute__((address_space(1))) const void nl_glb_void;

__device__ __forceinline__ int nl_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16 bytes of the LDS image -> the lane's channels as float32 (4 for float32 maps, 8 for 16-bit maps)
template <typename TIn> struct NlLane;
template <> struct NlLane<float> {
  static constexpr int kCh = 4;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
};
template <> struct NlLane<__half> {
  static constexpr int kCh = 8;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
      v[2 * i] = __low2float(h); v[2 * i + 1] = __high2float(h);
    }
  }
};
template <> struct NlLane<bf16_t> {
  static constexpr int kCh = 8;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};

template <typename TOut> __device__ __forceinline__ void nl_store4(TOut* d, float4 v);
template <> __device__ __forceinline__ void nl_store4<float>(float* d, float4 v) { *reinterpret_cast<float4*>(d) = v; }
template <> __device__ __forceinline__ void nl_store4<__half>(__half* d, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  uint2 r; r.x = *reinterpret_cast<const uint32_t*>(&a); r.y = *reinterpret_cast<const uint32_t*>(&b);
  *reinterpret_cast<uint2*>(d) = r;
}
template <> __device__ __forceinline__ void nl_store4<bf16_t>(bf16_t* d, float4 v) {
  uint2 r;
  r.x = (uint32_t)from_f32<bf16_t>(v.x).bits | ((uint32_t)from_f32<bf16_t>(v.y).bits << 16);
  r.y = (uint32_t)from_f32<bf16_t>(v.z).bits | ((uint32_t)from_f32<bf16_t>(v.w).bits << 16);
  *reinterpret_cast<uint2*>(d) = r;
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kNlThreads) void roi_align_fwd_nhwc_lds(RoiAlignParams p, int img_pixels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CL = NlLane<TIn>::kCh;                 // channels per lane
  constexpr int CB = 16 * CL;                           // channels per workgroup: one 256-byte chunk per pixel
  const int bins = p.pooled_h * p.pooled_w;
  // [axis samples: 2 x 32 AxisEntry][bin records: bins x 96 B][slab: CB x bins float32][image: img_pixels x 256 B]
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);
  AxisEntry* xtab = ytab + 32;
  unsigned char* brec = smem + 1024;
  float* slab = reinterpret_cast<float*>(smem + 1024 + kNlMaxBins * kNlBinRec);
  unsigned char* img = smem + 1024 + kNlMaxBins * kNlBinRec + (size_t)CB * bins * 4;
  const uint32_t img32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)img;
  const int tid = threadIdx.x, lane = tid & 63, wv = nl_uni(tid >> 6);
  const int nct = p.channels / CB;
  const int wi = xcd_work_item(blockIdx.x, gridDim.x, p.xcd_remap);
  const int ri = wi / nct;
  const int c0 = (wi - ri * nct) * CB;
  const RoiHead hd = load_roi_head(p, ri);
  TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)hd.r * p.channels + c0) * bins;
  if (hd.lvl < 0 || hd.lvl >= p.n_levels) {            // padding row of a fixed-shape batch: defined output
    for (int o = tid; o < CB * bins; o += kNlThreads) out[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[hd.lvl];
  const int H = L.height, W = L.width;
  // ---- axis samples (roi_align_cpu_loop.cpp:36-95), one thread per sample ------------------------------------------------------
  if (tid < 2 * p.pooled_h) ytab[tid] = make_axis(hd.sh, hd.bin_h, tid >> 1, tid & 1, 2, H);
  else if (tid >= 64 && tid < 64 + 2 * p.pooled_w) xtab[tid - 64] = make_axis(hd.sw, hd.bin_w, (tid - 64) >> 1, (tid - 64) & 1, 2, W);
  __syncthreads();
  const int x0 = xtab[0].lo, x1 = xtab[2 * p.pooled_w - 1].hi;        // sample positions are non-decreasing: first .lo / last .hi
  const int w = x1 - x0 + 1;
  const TIn* fbase = reinterpret_cast<const TIn*>(L.data) + (int64_t)hd.b * L.stride_n + c0;
  // plain channels_last strides and 16-byte alignment: what the LDS-DMA staging needs
  const bool dma_ok = L.stride_c == 1 && ((L.stride_w * (int64_t)sizeof(TIn)) & 15) == 0 && ((L.stride_h * (int64_t)sizeof(TIn)) & 15) == 0 &&
                      ((L.stride_n * (int64_t)sizeof(TIn)) & 15) == 0 && (reinterpret_cast<uintptr_t>(L.data) & 15) == 0;
  const int q = lane & 15;                               // this lane's 16-byte chunk of a pixel
  const int slot = (tid >> 4);                           // bin slot 0..15 of the workgroup

  int pa = 0;
  while (pa < p.pooled_h) {
    // ---- the strip: the longest run of bin rows [pa, pb) whose window fits the LDS image (uniform) -------------------------------
    const int ys = ytab[2 * pa].lo;
    int pb = pa, hs = 0;
    while (pb < p.pooled_h) {
      const int h2 = ytab[2 * pb + 1].hi - ys + 1;
      if (h2 * w > img_pixels) break;
      hs = h2; pb++;
    }
    const bool staged = dma_ok && pb > pa;
    if (!staged) pb = pa + 1;                           // not even one bin row fits (or odd strides): this bin row straight from global
    const int nb = (pb - pa) * p.pooled_w;              // bins of the strip
    const int b0 = pa * p.pooled_w;
    if (staged) {
      // ---- stage the window [ys, ys + hs) x [x0, x1] with LDS-DMA: a wave-instruction moves 4 pixels x 256 B, lane-linear ------
      const int np = hs * w;
      const float rw = 1.0f / (float)w;
      for (int k = wv; 4 * k < np; k += kNlThreads / 64) {
        const int pi = min(4 * k + (lane >> 4), np - 1);               // pixels past the window repeat its last one
        const int row = (int)(((float)pi + 0.5f) * rw);                  // exact: pi < 2^13
        const int col = pi - row * w;
        const TIn* g = fbase + (int64_t)(ys + row) * L.stride_h + (int64_t)(x0 + col) * L.stride_w;
        __builtin_amdgcn_global_load_lds((nl_glb_void*)(reinterpret_cast<const char*>(g) + q * 16),
                                         (nl_lds_void*)(img + (size_t)k * 1024), 16, 0, 0);
      }
    }
    // ---- bin records of the strip: thread <-> bin ---------------------------------------------------------------------------------
    if (tid < nb) {
      const int bl = tid, ph = pa + bl / p.pooled_w, pw = bl - (bl / p.pooled_w) * p.pooled_w;
      uint32_t* o32 = reinterpret_cast<uint32_t*>(brec + bl * kNlBinRec);
      float* wts = reinterpret_cast<float*>(brec + bl * kNlBinRec + 64);
#pragma unroll
      for (int iy = 0; iy < 2; iy++) {
        const AxisEntry ey = ytab[2 * ph + iy];
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
          const AxisEntry ex = xtab[2 * pw + ix];
          uint32_t t0, t1, t2, t3;
          if (staged) {   // byte offsets inside the LDS image
            t0 = (uint32_t)(((ey.lo - ys) * w + (ex.lo - x0)) * kNlChunk); t1 = (uint32_t)(((ey.lo - ys) * w + (ex.hi - x0)) * kNlChunk);
            t2 = (uint32_t)(((ey.hi - ys) * w + (ex.lo - x0)) * kNlChunk); t3 = (uint32_t)(((ey.hi - ys) * w + (ex.hi - x0)) * kNlChunk);
          } else {        // (row, column) packed: the direct path forms global addresses from them
            t0 = (uint32_t)ey.lo << 16 | (uint32_t)ex.lo; t1 = (uint32_t)ey.lo << 16 | (uint32_t)ex.hi;
            t2 = (uint32_t)ey.hi << 16 | (uint32_t)ex.lo; t3 = (uint32_t)ey.hi << 16 | (uint32_t)ex.hi;
          }
          o32[(iy * 2 + ix) * 4 + 0] = t0; o32[(iy * 2 + ix) * 4 + 1] = t1; o32[(iy * 2 + ix) * 4 + 2] = t2; o32[(iy * 2 + ix) * 4 + 3] = t3;
        }
        wts[iy * 2] = ey.l; wts[iy * 2 + 1] = ey.h;
      }
#pragma unroll
      for (int ix = 0; ix < 2; ix++) { const AxisEntry ex = xtab[2 * pw + ix]; wts[4 + ix * 2] = ex.l; wts[4 + ix * 2 + 1] = ex.h; }
    }
    __syncthreads();        // (an LDS-DMA in flight makes this fence wait vmcnt(0): the image is complete behind it)
    // ---- pool: 16 bins per pass of the workgroup, lane <-> (bin slot, 16-byte channel chunk) -----------------------------------------
    for (int bl0 = 0; bl0 < nb; bl0 += kNlThreads / 16) {
      const int bl = bl0 + slot;
      if (bl < nb) {
        const nu32x4* o4 = reinterpret_cast<const nu32x4*>(brec + bl * kNlBinRec);
        const nf32x4 wy = *reinterpret_cast<const nf32x4*>(brec + bl * kNlBinRec + 64);      // yl0 yh0 yl1 yh1
        const nf32x4 wx = *reinterpret_cast<const nf32x4*>(brec + bl * kNlBinRec + 80);      // xl0 xh0 xl1 xh1
        const float yl[2] = {wy.x, wy.z}, yh[2] = {wy.y, wy.w}, xl[2] = {wx.x, wx.z}, xh[2] = {wx.y, wx.w};
        float acc[CL];
#pragma unroll
        for (int c = 0; c < CL; c++) acc[c] = 0.f;
        // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
        for (int iy = 0; iy < 2; iy++)
#pragma unroll
          for (int ix = 0; ix < 2; ix++) {
            const nu32x4 off = o4[iy * 2 + ix];
            nu32x4 r1, r2, r3, r4;
            if (staged) {
              r1 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.x + q * 16);
              r2 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.y + q * 16);
              r3 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.z + q * 16);
              r4 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.w + q * 16);
            } else {
              const char* gb = reinterpret_cast<const char*>(fbase) + q * 16;
              auto ga = [&](uint32_t t) { return gb + ((int64_t)(t >> 16) * L.stride_h + (int64_t)(t & 0xffff) * L.stride_w) * (int64_t)sizeof(TIn); };
              if (dma_ok) {
                r1 = *reinterpret_cast<const nu32x4*>(ga(off.x)); r2 = *reinterpret_cast<const nu32x4*>(ga(off.y));
                r3 = *reinterpret_cast<const nu32x4*>(ga(off.z)); r4 = *reinterpret_cast<const nu32x4*>(ga(off.w));
              } else {      // unaligned chunks: 4-byte (2-byte) loads
                const uint32_t* a1 = reinterpret_cast<const uint32_t*>(ga(off.x)); const uint32_t* a2 = reinterpret_cast<const uint32_t*>(ga(off.y));
                const uint32_t* a3 = reinterpret_cast<const uint32_t*>(ga(off.z)); const uint32_t* a4 = reinterpret_cast<const uint32_t*>(ga(off.w));
                r1 = nu32x4{a1[0], a1[1], a1[2], a1[3]}; r2 = nu32x4{a2[0], a2[1], a2[2], a2[3]};
                r3 = nu32x4{a3[0], a3[1], a3[2], a3[3]}; r4 = nu32x4{a4[0], a4[1], a4[2], a4[3]};
              }
            }
            float v1[CL], v2[CL], v3[CL], v4[CL];
            NlLane<TIn>::widen(r1, v1); NlLane<TIn>::widen(r2, v2); NlLane<TIn>::widen(r3, v3); NlLane<TIn>::widen(r4, v4);
            const float w1 = yh[iy] * xh[ix], w2 = yh[iy] * xl[ix], w3 = yl[iy] * xh[ix], w4 = yl[iy] * xl[ix];   // :95
#pragma unroll
            for (int c = 0; c < CL; c++) acc[c] += w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];               // :208-211
          }
        // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32
        float* so = slab + (size_t)(q * CL) * bins + (b0 + bl);
#pragma unroll
        for (int c = 0; c < CL; c++) so[c * bins] = acc[c] * 0.25f;
      }
    }
    __syncthreads();        // every tap of the strip is read before the next strip's DMA (or the slab stores) go on
    pa = pb;
  }
  // ---- slab [CB][bins] is one contiguous run of the [R,C,PH,PW] output: 16-byte stores ------------------------------------------------
  const int n4 = (CB * bins) >> 2;
  if (((CB * bins) & 3) == 0) {
    for (int i = tid; i < n4; i += kNlThreads) nl_store4<TOut>(out + 4 * i, reinterpret_cast<const float4*>(slab)[i]);
  } else {
    for (int i = tid; i < CB * bins; i += kNlThreads) out[i] = from_f32<TOut>(slab[i]);
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// LDS per workgroup: 40 KB = four workgroups per CU (a 20 KB image: 2-3 strips per RoI).  The kernel is bound by the latency
// chain of a (RoI, channel block) item -- descriptor, DMA, pool, store: ~10 us -- times the items a CU holds, not by LDS or HBM
// throughput: measured on MI355X (8000 RoIs x 256 ch, float32) 32 KB 0.478 ms, 36 0.429, 40 0.401, 46 0.440, 52 0.419, 60 0.509,
// 78 0.494, 104-156 0.82; the direct-gather kernel 0.452; the NCHW cluster kernel on the same boxes 0.392.
struct NlConfig { int enabled = 1, lds_kb = 40; };
static const NlConfig& nl_config() {
  static const NlConfig cfg = [] {
    NlConfig c;
    if (const char* e = getenv("DTC_RA_NHWC_LDS")) c.enabled = atoi(e) != 0;
    if (const char* e = getenv("DTC_RA_NHWC_LDS_KB")) { const int v = atoi(e); if (v >= 24 && v <= 160) c.lds_kb = v; }
    return c;
  }();
  return cfg;
}

template <typename TIn> static int nl_cb() { return 16 * NlLane<TIn>::kCh; }

bool roi_align_nhwc_lds_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  if (!nl_config().enabled || p.sampling_ratio != 2) return false;
  // 16-bit maps: the direct-gather kernel is as fast (0.332 against 0.337 ms) -- this kernel takes them only when asked to
  static const bool force16 = [] { const char* e = getenv("DTC_RA_NHWC_LDS_16BIT"); return e && atoi(e) != 0; }();
  if (in_dtype != DTC_F32 && !force16) return false;
  const int bins = p.pooled_h * p.pooled_w;
  if (bins > kNlMaxBins || p.pooled_h > 16 || p.pooled_w > 16) return false;
  const int cb = in_dtype == DTC_F32 ? 64 : 128;
  if (p.channels % cb != 0) return false;
  // the tables and the output slab must leave room for a window image (else: the direct-gather kernel)
  if ((nl_config().lds_kb * 1024 - (1024 + kNlMaxBins * kNlBinRec + cb * bins * 4)) / kNlChunk < 16 + 3) return false;
  for (int l = 0; l < p.n_levels; l++)
    if (p.lv[l].stride_c != 1 || p.lv[l].height > 65535 || p.lv[l].width > 65535) return false;
  const bool f = in_dtype == DTC_F32, h = in_dtype == DTC_F16, b = in_dtype == DTC_BF16;
  return (f && (out_dtype == DTC_F32 || out_dtype == DTC_F16 || out_dtype == DTC_BF16)) ||
         (h && (out_dtype == DTC_F32 || out_dtype == DTC_F16)) || (b && (out_dtype == DTC_F32 || out_dtype == DTC_BF16));
}

template <typename TIn, typename TOut>
static int launch_nl_t(const RoiAlignParams& p, hipStream_t stream) {
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_nhwc_lds<TIn, TOut>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  const int bins = p.pooled_h * p.pooled_w;
  const int cb = nl_cb<TIn>();
  const int fixed = 1024 + kNlMaxBins * kNlBinRec + cb * bins * 4;
  const int lds_b = nl_config().lds_kb * 1024;
  int img_pixels = ((lds_b - fixed) / kNlChunk) & ~3;           // the DMA writes whole groups of 4 pixels
  if (img_pixels < 16) return DTC_EUNSUPPORTED;
  if (img_pixels > 8188) img_pixels = 8188;                      // pixel indices stay exact in the float reciprocal
  const int nct = p.channels / cb;
  hipLaunchKernelGGL((roi_align_fwd_nhwc_lds<TIn, TOut>), dim3((unsigned)p.n_rois * nct), dim3(kNlThreads), fixed + img_pixels * kNlChunk, stream, p, img_pixels);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

int launch_roi_align_nhwc_lds(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return launch_nl_t<float, float>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F16) return launch_nl_t<float, __half>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_BF16) return launch_nl_t<float, bf16_t>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return launch_nl_t<__half, float>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return launch_nl_t<__half, __half>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_F32) return launch_nl_t<bf16_t, float>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_BF16) return launch_nl_t<bf16_t, bf16_t>(p, stream);
  return DTC_EUNSUPPORTED;
}

}  // namespace dtc
