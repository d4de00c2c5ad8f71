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
//   * element strides instead of a fixed NCHW layout, fp16 or fp32 features, fp32 accumulation.
#include "dtc_common.h"

namespace dtc {

struct RoiAlignParams {
  dtc_feat_level lv[DTC_MAX_LEVELS];
  const float* rois;
  const int32_t* roi_levels;
  void* out;
  int n_levels, channels, roi_cols, n_rois, pooled_h, pooled_w, sampling_ratio, ch_tile;
};

// One axis of pre_calc_for_bilinear_interpolate (roi_align_cpu_loop.cpp:36-93).
struct AxisEntry {
  int lo, hi;   // element index along the axis (not yet multiplied by the stride)
  float l, h;   // l = v - lo ; h = 1 - l.  Both forced to 0 for an out-of-range sample (its PreCalc is all-zero, :49-63)
};

__device__ __forceinline__ AxisEntry make_axis(float start, float bin, int p, int i, int grid, int extent) {
  // :38-40  v = roi_start + p*bin + (i + .5f) * bin / grid     (float, left to right)
  float v = start + (float)p * bin;
  v = v + fdiv(((float)i + .5f) * bin, (float)grid);
  bool valid = !(v < -1.0f || v > (float)extent);  // :49 (float vs double -1.0 compares identically)
  if (v <= 0.f) v = 0.f;                           // :66-71
  int lo = (int)v, hi;
  if (lo >= extent - 1) { hi = lo = extent - 1; v = (float)lo; } else { hi = lo + 1; }  // :78-90
  float l = v - (float)lo;                         // :92
  float h = (float)(1.0 - (double)l);              // :94 "1. - ly": double subtract, rounded once to float
  AxisEntry e;
  e.lo = lo; e.hi = hi;
  e.l = valid ? l : 0.f;
  e.h = valid ? h : 0.f;
  return e;
}

constexpr int kRoiAlignThreads = 256;
constexpr int kMaxTableEntries = 2048;  // PH*gh + PW*gw ; larger (adaptive sampling on a huge RoI) -> on-the-fly path

// General kernel: any strides, any pooled size, adaptive sampling.  thread <-> output element (c, ph, pw) of the tile,
// consecutive threads -> consecutive addresses of the [R,C,PH,PW] output (coalesced stores).
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kRoiAlignThreads) void roi_align_fwd_general(RoiAlignParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);

  const int r = blockIdx.x;
  const int c0 = blockIdx.y * p.ch_tile;
  const int lvl = p.roi_levels ? p.roi_levels[r] : 0;
  if (lvl < 0 || lvl >= p.n_levels) {  // padding row of a fixed-shape batch (fpn.hip emits level -1): defined output
    const int bins0 = p.pooled_h * p.pooled_w;
    const int nc0 = min(p.ch_tile, p.channels - c0);
    TOut* o0 = reinterpret_cast<TOut*>(p.out) + ((size_t)r * p.channels + c0) * bins0;
    for (int o = threadIdx.x; o < nc0 * bins0; o += kRoiAlignThreads) o0[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[lvl];
  const float* roi = p.rois + (size_t)r * p.roi_cols;
  int b = 0;
  if (p.roi_cols == 5) { b = (int)roi[0]; roi++; }              // roi_align_cpu_loop.cpp:143-147
  const float s = L.spatial_scale;
  const float sw = roi[0] * s, sh = roi[1] * s, ew = roi[2] * s, eh = roi[3] * s;  // :150-153 no rounding
  const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);                // :160-161
  const float bin_h = fdiv(rh, (float)p.pooled_h), bin_w = fdiv(rw, (float)p.pooled_w);  // :162-163
  const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(fdiv(rh, (float)p.pooled_h));  // :166-170
  const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(fdiv(rw, (float)p.pooled_w));
  const float count = (float)(gh * gw);                                          // :173
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

}  // namespace dtc

DTC_API int dtc_roi_align_forward(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                     const float* rois, int roi_cols, const int32_t* roi_levels, int n_rois,
                                     int pooled_h, int pooled_w, int sampling_ratio, void* out, int out_dtype,
                                     dtc_stream_t stream) {
  if (!levels || n_levels < 1 || n_levels > DTC_MAX_LEVELS || channels < 1 || n_rois < 0 || pooled_h < 1 ||
      pooled_w < 1 || (roi_cols != 4 && roi_cols != 5) || (n_rois > 0 && (!rois || !out)))
    return DTC_EINVAL;
  dtc::RoiAlignParams p;
  for (int i = 0; i < n_levels; i++) {
    if (!levels[i].data || levels[i].height < 1 || levels[i].width < 1) return DTC_EINVAL;
    p.lv[i] = levels[i];
  }
  p.rois = rois; p.roi_levels = roi_levels; p.out = out;
  p.n_levels = n_levels; p.channels = channels; p.roi_cols = roi_cols; p.n_rois = n_rois;
  p.pooled_h = pooled_h; p.pooled_w = pooled_w; p.sampling_ratio = sampling_ratio;
  p.ch_tile = channels < 64 ? channels : 64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return dtc::launch_general<float, float>(p, s);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return dtc::launch_general<__half, float>(p, s);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return dtc::launch_general<__half, __half>(p, s);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F16) return dtc::launch_general<float, __half>(p, s);
  return DTC_EUNSUPPORTED;
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
