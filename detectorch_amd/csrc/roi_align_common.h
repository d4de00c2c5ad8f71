// Shared device helpers of the RoIAlign kernels (roi_align.hip, roi_align_tile.hip): launch parameters, the XCD-aware
// work assignment, and the RoI geometry of roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:36-173)
// restated operation for operation so that every kernel forms bit-identical sampling positions and weights.
#pragma once
#include "dtc_common.h"

namespace dtc {

struct RoiAlignParams {
  dtc_feat_level lv[DTC_MAX_LEVELS];
  const float* rois;
  const int32_t* roi_levels;
  const int32_t* roi_order;   // optional processing order: workgroup i handles RoI roi_order[i] (output row unchanged)
  const float* roi_desc;      // optional packed descriptors [R,8] in VISITING order: (batch, x1, y1, x2, y2, level, out_row, 0)
                              // one 32-byte load instead of three dependent global loads at the head of every workgroup
  void* out;
  int n_levels, channels, roi_cols, n_rois, pooled_h, pooled_w, sampling_ratio, ch_tile;
  int ch_block;   // channels per workgroup of the LDS kernel (multiple of 64): setup (tables, window) is paid once per block
  int cts64;      // 1: allow 64-channel sub-tiles (one bin per ds_read_b128 lane group: conflict-free taps)
  int xcd_remap;  // 1: workgroup -> work-item mapping keeps each XCD on a contiguous range of the visiting order
  // map-stationary kernel only: per-RoI records formed once per launch by map_prep_kernel (geometry, axis samples), so that the
  // 128 channel-group workgroups that pool a RoI do not each re-derive them (nullptr: derived in the kernel)
  const void* prep = nullptr;
};

// XCD-aware work assignment.  The dispatcher deals workgroups round-robin over the 8 XCDs (workgroup b runs on XCD b % 8)
// and each XCD has its own 4 MB L2.  RoIs are visited in (level, y) order so that neighbours share feature rows; with the
// identity mapping those neighbours land on 8 different L2s and every feature line is pulled through the fabric once per
// XCD that touches it.  Remapped, XCD x owns the contiguous slice [start(x), start(x+1)) of the visiting order and walks
// it front to back: a line is fetched by one L2 (two at a slice boundary).  Bijective for any grid size.
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_work_item(int b, int n, int enabled) {
  if (!enabled || n < 2 * kXcds) return b;
  const int x = b % kXcds, j = b / kXcds;
  const int q = n / kXcds, r = n - q * kXcds;
  return x * q + min(x, r) + j;
}

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

// What every workgroup derives from its RoI before touching features (roi_align_cpu_loop.cpp:143-173): which RoI / level /
// image, the scaled box (NO rounding, :150-153), bin sizes, sampling grid and the divisor.
struct RoiHead {
  int r, lvl, b;              // output row, level index (< 0: padding row), image index
  float sw, sh, rw, rh;       // scaled start (w, h) and size, size clamped to >= 1 (:160-161)
  float bin_h, bin_w;         // :162-163
  int gh, gw;                 // sampling grid (:166-170): fixed, or adaptive ceil(roi / pooled)
  float count, inv_count;     // :173 ; inv_count != 0 when count is a power of two (x * inv_count == x / count exactly)
};

// The raw record of RoI ri in processing order -- (batch, x1, y1, x2) (y2, level, output row, 0) -- split from the geometry
// so that a kernel can issue the loads of its NEXT RoI before it pools the current one.
struct RoiRaw { float4 d0, d1; };

__device__ __forceinline__ RoiRaw load_roi_raw(const RoiAlignParams& p, int ri) {
  RoiRaw w;
  if (p.roi_desc) {   // one packed 32-byte descriptor in visiting order
    w.d0 = reinterpret_cast<const float4*>(p.roi_desc)[(size_t)ri * 2];
    w.d1 = reinterpret_cast<const float4*>(p.roi_desc)[(size_t)ri * 2 + 1];
  } else {
    const int r = p.roi_order ? p.roi_order[ri] : ri;
    const int lvl = p.roi_levels ? p.roi_levels[r] : 0;
    const float* roi = p.rois + (size_t)r * p.roi_cols;
    float b = 0.f;
    if (p.roi_cols == 5) { b = roi[0]; roi++; }                 // :143-147
    w.d0 = make_float4(b, roi[0], roi[1], roi[2]);
    w.d1 = make_float4(roi[3], (float)lvl, (float)r, 0.f);
  }
  return w;
}

__device__ __forceinline__ RoiHead roi_head_from_raw(const RoiAlignParams& p, const RoiRaw& w) {
  RoiHead h;
  h.b = (int)w.d0.x; h.lvl = (int)w.d1.y; h.r = (int)w.d1.z;
  const float x1 = w.d0.y, y1 = w.d0.z, x2 = w.d0.w, y2 = w.d1.x;
  h.sw = h.sh = 0.f; h.rw = h.rh = 1.f; h.bin_h = h.bin_w = 1.f; h.gh = h.gw = 1; h.count = 1.f; h.inv_count = 1.f;
  if (h.lvl < 0 || h.lvl >= p.n_levels) return h;
  const float s = p.lv[h.lvl].spatial_scale;
  h.sw = x1 * s; h.sh = y1 * s;
  const float ew = x2 * s, eh = y2 * s;
  h.rw = fmaxf(ew - h.sw, 1.f); h.rh = fmaxf(eh - h.sh, 1.f);
  h.bin_h = fdiv(h.rh, (float)p.pooled_h); h.bin_w = fdiv(h.rw, (float)p.pooled_w);
  h.gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(fdiv(h.rh, (float)p.pooled_h));
  h.gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(fdiv(h.rw, (float)p.pooled_w));
  const int gg = h.gh * h.gw;
  h.count = (float)gg;
  h.inv_count = ((gg & (gg - 1)) == 0) ? fdiv(1.f, h.count) : 0.f;
  return h;
}

__device__ __forceinline__ RoiHead load_roi_head(const RoiAlignParams& p, int ri) {
  return roi_head_from_raw(p, load_roi_raw(p, ri));
}

// {w * float(low half), w * float(high half)} of a dword of two 16-bit elements, as a register pair for the packed fp32 adds.
// fp16: v_fma_mix_f32 converts and multiplies in ONE instruction -- fma(float(x), w, -0.0) is round(float(x) * w), the value
// v_cvt_f32_f16 + v_mul_f32 give (adding -0 never changes a sum).  bf16: two bit operations, one v_pk_mul_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ f32x2 mul_pair16(uint32_t u, float w);
template <> __device__ __forceinline__ f32x2 mul_pair16<__half>(uint32_t u, float w) {
  f32x2 r;
  asm("v_fma_mix_f32 %0, %2, %3, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(r.x), "=&v"(r.y) : "v"(u), "v"(w), "s"(-0.0f));
  return r;
}
template <> __device__ __forceinline__ f32x2 mul_pair16<bf16_t>(uint32_t u, float w) {
  const f32x2 v = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  return v * w;
}
template <> __device__ __forceinline__ f32x2 mul_pair16<float>(uint32_t, float) { return f32x2{0.f, 0.f}; }   // never instantiated for float maps

// CONTRACT mode (dtc_roi_align_set_exact(0)) on 16-bit maps:  acc += float(x) * w  as ONE fused operation per element.
// The reference is float-only (roi_align_forward_cuda.cu:199-208): on a 16-bit map there are no reference bits to match, north_star
// asks for <= 1e-4 on the float32-accumulated result, and the exact path spends a third of its vector instructions on keeping the
// multiply and the add of :75-77 apart (two roundings).  fp16: v_fma_mix_f32 converts, multiplies AND accumulates (one rounding);
// bf16: two bit operations + v_pk_fma_f32.  |fused - unfused| <= a few float32 ulp of the largest partial sum (measured: bench line).
template <typename T> __device__ __forceinline__ void fma_pair16(f32x2& acc, uint32_t u, float w);
template <> __device__ __forceinline__ void fma_pair16<__half>(f32x2& acc, uint32_t u, float w) {
  asm("v_fma_mix_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "+v"(acc.x), "+v"(acc.y) : "v"(u), "v"(w));
}
template <> __device__ __forceinline__ void fma_pair16<bf16_t>(f32x2& acc, uint32_t u, float w) {
  const f32x2 v = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  const f32x2 w2 = {w, w};
  acc = __builtin_elementwise_fma(v, w2, acc);
}
template <> __device__ __forceinline__ void fma_pair16<float>(f32x2&, uint32_t, float) {}   // never instantiated for float maps

// launchers of the cluster-stationary kernel (roi_align_tile.hip); in_dtype / out_dtype are DTC_* codes
bool roi_align_tile_supported(const RoiAlignParams& p, int in_dtype, int out_dtype);
int launch_roi_align_tile(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream);
// launchers of the map-stationary kernel (roi_align_map.hip): single-level inputs whose whole map fits LDS
bool roi_align_map_supported(const RoiAlignParams& p, int in_dtype, int out_dtype);
int launch_roi_align_map(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream);
void roi_align_set_exact(int exact);                   // 0: the map-stationary kernel may merge taps (<= 1e-5 from exact); default 1
int roi_align_get_exact();
size_t roi_align_map_workspace_bytes(int n_rois);      // per-launch preparation records (optional: workspace == nullptr -> none)
int launch_roi_align_map_ws(const RoiAlignParams& p, int in_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream);

// launchers of the channels_last kernel with an LDS-DMA staged window (roi_align_nhwc.hip): sampling_ratio 2, <= 64 bins
bool roi_align_nhwc_lds_supported(const RoiAlignParams& p, int in_dtype, int out_dtype);
int launch_roi_align_nhwc_lds(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream);

// launchers of the grouped 16-bit channels_last kernel (roi_align_nhwc16.hip): sampling_ratio 2, bins of several RoIs flattened over the lanes
bool roi_align_nhwc16_supported(const RoiAlignParams& p, int in_dtype, int out_dtype);
int launch_roi_align_nhwc16(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream);

}  // namespace dtc
