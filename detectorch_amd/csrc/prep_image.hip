// Network-input preparation on the device (the caller side of the hot path).
//
// Reference: prep_im_for_blob + im_list_to_blob, lib/utils/blob.py:62-87 and :27-59 (driven by lib/utils/preprocess_sample.py:
// 25-34): BGR image -> float32, minus the per-channel pixel means (:72-73), cv2.resize(fx = fy = im_scale, INTER_LINEAR)
// (:84-85), zero-padded to the batch's max shape rounded up to the FPN stride (:38-47), HWC -> CHW (:55-57).
// The resize arithmetic is OpenCV's (third-party, absent, unpinned: "parity unpinned" like A9): restated from its
// documented rule -- dsize = round(src * scale); src coordinate of a destination pixel = (d + 0.5) / scale - 0.5; float32
// weights (1 - f, f); taps clamped at the borders (f forced to 0); horizontal pass first, then vertical -- exactly the
// restatement oracle/oracle.c:orc_prep_image checks this kernel against.
//
// One launch for the whole batch: thread <-> one destination pixel (x fastest: coalesced stores into the three channel
// planes), 4 taps x 3 channels gathered from the interleaved source.  HBM-bound: reads h*w*3 source bytes (u8) once through
// L2, writes 3*Hb*Wb floats.
#include "dtc_common.h"

namespace dtc {

struct PrepImage {
  const void* data;       // HWC, 3 channels (BGR), uint8 or float32
  int h, w, dtype, row_stride;   // row_stride in elements
  int oh, ow;             // resized size
  float inv_scale_x, inv_scale_y;   // unused (kept for alignment)
  double scale;           // im_scale (fx = fy)
};

constexpr int kPrepMaxBatch = 32;      // descriptor table travels as a kernel argument (~1.6 KB)

struct PrepParams {
  PrepImage im[kPrepMaxBatch];
  double mean[3];
  float* blob;            // [B, 3, Hb, Wb]
  int batch, Hb, Wb;
};

template <typename T>
__device__ __forceinline__ float px_minus_mean(const T* p, double mean) {
  return (float)((double)(*p) - mean);        // blob.py:72-73: float32 image -= float64 means (computed in double, stored float32)
}

__global__ __launch_bounds__(256) void prep_image_kernel(const PrepParams p) {
  const int b = blockIdx.z;
  const PrepImage& I = p.im[b];
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= p.Wb) return;
  float* out = p.blob + ((size_t)b * 3 * p.Hb + y) * p.Wb + x;
  const size_t plane = (size_t)p.Hb * p.Wb;
  if (y >= I.oh || x >= I.ow) {               // blob.py:45-49: zero padding
    out[0] = 0.f; out[plane] = 0.f; out[2 * plane] = 0.f;
    return;
  }
  const double inv = 1.0 / I.scale;           // cv::resize with fx given: scale_x = 1 / inv_scale_x
  float fy = (float)(((double)y + 0.5) * inv - 0.5);
  int sy = (int)floorf(fy); fy -= (float)sy;
  if (sy < 0) { sy = 0; fy = 0.f; }
  if (sy >= I.h - 1) { sy = I.h - 1; fy = 0.f; }
  const int sy1 = min(sy + 1, I.h - 1);
  float fx = (float)(((double)x + 0.5) * inv - 0.5);
  int sx = (int)floorf(fx); fx -= (float)sx;
  if (sx < 0) { sx = 0; fx = 0.f; }
  if (sx >= I.w - 1) { sx = I.w - 1; fx = 0.f; }
  const int sx1 = min(sx + 1, I.w - 1);
  const float ax0 = 1.f - fx, ay0 = 1.f - fy;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float v00, v01, v10, v11;
    if (I.dtype == DTC_U8) {
      const uint8_t* s = reinterpret_cast<const uint8_t*>(I.data);
      v00 = px_minus_mean(s + (size_t)sy * I.row_stride + sx * 3 + c, p.mean[c]);
      v01 = px_minus_mean(s + (size_t)sy * I.row_stride + sx1 * 3 + c, p.mean[c]);
      v10 = px_minus_mean(s + (size_t)sy1 * I.row_stride + sx * 3 + c, p.mean[c]);
      v11 = px_minus_mean(s + (size_t)sy1 * I.row_stride + sx1 * 3 + c, p.mean[c]);
    } else {
      const float* s = reinterpret_cast<const float*>(I.data);
      v00 = px_minus_mean(s + (size_t)sy * I.row_stride + sx * 3 + c, p.mean[c]);
      v01 = px_minus_mean(s + (size_t)sy * I.row_stride + sx1 * 3 + c, p.mean[c]);
      v10 = px_minus_mean(s + (size_t)sy1 * I.row_stride + sx * 3 + c, p.mean[c]);
      v11 = px_minus_mean(s + (size_t)sy1 * I.row_stride + sx1 * 3 + c, p.mean[c]);
    }
    const float r0 = v00 * ax0 + v01 * fx;    // horizontal pass
    const float r1 = v10 * ax0 + v11 * fx;
    out[c * plane] = r0 * ay0 + r1 * fy;      // vertical pass
  }
}

}  // namespace dtc

// blob.py:75-82 -- the scale of one image; Python float (double) arithmetic, np.round = round half to even
static double prep_scale(int h, int w, int target_size, int max_size) {
  const int mn = h < w ? h : w, mx = h < w ? w : h;
  double s = (double)target_size / (double)mn;
  if (nearbyint(s * (double)mx) > (double)max_size) s = (double)max_size / (double)mx;
  return s;
}

DTC_API int dtc_prep_plan(const int32_t* heights, const int32_t* widths, int batch, int target_size, int max_size,
                          int pad_stride, double* im_scales, int32_t* out_hw, int32_t* blob_hw) {
  if (!heights || !widths || batch < 1 || target_size < 1 || max_size < 1 || !im_scales || !out_hw || !blob_hw)
    return DTC_EINVAL;
  int mh = 0, mw = 0;
  for (int b = 0; b < batch; b++) {
    if (heights[b] < 1 || widths[b] < 1) return DTC_EINVAL;
    const double s = prep_scale(heights[b], widths[b], target_size, max_size);
    im_scales[b] = s;
    // cv::resize: dsize = (saturate_cast<int>(w * fx), saturate_cast<int>(h * fy)) -- round half to even
    const int ow = (int)nearbyint((double)widths[b] * s), oh = (int)nearbyint((double)heights[b] * s);
    out_hw[2 * b] = oh < 1 ? 1 : oh; out_hw[2 * b + 1] = ow < 1 ? 1 : ow;
    if (out_hw[2 * b] > mh) mh = out_hw[2 * b];
    if (out_hw[2 * b + 1] > mw) mw = out_hw[2 * b + 1];
  }
  if (pad_stride > 1) {                       // blob.py:41-44
    mh = (mh + pad_stride - 1) / pad_stride * pad_stride;
    mw = (mw + pad_stride - 1) / pad_stride * pad_stride;
  }
  blob_hw[0] = mh; blob_hw[1] = mw;
  return DTC_OK;
}

DTC_API int dtc_prep_images(const dtc_image* images, int batch, const double* pixel_means, const double* im_scales,
                            const int32_t* out_hw, float* blob, int blob_h, int blob_w, dtc_stream_t stream) {
  if (!images || batch < 1 || !pixel_means || !im_scales || !out_hw || !blob || blob_h < 1 || blob_w < 1) return DTC_EINVAL;
  if (batch > dtc::kPrepMaxBatch) return DTC_EUNSUPPORTED;
  dtc::PrepParams p;
  for (int b = 0; b < batch; b++) {
    const dtc_image& s = images[b];
    if (!s.data || s.height < 1 || s.width < 1 || (s.dtype != DTC_U8 && s.dtype != DTC_F32) || s.row_stride < 3 * s.width ||
        out_hw[2 * b] > blob_h || out_hw[2 * b + 1] > blob_w)
      return DTC_EINVAL;
    dtc::PrepImage& d = p.im[b];
    d.data = s.data; d.h = s.height; d.w = s.width; d.dtype = s.dtype; d.row_stride = s.row_stride;
    d.oh = out_hw[2 * b]; d.ow = out_hw[2 * b + 1]; d.inv_scale_x = d.inv_scale_y = 0.f; d.scale = im_scales[b];
  }
  for (int c = 0; c < 3; c++) p.mean[c] = pixel_means[c];
  p.blob = blob; p.batch = batch; p.Hb = blob_h; p.Wb = blob_w;
  hipLaunchKernelGGL(dtc::prep_image_kernel, dim3((blob_w + 255) / 256, blob_h, batch), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
