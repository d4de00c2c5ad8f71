// A9  Mask resize + binarise + paste geometry for gfx950 -- replaces the per-detection host loop of segm_results
// (lib/utils/result_utils.py:170-214): D2H of the full [D,81,M,M] mask tensor (80/81 of it unused), cv2.resize of the
// zero-padded (M+2)^2 mask to the box size, `> 0.5`, paste into a full-frame uint8 image.
//
// One workgroup per detection.  Only the class-specific channel (:192) is read; the padded mask is staged in LDS; the
// output is the binarised mask restricted to the paste rectangle (:204-214) -- the only part of im_mask that is not
// zero -- written back-to-back into one byte buffer per image (offsets are emitted), so nothing full-frame is ever
// materialised.  RLE encoding (:217-220, pycocotools) stays on the host.
//
// PARITY NOTE: the interpolation is OpenCV's cv2.resize(CV_32F, INTER_LINEAR), which is not part of the reference tree
// (un-vendored, unpinned).  It is restated from OpenCV's documented rule, identically in oracle/oracle.c
// (orc_mask_resize_binarize): per axis  f = (dst + 0.5) * (src/dst) - 0.5 evaluated in double and rounded to float,
// s = floor(f), frac = f - s, clamp (s<0 -> 0, frac 0; s >= src-1 -> src-1, frac 0); horizontal lerp then vertical lerp
// in float32 without contraction.
#include "dtc_common.h"

namespace dtc {

constexpr int kPasteThreads = 256;
constexpr int kMaxMaskSide = 64;  // M + 2 <= 64
constexpr int kPasteSplit = 8;    // workgroups per detection, at most (row bands): one huge box does not set the kernel's duration
constexpr int kBandPixels = 4096; // a detection uses ceil(area / kBandPixels) of its kPasteSplit workgroups; the others exit at once
constexpr int kMaxTab = 2048;     // paste rectangle width + height served from LDS tables (larger: per-pixel math)

struct PasteParams {
  const float* masks;         // [n_masks, n_cls, M, M]
  const int32_t* mask_index;  // [B, max_out] row into masks, or NULL (b*max_out + d)
  const float* dets;          // [B, max_out, 6] (x1,y1,x2,y2,score,class)
  const int32_t* det_count;   // [B]
  const float* im_size;       // [B, 2] original (h, w)
  int n_cls, M, max_out, cls_specific;
  float thresh;
  uint8_t* crops;             // [B, per_image_capacity]
  long long per_image_capacity;
  int32_t* mask_boxes;        // [B, max_out, 4] expanded int box (x0,y0,x1,y1)  result_utils.py:183-184
  int32_t* mask_rects;        // [B, max_out, 4] paste rect (x_0,y_0,x_1,y_1)     :204-207
  long long* mask_offsets;    // [B, max_out] byte offset of the crop inside image b's region
  long long* mask_bytes;      // [B] total bytes image b needs (> capacity => overflow, nothing beyond capacity written)
};

// expand_boxes (lib/utils/boxes.py:245-261) on float32 boxes + astype(int32) truncation (result_utils.py:183-184)
__device__ __forceinline__ void expand_box_int(const float* rb, int M, int out[4]) {
  const float scale = (float)(((double)M + 2.0) / (double)M);
  float w_half = (rb[2] - rb[0]) * .5f, h_half = (rb[3] - rb[1]) * .5f;
  const float x_c = (rb[2] + rb[0]) * .5f, y_c = (rb[3] + rb[1]) * .5f;
  w_half = w_half * scale; h_half = h_half * scale;
  out[0] = (int)(x_c - w_half); out[2] = (int)(x_c + w_half);
  out[1] = (int)(y_c - h_half); out[3] = (int)(y_c + h_half);
}

__device__ __forceinline__ void paste_rect(const int eb[4], int im_h, int im_w, int r[4]) {
  r[0] = max(eb[0], 0); r[2] = min(eb[2] + 1, im_w);   // :204-205
  r[1] = max(eb[1], 0); r[3] = min(eb[3] + 1, im_h);   // :206-207
  if (r[2] < r[0]) r[2] = r[0];
  if (r[3] < r[1]) r[3] = r[1];
}

// one axis of OpenCV's INTER_LINEAR coordinate rule (see the header): source index pair + fraction for destination index dd
__device__ __forceinline__ void resize_axis(int dd, double scale, int S, int& s0, int& s1, float& f) {
  f = (float)(((double)dd + 0.5) * scale - 0.5);
  s0 = (int)floorf(f); f -= (float)s0;
  if (s0 < 0) { s0 = 0; f = 0.f; }
  if (s0 >= S - 1) { s0 = S - 1; f = 0.f; }
  s1 = min(s0 + 1, S - 1);
}

// LDS: [ (M+2)^2 mask | kMaxTab int | kMaxTab float ] = 20 KB at M = 28 -> eight workgroups per CU.  (Round 1/2 kept 48 KB of
// static tables, i.e. three workgroups per CU, and ran the full set-up in all kPasteSplit x max_out x B workgroups: 8192
// workgroups in 11 rounds = 43 us for 2.6 MB of output.  Now a workgroup that has nothing to paste leaves after one load.)
__global__ __launch_bounds__(kPasteThreads) void mask_paste_kernel(PasteParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char paste_smem[];
  const int S = p.M + 2;
  float* pm = reinterpret_cast<float*>(paste_smem);
  int* tab_i = reinterpret_cast<int*>(pm + S * S);
  float* tab_f = reinterpret_cast<float*>(tab_i + kMaxTab);
  __shared__ long long red[kPasteThreads / 64];
  const int d = blockIdx.x / kPasteSplit, band = blockIdx.x % kPasteSplit, b = blockIdx.y, tid = threadIdx.x;
  const int nd = min(p.det_count[b], p.max_out);
  const bool publisher = d == 0 && band == 0;               // also writes the image's total byte count
  if (d >= nd && !publisher) return;
  const int im_h = (int)p.im_size[b * 2 + 0], im_w = (int)p.im_size[b * 2 + 1];
  const float* det = p.dets + ((size_t)b * p.max_out + d) * 6;
  int eb[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0};
  int nbands = 1;
  if (d < nd) {
    expand_box_int(det, p.M, eb);
    paste_rect(eb, im_h, im_w, r);
    const long long a = (long long)(r[2] - r[0]) * (r[3] - r[1]);
    nbands = (int)min((long long)kPasteSplit, max(1ll, (a + kBandPixels - 1) / kBandPixels));
    if (band >= nbands) return;
  }
  // byte offset = sum of the paste-rect areas of the detections before this one (block 0 also publishes the total)
  const int upto = (d == 0) ? nd : min(d, nd);
  long long acc = 0;
  for (int q = tid; q < upto; q += kPasteThreads) {
    int qb[4], qr[4];
    expand_box_int(p.dets + ((size_t)b * p.max_out + q) * 6, p.M, qb);
    paste_rect(qb, im_h, im_w, qr);
    acc += (long long)(qr[2] - qr[0]) * (qr[3] - qr[1]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  long long sum = 0;
  for (int q = 0; q < kPasteThreads / 64; q++) sum += red[q];
  if (publisher && tid == 0) p.mask_bytes[b] = sum;
  if (d >= nd) return;
  const long long offset = (d == 0) ? 0 : sum;

  int w = eb[2] - eb[0] + 1, h = eb[3] - eb[1] + 1;      // :197-198
  w = max(w, 1); h = max(h, 1);                          // :199-200
  if (band == 0 && tid < 4) {
    p.mask_boxes[((size_t)b * p.max_out + d) * 4 + tid] = eb[tid];
    p.mask_rects[((size_t)b * p.max_out + d) * 4 + tid] = r[tid];
  }
  if (band == 0 && tid == 0) p.mask_offsets[(size_t)b * p.max_out + d] = offset;

  // stage the zero-padded (M+2)x(M+2) mask of the detection's class (:185-195)
  const int cls = p.cls_specific ? (int)det[5] : 0;
  const size_t row = p.mask_index ? (size_t)p.mask_index[(size_t)b * p.max_out + d] : (size_t)b * p.max_out + d;
  const float* src = p.masks + (row * p.n_cls + cls) * p.M * p.M;
  for (int i = tid; i < S * S; i += kPasteThreads) {
    const int y = i / S, x = i - y * S;
    pm[i] = (y >= 1 && y <= p.M && x >= 1 && x <= p.M) ? src[(y - 1) * p.M + (x - 1)] : 0.f;
  }
  __syncthreads();

  const int rw = r[2] - r[0], rh = r[3] - r[1];
  const long long area = (long long)rw * rh;
  if (area == 0 || offset + area > p.per_image_capacity) return;
  uint8_t* out = p.crops + (size_t)b * p.per_image_capacity + offset;
  const double scale_x = (double)S / (double)w, scale_y = (double)S / (double)h;
  // Per-axis source index / fraction tables (the fp64 coordinate math is done once per row and once per column of the
  // paste rectangle instead of once per pixel).  Entry: sx (low 16 bits), sx1 (high 16 bits), frac.
  const bool use_tab = rw + rh <= kMaxTab;
  if (use_tab) {
    for (int t = tid; t < rw + rh; t += kPasteThreads) {
      const bool is_x = t < rw;
      const int dd = is_x ? (r[0] + t - eb[0]) : (r[1] + (t - rw) - eb[1]);   // coordinate inside the resized (w x h) mask
      int s0, s1; float f;
      resize_axis(dd, is_x ? scale_x : scale_y, S, s0, s1, f);
      tab_i[t] = s0 | (s1 << 16);
      tab_f[t] = f;
    }
    __syncthreads();
  }
  // this workgroup's band of rows; 32-bit index math (area <= im_h * im_w < 2^31; a 64-bit division per pixel dominated
  // this loop before)
  const int row0 = (int)((long long)rh * band / nbands), row1 = (int)((long long)rh * (band + 1) / nbands);
  for (int i = row0 * rw + tid; i < row1 * rw; i += kPasteThreads) {
    const int py = i / rw, px = i - py * rw;
    int sx, sx1, sy, sy1; float fx, fy;
    if (use_tab) {
      const int xi = tab_i[px], yi = tab_i[rw + py];
      fx = tab_f[px]; fy = tab_f[rw + py];
      sx = xi & 0xffff; sx1 = xi >> 16; sy = yi & 0xffff; sy1 = yi >> 16;
    } else {
      resize_axis(r[0] + px - eb[0], scale_x, S, sx, sx1, fx);
      resize_axis(r[1] + py - eb[1], scale_y, S, sy, sy1, fy);
    }
    const float r0 = pm[sy * S + sx] * (1.f - fx) + pm[sy * S + sx1] * fx;     // horizontal pass
    const float r1 = pm[sy1 * S + sx] * (1.f - fx) + pm[sy1 * S + sx1] * fx;
    const float v = r0 * (1.f - fy) + r1 * fy;                                  // vertical pass
    out[i] = v > p.thresh ? 1 : 0;                                               // :203
  }
}

}  // namespace dtc

DTC_API int dtc_mask_paste(const float* masks, const int32_t* mask_index, int n_cls, int M, const float* dets,
                           const int32_t* det_count, const float* im_size, int batch, int max_out, float thresh_binarize,
                           int cls_specific_mask, uint8_t* crops, long long per_image_capacity, int32_t* mask_boxes,
                           int32_t* mask_rects, long long* mask_offsets, long long* mask_bytes, dtc_stream_t stream) {
  if (batch < 0 || max_out < 1 || n_cls < 1 || M < 1 || M + 2 > dtc::kMaxMaskSide || per_image_capacity < 0) return DTC_EINVAL;
  if (batch == 0) return DTC_OK;
  if (!masks || !dets || !det_count || !im_size || !crops || !mask_boxes || !mask_rects || !mask_offsets || !mask_bytes)
    return DTC_EINVAL;
  dtc::PasteParams p;
  p.masks = masks; p.mask_index = mask_index; p.dets = dets; p.det_count = det_count; p.im_size = im_size;
  p.n_cls = n_cls; p.M = M; p.max_out = max_out; p.cls_specific = cls_specific_mask; p.thresh = thresh_binarize;
  p.crops = crops; p.per_image_capacity = per_image_capacity; p.mask_boxes = mask_boxes; p.mask_rects = mask_rects;
  p.mask_offsets = mask_offsets; p.mask_bytes = mask_bytes;
  const size_t lds = (size_t)(M + 2) * (M + 2) * sizeof(float) + (size_t)dtc::kMaxTab * (sizeof(int) + sizeof(float));
  hipLaunchKernelGGL(dtc::mask_paste_kernel, dim3(max_out * dtc::kPasteSplit, batch), dim3(dtc::kPasteThreads), lds,
                     reinterpret_cast<hipStream_t>(stream), p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
