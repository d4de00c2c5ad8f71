// A9  Mask resize + binarise + paste geometry for gfx950 -- replaces the per-detection host loop of segm_results
// (lib/utils/result_utils.py:170-214): D2H of the full [D,81,M,M] mask tensor (80/81 of it unused), cv2.resize of the
// zero-padded (M+2)^2 mask to the box size, `> 0.5`, paste into a full-frame uint8 image.
//
// One workgroup per (detection, band of rows) -- band 0 by the detection's own workgroup, further bands of large boxes by
// helper workgroups (see the kernel).  Only the class-specific channel (:192) is read; the padded mask is staged in LDS; the
// output is the binarised mask restricted to the paste rectangle (:204-214) -- the only part of im_mask that is not
// zero -- written back-to-back into one byte buffer per image (offsets are emitted), so nothing full-frame is ever
// materialised.  RLE encoding (:217-220, pycocotools): dtc_mask_rle (mask_rle.hip) on the same crops.
//
// PARITY NOTE: the interpolation is OpenCV's cv2.resize(CV_32F, INTER_LINEAR), which is not part of the reference tree
// (un-vendored, unpinned).  It is restated from OpenCV's documented rule, identically in oracle/oracle.c
// (orc_mask_resize_binarize): per axis  f = (dst + 0.5) * (src/dst) - 0.5 evaluated in double and rounded to float,
// s = floor(f), frac = f - s, clamp (s<0 -> 0, frac 0; s >= src-1 -> src-1, frac 0); horizontal lerp then vertical lerp
// in float32 without contraction.
#include "dtc_common.h"

namespace dtc {

DTC_PT_TABLE(mask_paste)

constexpr int kPasteThreads = 256;
constexpr int kMaxMaskSide = 64;   // M + 2 <= 64
constexpr int kBandPixels = 4096;  // a detection's paste rectangle is cut into row bands of about this many pixels ...
constexpr int kMaxBands = 64;      // ... at most this many: one huge box does not set the kernel's duration
constexpr int kHelpers = 64;       // helper workgroups per image: they take the bands 1.. of every detection of their image
constexpr int kMaxDets = 512;      // detections per image the helper scheme handles (LDS prefix tables); above: no helpers
constexpr int kMaxRowTab = 256;    // rows of a band served from the LDS row table (taller bands: per-row math)
constexpr int kRing = 16;          // source rows of the column lerps resident per lane (ring indexed by source row & 15)

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

// Separable form of the same arithmetic.  cv2.resize's value at (py, px) is  r0 * (1 - fy) + r1 * fy  with
// r = pm[sy][sx] * (1 - fx) + pm[sy][sx1] * fx: the horizontal lerp depends on (source row, px) only, so a lane owns ONE
// column of the paste rectangle, forms the horizontal lerp of its column against the source rows its share of the band
// samples (values in an LDS column it alone reads and writes: no barrier), and then walks down the rows with two LDS reads,
// one subtract, two multiplies and one add per pixel -- the very float32 operations the per-pixel form performs, in the same
// order, so the bytes are identical.  (The per-pixel form re-did the horizontal pass, four table reads and an index division
// for every pixel: 8 LDS reads + ~25 VALU instructions per byte written.)
//
// Work distribution.  grid = (kHelpers + max_out, B).  Main workgroup d < max_out pastes band 0 of detection d (and leaves after one
// load if the image has fewer detections); the kHelpers helper workgroups of an image share the bands 1.. of all its
// detections round-robin.  Every live workgroup derives the same per-image tables (rectangle areas -> byte offsets, extra
// bands -> item prefix) from the <= kMaxDets detection rows, so there is no inter-workgroup communication.  Round 2 launched
// 8 workgroups per detection slot (8192 for a batch of 8, 7300 of them leaving at once, in 4+ residency rounds: 30 us for
// 0.7 MB of crops); now 1536 are launched and a large box is cut into up to 64 bands instead of 8.
// LDS: [ kMaxRowTab (int, float) | (M+2)^2 mask | kRing x 256 column lerps ] = 22 KB at M = 28, + 6 KB of static tables:
// five workgroups per CU, the launch is resident at once.
struct PasteRect { int eb[4], r[4]; long long area; int nbands; };

__device__ __forceinline__ PasteRect paste_rect_of(const PasteParams& p, int b, int q, int im_h, int im_w) {
  PasteRect t;
  expand_box_int(p.dets + ((size_t)b * p.max_out + q) * 6, p.M, t.eb);
  paste_rect(t.eb, im_h, im_w, t.r);
  t.area = (long long)(t.r[2] - t.r[0]) * (t.r[3] - t.r[1]);
  t.nbands = (int)min((long long)kMaxBands, max(1ll, (t.area + kBandPixels - 1) / kBandPixels));
  return t;
}

__global__ __launch_bounds__(kPasteThreads) void mask_paste_kernel(PasteParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char paste_smem[];
  const int S = p.M + 2;
  int2* tab = reinterpret_cast<int2*>(paste_smem);               // [kMaxRowTab]: (byte offsets of rows sy | sy1 << 16 in hcol, fraction)
  float* pm = reinterpret_cast<float*>(tab + kMaxRowTab);          // [S][S]
  float* hcol = pm + S * S;                                        // [kRing][kPasteThreads]
  __shared__ long long s_off[kMaxDets];                            // exclusive prefix of the rectangle areas = byte offsets
  __shared__ int s_xi[kMaxDets];                                   // inclusive prefix of (nbands - 1) = helper items
  __shared__ long long red_a[kPasteThreads / 64];
  __shared__ int red_x[kPasteThreads / 64];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nd = min(p.det_count[b], p.max_out);
  const bool helpers = p.max_out <= kMaxDets;       // else: workgroup d pastes ALL bands of detection d, one after the other
  // the helpers are the first workgroups of an image in dispatch order: they carry the longest chains (several bands of the
  // large boxes), the one-band main workgroups fill in behind them
  const bool is_helper = helpers && (int)blockIdx.x < kHelpers;
  const int wg = is_helper ? p.max_out + (int)blockIdx.x : (int)blockIdx.x - (helpers ? kHelpers : 0);   // main: detection index
  if (is_helper ? (!helpers || nd == 0) : (wg >= nd && wg != 0)) return;
  const int im_h = (int)p.im_size[b * 2 + 0], im_w = (int)p.im_size[b * 2 + 1];
  const int ptk = is_helper ? 1 : 0, ptb = b == 0 ? (int)blockIdx.x - (is_helper ? 0 : kHelpers) : 1 << 20;   // phase trace: image 0 only
  (void)ptk; (void)ptb;
  DTC_PT(ptk, ptb, 0);

  // ---- per-image tables (every live workgroup: same inputs, same results) ---------------------------------------------------
  long long my_off = 0, total = 0;
  int n_items = 0;
  if (helpers) {
    // thread t owns detections kPer*t ..: serial inside the thread, shuffle scan across the wave, LDS across the waves
    constexpr int kPer = kMaxDets / kPasteThreads;
    long long a[kPer]; int x[kPer];
    long long ta = 0; int tx = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int q = tid * kPer + k;
      a[k] = 0; x[k] = 0;
      if (q < nd) { const PasteRect t = paste_rect_of(p, b, q, im_h, im_w); a[k] = t.area; x[k] = t.nbands - 1; }
      ta += a[k]; tx += x[k];
    }
    long long sa = ta; int sx = tx;                   // inclusive wave scan
    for (int off = 1; off < 64; off <<= 1) {
      const long long va = __shfl_up(sa, off, 64);
      const int vx = __shfl_up(sx, off, 64);
      if (lane >= off) { sa += va; sx += vx; }
    }
    if (lane == 63) { red_a[wv] = sa; red_x[wv] = sx; }
    __syncthreads();
    long long base_a = 0; int base_x = 0;
    for (int q = 0; q < kPasteThreads / 64; q++) {
      if (q < wv) { base_a += red_a[q]; base_x += red_x[q]; }
      total += red_a[q]; n_items += red_x[q];
    }
    long long ea = base_a + sa - ta; int ex = base_x + sx - tx;    // exclusive prefixes at detection kPer * t
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int q = tid * kPer + k;
      if (q < nd) { s_off[q] = ea; s_xi[q] = ex + x[k]; }
      ea += a[k]; ex += x[k];
    }
    __syncthreads();
  } else {
    // no tables: the byte offset is the sum of the areas of the detections in front of this one (workgroup 0: the total)
    const int upto = (wg == 0) ? nd : min(wg, nd);
    long long acc = 0;
    for (int q = tid; q < upto; q += kPasteThreads) acc += paste_rect_of(p, b, q, im_h, im_w).area;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) red_a[wv] = acc;
    __syncthreads();
    for (int q = 0; q < kPasteThreads / 64; q++) total += red_a[q];
    my_off = (wg == 0) ? 0 : total;
  }
  if (wg == 0 && tid == 0) p.mask_bytes[b] = total;
  DTC_PT(ptk, ptb, 1);
  if (!is_helper && wg >= nd) return;

  // ---- items of this workgroup: (detection, band) ---------------------------------------------------------------------------
  const int first = is_helper ? wg - p.max_out : 0;
  const int step = is_helper ? kHelpers : 1;
  int limit;
  if (is_helper) limit = n_items;
  else limit = helpers ? 1 : kMaxBands;             // main workgroup: band 0 (helpers take the rest) or all bands
  for (int j = first; j < limit; j += step) {
    int d, band;
    if (is_helper) {                                // upper bound of j in the inclusive item prefix (uniform, LDS broadcast)
      int lo = 0, hi = nd - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_xi[mid] > j) hi = mid; else lo = mid + 1; }
      d = lo;
      band = j - (d > 0 ? s_xi[d - 1] : 0) + 1;
    } else { d = wg; band = j; }
    const PasteRect t = paste_rect_of(p, b, d, im_h, im_w);
    if (band >= t.nbands) break;                    // (main workgroup without helpers: past its last band)
    const long long offset = helpers ? s_off[d] : my_off;
    const float* det = p.dets + ((size_t)b * p.max_out + d) * 6;
    const int* eb = t.eb; const int* r = t.r;
    int w = eb[2] - eb[0] + 1, h = eb[3] - eb[1] + 1;      // :197-198
    w = max(w, 1); h = max(h, 1);                          // :199-200
    if (band == 0 && tid < 4) {            // selects, not eb[tid]: a dynamic index would put the two arrays in scratch memory
      p.mask_boxes[((size_t)b * p.max_out + d) * 4 + tid] = tid == 0 ? eb[0] : tid == 1 ? eb[1] : tid == 2 ? eb[2] : eb[3];
      p.mask_rects[((size_t)b * p.max_out + d) * 4 + tid] = tid == 0 ? r[0] : tid == 1 ? r[1] : tid == 2 ? r[2] : r[3];
    }
    if (band == 0 && tid == 0) p.mask_offsets[(size_t)b * p.max_out + d] = offset;
    const int rw = r[2] - r[0], rh = r[3] - r[1];
    if (t.area == 0 || offset + t.area > p.per_image_capacity) continue;       // uniform

    __syncthreads();                                       // the previous item's readers are done with pm / tab
    // stage the zero-padded (M+2)x(M+2) mask of the detection's class (:185-195)
    const int cls = p.cls_specific ? (int)det[5] : 0;
    const size_t row = p.mask_index ? (size_t)p.mask_index[(size_t)b * p.max_out + d] : (size_t)b * p.max_out + d;
    const float* src = p.masks + (row * p.n_cls + cls) * p.M * p.M;
    for (int i = tid; i < S * S; i += kPasteThreads) {
      const int y = i / S, x = i - y * S;
      pm[i] = (y >= 1 && y <= p.M && x >= 1 && x <= p.M) ? src[(y - 1) * p.M + (x - 1)] : 0.f;
    }
    uint8_t* out = p.crops + (size_t)b * p.per_image_capacity + offset;
    const double scale_x = (double)S / (double)w, scale_y = (double)S / (double)h;
    // the band's rows and its row table: source row pair (as byte offsets of the two hcol rows) + fraction; the fp64 coordinate
    // math runs once per row of the band and once per column, not once per pixel
    const int row0 = (int)((long long)rh * band / t.nbands), row1 = (int)((long long)rh * (band + 1) / t.nbands);
    const int nr = row1 - row0;
    const bool use_tab = nr <= kMaxRowTab;
    if (use_tab) {
      for (int i = tid; i < nr; i += kPasteThreads) {
        int s0, s1; float f;
        resize_axis(r[1] + row0 + i - eb[1], scale_y, S, s0, s1, f);     // coordinate inside the resized (w x h) mask
        tab[i] = make_int2((int)((unsigned)((s0 & (kRing - 1)) * kPasteThreads * 4) | ((unsigned)((s1 & (kRing - 1)) * kPasteThreads * 4) << 16)), __float_as_int(f));   // ring slot offsets < 2^14
      }
    }
    __syncthreads();
    DTC_PT(ptk, ptb, 2);
    // From here on the wavefronts work on their own: a share is a 64-column chunk of the rectangle x a part of the band's rows
    // (narrow rectangles split the rows over the waves instead of leaving waves idle); a lane only reads the hcol column it wrote.
    const int ncc = (rw + 63) >> 6;
    const int nrs = ncc >= 4 ? 1 : 4 / ncc;
    const char* hme = reinterpret_cast<const char*>(hcol + tid);
    auto pix = [&](int2 e, uint8_t* o) {
      const float fy = __int_as_float(e.y);
      const float r0 = *reinterpret_cast<const float*>(hme + (e.x & 0xffff));
      const float r1 = *reinterpret_cast<const float*>(hme + ((unsigned)e.x >> 16));
      const float v = r0 * (1.f - fy) + r1 * fy;                                    // vertical pass
      *o = v > p.thresh ? 1 : 0;                                                    // :203
    };
    for (int share = wv; share < ncc * nrs; share += kPasteThreads / 64) {
      const int cc = share % ncc, rs = share / ncc;
      const int ta = nr * rs / nrs, tb = nr * (rs + 1) / nrs;       // rows [ta, tb) of the band
      const int px = cc * 64 + lane;
      if (tb <= ta || px >= rw) continue;
      int sx, sx1; float fx;
      resize_axis(r[0] + px - eb[0], scale_x, S, sx, sx1, fx);
      // The share's rows go in chunks whose source rows fit the ring: rows i0 .. i0 + R - 1 sample source rows
      // sy(i0) .. sy(i0) + (R - 1) * S / h + 2 at most (the row map is monotone with slope S / h), so R <= 13 h / S keeps the
      // span <= kRing.  Per chunk: horizontal pass of column px against exactly those source rows, then the vertical pass.
      const int R = max(1, (int)(13.0 * (double)h / (double)S));
      for (int i0 = ta; i0 < tb; i0 += R) {
        const int i1 = min(tb, i0 + R);
        int sy_lo, sy_hi, unused; float unused_f;
        resize_axis(r[1] + row0 + i0 - eb[1], scale_y, S, sy_lo, unused, unused_f);
        resize_axis(r[1] + row0 + i1 - 1 - eb[1], scale_y, S, unused, sy_hi, unused_f);
        for (int sy = sy_lo; sy <= sy_hi; sy++)
          hcol[(sy & (kRing - 1)) * kPasteThreads + tid] = pm[sy * S + sx] * (1.f - fx) + pm[sy * S + sx1] * fx;
        uint8_t* o = out + (size_t)(row0 + i0) * rw + px;
        if (use_tab) {
          int i = i0;
          for (; i + 4 <= i1; i += 4, o += 4 * (size_t)rw) {          // four rows in flight: the chain per row is two LDS latencies
            const int2 e0 = tab[i], e1 = tab[i + 1], e2 = tab[i + 2], e3 = tab[i + 3];   // uniform addresses: LDS broadcast
            pix(e0, o); pix(e1, o + rw); pix(e2, o + 2 * (size_t)rw); pix(e3, o + 3 * (size_t)rw);
          }
          for (; i < i1; i++, o += rw) pix(tab[i], o);
        } else {
          for (int i = i0; i < i1; i++, o += rw) {
            int sy, sy1; float fy;
            resize_axis(r[1] + row0 + i - eb[1], scale_y, S, sy, sy1, fy);
            pix(make_int2((int)((unsigned)((sy & (kRing - 1)) * kPasteThreads * 4) | ((unsigned)((sy1 & (kRing - 1)) * kPasteThreads * 4) << 16)), __float_as_int(fy)), o);
          }
        }
      }
    }
  }
  __syncthreads();
  DTC_PT(ptk, ptb, 3);
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
  const size_t lds = (size_t)(M + 2) * (M + 2) * sizeof(float) + (size_t)dtc::kRing * dtc::kPasteThreads * sizeof(float) +
                     (size_t)dtc::kMaxRowTab * (sizeof(int) + sizeof(float));
  hipLaunchKernelGGL(dtc::mask_paste_kernel, dim3(max_out + (max_out <= dtc::kMaxDets ? dtc::kHelpers : 0), batch), dim3(dtc::kPasteThreads), lds,
                     reinterpret_cast<hipStream_t>(stream), p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
