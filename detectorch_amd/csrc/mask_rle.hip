// COCO run-length encoding of the pasted masks, on the device (the result wire format).
//
// Reference: segm_results, lib/utils/result_utils.py:204-221 -- the binarised crop is pasted into a zero (im_h, im_w) uint8
// frame (:204-214), converted to Fortran order and handed to pycocotools `mask.encode` (:217-220), whose published
// algorithm (cocoapi common/maskApi.c, rleEncode + rleToString; the submodule is absent from the reference checkout, see
// oracle/oracle.c:orc_rle_*) is: walk the frame column-major, emit the lengths of alternating runs starting with a run of
// zeros, then write every count (from the 4th on: minus the count two places back) as little-endian 5-bit groups with a
// continuation bit, offset by '0'.
//
// The frame is never materialised.  A detection's mask is zero outside its paste rectangle, so only the crop
// dtc_mask_paste wrote (row-major, rect (x0,y0,x1,y1)) is read; every crop pixel contributes at most two value changes of
// the column-major frame sequence: one in front of it (frame position F = x*im_h + y) when it differs from the frame pixel
// before it, and -- for the last pixel of a crop column that is set -- one behind it (F + 1) when the next frame pixel lies
// outside the crop.  The changes are compacted IN ORDER (wave ballots + cross-wave prefix), differenced into run lengths,
// and the count string is produced with a second ordered compaction over the per-count character lengths.
// One workgroup per detection; host receives ~100 bytes per mask instead of a dense bitmap.
#include "dtc_common.h"

namespace dtc {

constexpr int kRleThreads = 256;

struct RleParams {
  const uint8_t* crops;          // [B][cap]
  const int32_t* rects;          // [B][D][4]  x0,y0,x1,y1
  const long long* offsets;      // [B][D]
  const int32_t* det_count;      // [B]
  const float* im_size;          // [B][2] h,w
  long long cap;
  int batch, max_out, runs_stride, str_stride;
  uint32_t* counts;              // [B][D][runs_stride]
  int32_t* n_runs;               // [B][D]   (< 0: -needed, buffer too small; nothing valid written)
  uint8_t* str;                  // [B][D][str_stride]
  int32_t* str_len;              // [B][D]   (< 0: -needed)
};

// ordered block-wide exclusive offset for per-thread item counts (0..n each); returns the thread's first slot, updates
// `running` (shared) by the block total.  All threads must call.
__device__ __forceinline__ int ordered_slots(int mine, int* wave_tot, int* running) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // inclusive scan inside the wave (counts are tiny: shuffle-free ballot tricks do not apply to values > 1)
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int base = *running;
  for (int q = 0; q < wv; q++) base += wave_tot[q];
  const int slot = base + incl - mine;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < kRleThreads / 64; q++) t += wave_tot[q]; *running += t; }
  __syncthreads();
  return slot;
}

__device__ __forceinline__ int rle_chars(long long x) {   // number of characters rleToString emits for x
  int n = 0;
  bool more = true;
  while (more) {
    const int c = (int)(x & 0x1f);
    x >>= 5;
    more = (c & 0x10) ? (x != -1) : (x != 0);
    n++;
  }
  return n;
}

__global__ __launch_bounds__(kRleThreads) void mask_rle_kernel(RleParams p) {
  __shared__ int wave_tot[kRleThreads / 64];
  __shared__ int running;
  __shared__ uint32_t carry;
  const int b = blockIdx.y, d = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t di = (size_t)b * p.max_out + d;
  if (d >= p.det_count[b]) {
    if (tid == 0) { p.n_runs[di] = 0; p.str_len[di] = 0; }
    return;
  }
  const int im_h = (int)p.im_size[b * 2], im_w = (int)p.im_size[b * 2 + 1];
  const long long N = (long long)im_h * im_w;
  const int x0 = p.rects[di * 4], y0 = p.rects[di * 4 + 1], x1 = p.rects[di * 4 + 2], y1 = p.rects[di * 4 + 3];
  const int cw = max(x1 - x0, 0), ch = max(y1 - y0, 0);
  const uint8_t* crop = p.crops + (size_t)b * p.cap + p.offsets[di];
  uint32_t* cnt = p.counts + di * p.runs_stride;
  uint8_t* str = p.str + di * p.str_stride;
  const long long npx = (long long)cw * ch;
  // dtc_mask_paste skips a crop that does not fit image b's region (offset + area > capacity) but still publishes its offset:
  // encoding from there would read stale bytes, the next image's crops, or past the buffer.  Report "no valid data".
  if (p.offsets[di] < 0 || p.offsets[di] + npx > p.cap) {
    if (tid == 0) { p.n_runs[di] = -1; p.str_len[di] = -1; }
    return;
  }
  const bool full_h = (y0 == 0 && y1 == im_h);      // crop columns are contiguous in the frame sequence
  if (tid == 0) running = 0;
  __syncthreads();

  // ---- pass 1: positions of the value changes, in order, into cnt[] --------------------------------------------------
  for (long long j0 = 0; j0 < npx; j0 += kRleThreads) {
    const long long j = j0 + tid;
    int ns = 0;
    uint32_t pos_s = 0, pos_e = 0;
    bool s = false, e = false;
    if (j < npx) {
      const int cx = (int)(j / ch), cy = (int)(j - (long long)cx * ch);
      const uint8_t v = crop[(size_t)cy * cw + cx] != 0;
      uint8_t prev = 0;
      if (cy > 0) prev = crop[(size_t)(cy - 1) * cw + cx] != 0;
      else if (full_h && cx > 0) prev = crop[(size_t)(ch - 1) * cw + cx - 1] != 0;
      const long long F = (long long)(x0 + cx) * im_h + (y0 + cy);
      s = v != prev;
      pos_s = (uint32_t)F;
      const bool next_in = (cy + 1 < ch) || (full_h && cx + 1 < cw);
      e = v && !next_in && (F + 1 < N);
      pos_e = (uint32_t)(F + 1);
      ns = (s ? 1 : 0) + (e ? 1 : 0);
    }
    const int slot = ordered_slots(ns, wave_tot, &running);
    if (s && slot < p.runs_stride) cnt[slot] = pos_s;
    if (e && slot + (s ? 1 : 0) < p.runs_stride) cnt[slot + (s ? 1 : 0)] = pos_e;
  }
  __syncthreads();
  const int m = running;                   // number of value changes; runs = m + 1
  if (m + 1 > p.runs_stride) {
    if (tid == 0) { p.n_runs[di] = -(m + 1); p.str_len[di] = -1; }
    return;
  }
  // ---- pass 2: positions -> run lengths, in place (cnt[k] = P_k - P_{k-1}; cnt[m] = N - P_{m-1}) ----------------------
  __threadfence_block();
  if (tid == 0) carry = 0;                 // P_{-1} = 0: the first run (zeros) has length P_0
  __syncthreads();
  for (int k0 = 0; k0 <= m; k0 += kRleThreads) {
    const int k = k0 + tid;
    uint32_t cur = 0, prv = 0;
    if (k <= m) {
      cur = k < m ? cnt[k] : (uint32_t)N;
      prv = k == 0 ? 0u : (tid == 0 ? carry : cnt[k - 1]);
    }
    __syncthreads();
    if (k <= m) cnt[k] = cur - prv;
    if (tid == kRleThreads - 1) carry = cur;          // original P of the chunk's last element for the next chunk
    __syncthreads();
  }
  // ---- pass 3: the count string ---------------------------------------------------------------------------------------
  if (tid == 0) running = 0;
  __syncthreads();
  for (int k0 = 0; k0 <= m; k0 += kRleThreads) {
    const int k = k0 + tid;
    long long x = 0;
    int nc = 0;
    if (k <= m) {
      x = (long long)cnt[k];
      if (k > 2) x -= (long long)cnt[k - 2];
      nc = rle_chars(x);
    }
    const int slot = ordered_slots(nc, wave_tot, &running);
    if (k <= m && slot + nc <= p.str_stride) {
      bool more = true;
      int o = slot;
      while (more) {
        int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? (x != -1) : (x != 0);
        if (more) c |= 0x20;
        str[o++] = (uint8_t)(c + 48);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    p.n_runs[di] = m + 1;
    p.str_len[di] = running <= p.str_stride ? running : -running;
  }
}

}  // namespace dtc

DTC_API int dtc_mask_rle(const uint8_t* crops, long long per_image_capacity, const int32_t* mask_rects,
                         const long long* mask_offsets, const int32_t* det_count, const float* im_size, int batch,
                         int max_out, uint32_t* rle_counts, int runs_stride, int32_t* rle_n_runs, uint8_t* rle_str,
                         int str_stride, int32_t* rle_str_len, dtc_stream_t stream) {
  if (!crops || !mask_rects || !mask_offsets || !det_count || !im_size || !rle_counts || !rle_n_runs || !rle_str ||
      !rle_str_len || batch < 1 || max_out < 1 || runs_stride < 1 || str_stride < 1)
    return DTC_EINVAL;
  dtc::RleParams p;
  p.crops = crops; p.rects = mask_rects; p.offsets = mask_offsets; p.det_count = det_count; p.im_size = im_size;
  p.cap = per_image_capacity; p.batch = batch; p.max_out = max_out; p.runs_stride = runs_stride; p.str_stride = str_stride;
  p.counts = rle_counts; p.n_runs = rle_n_runs; p.str = rle_str; p.str_len = rle_str_len;
  hipLaunchKernelGGL(dtc::mask_rle_kernel, dim3(max_out, batch), dim3(dtc::kRleThreads), 0,
                     reinterpret_cast<hipStream_t>(stream), p);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
