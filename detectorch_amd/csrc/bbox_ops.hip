// bbox_overlaps and box_voting -- the IoU primitive and the optional bbox-vote refinement of the detection post-processing
// (SURVEY 8f-4).  Reference: lib/utils_cython/cython_bbox.pyx:32-72 (bbox_overlaps), lib/utils/boxes.py:280-329 (box_voting,
// called from box_results_with_nms_and_limit, lib/utils/result_utils.py:147-153 when do_bbox_vote is set).
//
// Numerics (oracle/oracle.c:orc_bbox_overlaps has the derivation, pinned against the reference's own Cython build): every
// `a - b + 1` is (double)(a - b) + 1.0 -- the subtraction in float32, the +1 and span products in double -- rounded to float32
// where the reference stores into a DTYPE_t variable; iw*ih is a float32 product; the division is IEEE float32.
// box_voting reproduces numpy's evaluation order: the weighted coordinate sums add the voters in index order (axis-0
// reduction of the [m,4] product), the weight sum is numpy's pairwise float32 summation.
#include "dtc_common.h"

namespace dtc {

__device__ __forceinline__ double span1(float hi, float lo) { return (double)(hi - lo) + 1.0; }

// one element of bbox_overlaps: box B vs query Q (cython_bbox.pyx:54-74)
__device__ __forceinline__ float iou_bbox(float4 B, float4 Q) {
  const float box_area = (float)(span1(Q.z, Q.x) * span1(Q.w, Q.y));          // :54-57
  const float iw = (float)span1(fminf(B.z, Q.z), fmaxf(B.x, Q.x));            // :59-62
  if (!(iw > 0.f)) return 0.f;
  const float ih = (float)span1(fminf(B.w, Q.w), fmaxf(B.y, Q.y));            // :64-67
  if (!(ih > 0.f)) return 0.f;
  const float ua = (float)(span1(B.z, B.x) * span1(B.w, B.y) + (double)box_area - (double)(iw * ih));   // :69-73
  return fdiv(iw * ih, ua);                                                   // :74
}

__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* __restrict__ boxes, int n, int box_stride,
                                                            const float* __restrict__ query, int k, int query_stride,
                                                            float* __restrict__ out, int vec_ok) {
  // thread <-> (row i, 4 consecutive queries): coalesced 16-byte stores of the row-major [n, k] matrix
  const long long total = (long long)n * ((k + 3) / 4);
  const int kq = (k + 3) / 4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / kq), q0 = (int)(t - (long long)i * kq) * 4;
    const float* bp = boxes + (size_t)i * box_stride;
    const float4 B = make_float4(bp[0], bp[1], bp[2], bp[3]);
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = min(q0 + u, k - 1);
      const float* qp = query + (size_t)q * query_stride;
      r[u] = iou_bbox(B, make_float4(qp[0], qp[1], qp[2], qp[3]));
    }
    float* o = out + (size_t)i * k + q0;
    if (vec_ok) *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
    else
      for (int u = 0; u < 4 && q0 + u < k; u++) o[u] = r[u];
  }
}

constexpr int kVoteMax = 8192;     // all_dets per call (LDS voter list: 32 KB)

// numpy float32 pairwise add-reduce over ws[vl[i]] (see oracle/oracle.c:np_pairwise_sum_f32), explicit stack instead of
// recursion: blocks of <= 128 elements, split at n/2 rounded down to a multiple of 8.
__device__ float np_sum_f32(const float* all, const int* vl, int n) {
  auto W = [&](int i) { return all[(size_t)vl[i] * 5 + 4]; };
  auto leaf = [&](int s, int m) {
    if (m < 8) {
      float r = 0.f;
      for (int i = 0; i < m; i++) r += W(s + i);
      return r;
    }
    float r0 = W(s), r1 = W(s + 1), r2 = W(s + 2), r3 = W(s + 3), r4 = W(s + 4), r5 = W(s + 5), r6 = W(s + 6), r7 = W(s + 7);
    int i = 8;
    for (; i < m - (m % 8); i += 8) {
      r0 += W(s + i); r1 += W(s + i + 1); r2 += W(s + i + 2); r3 += W(s + i + 3);
      r4 += W(s + i + 4); r5 += W(s + i + 5); r6 += W(s + i + 6); r7 += W(s + i + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < m; i++) res += W(s + i);
    return res;
  };
  int fs[16], fn[16], fstage[16];
  float val[16];
  int sp = 0, vp = 0;
  fs[0] = 0; fn[0] = n; fstage[0] = 0; sp = 1;
  while (sp > 0) {
    const int s = fs[sp - 1], m = fn[sp - 1];
    if (m <= 128) { val[vp++] = leaf(s, m); sp--; continue; }
    int n2 = m / 2; n2 -= n2 % 8;
    if (fstage[sp - 1] == 0) { fstage[sp - 1] = 1; fs[sp] = s; fn[sp] = n2; fstage[sp] = 0; sp++; }
    else if (fstage[sp - 1] == 1) { fstage[sp - 1] = 2; fs[sp] = s + n2; fn[sp] = m - n2; fstage[sp] = 0; sp++; }
    else { const float b = val[--vp]; const float a = val[--vp]; val[vp++] = a + b; sp--; }
  }
  return val[0];
}

// one wave per top det
__global__ __launch_bounds__(64) void box_voting_kernel(const float* __restrict__ top, int t, const float* __restrict__ all,
                                                        int a, float thresh, float* __restrict__ out,
                                                        int32_t* __restrict__ n_voters) {
  __shared__ int vl[kVoteMax];
  const int k = blockIdx.x, lane = threadIdx.x;
  const float* tp = top + (size_t)k * 5;
  const float4 B = make_float4(tp[0], tp[1], tp[2], tp[3]);
  int m = 0;
  for (int j0 = 0; j0 < a; j0 += 64) {
    const int j = j0 + lane;
    bool vote = false;
    if (j < a) {
      const float* ap = all + (size_t)j * 5;
      vote = iou_bbox(B, make_float4(ap[0], ap[1], ap[2], ap[3])) >= thresh;          // boxes.py:292
    }
    const uint64_t bal = __ballot(vote);
    if (vote) vl[m + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = j;
    m += __builtin_popcountll(bal);
  }
  __syncthreads();
  float res = 0.f;
  if (lane < 4) {            // weighted coordinate sums: voters in index order (axis-0 reduce of boxes * ws[:, None])
    float acc = 0.f;
    for (int i = 0; i < m; i++) {
      const float* ap = all + (size_t)vl[i] * 5;
      const float prod = ap[lane] * ap[4];
      acc = i == 0 ? prod : acc + prod;
    }
    res = acc;
  }
  float scl = 0.f;
  if (lane == 4 && m > 0) scl = np_sum_f32(all, vl, m);                                // ws.sum(): numpy pairwise
  scl = __shfl(scl, 4, 64);
  if (lane < 4) out[(size_t)k * 5 + lane] = m > 0 ? fdiv(res, scl) : tp[lane];        // :295 np.average
  if (lane == 4) out[(size_t)k * 5 + 4] = tp[4];                                       // 'ID' scoring: score unchanged
  if (lane == 5 && n_voters) n_voters[k] = m;
}

}  // namespace dtc

DTC_API int dtc_bbox_overlaps(const float* boxes, int n, int box_cols, const float* query_boxes, int k, int query_cols,
                              float* overlaps, dtc_stream_t stream) {
  if (n < 0 || k < 0 || box_cols < 4 || query_cols < 4) return DTC_EINVAL;
  if (n == 0 || k == 0) return DTC_OK;
  if (!boxes || !query_boxes || !overlaps) return DTC_EINVAL;
  const long long total = (long long)n * ((k + 3) / 4);
  const int blocks = (int)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
  hipLaunchKernelGGL(dtc::bbox_overlaps_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), boxes, n,
                     box_cols, query_boxes, k, query_cols, overlaps,
                     ((k & 3) == 0 && (reinterpret_cast<uintptr_t>(overlaps) & 15) == 0) ? 1 : 0);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

DTC_API int dtc_box_voting(const float* top_dets, int n_top, const float* all_dets, int n_all, float thresh,
                           float* top_dets_out, int32_t* n_voters, dtc_stream_t stream) {
  if (n_top < 0 || n_all < 0) return DTC_EINVAL;
  if (n_top == 0) return DTC_OK;
  if (!top_dets || !top_dets_out || (n_all > 0 && !all_dets)) return DTC_EINVAL;
  if (n_all > dtc::kVoteMax) return DTC_EUNSUPPORTED;
  hipLaunchKernelGGL(dtc::box_voting_kernel, dim3(n_top), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), top_dets, n_top,
                     all_dets, n_all, thresh, top_dets_out, n_voters);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}
