// FPN level mapping of a SHORT list of rois (the mask branch: the <= ~100 detections of an image) as a device function, so that the
// kernel that produces the rows can emit what the multi-level RoIAlign consumes without another launch.
//   map_rois_to_fpn_levels        lib/utils/multilevel_rois.py:41-53
//   add_multilevel_rois_for_test  lib/utils/multilevel_rois.py:19-39  (mask branch, eval_mask_FPN.ipynb:249)
// The outputs are exactly those of dtc_fpn_collect_distribute(in_scores = NULL) -- fpn.hip uses the same two formulas below.
#pragma once
#include "dtc_common.h"

namespace dtc {

// lib/utils/multilevel_rois.py:47-52 in float32 numpy arithmetic (boxes_area: lib/utils/boxes.py:77-79)
__device__ __forceinline__ int fpn_level(float x1, float y1, float x2, float y2, int k_min, int k_max) {
  const float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
  const float s = fsqrt(area);
  float t = floorf(4.f + flog2_cr(fdiv(s, 224.f) + 1e-6f));
  t = fminf(fmaxf(t, (float)k_min), (float)k_max);
  return (int)t;
}

// Locality code of a RoI for the RoIAlign VISITING order (a performance hint, fpn.hip): level:3 | band of 2^band_log2 feature rows:6 |
// x centre in feature pixels:12 | row:11.  lvl < 0 (padding row): sorts last.
__device__ __forceinline__ uint32_t fpn_order_key(float4 bx, int lvl, int k_min, int band_log2, int r) {
  const uint32_t yc = (uint32_t)fminf(fmaxf((bx.y + bx.w) * 0.5f, 0.f), 65535.f);
  const uint32_t xc = (uint32_t)fminf(fmaxf((bx.x + bx.z) * 0.5f, 0.f), 65535.f);
  const uint32_t lv4 = lvl < 0 ? 15u : (uint32_t)lvl;
  const uint32_t fs = min((uint32_t)k_min + lv4, 15u);
  const uint32_t band = min((yc >> fs) >> band_log2, 63u), xf = min(xc >> fs, 4095u);
  return (min(lv4, 7u) << 29) | (((band << 12) | xf) << 11) | (uint32_t)r;      // r < 2048: 11 bits
}

struct FpnMapOut {          // dtc_fpn_collect_distribute's outputs (in_scores == NULL form), all [B, D, ...]
  float* rois5; int32_t* roi_levels; int32_t* n_out; float* rois_by_level; int32_t* level_counts; int32_t* idx_restore;
  int32_t* roi_order; float* roi_desc;
  int k_min, k_max, band_log2, on;
};

constexpr int kFpnMapMaxRows = 512;

// Rows [0, D) of image b; rows t < m hold box_of_row(t), the rest are padding rows (level -1).  Every thread of the workgroup calls,
// AFTER a barrier behind which box_of_row's source is complete; blockDim >= D; code_s / key_s: LDS, D + 4 words each, D <= 512.
template <typename BoxFn>
__device__ __forceinline__ void fpn_map_rows(const FpnMapOut& o, int b, int D, int m, BoxFn box_of_row, uint32_t* code_s, uint32_t* key_s) {
  const int t = threadIdx.x;
  const int nl = o.k_max - o.k_min + 1;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  int lvl = -1;
  uint32_t code = 0xffffffffu, key = 0xffffffffu;
  if (t < D) {
    if (t < m) {
      bx = box_of_row(t);
      lvl = fpn_level(bx.x, bx.y, bx.z, bx.w, o.k_min, o.k_max) - o.k_min;
      code = ((uint32_t)lvl << 16) | (uint32_t)t;               // (level, row): np.where(lvls == lvl)[0] is ascending (:123)
    }
    key = fpn_order_key(bx, lvl, o.k_min, o.band_log2, t);
    code_s[t] = code; key_s[t] = key;
  }
  if (t >= D && t < D + 4) { code_s[t] = 0xffffffffu; key_s[t] = 0xffffffffu; }   // pad to a multiple of four: never smaller
  __syncthreads();
  const int n4 = (D + 3) >> 2;
  if (t < D) {
    int dst = 0, rank = 0;
    for (int j = 0; j < n4; j++) {
      const uint4 c = reinterpret_cast<const uint4*>(code_s)[j], k = reinterpret_cast<const uint4*>(key_s)[j];
      dst += (c.x < code ? 1 : 0) + (c.y < code ? 1 : 0) + (c.z < code ? 1 : 0) + (c.w < code ? 1 : 0);
      rank += (k.x < key ? 1 : 0) + (k.y < key ? 1 : 0) + (k.z < key ? 1 : 0) + (k.w < key ? 1 : 0);
    }
    if (lvl < 0) dst = -1;
    const size_t g = (size_t)b * D + t;
    float* r5 = o.rois5 + g * 5;
    r5[0] = (float)b; r5[1] = bx.x; r5[2] = bx.y; r5[3] = bx.z; r5[4] = bx.w;
    o.roi_levels[g] = lvl;
    o.idx_restore[g] = dst;                                       // argsort(concat(idx_lvl)) == inverse permutation (:127)
    if (dst >= 0) reinterpret_cast<float4*>(o.rois_by_level)[(size_t)b * D + dst] = bx;
    if (o.roi_order) {
      const size_t q = (size_t)b * D + rank;
      o.roi_order[q] = b * D + t;
      if (o.roi_desc) {
        float4* d = reinterpret_cast<float4*>(o.roi_desc + q * 8);
        d[0] = make_float4((float)b, bx.x, bx.y, bx.z);
        d[1] = make_float4(bx.w, (float)lvl, (float)(b * D + t), 0.f);
      }
    }
  }
  if (t < nl) {
    int cnt = 0;
    for (int j = 0; j < D; j++) cnt += (code_s[j] >> 16) == (uint32_t)t ? 1 : 0;
    o.level_counts[b * nl + t] = cnt;
  }
  if (t == 0) o.n_out[b] = m;
}

}  // namespace dtc
