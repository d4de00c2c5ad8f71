// Block-cooperative helper for MSB-first radix selection of the k-th LARGEST key (used by the RPN pre-NMS top-k,
// generate_proposals.py:77-86, and by the per-image detection limit, result_utils.py:159).
#pragma once
#include "dtc_common.h"

namespace dtc {

// Wave 0 of the block: find the digit d with  sum(h[d+1..]) < k <= sum(h[d..])  and the remaining rank inside it.
// Result broadcast through sh[0..1].  nbins <= 4096, a multiple of 64.
__device__ __forceinline__ void select_digit(const uint32_t* __restrict__ h, int nbins, uint32_t k, uint32_t* sh) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int per = nbins / 64;
    uint32_t local[64];
    uint32_t tot = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) {
      local[i] = i < per ? h[lane * per + i] : 0u;
      tot += local[i];
    }
    // inclusive suffix sum over lanes: suf = sum of tot for lanes >= lane
    uint32_t suf = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_down(suf, off, 64);
      if (lane + off < 64) suf += o;
    }
    const uint32_t above = suf - tot;  // elements in bins owned by higher lanes
    if (above < k && k <= suf) {
      uint32_t acc = above;
      int d = 0;
      uint32_t rem = 0;
      bool found = false;
#pragma unroll
      for (int i = 63; i >= 0; i--) {
        if (i < per && !found) {
          if (acc + local[i] >= k) { d = lane * per + i; rem = k - acc; found = true; }
          acc += local[i];
        }
      }
      sh[0] = (uint32_t)d;
      sh[1] = rem;
    }
  }
  __syncthreads();
}


}  // namespace dtc
