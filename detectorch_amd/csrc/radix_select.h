// Block-cooperative helper for MSB-first radix selection of the k-th LARGEST key (used by the RPN pre-NMS top-k,
// generate_proposals.py:77-86, and by the per-image detection limit, result_utils.py:159).
#pragma once
#include "dtc_common.h"

namespace dtc {

// Find the digit d with  sum(h[d+1..]) < k <= sum(h[d..])  and the remaining rank inside it; result broadcast through
// sh[0..1].  Every thread of the block calls (blockDim >= 256); nbins is a multiple of 256, <= 4096; h (global or LDS) is
// 16-byte aligned.  The first 256 threads own nbins / 256 consecutive bins each (vector loads), sum them, suffix-scan the sums
// with wave shuffles + one LDS hand-over of the four wave totals, and the one thread whose range holds the k-th element walks
// its <= 16 bins.  (Rounds 1-2 had wave 0 alone read 64 bins per lane with a stride of 64 -- 64 uncoalesced loads per lane from
// the global histogram, 32-way bank conflicts on an LDS one -- and walk them serially: 3.3-4.4 us per call on the global
// histogram, paid by every workgroup of rpn_hist<1> once and of rpn_compact twice.)
// The bins a thread owns, fetched ahead of the selection (two dependent selections -- rpn_compact's -- then cost ONE global round trip:
// both histograms are requested before the first is consumed).
struct DigitBins { uint32_t loc[16]; };
__device__ __forceinline__ void load_digit_bins(const uint32_t* __restrict__ h, int nbins, DigitBins& b) {
  const int t = threadIdx.x;
  const int per = nbins >> 8;
  // contract: every thread of the workgroup calls this, blockDim.x >= 256, nbins a multiple of 256 and <= 4096,
  // a 16-byte aligned histogram when nbins is a multiple of 1024
  if (blockDim.x < 256 || (nbins & 255) != 0 || per > 16 || ((per & 3) == 0 && (reinterpret_cast<uintptr_t>(h) & 15) != 0)) __builtin_trap();
#pragma unroll
  for (int i = 0; i < 16; i++) b.loc[i] = 0u;
  if (t < 256) {
    if ((per & 3) == 0) {
      const uint4* h4 = reinterpret_cast<const uint4*>(h + t * per);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint4 q = i < (per >> 2) ? h4[i] : make_uint4(0u, 0u, 0u, 0u);
        b.loc[4 * i] = q.x; b.loc[4 * i + 1] = q.y; b.loc[4 * i + 2] = q.z; b.loc[4 * i + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) b.loc[i] = i < per ? h[t * per + i] : 0u;
    }
  }
}

__device__ __forceinline__ void select_digit_loaded(const DigitBins& b, int nbins, uint32_t k, uint32_t* sh) {
  __shared__ uint32_t sd_wtot[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int per = nbins >> 8;
  const uint32_t* loc = b.loc;
  uint32_t part = 0, suf = 0;
  if (t < 256) {
#pragma unroll
    for (int i = 0; i < 16; i++) part += loc[i];
    suf = part;                                   // inclusive suffix sum over the lanes of this wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_down(suf, off, 64);
      if (lane + off < 64) suf += o;
    }
    if (lane == 0) sd_wtot[wv] = suf;
  }
  __syncthreads();
  if (t < 256) {
    uint32_t above = suf - part;                  // elements in bins owned by higher threads
    for (int q = wv + 1; q < 4; q++) above += sd_wtot[q];
    if (above < k && k <= above + part) {
      uint32_t acc = above, rem = 0;
      int d = 0;
      bool found = false;
#pragma unroll
      for (int i = 15; i >= 0; i--) {
        if (i < per && !found) {
          if (acc + loc[i] >= k) { d = t * per + i; rem = k - acc; found = true; }
          acc += loc[i];
        }
      }
      sh[0] = (uint32_t)d;
      sh[1] = rem;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void select_digit(const uint32_t* __restrict__ h, int nbins, uint32_t k, uint32_t* sh) {
  DigitBins b;
  load_digit_bins(h, nbins, b);
  select_digit_loaded(b, nbins, k, sh);
}


}  // namespace dtc
