// Workgroup-wide bitonic sort of 64-bit keys held in LDS (gfx950, wave64).
//
// All ordering decisions of the hot path -- pre-NMS top-k (generate_proposals.py:77-86), the argsort inside NMS
// (cython_nms.pyx:45), collect's torch.sort (collect_and_distribute_fpn_rpn_proposals.py:102), the per-image detection
// limit (result_utils.py:159) -- go through ONE key format so the tie rule is the same everywhere:
//     key = (~monotone(score) << 32) | index        sorted ASCENDING  ==  score descending, index ascending.
#pragma once
#include "dtc_common.h"

namespace dtc {

__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  uint32_t b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0;  // -0.0 == +0.0 for every comparison the reference makes
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t o) {
  uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(b);
}
// descending-score / ascending-index key
__device__ __forceinline__ uint64_t make_desc_key(float score, uint32_t index) {
  return ((uint64_t)(~float_to_ordered(score)) << 32) | index;
}
__device__ __forceinline__ float desc_key_score(uint64_t k) { return ordered_to_float(~(uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t desc_key_index(uint64_t k) { return (uint32_t)k; }
constexpr uint64_t kPadKey = ~0ull;

// Sort keys[0..n_pow2) ascending.  n_pow2 is a power of two >= 2; every thread of the block must call.
// Comparator t of a step belongs to thread t % THREADS.  For distances j <= 64 the 64 comparators of a wave touch exactly the
// 128 consecutive keys [128 * (t / 64), +128) -- the same keys in every such step -- so consecutive steps with j <= 64 are
// private to the wave: LDS executes a wave's accesses in order, no workgroup barrier is needed between them (a sort of 1024
// keys has 55 steps, 9 of them with j >= 128).  A barrier separates steps only where the partition changes.
template <int THREADS>
__device__ __forceinline__ void block_bitonic_sort(uint64_t* keys, int n_pow2) {
  const int tid = threadIdx.x;
  bool wide_prev = true;                                  // keys were written by arbitrary threads before the call
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool wide = j > 64;
      if (wide || wide_prev) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
      wide_prev = wide;
      for (int t = tid; t < (n_pow2 >> 1); t += THREADS) {
        // t-th comparator of this stage: i has bit j clear
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const uint64_t a = keys[i], b = keys[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
    }
  }
  __syncthreads();
}

__host__ __device__ __forceinline__ int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace dtc
