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

// Sort keys[0..n_pow2) ascending.  n_pow2 is a power of two >= 2, <= 16 * THREADS; every thread of the block must call.
//
// The keys live in REGISTERS during the sort: thread t holds elements e * THREADS + t (E = n_pow2 / THREADS of them, one when
// n_pow2 <= THREADS).  A compare-exchange step of distance j is
//   j <  64        a wave shuffle (the partner is lane ^ j): no LDS round trip, no barrier  -- 45 of the 55 steps of a 1024-key sort
//   64 <= j < THREADS   an exchange through the LDS array (write, barrier, read the partner, barrier)
//   j >= THREADS   between two registers of the same thread.
// Rounds 1-2 ran every step as two LDS reads + two conditional LDS writes per comparator with a wave or workgroup barrier in
// between: 11.7 us for the 2048 keys of rpn_sort_decode, 4.8 us for the <= 64 candidates of a det_candidates segment.
template <int THREADS, int E>
__device__ __forceinline__ void block_bitonic_sort_regs(uint64_t* keys, int n_pow2) {
  const int tid = threadIdx.x;
  uint64_t v[E];
#pragma unroll
  for (int e = 0; e < E; e++) { const int idx = e * THREADS + tid; v[e] = idx < n_pow2 ? keys[idx] : kPadKey; }
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= THREADS) {                                  // both elements in this thread (only when E > 1)
        const int je = j / THREADS;
#pragma unroll
        for (int jb = 1; jb < E; jb <<= 1) {               // unrolled over the possible distances: register indices stay constants
          if (je == jb) {
#pragma unroll
            for (int e = 0; e < E; e++) {
              if ((e & jb) == 0) {
                const int idx = e * THREADS + tid;
                const bool up = (idx & k) == 0;
                const uint64_t a = v[e], b = v[e | jb];
                if ((a > b) == up) { v[e] = b; v[e | jb] = a; }
              }
            }
          }
        }
      } else if (j >= 64) {                                // partner in another wave: through LDS
        __syncthreads();                                   // the previous exchange's reads are done
#pragma unroll
        for (int e = 0; e < E; e++) { const int idx = e * THREADS + tid; if (idx < n_pow2) keys[idx] = v[e]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int idx = e * THREADS + tid;
          if (idx < n_pow2) {
            const uint64_t o = keys[idx ^ j];
            const bool take_min = ((idx & j) == 0) == ((idx & k) == 0);
            v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
          }
        }
      } else {                                             // partner in this wave
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int idx = e * THREADS + tid;
          const uint32_t lo = __shfl_xor((uint32_t)v[e], j, 64), hi = __shfl_xor((uint32_t)(v[e] >> 32), j, 64);
          const uint64_t o = ((uint64_t)hi << 32) | lo;
          const bool take_min = ((idx & j) == 0) == ((idx & k) == 0);
          v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) { const int idx = e * THREADS + tid; if (idx < n_pow2) keys[idx] = v[e]; }
  __syncthreads();
}

template <int THREADS>
__device__ __forceinline__ void block_bitonic_sort(uint64_t* keys, int n_pow2) {
  __syncthreads();                                         // keys were written by arbitrary threads before the call
  const int per = (n_pow2 + THREADS - 1) / THREADS;
  if (per <= 1) block_bitonic_sort_regs<THREADS, 1>(keys, n_pow2);
  else if (per == 2) block_bitonic_sort_regs<THREADS, 2>(keys, n_pow2);
  else if (per == 4) block_bitonic_sort_regs<THREADS, 4>(keys, n_pow2);
  else if (per == 8) block_bitonic_sort_regs<THREADS, 8>(keys, n_pow2);
  else if (per == 16) block_bitonic_sort_regs<THREADS, 16>(keys, n_pow2);
  else __builtin_trap();        // more than 16 keys per thread: no instantiation sorts that (callers bound n_pow2 on the host)
}

__host__ __device__ __forceinline__ int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace dtc
