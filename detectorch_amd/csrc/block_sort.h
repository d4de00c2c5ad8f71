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

// value of lane (lane ^ J) for a COMPILE-TIME distance J < 64.  J = 1, 2, 4, 8 are DPP moves on the VALU (quad_perm / row_mirror /
// row_half_mirror: ~8 cycles); 16 is a ds_swizzle, 32 a ds_bpermute (both through the LDS crossbar, ~100+ cycles) -- the run-time
// distance of __shfl_xor is always the latter.  34 of the 45 wave-local steps of a 1024-key sort, 18 of the 21 of a 64-key sort, are
// distances 1-8.  Every lane of the wavefront must be active.
template <int J> __device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  const int x = (int)v;
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {                                                                                 // (l ^ 7) ^ 3
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);                                   // row_half_mirror
    return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, false);                                 // quad_perm [3,2,1,0]
  } else if constexpr (J == 8) {                                                                               // (l ^ 15) ^ 7
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);                                   // row_mirror
    return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x141, 0xf, 0xf, false);                                // row_half_mirror
  } else if constexpr (J == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle(x, 0x401F);                       // bit mode: xor 0x10, and 0x1f
  else return (uint32_t)__shfl_xor(x, J, 64);
}

// one compare-exchange step of distance J < 64 on the register-resident keys (see block_bitonic_sort_regs)
template <int J, int THREADS, int E>
__device__ __forceinline__ void bitonic_wave_step(uint64_t (&v)[E], int tid, int k) {
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int idx = e * THREADS + tid;
    const uint32_t lo = lane_xor<J>((uint32_t)v[e]), hi = lane_xor<J>((uint32_t)(v[e] >> 32));
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    const bool take_min = ((idx & J) == 0) == ((idx & k) == 0);
    v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
  }
}

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
      } else {                                             // partner in this wave: the rest of the stage, compile-time distances
        if (j >= 32) bitonic_wave_step<32, THREADS, E>(v, tid, k);
        if (j >= 16) bitonic_wave_step<16, THREADS, E>(v, tid, k);
        if (j >= 8) bitonic_wave_step<8, THREADS, E>(v, tid, k);
        if (j >= 4) bitonic_wave_step<4, THREADS, E>(v, tid, k);
        if (j >= 2) bitonic_wave_step<2, THREADS, E>(v, tid, k);
        bitonic_wave_step<1, THREADS, E>(v, tid, k);
        break;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) { const int idx = e * THREADS + tid; if (idx < n_pow2) keys[idx] = v[e]; }
  __syncthreads();
}

// One key per thread, 256 <= n_pow2 <= THREADS: every wavefront sorts its 64 keys in registers (the 21 wave-local steps of the network,
// ascending in every wave), then log2(n / 64) MERGE rounds through the LDS array -- every thread finds ITS output of the merge of two
// sorted runs by a merge-path search (<= log2(run) + 1 steps of two 8-byte LDS reads), two barriers per round.  A 1024-key sort: 21 + 4
// rounds instead of the network's 21 + 24 wave-local steps + 10 LDS exchanges with two barriers each (rpn_sort: 8.7 -> see tools/r06).
// Equal keys only occur as padding (kPadKey, sorted last): their mutual order is immaterial.
template <int THREADS>
__device__ __forceinline__ void block_merge_sort_e1(uint64_t* keys, int n_pow2) {
  const int tid = threadIdx.x;
  uint64_t v[1];
  v[0] = tid < n_pow2 ? keys[tid] : kPadKey;
  constexpr int kUp = 1 << 30;                             // (idx & kUp) == 0 for every idx: the last stage sorts every wave ascending
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
    const int kk = k == 64 ? kUp : k;
    if (k >= 64) bitonic_wave_step<32, THREADS, 1>(v, tid, kk);
    if (k >= 32) bitonic_wave_step<16, THREADS, 1>(v, tid, kk);
    if (k >= 16) bitonic_wave_step<8, THREADS, 1>(v, tid, kk);
    if (k >= 8) bitonic_wave_step<4, THREADS, 1>(v, tid, kk);
    if (k >= 4) bitonic_wave_step<2, THREADS, 1>(v, tid, kk);
    bitonic_wave_step<1, THREADS, 1>(v, tid, kk);
  }
  for (int run = 64; run < n_pow2; run <<= 1) {
    __syncthreads();                                       // the previous round's reads are done
    if (tid < n_pow2) keys[tid] = v[0];
    __syncthreads();
    if (tid < n_pow2) {
      const int k = tid & (2 * run - 1);
      const uint64_t* A = keys + (tid - k);
      const uint64_t* B = A + run;
      int lo = max(0, k - run), hi = min(k, run);          // i = number of elements taken from A among the first k outputs
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] < B[k - mid - 1]) lo = mid + 1; else hi = mid;
      }
      const int i = lo, j = k - lo;
      if (j >= run) v[0] = A[i];
      else if (i >= run) v[0] = B[j];
      else { const uint64_t x = A[i], y = B[j]; v[0] = x < y ? x : y; }
    }
  }
  __syncthreads();
  if (tid < n_pow2) keys[tid] = v[0];
  __syncthreads();
}

template <int THREADS>
__device__ __forceinline__ void block_bitonic_sort(uint64_t* keys, int n_pow2) {
  __syncthreads();                                         // keys were written by arbitrary threads before the call
  const int per = (n_pow2 + THREADS - 1) / THREADS;
  if (per <= 1 && n_pow2 >= 256) block_merge_sort_e1<THREADS>(keys, n_pow2);
  else if (per <= 1) block_bitonic_sort_regs<THREADS, 1>(keys, n_pow2);
  else if (per == 2) block_bitonic_sort_regs<THREADS, 2>(keys, n_pow2);
  else if (per == 4) block_bitonic_sort_regs<THREADS, 4>(keys, n_pow2);
  else if (per == 8) block_bitonic_sort_regs<THREADS, 8>(keys, n_pow2);
  else if (per == 16) block_bitonic_sort_regs<THREADS, 16>(keys, n_pow2);
  else __builtin_trap();        // more than 16 keys per thread: no instantiation sorts that (callers bound n_pow2 on the host)
}

// Stable LSD radix sort of n 64-bit keys (ascending) for LONG lists -- the 6000 + candidate keys of a C4 RPN segment, where the bitonic
// network above needs 91 steps over 8192 padded keys (85 us in one workgroup).  8-bit digits; `digit_mask` bit d set = digit d (key bits
// [8d, 8d + 8)) can differ between keys (the caller knows how many index bits are in use: the all-zero digits are skipped).  Two LDS key
// buffers (src = a on entry; returns the buffer that holds the sorted keys) + cnt[(THREADS / 64) * 256] counters.
// Wave w owns the contiguous positions [w * 64 * E, (w + 1) * 64 * E) of the current order, lane l its elements w * 64 * E + 64 e + l: inside a
// wave the order is (e, lane), so the rank of an element among the wave's equal digits = (equal digits of the wave's earlier rounds: a
// running per-wave counter) + (equal digits in lower lanes of this round: 8 ballots); across waves one exclusive scan per digit.
// Every thread of the block must call; n <= THREADS * E.
template <int THREADS, int E>
__device__ __forceinline__ uint64_t* block_radix_sort_u64(uint64_t* a, uint64_t* b, uint32_t* cnt, uint32_t* dig_base, int n, uint32_t digit_mask) {
  constexpr int NW = THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint64_t* src = a;
  uint64_t* dst = b;
  uint32_t* wc = cnt + w * 256;
  if (n > THREADS * E) __builtin_trap();                   // like the bitonic sort: callers bound n on the host
  __syncthreads();                                         // keys were written by arbitrary threads before the call
  for (int d = 0; d < 8; d++) {
    if (!((digit_mask >> d) & 1u)) continue;               // uniform
    const int shift = 8 * d;
    for (int i = tid; i < NW * 256; i += THREADS) cnt[i] = 0;
    __syncthreads();
    uint64_t key[E];
    uint32_t rank[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int pos = w * 64 * E + 64 * e + lane;
      const bool valid = pos < n;
      key[e] = valid ? src[pos] : ~0ull;
      const uint32_t dg = (uint32_t)(key[e] >> shift) & 255u;
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
        const bool one = (dg >> bit) & 1u;
        const uint64_t m = __ballot(one);
        peers &= one ? m : ~m;
      }
      const int leader = __builtin_ctzll(peers | (1ull << 63));     // (peers != 0 for a valid lane: it contains the lane itself)
      const uint32_t below = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
      uint32_t prior = 0;
      if (valid && lane == leader) { prior = wc[dg]; wc[dg] = prior + (uint32_t)__builtin_popcountll(peers); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's counter updates retire in order, round by round
      prior = (uint32_t)__shfl((int)prior, leader, 64);
      rank[e] = prior + below;
    }
    __syncthreads();
    // per digit: exclusive scan of the wave counts, digit totals; then the exclusive scan of the 256 totals (wave 0)
    if (tid < 256) {
      uint32_t tot = 0;
#pragma unroll
      for (int ww = 0; ww < NW; ww++) { const uint32_t c = cnt[ww * 256 + tid]; cnt[ww * 256 + tid] = tot; tot += c; }
      dig_base[tid] = tot;
    }
    __syncthreads();
    if (w == 0) {
      uint32_t v4[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) { v4[q] = dig_base[4 * lane + q]; sum += v4[q]; }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += o; }
      uint32_t run = incl - sum;
#pragma unroll
      for (int q = 0; q < 4; q++) { dig_base[4 * lane + q] = run; run += v4[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int pos = w * 64 * E + 64 * e + lane;
      if (pos < n) {
        const uint32_t dg = (uint32_t)(key[e] >> shift) & 255u;
        dst[dig_base[dg] + wc[dg] + rank[e]] = key[e];
      }
    }
    __syncthreads();
    uint64_t* t = src; src = dst; dst = t;
  }
  return src;
}

__host__ __device__ __forceinline__ int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace dtc
