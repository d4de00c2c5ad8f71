// Shared device/host helpers for libdetectorch_hip (gfx950 only; wave = 64 lanes).
//
// Numerics contract: the whole library is compiled with -ffp-contract=off.  The reference's float32 results are the
// product of one rounding per operation in a fixed order (x86 SSE, no FMA), and NMS keep-indices are chaotic in the last
// bit, so nothing here may be contracted into v_fma/v_mad or re-associated.  Division and sqrt use the IEEE-correct
// intrinsics explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <mutex>

#include "../../include/detectorch_hip.h"

#define DTC_WAVE 64
// One target: the inline assembly below (v_cvt_pk_bf16_f32, v_fma_mix_f32 with an SGPR operand, global_load_lds_dwordx4) exists on
// gfx950 only.  Any other --offload-arch stops HERE with a readable message instead of somewhere inside the assembler.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdetectorch_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
#define DTC_API extern "C" __attribute__((visibility("default")))

#define DTC_CHECK_LAUNCH()                                 \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return DTC_ELAUNCH;             \
  } while (0)

// Raise a (non-template) kernel's dynamic-LDS limit once per process; thread-safe (std::call_once), the status of the one
// hipFuncSetAttribute call is remembered, so every caller of a failed raise gets DTC_ELAUNCH.
#define DTC_RAISE_LDS_ONCE(kernel, bytes)                                                                            \
  do {                                                                                                               \
    static std::once_flag once__;                                                                                    \
    static hipError_t rc__ = hipSuccess;                                                                             \
    std::call_once(once__, [] {                                                                                      \
      rc__ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                 (bytes));                                                                           \
    });                                                                                                              \
    if (rc__ != hipSuccess) return DTC_ELAUNCH;                                                                      \
  } while (0)

namespace dtc {

__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float fsqrt(float a) { return __fsqrt_rn(a); }
// exp / log2 evaluated in double and rounded once: correctly-rounded float32 in all but ~1e-8 of inputs, which is what
// oracle/oracle.c does too (see its header), so HIP == oracle bit-for-bit.
__device__ __forceinline__ float fexp_cr(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float flog2_cr(float x) { return (float)log2((double)x); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }

// bfloat16 storage type (DTC_BF16): the upper 16 bits of a float32; conversion to float32 is exact, from float32 rounds to
// nearest even (NaN stays NaN).  No arithmetic happens in bf16: the kernels accumulate in float32.
struct bf16_t { uint16_t bits; };
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return __uint_as_float((uint32_t)v.bits << 16); }
// round to nearest even on the bit pattern, NaN -> truncated payload | quiet bit (what torch's .to(bfloat16) does).  gfx950 has the
// instruction: v_cvt_pk_bf16_f32 equals the six-instruction bit arithmetic it replaces on ALL 2^32 inputs (NaNs, denormals and
// infinities included; exhaustive check on the GPU: tools/micro/bf16_cvt_check.hip)
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  uint32_t u;
  asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u) : "v"(v));
  bf16_t r;
  r.bits = (uint16_t)u;
  return r;
}
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 r) {
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}

// Streaming (non-temporal, `nt`) stores for outputs that are written once and never re-read by the kernel that writes them: the pooled
// features of a RoIAlign launch are 0.4-1.6 GB that would otherwise push the feature-map lines the neighbouring workgroups are about
// to re-use out of the XCD's 4 MB L2 (round 5, box-head launch: fabric read requests -10 %, L2 hit 0.51 -> 0.54, launch -2 %).
typedef uint32_t dtc_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t dtc_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream16(void* d, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  const dtc_u32x4 v = {x, y, z, w};
  __builtin_nontemporal_store(v, reinterpret_cast<dtc_u32x4*>(d));
}
__device__ __forceinline__ void store_stream16(void* d, float4 f) {
  store_stream16(d, __float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
}
__device__ __forceinline__ void store_stream8(void* d, uint32_t x, uint32_t y) {
  const dtc_u32x2 v = {x, y};
  __builtin_nontemporal_store(v, reinterpret_cast<dtc_u32x2*>(d));
}

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Development aid (-DDTC_PHASE_TRACE, tools/r02b/phase_trace.py): thread 0 of the first workgroups of an instrumented kernel
// stamps the 100 MHz wall clock at its phase boundaries into a per-file table [kernel id][workgroup][mark] that
// dtc_debug_phase_trace_<file>() copies out.  Compiled out of the product (the macros expand to nothing).
#ifdef DTC_PHASE_TRACE
constexpr int kPtKernels = 4, kPtBlocks = 64, kPtMarks = 24;
#define DTC_PT_TABLE(name)                                                                                         \
  static __device__ unsigned long long g_pt[dtc::kPtKernels * dtc::kPtBlocks * dtc::kPtMarks];                    \
  }                                                                                                                \
  DTC_API int dtc_debug_phase_trace_##name(void* dst, size_t bytes) {                                              \
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(dtc::g_pt), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; \
  }                                                                                                                \
  namespace dtc {
#define DTC_PT(kid, blk, mark)                                                                                     \
  do {                                                                                                             \
    if (threadIdx.x == 0 && (blk) < dtc::kPtBlocks)                                                                \
      dtc::g_pt[((kid) * dtc::kPtBlocks + (blk)) * dtc::kPtMarks + (mark)] = __builtin_amdgcn_s_memrealtime();    \
  } while (0)
#else
#define DTC_PT_TABLE(name)
#define DTC_PT(kid, blk, mark) ((void)0)
#endif

// Zero-fill as a KERNEL node.  hipMemsetAsync must not be used on any path that can be captured into a hipGraph: on ROCm
// 7.0 / gfx950 a captured memset node followed by kernels that are ALSO launched eagerly between replays was observed to
// run out of order with them (round 2: stale radix-select histograms after `replay B, eager A, eager B, replay A`).
template <int kUnused = 0>
__global__ void zero_words_kernel(uint32_t* p, size_t n_words) {
  // 16-byte stores where the buffer allows it (the radix-select histograms of a batch are 1.3 MB: a quarter of the workgroups)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n4 = (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? n_words >> 2 : 0;
  if (i < n4) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
  const size_t tail = 4 * n4 + i;                       // the first threads also clear what the 16-byte part leaves over
  if (i < n_words - 4 * n4 && n4 == 0) p[tail] = 0u;    // unaligned buffer: one word per thread (grid sized for it below)
  else if (n4 != 0 && i < (n_words & 3)) p[tail] = 0u;
}
inline int zero_async(void* p, size_t bytes, hipStream_t s) {   // bytes: multiple of 4, p 4-byte aligned
  if (bytes == 0) return DTC_OK;
  const size_t n = bytes / 4;
  const bool al16 = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const size_t threads = al16 ? (n >> 2) + 3 : n;      // >= the 16-byte stores, and >= the (< 4) left-over words
  hipLaunchKernelGGL(zero_words_kernel<0>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), n);
  return hipGetLastError() == hipSuccess ? DTC_OK : DTC_ELAUNCH;
}

}  // namespace dtc
