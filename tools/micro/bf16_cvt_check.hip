// Exhaustive check: is gfx950's v_cvt_pk_bf16_f32 the same function as from_f32<bf16_t> (dtc_common.h: round to nearest even on the
// bit pattern, NaN -> truncated payload | quiet bit) for ALL 2^32 float32 inputs?   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16chk tools/micro/bf16_cvt_check.hip && /tmp/bf16chk
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint16_t sw(float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t hw2(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ void check(unsigned long long* mism, uint32_t* first) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 256;
  unsigned long long m = 0;
  for (int k = 0; k < 256; k += 2) {
    const uint32_t ua = (uint32_t)(i0 + k), ub = (uint32_t)(i0 + k + 1);
    const uint32_t h = hw2(__uint_as_float(ua), __uint_as_float(ub));
    const uint16_t sa = sw(__uint_as_float(ua)), sb = sw(__uint_as_float(ub));
    if ((uint16_t)h != sa) { if (atomicAdd(mism, 1ull) < 8) { first[0] = ua; first[1] = h & 0xffff; first[2] = sa; } m++; }
    if ((uint16_t)(h >> 16) != sb) { if (atomicAdd(mism, 1ull) < 8) { first[0] = ub; first[1] = h >> 16; first[2] = sb; } m++; }
  }
}

int main() {
  unsigned long long* d; uint32_t* f;
  hipMalloc(&d, 8); hipMalloc(&f, 12); hipMemset(d, 0, 8); hipMemset(f, 0, 12);
  hipLaunchKernelGGL(check, dim3(65536), dim3(256), 0, 0, d, f);           // 65536 * 256 * 256 = 2^32 patterns
  unsigned long long h = 0; uint32_t hf[3];
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, f, 12, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_bf16_f32 vs software round-to-nearest-even: %llu mismatches of 4294967296", h);
  if (h) printf("  (e.g. input 0x%08x: hardware 0x%04x, software 0x%04x)", hf[0], hf[1], hf[2]);
  printf("\n");
  return 0;
}
