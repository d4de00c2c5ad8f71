// Vector-L1 (TCP) access accounting on gfx950 for the load shapes RoIAlign can choose from.  Development aid:
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/tcp_access_patterns tools/micro/tcp_access_patterns.hip
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d out -o p -- tools/micro/tcp_access_patterns
// Every kernel issues exactly ONE load instruction per thread, 1024 workgroups x 256 threads, over a 1 GiB buffer (cold
// lines).  Counters / (1024*4 wave-instructions) = accesses and L2 requests per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void probe(const float* __restrict__ src, float* __restrict__ dst, int stride_elems) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const float* base = src + (size_t)wave * 4096 * 4;          // 64 KB apart: no sharing between waves
  float acc = 0.f;
  if (MODE == 0) acc = base[lane];                             // 64 consecutive dwords, 256-B aligned
  if (MODE == 1) acc = base[lane + 1];                         // same, shifted by one dword (quads straddle 16-B units)
  if (MODE == 2) acc = base[(lane >> 4) * stride_elems + (lane & 15) + 3];   // 4 row pieces of 16 px, misaligned (the dword stager)
  if (MODE == 3) acc = base[(lane / 9) * stride_elems + (lane % 9) + 3];     // row pieces of 9 px (a typical window row)
  if (MODE == 4) { const float4 v = reinterpret_cast<const float4*>(base)[lane]; acc = v.x + v.y + v.z + v.w; }   // 64 x 16 B contiguous
  if (MODE == 5) { const float4 v = *reinterpret_cast<const float4*>(base + (size_t)lane * stride_elems); acc = v.x + v.y + v.z + v.w; }  // 16 B per lane, every lane its own line
  if (MODE == 6) { const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(lane >> 2) * stride_elems + (lane & 3) * 4); acc = v.x + v.y + v.z + v.w; }  // quads read 64 contiguous bytes, 16 different lines
  if (MODE == 7) acc = base[(size_t)(lane >> 2) * stride_elems + (lane & 3)];   // quads read 16 aligned bytes, 16 different lines
  if (MODE == 8) acc = base[(size_t)lane * stride_elems];        // every lane its own line, dword
  dst[t] = acc;
}

int main() {
  const size_t n = (size_t)1 << 28;   // 1 GiB of floats
  float *src, *dst;
  hipMalloc(&src, n * sizeof(float));
  hipMalloc(&dst, 1024 * 256 * sizeof(float));
  hipMemset(src, 0, n * sizeof(float));
  const int stride = 336;             // a P2 feature row
#define RUN(M) hipLaunchKernelGGL(probe<M>, dim3(1024), dim3(256), 0, 0, src, dst, stride); hipDeviceSynchronize();
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
  printf("done\n");
  return 0;
}
