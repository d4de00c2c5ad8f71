// Microbenchmark (round 5): how many 128-byte lines per second can SCALAR loads pull into L2?
// Idea under test: the box-head RoIAlign is bound by the vector L1's ~64 outstanding line fills x the fill latency; a
// scalar load of one dword per line goes through the scalar data cache, not the vector L1, and leaves the line in the XCD's L2 --
// a prefetch that costs no vector-L1 miss slot.  Worth building only if the scalar path sustains a useful rate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/scalar_prefetch_rate.hip -o /tmp/spr && /tmp/spr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int G>   // G scalar loads in flight per wave
__global__ __launch_bounds__(64) void scalar_walk(const char* base, size_t bytes_per_wave, int iters, unsigned* sink) {
  const char* p = base + (size_t)blockIdx.x * bytes_per_wave;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    unsigned v[G];
#pragma unroll
    for (int k = 0; k < G; k++) asm volatile("s_load_dword %0, %1, %2" : "=s"(v[k]) : "s"(p), "n"(k * 128));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < G; k++) acc ^= v[k];
    p += G * 128;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(64) void vector_walk(const char* base, size_t bytes_per_wave, int iters, unsigned* sink) {
  // same lines through the vector path: lane k of 8-line groups, one dword per line
  const char* p = base + (size_t)blockIdx.x * bytes_per_wave + (size_t)threadIdx.x * 128;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) { acc ^= *reinterpret_cast<const unsigned*>(p); p += 64 * 128; }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

int main() {
  const size_t total = (size_t)3 << 30;
  char* buf; unsigned* sink;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 1, total));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int waves_per_cu : {1, 4, 8, 16}) {
    const int nblk = 256 * waves_per_cu;
    const size_t per = (total / nblk) & ~(size_t)1023;
    auto run = [&](auto kern, int G, const char* name) {
      const int iters = (int)(per / (G * 128));
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, 0, buf, per, iters, sink); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double lines = (double)nblk * iters * G;
      printf("%-22s waves/CU %2d  in flight/wave %2d : %7.2f G lines/s = %6.2f TB/s of 128-B lines (%.2f ms)\n", name, waves_per_cu, G, lines / best / 1e6,
             lines * 128 / best / 1e9, best);
    };
    run(scalar_walk<4>, 4, "scalar s_load_dword");
    run(scalar_walk<8>, 8, "scalar s_load_dword");
    run(scalar_walk<15>, 15, "scalar s_load_dword");
    {
      const int iters = (int)(per / (64 * 128));
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(vector_walk, dim3(nblk), dim3(64), 0, 0, buf, per, iters, sink); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double lines = (double)nblk * iters * 64;
      printf("%-22s waves/CU %2d  (one line per lane)  : %7.2f G lines/s = %6.2f TB/s of 128-B lines (%.2f ms)\n", "vector global_load", waves_per_cu, lines / best / 1e6, lines * 128 / best / 1e9, best);
    }
  }
  return 0;
}
