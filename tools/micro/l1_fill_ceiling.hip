// How many bytes per second can the vector L1s of an MI355X pull in, whatever a kernel then does with them?  (DESIGN.md 3.1:
// the box-head RoIAlign launch requests 20.7 M line fills = 2.65 GB and takes 0.36-0.37 ms = 7.2 TB/s of L1 fills.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/l1_fill_ceiling tools/micro/l1_fill_ceiling.hip && tools/micro/l1_fill_ceiling
// A pure-load kernel with the RoIAlign kernel's residency (256-thread workgroups, 3 or 4 per CU) and its load shape: a wave
// instruction fetches 16 ROW PIECES of 64 B (4 lanes x 16 B) from 16 different rows of a "feature plane" whose rows are
// `pitch` bytes apart, i.e. a patch of a window; a thread keeps U such loads in flight, a workgroup sweeps `rows` x 128 B
// patches over its own channel planes.  Nothing is computed: the loaded values are summed into one store per thread.
// Variants: working set resident in the Infinity Cache (64 MB region: 2 x the aggregate L2, so nearly every fill MISSES L2 -- a
// MALL -> L2 -> L1 rate) / streaming from HBM (2 GB region) / **L2-resident** (round 4: every XCD's workgroups -- block b runs on
// XCD b % 8 -- walk a slice of 1-3 MB of their own, i.e. every fill after the first sweep hits that XCD's 4 MB L2: the rate an
// L1 can pull from a HITTING L2); contiguous 1-KB wave loads for comparison (the plain streaming-copy read pattern).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: patch rows (16 rows x 64 B per wave instruction).  MODE 1: 1 KB contiguous per wave instruction.
template <int MODE, int U>
__global__ __launch_bounds__(256) void fill_kernel(const float4* __restrict__ src, size_t region_f4, int pitch_f4, int iters,
                                                   float* __restrict__ sink, int per_xcd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wg = blockIdx.x;
  float acc = 0.f;
  // per_xcd: region_f4 is the size of ONE XCD's private slice; the workgroups of XCD x (= blockIdx % 8) stay inside slice x
  if (per_xcd) src += (size_t)(blockIdx.x % 8) * region_f4;
  // every workgroup starts somewhere else in the region and walks it with a large odd stride: no two workgroups share lines
  size_t pos = ((wg * 0x9E3779B97F4A7C15ull) % region_f4) & ~(size_t)7;      // 128-byte aligned: a 256-byte row strip = 2 lines
  for (int it = 0; it < iters; it++) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t idx;
      if (MODE == 0) {
        // wave w, load u: 16 consecutive rows of one 64-byte column strip; the four waves take four neighbouring strips
        const size_t row0 = (size_t)(u * 16 + (lane >> 2));
        idx = pos + row0 * pitch_f4 + wave * 4 + (lane & 3);
      } else {
        idx = pos + (size_t)(u * 4 + wave) * 64 + lane;
      }
      if (idx >= region_f4) idx -= region_f4;
      v[u] = src[idx];
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    pos += MODE == 0 ? (size_t)U * 16 * pitch_f4 + 976 : (size_t)U * 4 * 64 + 976 * 64;
    if (pos >= region_f4) pos -= region_f4;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int U>
static int run(const char* name, const float4* src, size_t region_bytes, int pitch_bytes, int wgs_per_cu, float* sink, int per_xcd = 0) {
  const int cus = 256, blocks = cus * wgs_per_cu * 8, iters = 64;
  const size_t region_f4 = region_bytes / 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // occupancy is set through the dynamic LDS request: 160 KB / wgs_per_cu
  const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) - 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<MODE, U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((fill_kernel<MODE, U>), dim3(blocks), dim3(256), lds, 0, src, region_f4, pitch_bytes / 16, iters, sink, per_xcd);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((fill_kernel<MODE, U>), dim3(blocks), dim3(256), lds, 0, src, region_f4, pitch_bytes / 16, iters, sink, per_xcd);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  // bytes REQUESTED by the lanes; lines touched: MODE 0 a 64-byte strip is half a line -> the four waves of a workgroup
  // share two lines per row, so the lines filled are 2 per row and 16 rows x U per iteration (= requested bytes: 4 x 64 B = 2 lines)
  const double bytes = (double)blocks * 256 * iters * U * 16;
  printf("%-58s region %5zu MB  %d workgroups/CU  U=%d: %.3f ms  %.2f TB/s of L1 fills\n", name, region_bytes >> 20, wgs_per_cu, U,
         ms, bytes / ms / 1e9);
  fflush(stdout);
  return 0;
}

int main() {
  const size_t big = (size_t)2 << 30;
  float4* src;
  float* sink;
  CHECK(hipMalloc(&src, big));
  CHECK(hipMalloc(&sink, (size_t)256 * 8 * 8 * 256 * sizeof(float)));       // up to 8 workgroups per CU x 8 rounds
  CHECK(hipMemset(src, 0, big));
  const int pitch = 352 * 4;      // ~ a P2 row of the bench's feature maps (336 float32), rounded up to whole 128-byte lines
  for (int wgs = 3; wgs <= 4; wgs++) {
    if (run<0, 8>("patch rows (16 x 64 B per wave load), cache-resident", src, (size_t)64 << 20, pitch, wgs, sink)) return 1;
    if (run<0, 8>("patch rows (16 x 64 B per wave load), from HBM", src, big, pitch, wgs, sink)) return 1;
    if (run<1, 8>("contiguous 1 KB per wave load, cache-resident", src, (size_t)64 << 20, pitch, wgs, sink)) return 1;
    if (run<1, 8>("contiguous 1 KB per wave load, from HBM", src, big, pitch, wgs, sink)) return 1;
  }
  if (run<0, 4>("patch rows, 4 loads in flight per thread, cache-resident", src, (size_t)64 << 20, pitch, 3, sink)) return 1;
  if (run<0, 4>("patch rows, 4 loads in flight per thread, from HBM", src, big, pitch, 3, sink)) return 1;
  if (run<1, 8>("contiguous, 8 workgroups/CU, from HBM", src, big, pitch, 8, sink)) return 1;
  if (run<1, 8>("contiguous, 8 workgroups/CU, cache-resident", src, (size_t)64 << 20, pitch, 8, sink)) return 1;
  // round 4: L2-RESIDENT -- each XCD's workgroups confined to their own slice (the slice is swept ~25-75 times per launch)
  for (int mb = 1; mb <= 3; mb++) {
    if (run<0, 8>("patch rows, L2-resident (per-XCD slice)", src, (size_t)mb << 20, pitch, 3, sink, 1)) return 1;
    if (run<1, 8>("contiguous 1 KB, L2-resident (per-XCD slice)", src, (size_t)mb << 20, pitch, 3, sink, 1)) return 1;
  }
  if (run<0, 8>("patch rows, L2-resident (per-XCD slice), 4 workgroups/CU", src, (size_t)2 << 20, pitch, 4, sink, 1)) return 1;
  if (run<0, 4>("patch rows, L2-resident, 4 loads in flight", src, (size_t)2 << 20, pitch, 3, sink, 1)) return 1;
  if (run<0, 8>("patch rows, 8 MB per XCD (2 x L2: Infinity Cache)", src, (size_t)8 << 20, pitch, 3, sink, 1)) return 1;
  return 0;
}
