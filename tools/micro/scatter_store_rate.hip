// Feasibility probe for a feature-stationary RoIAlign: how fast can the [R, C, 7, 7] output be written
// when a workgroup owns a feature TILE and therefore only FRAGMENTS (runs of RUN consecutive bins) of many RoIs?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/scatter_store_rate tools/micro/scatter_store_rate.hip && tools/micro/scatter_store_rate
// Emulates: 1144 tiles x 4 channel blocks of 64; per tile ~336 items = fragments of RUN bins of random RoIs; lane <-> item,
// wave <-> channel quad of a 16-channel pass; every (roi, c, bin) written exactly once overall = 8000*256*49 dwords (0.4 GB).
// Prints the time per full output for several RUN lengths and for the contiguous slab store the present kernel does.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

constexpr int R = 8000, C = 256, BINS = 49;

__global__ __launch_bounds__(256) void scatter_kernel(const int* __restrict__ item_roi, const int* __restrict__ item_bin,
                                                      const int* __restrict__ tile_off, float* __restrict__ out) {
  const int tile = blockIdx.x, cb = blockIdx.y;                  // channel block of 64
  const int beg = tile_off[tile], end = tile_off[tile + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int pass = 0; pass < 4; pass++) {                         // 4 passes of 16 channels
    const int c0 = cb * 64 + pass * 16 + wave * 4;               // this wave's channel quad
    for (int i = beg + lane; i < end; i += 64) {
      const int r = item_roi[i], b = item_bin[i];
      float* o = out + ((size_t)r * C + c0) * BINS + b;
      const float v = (float)(r + b);
      o[0] = v; o[BINS] = v + 1.f; o[2 * BINS] = v + 2.f; o[3 * BINS] = v + 3.f;
    }
  }
}

__global__ __launch_bounds__(256) void slab_kernel(float* __restrict__ out) {   // what roi_align_fwd_lds does: [32][49] slabs
  const size_t base = ((size_t)blockIdx.x * C + blockIdx.y * 128) * BINS;
  for (int pass = 0; pass < 4; pass++)
    for (int i = threadIdx.x; i < 32 * BINS / 4; i += 256)
      reinterpret_cast<float4*>(out + base + (size_t)pass * 32 * BINS)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  float* out;
  hipMalloc(&out, (size_t)R * C * BINS * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int run : {49, 21, 7, 3, 1}) {
    // every RoI's 49 bins are cut into fragments of `run` bins; fragments are dealt to tiles in a shuffled order
    std::vector<int> fr_roi, fr_start, fr_len;
    for (int r = 0; r < R; r++)
      for (int s = 0; s < BINS; s += run) { fr_roi.push_back(r); fr_start.push_back(s); fr_len.push_back(s + run <= BINS ? run : BINS - s); }
    const int nf = (int)fr_roi.size();
    std::vector<int> perm(nf);
    for (int i = 0; i < nf; i++) perm[i] = i;
    srand(1);
    for (int i = 0; i < nf; i++) {   // local shuffle: fragments of nearby RoIs end up in the same tile (window 64 fragments)
      const int j = i + rand() % (nf - i < 64 ? nf - i : 64);
      const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    const int n_tiles = 1144;
    std::vector<int> ir, ib, toff(n_tiles + 1);
    for (int t = 0; t < n_tiles; t++) {
      toff[t] = (int)ir.size();
      const int f0 = (int)((long long)nf * t / n_tiles), f1 = (int)((long long)nf * (t + 1) / n_tiles);
      for (int f = f0; f < f1; f++)
        for (int k = 0; k < fr_len[perm[f]]; k++) { ir.push_back(fr_roi[perm[f]]); ib.push_back(fr_start[perm[f]] + k); }
    }
    toff[n_tiles] = (int)ir.size();
    int *d_ir, *d_ib, *d_off;
    hipMalloc(&d_ir, ir.size() * 4); hipMalloc(&d_ib, ib.size() * 4); hipMalloc(&d_off, toff.size() * 4);
    hipMemcpy(d_ir, ir.data(), ir.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ib, ib.data(), ib.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_off, toff.data(), toff.size() * 4, hipMemcpyHostToDevice);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(scatter_kernel, dim3(n_tiles, 4), dim3(256), 0, 0, d_ir, d_ib, d_off, out);
    hipEventRecord(e0);
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL(scatter_kernel, dim3(n_tiles, 4), dim3(256), 0, 0, d_ir, d_ib, d_off, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("fragments of %2d bins: %.3f ms per 0.4 GB output (%d items, %.0f per tile)\n", run, ms / 10, (int)ir.size(), ir.size() / (double)n_tiles);
    hipFree(d_ir); hipFree(d_ib); hipFree(d_off);
  }
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(slab_kernel, dim3(R, 2), dim3(256), 0, 0, out);
  hipEventRecord(e0);
  for (int it = 0; it < 10; it++) hipLaunchKernelGGL(slab_kernel, dim3(R, 2), dim3(256), 0, 0, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("contiguous [32][49] slabs (present kernel's store pattern): %.3f ms per 0.4 GB output\n", ms / 10);
  return 0;
}
