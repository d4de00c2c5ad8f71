// DRAFT for round 3 -- NOT part of libdetectorch_hip.so, never run on a GPU yet (the round-2 GPU budget was spent when the
// model below said this is the structure to build).  It compiles against the product headers:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I detectorch_amd/csrc -c tools/r03_draft/roi_align_band.hip
//
// A1  RoIAlign forward, BAND-SWEEP kernel for the FPN box / mask heads (sampling_ratio 2, NCHW maps).
//
// Why (tools/r02b/fill_models.py).  The cluster kernel (roi_align_tile.hip) stages the union patch of ~5
// neighbouring RoIs per workgroup: 9.9 line fills per (RoI, channel) where 2.7 are compulsory, and it sits at the fill rate
// / LDS occupancy that count implies (0.36 ms per 8000-RoI launch).  A workgroup that owns a BAND of feature rows of one
// level and one image, and sweeps it in x with a sliding LDS window, stages every row of the band (+ the halo its windows
// reach below it) once per channel: 4.7 fills per (RoI, channel) with 32-row bands.  The map-stationary C4 kernel
// (roi_align_map.hip) shows what the inner loop costs when the data is simply there: 0.155 ns per (RoI, channel) with 25
// samples per bin, against 0.176 ns for the cluster kernel with 4.
//
// Structure (one workgroup = 1024 threads = 16 wavefronts, one per CU, like roi_align_fwd_map):
//   work item   = (band instance, channel group of 4 NQ channels); a band instance is a run [first, first + count) of the packed
//                 descriptors in visiting order (level, band, x) -- dtc_fpn_collect_distribute's order -- with the same image /
//                 level / band (its bucket table IS the list of runs: export it, or run band_items_kernel below)
//   LDS         = ring image [NQ][rows_cap][kRingCols][4 channels] float32 + a per-wave output slab
//   sweep       = repeat { batch := the longest run of unprocessed RoIs whose windows fit kRingCols columns together;
//                 load the columns the ring does not hold yet; barrier; the wavefronts pool the batch's RoIs one each
//                 (lane <-> bin, 2 x 2 samples x 4 taps x NQ quads, the reference's operation order); barrier }
// Offline statistics on the bench distribution (32-row bands, 64-column ring): P2 bands hold 135 RoIs on average, their
// windows span 46 rows (max 71), a batch holds 20 RoIs; the 6 % of RoIs on P3-P5 form bands of 3-14 RoIs (batches of 3-4):
// they can stay on the cluster kernel (a second launch over their contiguous descriptor range) or ride along here.
//
// TODO before it can ship: (1) run it -- nothing below has executed; (2) rows_cap: bands whose windows reach deeper than the
// LDS image holds (rare: a tall P2 box at the top of a band) must hand those RoIs to the per-output gather; (3) per-RoI axis
// tables from a prep kernel instead of 4 x make_axis per lane and channel group; (4) channel-group-major dispatch + XCD
// slices; (5) the pad-slot trick of the cluster kernel against the transposing-commit bank conflicts; (6) fp16 / bf16 inputs;
// (7) raw buffer loads issued under the pooling of the previous batch (here: 16-byte global loads, four per thread in flight,
// between the two barriers); (8) registers: at the 128-VGPR cap of a
// 1024-thread workgroup this draft spills 92 bytes per lane (-Rpass-analysis=kernel-resource-usage) -- the four make_axis results
// per lane live across the quad loop; tables (3) remove them.
#include <string.h>

#include "roi_align_common.h"

namespace dtc {

constexpr int kBandThreads = 1024;
constexpr int kBandWaves = kBandThreads / 64;
constexpr int kRingCols = 64;                       // columns of the sliding window (power of two, multiple of 4)

struct BandItem { int first, count, b, lvl, row0, pad; };   // RoIs [first, first + count) of roi_desc; first feature row of the band

typedef float bf32x2 __attribute__((ext_vector_type(2)));
typedef float bf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int band_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// (level, band) key of a packed descriptor: the formula of fpn.hip's visiting order
__device__ __forceinline__ int band_key(const RoiAlignParams& p, int ri, int band_log2, int k_min) {
  const RoiRaw w = load_roi_raw(p, ri);
  const int lvl = (int)w.d1.y;
  if (lvl < 0) return 0x7fffffff;                    // padding rows: the visiting order puts them last
  const float yc = fminf(fmaxf((w.d0.z + w.d1.x) * 0.5f, 0.f), 65535.f);
  const int fs = min(k_min + lvl, 15);
  const int band = min(((int)yc >> fs) >> band_log2, 63);
  return ((int)w.d0.x << 12) | (lvl << 6) | band;    // image | level | band
}

// One thread per RoI: a run starts where the key differs from the predecessor's.  items[] is filled in any order.
__global__ void band_items_kernel(RoiAlignParams p, int band_log2, int k_min, BandItem* items, int* n_items, int max_items) {
  const int ri = blockIdx.x * blockDim.x + threadIdx.x;
  if (ri >= p.n_rois) return;
  const int key = band_key(p, ri, band_log2, k_min);
  if (key == 0x7fffffff) return;
  if (ri > 0 && band_key(p, ri - 1, band_log2, k_min) == key) return;
  int n = 1;                                         // run length: a serial walk by the run's first thread (<= ~200 RoIs)
  while (ri + n < p.n_rois && band_key(p, ri + n, band_log2, k_min) == key) n++;
  const int slot = atomicAdd(n_items, 1);
  if (slot >= max_items) return;
  BandItem it;
  it.first = ri; it.count = n; it.b = key >> 12; it.lvl = (key >> 6) & 63; it.pad = 0;
  it.row0 = (key & 63) << band_log2;
  items[slot] = it;
}

template <typename TOut> __device__ __forceinline__ void band_store4(TOut* d, float4 v);
template <> __device__ __forceinline__ void band_store4<float>(float* d, float4 v) { *reinterpret_cast<float4*>(d) = v; }

// window of a RoI in feature pixels of its level (inclusive), sampling_ratio 2: first .lo / last .hi of the sample positions
struct BandWin { int x0, x1, y0, y1; };
__device__ __forceinline__ BandWin band_window(const RoiAlignParams& p, const RoiHead& hd, int H, int W) {
  BandWin w;
  w.y0 = make_axis(hd.sh, hd.bin_h, 0, 0, 2, H).lo;
  w.y1 = make_axis(hd.sh, hd.bin_h, p.pooled_h - 1, 1, 2, H).hi;
  w.x0 = make_axis(hd.sw, hd.bin_w, 0, 0, 2, W).lo;
  w.x1 = make_axis(hd.sw, hd.bin_w, p.pooled_w - 1, 1, 2, W).hi;
  return w;
}

template <typename TOut, int NQ>
__global__ __launch_bounds__(kBandThreads) void roi_align_fwd_band(RoiAlignParams p, const BandItem* __restrict__ items,
                                                                   int rows_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CG = 4 * NQ;
  __shared__ int s_next, s_i1, s_xa, s_xb, s_rmin, s_rmax;
  __shared__ int s_wx0[64], s_wx1[64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ncg = ceil_div(p.channels, CG);
  const int item = blockIdx.x / ncg, cg = blockIdx.x - item * ncg;
  const int c0 = cg * CG;
  const BandItem it = items[item];
  const dtc_feat_level L = p.lv[it.lvl];
  const int H = L.height, W = L.width;
  const int bins = p.pooled_h * p.pooled_w;
  const int plane_bytes = rows_cap * kRingCols * 16;                       // one channel quad of the ring image
  char* ring = reinterpret_cast<char*>(smem);                             // [NQ][rows_cap][kRingCols][4] float32
  float* slab = reinterpret_cast<float*>(smem + NQ * plane_bytes) + (size_t)wv * CG * bins;   // [CG][bins] per wave
  const float* fbase = reinterpret_cast<const float*>(L.data) + (int64_t)it.b * L.stride_n + (int64_t)c0 * L.stride_c;
  float* out = reinterpret_cast<float*>(p.out);
  const float rpw = __frcp_rn((float)p.pooled_w);

  // ---- rows the band's windows reach: [rbase, rbase + rows).  The visiting order bands a RoI by its CENTRE row, so windows
  // stick out above the band's first row as well as below its last one.
  if (tid == 0) { s_rmin = 0x7fffffff; s_rmax = -1; s_next = it.first; }
  __syncthreads();
  for (int i = tid; i < it.count; i += kBandThreads) {
    const BandWin w = band_window(p, load_roi_head(p, it.first + i), H, W);
    atomicMin(&s_rmin, w.y0);
    atomicMax(&s_rmax, w.y1);
  }
  __syncthreads();
  const int rbase = s_rmin;
  const int rows = min(s_rmax - rbase + 1, rows_cap);   // TODO (2): RoIs whose window leaves the image take the gather path
  int res_a = 0, res_b = 0;                          // columns [res_a, res_b) are resident in the ring (uniform)
  int i0 = it.first;
  const int i_end = it.first + it.count;
  while (i0 < i_end) {
    // ---- the batch: the longest run [i0, i1) whose windows fit the ring together (wavefront 0, <= 64 candidates) ----------
    if (wv == 0) {
      const int ri = i0 + lane;
      int x0 = 0x7fffffff, x1 = -1;
      if (ri < i_end) { const BandWin w = band_window(p, load_roi_head(p, ri), H, W); x0 = w.x0 & ~3; x1 = w.x1 | 3; }
      int mn = x0, mx = x1;                          // inclusive prefix min / max over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int a = __shfl_up(mn, o, 64), b = __shfl_up(mx, o, 64);
        if (lane >= o) { mn = min(mn, a); mx = max(mx, b); }
      }
      const bool fits = ri < i_end && mx - mn + 1 <= kRingCols;
      const uint64_t m = __ballot(fits);
      const int n = m == ~0ull ? 64 : __builtin_ctzll(~m);          // leading run of fitting prefixes
      const int nn = max(n, 1);                      // a single window wider than the ring: TODO (2), taken alone for now
      s_wx0[lane] = x0; s_wx1[lane] = x1;
      if (lane == nn - 1) { s_i1 = i0 + nn; s_xa = mn; s_xb = mx; }
    }
    __syncthreads();
    const int i1 = band_uni(s_i1), xa = band_uni(s_xa), xb = min(band_uni(s_xb), (W - 1) | 3);
    // ---- load the columns of [xa, xb] the ring does not hold: 4-pixel pieces, transposed into LDS ---------------------------
    // resident and required intervals are multiples of 4 columns; what stays is their intersection, what is loaded is the
    // (at most two) column runs left and right of it: gA + gB groups of 4 columns
    const int keep_a = max(xa, res_a), keep_b = min(xb + 1, res_b);
    const bool overlap = keep_b > keep_a;
    const int ngx = (xb + 1 - xa) >> 2;
    const int gA = overlap ? (keep_a - xa) >> 2 : ngx, gB = overlap ? (xb + 1 - keep_b) >> 2 : 0;
    const int ng_new = gA + gB;
    const bool vec = L.stride_w == 1 && (W & 3) == 0 && ((L.stride_h | L.stride_c | L.stride_n) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(L.data) & 15) == 0;
    if (ng_new > 0) {
      // a wave-level unit = (channel quad, row, 16 consecutive groups): lane = (channel of the quad, group) -> the 64 loads of a
      // unit are four 256-byte runs; SU units of a wave are in flight together
      constexpr int SU = 4;
      const int n16 = (ng_new + 15) >> 4;
      const int units = NQ * rows * n16;
      const int gl = lane & 15, cl = lane >> 4;
      for (int u0 = wv; u0 < units; u0 += SU * kBandWaves) {
        float4 v[SU];
        int dsto[SU];
#pragma unroll
        for (int k = 0; k < SU; k++) {
          const int u = min(u0 + k * kBandWaves, units - 1);
          const int g16 = u % n16, t = u / n16, row = t % rows, cq = t / rows;
          const int g = min(g16 * 16 + gl, ng_new - 1);                  // lanes past the run repeat its last group
          const int col = g < gA ? xa + 4 * g : keep_b + 4 * (g - gA);
          const int frow = min(rbase + row, H - 1);
          const float* src = fbase + (int64_t)(4 * cq + cl) * L.stride_c + (int64_t)frow * L.stride_h;
          if (vec) v[k] = *reinterpret_cast<const float4*>(src + col);
          else {
            v[k].x = src[(int64_t)min(col, W - 1) * L.stride_w]; v[k].y = src[(int64_t)min(col + 1, W - 1) * L.stride_w];
            v[k].z = src[(int64_t)min(col + 2, W - 1) * L.stride_w]; v[k].w = src[(int64_t)min(col + 3, W - 1) * L.stride_w];
          }
          dsto[k] = cq * plane_bytes + (row * kRingCols + (col & (kRingCols - 1))) * 16 + cl * 4;
        }
#pragma unroll
        for (int k = 0; k < SU; k++) {
          if (u0 + k * kBandWaves < units) {                             // (duplicates of the last group rewrite the same words)
            float* d = reinterpret_cast<float*>(ring + dsto[k]);
            d[0] = v[k].x; d[4] = v[k].y; d[8] = v[k].z; d[12] = v[k].w;
          }
        }
      }
    }
    res_a = xa; res_b = xb + 1;
    if (tid == 0) s_next = i0;
    __syncthreads();
    // ---- pool the batch: wavefronts take RoIs from the shared counter, lane <-> bin -----------------------------------------
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&s_next, 1);
      const int ri = band_uni(t);
      if (ri >= i1) break;
      const RoiHead hd = load_roi_head(p, ri);
      float* orow = out + ((size_t)hd.r * p.channels + c0) * bins;
#pragma unroll 1
      for (int b0 = 0; b0 < bins; b0 += 64) {
        const int bin = min(b0 + lane, bins - 1);
        const bool on = b0 + lane < bins;
        const int ph = (int)(((float)bin + 0.5f) * rpw), pw = bin - ph * p.pooled_w;
        int ylo[2], yhi[2], xlo[2], xhi[2];
        float yl[2], yh[2], xl[2], xh[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const AxisEntry ey = make_axis(hd.sh, hd.bin_h, ph, i, 2, H);
          const AxisEntry ex = make_axis(hd.sw, hd.bin_w, pw, i, 2, W);
          yl[i] = ey.l; yh[i] = ey.h; xl[i] = ex.l; xh[i] = ex.h;
          ylo[i] = min(ey.lo - rbase, rows - 1) * (kRingCols * 16); yhi[i] = min(ey.hi - rbase, rows - 1) * (kRingCols * 16);
          xlo[i] = (ex.lo & (kRingCols - 1)) << 4; xhi[i] = (ex.hi & (kRingCols - 1)) << 4;
        }
        bf32x2 acc[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; q++) { acc[q][0] = bf32x2{0.f, 0.f}; acc[q][1] = bf32x2{0.f, 0.f}; }
        // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
        for (int iy = 0; iy < 2; iy++)
#pragma unroll
          for (int ix = 0; ix < 2; ix++) {
            const float w1 = yh[iy] * xh[ix], w2 = yh[iy] * xl[ix], w3 = yl[iy] * xh[ix], w4 = yl[iy] * xl[ix];   // :95
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              const char* m = ring + q * plane_bytes;
              const bf32x4 v1 = *reinterpret_cast<const bf32x4*>(__builtin_assume_aligned(m + ylo[iy] + xlo[ix], 16));
              const bf32x4 v2 = *reinterpret_cast<const bf32x4*>(__builtin_assume_aligned(m + ylo[iy] + xhi[ix], 16));
              const bf32x4 v3 = *reinterpret_cast<const bf32x4*>(__builtin_assume_aligned(m + yhi[iy] + xlo[ix], 16));
              const bf32x4 v4 = *reinterpret_cast<const bf32x4*>(__builtin_assume_aligned(m + yhi[iy] + xhi[ix], 16));
              acc[q][0] += w1 * v1.lo + w2 * v2.lo + w3 * v3.lo + w4 * v4.lo;                                     // :208-211
              acc[q][1] += w1 * v1.hi + w2 * v2.hi + w3 * v3.hi + w4 * v4.hi;
            }
          }
        // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32
        if (on) {
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            slab[(4 * q + 0) * bins + bin] = acc[q][0].x * 0.25f; slab[(4 * q + 1) * bins + bin] = acc[q][0].y * 0.25f;
            slab[(4 * q + 2) * bins + bin] = acc[q][1].x * 0.25f; slab[(4 * q + 3) * bins + bin] = acc[q][1].y * 0.25f;
          }
        }
      }
      // [CG][bins] is one contiguous run of the [R, C, PH, PW] output (bins <= 64 in this draft): 16-byte stores from the slab
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int n4 = (CG * bins) >> 2;
      for (int i = lane; i < n4; i += 64) band_store4<float>(orow + 4 * i, reinterpret_cast<const float4*>(slab)[i]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                  // every wave is done with the ring before the next batch overwrites columns
    i0 = i1;
  }
}

}  // namespace dtc

// ---- experiment entry (tools/r03_draft/run_band.py): fp32 NCHW maps, packed descriptors, bins <= 64 ----------------------------
// ws: [ int n_items | pad to 16 | BandItem items[max_items] ].  Synchronises once to read the item count (an experiment, not a
// product entry: the product would take the run table from dtc_fpn_collect_distribute and launch without a host round trip).
extern "C" __attribute__((visibility("default"))) int dtc_draft_roi_align_band(
    const dtc_feat_level* levels, int n_levels, int channels, const float* roi_desc, int n_rois, int pooled_h, int pooled_w,
    float* out, void* ws, int max_items, int band_log2, int k_min, int rows_cap, int* n_items_out, void* stream) {
  using namespace dtc;
  if (n_levels < 1 || n_levels > DTC_MAX_LEVELS || pooled_h * pooled_w > 64 || (channels & 7) != 0) return DTC_EUNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  RoiAlignParams p;
  memset(&p, 0, sizeof(p));
  for (int l = 0; l < n_levels; l++) p.lv[l] = levels[l];
  p.rois = nullptr; p.roi_levels = nullptr; p.roi_order = nullptr; p.roi_desc = roi_desc; p.out = out;
  p.n_levels = n_levels; p.channels = channels; p.roi_cols = 5; p.n_rois = n_rois; p.pooled_h = pooled_h; p.pooled_w = pooled_w;
  p.sampling_ratio = 2;
  int* n_items = reinterpret_cast<int*>(ws);
  BandItem* items = reinterpret_cast<BandItem*>(reinterpret_cast<unsigned char*>(ws) + 16);
  if (zero_async(n_items, 16, s) != DTC_OK) return DTC_ELAUNCH;
  hipLaunchKernelGGL(band_items_kernel, dim3((unsigned)((n_rois + 255) / 256)), dim3(256), 0, s, p, band_log2, k_min, items, n_items, max_items);
  int n = 0;
  if (hipMemcpyAsync(&n, n_items, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return DTC_ELAUNCH;
  if (n_items_out) *n_items_out = n;
  if (n > max_items) return DTC_EWORKSPACE;
  if (n == 0) return DTC_OK;
  constexpr int NQ = 2;
  const int bins = pooled_h * pooled_w;
  const size_t lds = (size_t)NQ * rows_cap * kRingCols * 16 + (size_t)kBandWaves * 4 * NQ * bins * 4;
  if (lds > 158 * 1024) return DTC_EUNSUPPORTED;
  static bool raised = false;     // experiment code: single-threaded harness
  if (!raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_band<float, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess) return DTC_ELAUNCH;
    raised = true;
  }
  const int ncg = channels / (4 * NQ);
  hipLaunchKernelGGL((roi_align_fwd_band<float, NQ>), dim3((unsigned)(n * ncg)), dim3(kBandThreads), lds, s, p, items, rows_cap);
  return hipGetLastError() == hipSuccess ? DTC_OK : DTC_ELAUNCH;
}
