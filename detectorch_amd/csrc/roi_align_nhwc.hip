// A1  RoIAlign forward for gfx950 -- channels_last (NHWC) feature maps, sampling_ratio 2, <= 64 bins (the FPN box head when the
// backbone emits channels_last tensors: MIOpen's preferred layout for 16-bit convolutions on MI355X).
//
// Replaces roi_align_forward_kernel (lib/cppcuda/roi_align_forward_cuda.cu:82-159); bit-compatible with the CPU path
// roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:118-219): same float32 operations, same order.
//
// Why a kernel of its own.  With NCHW maps the two things the cluster-stationary kernel (roi_align_tile.hip) cannot get rid of are
// the transposing LDS commit (316 staged pixels per (RoI, channel), four ds_write_b32 per 16-byte piece) and the bank-conflicted
// tap gather (lane <-> (RoI, bin): 9.5 LDS cycles per ds_read_b128 where 4 are conflict-free, tools/r03/lds_taps.py).  With the
// channels innermost both disappear:
//   * a pixel's 256-byte chunk (64 float32 / 128 16-bit channels) is contiguous in memory AND is what the LDS image wants
//     ([pixel][channels]): the window is staged with LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane land lane-linear, four
//     pixels per wave-instruction) -- no staging registers, no commit instructions, every fetched byte is used: 5.3 line fills
//     per (RoI, channel) where the cluster kernel needs 9.9;
//   * lane <-> (16-byte channel chunk q, bin slot): the 16 lanes of a bin slot read the 16 consecutive 16-byte slots of ONE pixel,
//     and the lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) take complementary chunk ranges from two pixels, so every
//     16-lane group covers the 16 bank-slots exactly once whatever the pixels are: 4 LDS cycles per tap read, by construction;
//   * a bin's 16 tap offsets and 8 axis weights are the same for its 16 lanes: formed once per workgroup into an LDS table
//     (thread <-> bin), fetched with six broadcast reads.
// workgroup = (RoI, 256-byte channel block); a window that does not fit the LDS image is pooled in STRIPS of bin rows; a RoI whose
// single bin row does not fit (or a level whose strides are not the plain channels_last ones) takes the direct-gather kernel
// roi_align_fwd_nhwc's path per output.  Results leave through an LDS slab [channels][bins] as 16-byte stores.
#include <stdlib.h>

#include <mutex>

#include "roi_align_common.h"

namespace dtc {

constexpr int kNlThreads = 256;
constexpr int kNlChunk = 256;                  // bytes of one pixel's channel block in the LDS image
constexpr int kNlMaxBins = 64;
constexpr int kNlBinRec = 96;                  // bytes per bin record: 16 tap offsets (uint32) + y/x weights (8 floats)

typedef float nf32x2 __attribute__((ext_vector_type(2)));
typedef float nf32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t nu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void nl_lds_void;
typedef __attribute__((address_space(1))) const void nl_glb_void;

__device__ __forceinline__ int nl_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// 16 bytes of the LDS image -> the lane's channels as float32 (4 for float32 maps, 8 for 16-bit maps)
template <typename TIn> struct NlLane;
template <> struct NlLane<float> {
  static constexpr int kCh = 4;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
};
template <> struct NlLane<__half> {
  static constexpr int kCh = 8;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
      v[2 * i] = __low2float(h); v[2 * i + 1] = __high2float(h);
    }
  }
};
template <> struct NlLane<bf16_t> {
  static constexpr int kCh = 8;
  static __device__ __forceinline__ void widen(const nu32x4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};

template <typename TOut> __device__ __forceinline__ void nl_store4(TOut* d, float4 v);
template <> __device__ __forceinline__ void nl_store4<float>(float* d, float4 v) { store_stream16(d, v); }     // streaming stores: dtc_common.h
template <> __device__ __forceinline__ void nl_store4<__half>(__half* d, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  store_stream8(d, *reinterpret_cast<const uint32_t*>(&a), *reinterpret_cast<const uint32_t*>(&b));
}
template <> __device__ __forceinline__ void nl_store4<bf16_t>(bf16_t* d, float4 v) {
  store_stream8(d, (uint32_t)from_f32<bf16_t>(v.x).bits | ((uint32_t)from_f32<bf16_t>(v.y).bits << 16),
                (uint32_t)from_f32<bf16_t>(v.z).bits | ((uint32_t)from_f32<bf16_t>(v.w).bits << 16));
}


template <typename TIn, typename TOut>
__global__ __launch_bounds__(kNlThreads) void roi_align_fwd_nhwc_lds(RoiAlignParams p, int img_pixels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CL = NlLane<TIn>::kCh;                 // channels per lane
  constexpr int CB = 16 * CL;                           // channels per workgroup: one 256-byte chunk per pixel
  const int bins = p.pooled_h * p.pooled_w;
  // [axis samples: 2 x 32 AxisEntry][bin records: bins x 96 B][slab: CB x bins float32][image: img_pixels x 256 B]
  AxisEntry* ytab = reinterpret_cast<AxisEntry*>(smem);
  AxisEntry* xtab = ytab + 32;
  unsigned char* brec = smem + 1024;
  float* slab = reinterpret_cast<float*>(smem + 1024 + kNlMaxBins * kNlBinRec);
  unsigned char* img = smem + 1024 + kNlMaxBins * kNlBinRec + (size_t)CB * bins * 4;
  const uint32_t img32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)img;
  const int tid = threadIdx.x, lane = tid & 63, wv = nl_uni(tid >> 6);
  const int nct = p.channels / CB;
  const int wi = xcd_work_item(blockIdx.x, gridDim.x, p.xcd_remap);
  const int ri = wi / nct;
  const int c0 = (wi - ri * nct) * CB;
  const RoiHead hd = load_roi_head(p, ri);
  TOut* out = reinterpret_cast<TOut*>(p.out) + ((size_t)hd.r * p.channels + c0) * bins;
  if (hd.lvl < 0 || hd.lvl >= p.n_levels) {            // padding row of a fixed-shape batch: defined output
    for (int o = tid; o < CB * bins; o += kNlThreads) out[o] = from_f32<TOut>(0.f);
    return;
  }
  const dtc_feat_level L = p.lv[hd.lvl];
  const int H = L.height, W = L.width;
  // ---- axis samples (roi_align_cpu_loop.cpp:36-95), one thread per sample ------------------------------------------------------
  if (tid < 2 * p.pooled_h) ytab[tid] = make_axis(hd.sh, hd.bin_h, tid >> 1, tid & 1, 2, H);
  else if (tid >= 64 && tid < 64 + 2 * p.pooled_w) xtab[tid - 64] = make_axis(hd.sw, hd.bin_w, (tid - 64) >> 1, (tid - 64) & 1, 2, W);
  __syncthreads();
  const int x0 = xtab[0].lo, x1 = xtab[2 * p.pooled_w - 1].hi;        // sample positions are non-decreasing: first .lo / last .hi
  const int w = x1 - x0 + 1;
  const TIn* fbase = reinterpret_cast<const TIn*>(L.data) + (int64_t)hd.b * L.stride_n + c0;
  // plain channels_last strides and 16-byte alignment: what the LDS-DMA staging needs
  const bool dma_ok = L.stride_c == 1 && ((L.stride_w * (int64_t)sizeof(TIn)) & 15) == 0 && ((L.stride_h * (int64_t)sizeof(TIn)) & 15) == 0 &&
                      ((L.stride_n * (int64_t)sizeof(TIn)) & 15) == 0 && (reinterpret_cast<uintptr_t>(L.data) & 15) == 0;
  const int q = lane & 15;                               // this lane's 16-byte chunk of a pixel
  const int slot = (tid >> 4);                           // bin slot 0..15 of the workgroup

  int pa = 0;
  while (pa < p.pooled_h) {
    // ---- the strip: the longest run of bin rows [pa, pb) whose window fits the LDS image (uniform) -------------------------------
    const int ys = ytab[2 * pa].lo;
    int pb = pa, hs = 0;
    while (pb < p.pooled_h) {
      const int h2 = ytab[2 * pb + 1].hi - ys + 1;
      if (h2 * w > img_pixels) break;
      hs = h2; pb++;
    }
    const bool staged = dma_ok && pb > pa;
    if (!staged) pb = pa + 1;                           // not even one bin row fits (or odd strides): this bin row straight from global
    const int nb = (pb - pa) * p.pooled_w;              // bins of the strip
    const int b0 = pa * p.pooled_w;
    if (staged) {
      // ---- stage the window [ys, ys + hs) x [x0, x1] with LDS-DMA: a wave-instruction moves 4 pixels x 256 B, lane-linear ------
      const int np = hs * w;
      const float rw = 1.0f / (float)w;
      for (int k = wv; 4 * k < np; k += kNlThreads / 64) {
        const int pi = min(4 * k + (lane >> 4), np - 1);               // pixels past the window repeat its last one
        const int row = (int)(((float)pi + 0.5f) * rw);                  // exact: pi < 2^13
        const int col = pi - row * w;
        const TIn* g = fbase + (int64_t)(ys + row) * L.stride_h + (int64_t)(x0 + col) * L.stride_w;
        __builtin_amdgcn_global_load_lds((nl_glb_void*)(reinterpret_cast<const char*>(g) + q * 16),
                                         (nl_lds_void*)(img + (size_t)k * 1024), 16, 0, 0);
      }
    }
    // ---- bin records of the strip: thread <-> bin ---------------------------------------------------------------------------------
    if (tid < nb) {
      const int bl = tid, ph = pa + bl / p.pooled_w, pw = bl - (bl / p.pooled_w) * p.pooled_w;
      uint32_t* o32 = reinterpret_cast<uint32_t*>(brec + bl * kNlBinRec);
      float* wts = reinterpret_cast<float*>(brec + bl * kNlBinRec + 64);
#pragma unroll
      for (int iy = 0; iy < 2; iy++) {
        const AxisEntry ey = ytab[2 * ph + iy];
#pragma unroll
        for (int ix = 0; ix < 2; ix++) {
          const AxisEntry ex = xtab[2 * pw + ix];
          uint32_t t0, t1, t2, t3;
          if (staged) {   // byte offsets inside the LDS image
            t0 = (uint32_t)(((ey.lo - ys) * w + (ex.lo - x0)) * kNlChunk); t1 = (uint32_t)(((ey.lo - ys) * w + (ex.hi - x0)) * kNlChunk);
            t2 = (uint32_t)(((ey.hi - ys) * w + (ex.lo - x0)) * kNlChunk); t3 = (uint32_t)(((ey.hi - ys) * w + (ex.hi - x0)) * kNlChunk);
          } else {        // (row, column) packed: the direct path forms global addresses from them
            t0 = (uint32_t)ey.lo << 16 | (uint32_t)ex.lo; t1 = (uint32_t)ey.lo << 16 | (uint32_t)ex.hi;
            t2 = (uint32_t)ey.hi << 16 | (uint32_t)ex.lo; t3 = (uint32_t)ey.hi << 16 | (uint32_t)ex.hi;
          }
          o32[(iy * 2 + ix) * 4 + 0] = t0; o32[(iy * 2 + ix) * 4 + 1] = t1; o32[(iy * 2 + ix) * 4 + 2] = t2; o32[(iy * 2 + ix) * 4 + 3] = t3;
        }
        wts[iy * 2] = ey.l; wts[iy * 2 + 1] = ey.h;
      }
#pragma unroll
      for (int ix = 0; ix < 2; ix++) { const AxisEntry ex = xtab[2 * pw + ix]; wts[4 + ix * 2] = ex.l; wts[4 + ix * 2 + 1] = ex.h; }
    }
    __syncthreads();        // (an LDS-DMA in flight makes this fence wait vmcnt(0): the image is complete behind it)
    // ---- pool: 16 bins per pass of the workgroup, lane <-> (bin slot, 16-byte channel chunk) -----------------------------------------
    for (int bl0 = 0; bl0 < nb; bl0 += kNlThreads / 16) {
      const int bl = bl0 + slot;
      if (bl < nb) {
        const nu32x4* o4 = reinterpret_cast<const nu32x4*>(brec + bl * kNlBinRec);
        const nf32x4 wy = *reinterpret_cast<const nf32x4*>(brec + bl * kNlBinRec + 64);      // yl0 yh0 yl1 yh1
        const nf32x4 wx = *reinterpret_cast<const nf32x4*>(brec + bl * kNlBinRec + 80);      // xl0 xh0 xl1 xh1
        const float yl[2] = {wy.x, wy.z}, yh[2] = {wy.y, wy.w}, xl[2] = {wx.x, wx.z}, xh[2] = {wx.y, wx.w};
        float acc[CL];
#pragma unroll
        for (int c = 0; c < CL; c++) acc[c] = 0.f;
        // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
        for (int iy = 0; iy < 2; iy++)
#pragma unroll
          for (int ix = 0; ix < 2; ix++) {
            const nu32x4 off = o4[iy * 2 + ix];
            nu32x4 r1, r2, r3, r4;
            if (staged) {
              r1 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.x + q * 16);
              r2 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.y + q * 16);
              r3 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.z + q * 16);
              r4 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(img32 + off.w + q * 16);
            } else {
              const char* gb = reinterpret_cast<const char*>(fbase) + q * 16;
              auto ga = [&](uint32_t t) { return gb + ((int64_t)(t >> 16) * L.stride_h + (int64_t)(t & 0xffff) * L.stride_w) * (int64_t)sizeof(TIn); };
              if (dma_ok) {
                r1 = *reinterpret_cast<const nu32x4*>(ga(off.x)); r2 = *reinterpret_cast<const nu32x4*>(ga(off.y));
                r3 = *reinterpret_cast<const nu32x4*>(ga(off.z)); r4 = *reinterpret_cast<const nu32x4*>(ga(off.w));
              } else {      // unaligned chunks: 4-byte (2-byte) loads
                const uint32_t* a1 = reinterpret_cast<const uint32_t*>(ga(off.x)); const uint32_t* a2 = reinterpret_cast<const uint32_t*>(ga(off.y));
                const uint32_t* a3 = reinterpret_cast<const uint32_t*>(ga(off.z)); const uint32_t* a4 = reinterpret_cast<const uint32_t*>(ga(off.w));
                r1 = nu32x4{a1[0], a1[1], a1[2], a1[3]}; r2 = nu32x4{a2[0], a2[1], a2[2], a2[3]};
                r3 = nu32x4{a3[0], a3[1], a3[2], a3[3]}; r4 = nu32x4{a4[0], a4[1], a4[2], a4[3]};
              }
            }
            float v1[CL], v2[CL], v3[CL], v4[CL];
            NlLane<TIn>::widen(r1, v1); NlLane<TIn>::widen(r2, v2); NlLane<TIn>::widen(r3, v3); NlLane<TIn>::widen(r4, v4);
            const float w1 = yh[iy] * xh[ix], w2 = yh[iy] * xl[ix], w3 = yl[iy] * xh[ix], w4 = yl[iy] * xl[ix];   // :95
#pragma unroll
            for (int c = 0; c < CL; c++) acc[c] += w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];               // :208-211
          }
        // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32
        float* so = slab + (size_t)(q * CL) * bins + (b0 + bl);
#pragma unroll
        for (int c = 0; c < CL; c++) so[c * bins] = acc[c] * 0.25f;
      }
    }
    __syncthreads();        // every tap of the strip is read before the next strip's DMA (or the slab stores) go on
    pa = pb;
  }
  // ---- slab [CB][bins] is one contiguous run of the [R,C,PH,PW] output: 16-byte stores ------------------------------------------------
  const int n4 = (CB * bins) >> 2;
  if (((CB * bins) & 3) == 0) {
    for (int i = tid; i < n4; i += kNlThreads) nl_store4<TOut>(out + 4 * i, reinterpret_cast<const float4*>(slab)[i]);
  } else {
    for (int i = tid; i < CB * bins; i += kNlThreads) out[i] = from_f32<TOut>(slab[i]);
  }
}

// ======================================================================================================================================
// Round 4: the same arithmetic as a PERSISTENT, software-pipelined, wave-specialised kernel (float32 output).
//
// What bounded roi_align_fwd_nhwc_lds (counters: profiles/r04_a_*): not its fills (1.36 GB per box-head launch, 71 % of them L2 hits,
// fabric traffic = the compulsory 0.72 + 0.40 GB) and not the LDS or the VALU -- the serial chain of a (RoI, channel block) item:
// descriptor -> axis samples -> barrier -> DMA of the window -> barrier -> pool -> barrier -> stores, of which a tenth is work, times
// the 31 items a workgroup slot runs through.  Here a workgroup (512 threads, two per CU) owns a static, XCD-local, interleaved list
// of items and works through it as a pipeline of UNITS (an item, or a strip of its bin rows when the window does not fit an image):
//
//   wave 0 = the PLANNER, two units ahead: it alone forms an item's geometry (descriptor through the scalar cache, the 28 axis
//            samples in its lanes, strip extents by v_readlane) and publishes axis tables + a 48-byte unit descriptor in LDS;
//   waves 1-7 = the POOLERS.  At the top of iteration k they put the window of unit k+1 in flight -- LDS-DMA into the other of two
//            images, the instructions dealt round-robin over the seven waves -- and then pool unit k: 28 bin slots x 16-byte channel
//            chunks, slot <-> consecutive bins of the unit.  A wave's four slots are four CONSECUTIVE bins, i.e. 16 contiguous output
//            bytes per channel: the wave transposes its 4 bins x 64 channels through 1.25 KB of LDS of its own and stores them at
//            once -- no workgroup-wide output slab, no store phase, no second barrier.
//   One barrier per unit (T: every pooler has waited for its own DMA instructions; descriptors, tables and image are visible).
//
// The DMA instructions are inline assembly (global_load_lds_dwordx4 with an SGPR base): hipcc does not count them, so neither
// __syncthreads() nor a later LDS read drains them early (cdna_hip_programming.md 5.7); the only wait is the one in front of T.
// What the first versions taught (commit history; profiles/r04_c_*): with every wave planning and issuing, the uniform work cost as
// much as the pooling; with the planner issuing all the DMA it was the bottleneck; a workgroup-wide slab cost a store phase and a
// barrier per item and 12 KB of image.  Serves every bin count (7 x 7 and 14 x 14 alike).
constexpr int kNpThreads = 512;
constexpr int kNpPoolWaves = kNpThreads / 64 - 1;          // 7
constexpr int kNpPoolSlots = kNpPoolWaves * 4;             // 28 bins per round
constexpr int kNpTabEntries = 32;              // axis samples per axis (2 x pooled <= 32)
struct NpTab { int off_lo, off_hi; float l, h; };          // byte offsets inside the item's window image, weights
struct NpRc { int lo, hi; };                               // feature row / column (the straight-from-global path)
struct NpDesc {                                            // what the poolers need to know about a unit (wave-uniform), 48 bytes
  int flags;                                               // bit 0 valid, 1 staged, 2 last unit of its item, 3-4 mode, 5-6 table slot
  int rows;                                                // pa | pb << 8
  int strip_off;                                           // bytes of the item's window image above the strip
  int wh;                                                  // window width | rows of the strip << 16
  int shb, swb;                                            // row / pixel stride in bytes
  uint32_t out_lo, out_hi, wb_lo, wb_hi;                   // output pointer of (row, channel block); address of the strip's first pixel
  int pad[2];
};
static_assert(sizeof(NpDesc) == 48, "NpDesc layout");
struct NpGlb { const void* fbase; int64_t sh, sw; };       // straight-from-global path only: image + channel block base, strides
constexpr int kNpSlots = 3;                                 // the planner runs two units ahead of the pooling: three descriptor / table slots
constexpr int kNpTabBytes = kNpSlots * 2 * kNpTabEntries * (int)(sizeof(NpTab) + sizeof(NpRc));     // 3 slots x (y, x): 4.5 KB
constexpr int kNpHdrBytes = kNpTabBytes + kNpSlots * (int)sizeof(NpDesc) + kNpSlots * 32;             // + descriptors + NpGlb (24 -> 32 B)
constexpr int kNpScrPitch = 5;                              // dwords per channel of a wave's transposition scratch (4 bins + 1: conflict-free)

__device__ __forceinline__ float np_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int np_rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// 16 bytes per lane, lane-linear into LDS at lds_dst (wave-uniform); source = wave-uniform 64-bit base (SGPR pair) + 32-bit lane
// offset: no 64-bit vector arithmetic per instruction.  Not counted by the compiler.
__device__ __forceinline__ void np_glds16s(uint32_t voff, const void* sbase_, uint32_t lds_dst_) {
  // (readfirstlane: free when the value already lives in SGPRs, and makes the operands provably uniform when it is loop-carried)
  const uint64_t sb = reinterpret_cast<uint64_t>(sbase_);
  const void* sbase = reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32)) << 32) |
                                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb));
  const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

struct __attribute__((packed, aligned(4))) NpF4 { float x, y, z, w; };     // 16-byte store at 4-byte alignment (global_store_dwordx4)

// Development aid (-DDTC_NP_TRACE, tools/r04/np_trace.py): cycle counter of the planner's lane 0 and of the first pooler wave's
// lane 0 at their phase boundaries, summed per phase.
#ifdef DTC_NP_TRACE
constexpr int kNpTraceSlots = 16;
__device__ unsigned long long g_np_trace[2 * kNpTraceSlots * 4096];
struct NpTrace {
  unsigned long long last, acc[kNpTraceSlots];
  __device__ __forceinline__ void start() { for (int i = 0; i < kNpTraceSlots; i++) acc[i] = 0; last = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += n - last; last = n; }
};
#define NP_MARK(i) tt.mark(i)
#else
#define NP_MARK(i) ((void)0)
#endif

template <typename TIn, bool DESC>      // DESC: packed descriptors (p.roi_desc) -- the only vector-memory reads are then the DMA
__global__ __launch_bounds__(kNpThreads, 4) void roi_align_fwd_nhwc_pipe(RoiAlignParams p, int img_pixels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CL = NlLane<TIn>::kCh;                 // channels per lane
  constexpr int CB = 16 * CL;                           // channels per item: one 256-byte chunk per pixel
  const int PH = p.pooled_h, PW = p.pooled_w, bins = PH * PW;
  // [axis tables: 3 slots][unit descriptors x 3][item globals x 3][transposition scratch: 7 waves x CB x 5 dwords][image 0][image 1]
  NpTab* tab = reinterpret_cast<NpTab*>(smem);                                             // [slot][y|x][32]
  NpRc* rct = reinterpret_cast<NpRc*>(smem + kNpSlots * 2 * kNpTabEntries * sizeof(NpTab));      // [slot][y|x][32]
  NpDesc* ud = reinterpret_cast<NpDesc*>(smem + kNpTabBytes);
  NpGlb* ig = reinterpret_cast<NpGlb*>(smem + kNpTabBytes + kNpSlots * sizeof(NpDesc));          // [slot], 32-byte slots
  float* scr0 = reinterpret_cast<float*>(smem + kNpHdrBytes);
  unsigned char* img0 = smem + kNpHdrBytes + (size_t)kNpPoolWaves * CB * kNpScrPitch * 4;
  const uint32_t img32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)img0;
  const uint32_t img_bytes = (uint32_t)img_pixels * kNlChunk;
  const int tid = threadIdx.x, lane = tid & 63, wv = nl_uni(tid >> 6);
  const int q = lane & 15;
  const int nct = p.channels / CB;
  // ---- this workgroup's items: XCD x (= block % 8) owns a contiguous slice of the (RoI, channel block) items in visiting order; its
  // workgroups take them interleaved, so the workgroups resident on an XCD pool neighbouring RoIs at the same time (shared lines: L2)
  const int n_items = p.n_rois * nct;
  int first_item, item_end, step;
  if (p.xcd_remap && (gridDim.x % kXcds) == 0) {
    const int x = blockIdx.x % kXcds, j = blockIdx.x / kXcds;
    const int qn = n_items / kXcds, rn = n_items - qn * kXcds;
    const int start = x * qn + min(x, rn);
    first_item = start + j; item_end = start + qn + (x < rn ? 1 : 0); step = gridDim.x / kXcds;
  } else {
    first_item = blockIdx.x; item_end = n_items; step = gridDim.x;
  }
  if (first_item >= item_end) return;
#ifdef DTC_NP_TRACE
  NpTrace tt; tt.start();
  unsigned long long n_units = 0, n_items_done = 0, n_dma = 0;
#endif

  if (wv == 0) {
    // =================================================== the planner =========================================================
    __builtin_amdgcn_s_setprio(3);          // one wave ahead of seven: its instructions go first
    // descriptors through the scalar cache (constant address space, uniform index: s_load) when they are packed
    auto load_raw = [&](int ri) {
      if (!DESC) return load_roi_raw(p, ri);
      typedef __attribute__((address_space(4))) const float cfl;
      const cfl* d = reinterpret_cast<cfl*>(reinterpret_cast<uintptr_t>(p.roi_desc)) + (size_t)nl_uni(ri) * 8;
      RoiRaw w;
      w.d0 = make_float4(d[0], d[1], d[2], d[3]); w.d1 = make_float4(d[4], d[5], d[6], d[7]);
      return w;
    };
    RoiRaw raw_pre = load_raw(first_item / nct);      // descriptor of the next item to set up, fetched one item ahead
    // the item being planned (all uniform)
    int i_mode = 2, i_par = 0, i_x0 = 0, i_w = 1, i_y0 = 0;
    const TIn* i_fbase = nullptr; int64_t i_sh = 0, i_sw = 0; float* i_out = nullptr;
    int ylo_r = 0, yhi_r = 0;       // its y samples in lanes 0 .. 2 PH - 1
    int next_par = 0;
    auto setup_item = [&](int wi) {
      const int ri = wi / nct;
      const int c0 = (wi - ri * nct) * CB;
      const RoiHead hd = roi_head_from_raw(p, raw_pre);
      if (wi + step < item_end) raw_pre = load_raw((wi + step) / nct);
      const int lvl = nl_uni(hd.lvl);
      i_par = next_par; next_par = next_par == kNpSlots - 1 ? 0 : next_par + 1;
      i_out = reinterpret_cast<float*>(p.out) + ((size_t)nl_uni(hd.r) * p.channels + c0) * bins;
      if (lvl < 0 || lvl >= p.n_levels) { i_mode = 2; i_x0 = i_y0 = 0; i_w = 1; i_fbase = nullptr; i_sh = i_sw = 0; return; }
      const dtc_feat_level L = p.lv[lvl];
      const float sh = np_unif(hd.sh), sw = np_unif(hd.sw), bh = np_unif(hd.bin_h), bw = np_unif(hd.bin_w);
      // lanes 0 .. 2 PH - 1: y samples; lanes 32 .. 32 + 2 PW - 1: x samples (roi_align_cpu_loop.cpp:36-95)
      const bool isx = lane >= 32;
      const int si = isx ? lane - 32 : lane;
      const int LH = nl_uni(L.height), LW = nl_uni(L.width);      // (opaque: a select of the two FIELDS becomes a vector load of a selected address)
      const AxisEntry e = make_axis(isx ? sw : sh, isx ? bw : bh, si >> 1, si & 1, 2, isx ? LW : LH);
      ylo_r = e.lo; yhi_r = e.hi;
      i_x0 = np_rl(e.lo, 32); i_y0 = np_rl(e.lo, 0);      // sample positions are non-decreasing: first .lo / last .hi
      const int x1 = np_rl(e.hi, nl_uni(32 + 2 * PW - 1));
      i_w = x1 - i_x0 + 1;
      i_fbase = reinterpret_cast<const TIn*>(L.data) + (int64_t)nl_uni(hd.b) * L.stride_n + c0;
      i_sh = L.stride_h; i_sw = L.stride_w;
      // plain channels_last strides and 16-byte alignment: what the LDS-DMA staging needs
      const bool dma_ok = L.stride_c == 1 && ((L.stride_w * (int64_t)sizeof(TIn)) & 15) == 0 && ((L.stride_h * (int64_t)sizeof(TIn)) & 15) == 0 &&
                          ((L.stride_n * (int64_t)sizeof(TIn)) & 15) == 0 && (reinterpret_cast<uintptr_t>(L.data) & 15) == 0 &&
                          L.stride_h > 0 && L.stride_w > 0 && (L.stride_h + L.stride_w) * (int64_t)sizeof(TIn) * (img_pixels + 4) < (1ll << 31);   // 32-bit lane offsets
      i_mode = dma_ok ? 0 : 1;
      if (si < 2 * (isx ? PW : PH)) {
        const int o = (i_par * 2 + (isx ? 1 : 0)) * kNpTabEntries + si;
        NpTab t;
        if (isx) { t.off_lo = (e.lo - i_x0) * kNlChunk; t.off_hi = (e.hi - i_x0) * kNlChunk; }
        else { t.off_lo = (e.lo - i_y0) * i_w * kNlChunk; t.off_hi = (e.hi - i_y0) * i_w * kNlChunk; }
        t.l = e.l; t.h = e.h;
        tab[o] = t;
        NpRc rc; rc.lo = e.lo; rc.hi = e.hi;
        rct[o] = rc;
      }
      if (lane == 0) { NpGlb g; g.fbase = i_fbase; g.sh = i_sh; g.sw = i_sw; *reinterpret_cast<NpGlb*>(reinterpret_cast<char*>(ig) + i_par * 32) = g; }
    };
    // plan the strip of the current item that starts at bin row pa -- the longest run of bin rows whose window fits an image -- and
    // publish its descriptor in slot `ds`
    auto plan_unit = [&](int pa, int ds) {
      int pb, ys = 0, hs = 0, staged = 0;
      if (i_mode == 2) {
        pb = PH;
      } else {
        ys = np_rl(ylo_r, nl_uni(2 * pa));
        pb = pa;
        while (pb < PH) {
          const int h2 = np_rl(yhi_r, nl_uni(2 * pb + 1)) - ys + 1;
          if (h2 * i_w > img_pixels) break;
          hs = h2; pb++;
        }
        staged = i_mode == 0 && pb > pa;
        if (!staged) { pb = pa + 1; hs = 0; }               // not even one bin row fits (or odd strides): this bin row straight from global
      }
      if (lane == 0) {
        NpDesc d;
        d.flags = 1 | (staged << 1) | ((pb >= PH ? 1 : 0) << 2) | (i_mode << 3) | (i_par << 5);
        d.rows = pa | (pb << 8);
        d.strip_off = (ys - i_y0) * i_w * kNlChunk; d.wh = i_w | (hs << 16);
        d.shb = (int)(i_sh * (int64_t)sizeof(TIn)); d.swb = (int)(i_sw * (int64_t)sizeof(TIn));
        const uint64_t o = reinterpret_cast<uint64_t>(i_out);
        d.out_lo = (uint32_t)o; d.out_hi = (uint32_t)(o >> 32);
        const uint64_t wb = reinterpret_cast<uint64_t>(i_fbase + (int64_t)ys * i_sh + (int64_t)i_x0 * i_sw);
        d.wb_lo = (uint32_t)wb; d.wb_hi = (uint32_t)(wb >> 32);
        d.pad[0] = d.pad[1] = 0;
        ud[ds] = d;
      }
#ifdef DTC_NP_TRACE
      n_dma += (unsigned long long)(hs * i_w);
#endif
      return pb;
    };
    // the unit after the one just planned: next strip, or first strip of the next item, or none
    int next_item = first_item, cur_pb = PH, more = 1;
    auto plan_next = [&](int ds) {
      if (!more) return;
      if (cur_pb < PH) {
        cur_pb = plan_unit(cur_pb, ds);
      } else if (next_item < item_end) {
        setup_item(next_item); next_item += step;
        cur_pb = plan_unit(0, ds);
#ifdef DTC_NP_TRACE
        n_items_done++;
#endif
      } else {
        if (lane == 0) ud[ds].flags = 0;
        more = 0;
      }
    };
    // ---- prologue: units 0 and 1 are planned before the first barrier (P0), the poolers issue unit 0's DMA between P0 and T(0)
    plan_next(0);
    plan_next(1);
    __syncthreads();                                        // P0
    int k = 0;
    while (true) {
      NP_MARK(0);
      __syncthreads();                                        // T(k)
      NP_MARK(2);
      if (!(nl_uni(ud[k % kNpSlots].flags) & 1)) break;
      plan_next((k + 2) % kNpSlots);                         // unit k+2 (the slot of unit k-1, pooled before T(k))
      NP_MARK(4);
      k++;
#ifdef DTC_NP_TRACE
      n_units++;
#endif
    }
#ifdef DTC_NP_TRACE
    if (lane == 0 && blockIdx.x < 4096) {
      unsigned long long* o = g_np_trace + (size_t)blockIdx.x * 2 * kNpTraceSlots;
      for (int i = 0; i < 8; i++) o[i] = tt.acc[i];
      o[8] = n_units; o[9] = n_items_done; o[10] = n_dma;
    }
#endif
    return;
  }

  // ======================================================= the poolers ===========================================================
  const int ptid = tid - 64;                            // 0 .. 447
  const int pwv = wv - 1;                               // 0 .. 6
  const int slot = ptid >> 4;                           // 0 .. 27: this lane's bin of a round
  const int sl = lane >> 4;                             // 0 .. 3: its bin among the wave's four
  float* scr = scr0 + (size_t)pwv * CB * kNpScrPitch;   // this wave's transposition scratch [CB channels][4 bins + 1]
  const float rpwf = 1.0f / (float)PW;
  // a descriptor as the poolers use it: every field in SGPRs
  struct NpU { int flags, pa, pb, strip_off, w, hs, shb, swb; uint32_t out_lo, out_hi, wb_lo, wb_hi; };
  auto read_desc = [&](int slot_i) {
    const NpDesc d = ud[slot_i];
    NpU u;
    u.flags = nl_uni(d.flags);
    const int rows = nl_uni(d.rows), wh = nl_uni(d.wh);
    u.pa = rows & 255; u.pb = rows >> 8; u.strip_off = nl_uni(d.strip_off); u.w = wh & 0xffff; u.hs = wh >> 16;
    u.shb = nl_uni(d.shb); u.swb = nl_uni(d.swb);
    u.out_lo = (uint32_t)nl_uni((int)d.out_lo); u.out_hi = (uint32_t)nl_uni((int)d.out_hi);
    u.wb_lo = (uint32_t)nl_uni((int)d.wb_lo); u.wb_hi = (uint32_t)nl_uni((int)d.wb_hi);
    return u;
  };
  // LDS-DMA of a unit's window [ys, ys + hs) x [x0, x0 + w) into image `buf`: a wave-instruction moves 4 pixels x 256 B, lane-linear;
  // pixel pi = 4 kk + (lane >> 4) of the window, row-major.  The seven pooler waves take the instructions round-robin: a wave's next
  // instruction is 28 pixels on -- a constant byte step with one conditional row wrap, 5 VALU instructions per DMA instruction.
  auto issue_dma = [&](const NpU& dn, int buf) {
    const int w = dn.w, np = dn.hs * w;
    const uint32_t shb = (uint32_t)dn.shb, swb = (uint32_t)dn.swb;
    const char* wbase = reinterpret_cast<const char*>(((uint64_t)dn.wb_hi << 32) | (uint64_t)dn.wb_lo);
    const uint32_t dst0 = img32 + (uint32_t)buf * img_bytes;
    const float rw = 1.0f / (float)w;
    const int nk = (np + 3) >> 2;
    if (pwv >= nk) return;
    const int p0 = 4 * pwv + sl;
    int row = (int)(((float)p0 + 0.5f) * rw);                          // exact: p0 < 2^13
    int col = p0 - row * w;
    uint32_t off = (uint32_t)row * shb + (uint32_t)col * swb + (uint32_t)q * 16u;
    const int drow = nl_uni((int)(((float)(4 * kNpPoolWaves) + 0.5f) * rw)), dcol = 4 * kNpPoolWaves - drow * w;      // 28 = drow * w + dcol
    const uint32_t dstep = (uint32_t)drow * shb + (uint32_t)dcol * swb, wrapfix = shb - (uint32_t)w * swb;
    int kk = pwv;
    for (; kk + kNpPoolWaves < nk; kk += kNpPoolWaves) {               // every group but possibly the window's last one is whole
      np_glds16s(off, wbase, dst0 + (uint32_t)kk * 1024u);
      col += dcol; off += dstep;
      if (col >= w) { col -= w; off += wrapfix; }
    }
    if (kk == nk - 1) {   // the window's last group: pixels past the window repeat its last one
      const int pi = min(4 * kk + sl, np - 1);
      const int r2 = (int)(((float)pi + 0.5f) * rw);
      const int c2 = pi - r2 * w;
      off = (uint32_t)r2 * shb + (uint32_t)c2 * swb + (uint32_t)q * 16u;
    }
    np_glds16s(off, wbase, dst0 + (uint32_t)kk * 1024u);
  };
  // a round's results wait in the wave's scratch ([channel][4 bins], written by lane <-> (bin, chunk)) until flush(): lane <-> channel
  // reads its four bins and stores them as 16 contiguous bytes (LDS operations of one wave execute in order: no wait in between)
  float* pend_dst = nullptr; int pend_nv = 0;
  auto flush = [&]() {
    if (pend_dst == nullptr) return;
#pragma unroll
    for (int cc = 0; cc < CB; cc += 64) {
      const int ch = cc + lane;
      const float* si = scr + (size_t)ch * kNpScrPitch;
      float* dst = pend_dst + (size_t)ch * bins;
      if (pend_nv == 4) {
        NpF4 v; v.x = si[0]; v.y = si[1]; v.z = si[2]; v.w = si[3];
        *reinterpret_cast<NpF4*>(dst) = v;
      } else {
        for (int j = 0; j < pend_nv; j++) dst[j] = si[j];
      }
    }
    pend_dst = nullptr;
  };
  __syncthreads();                                         // P0: the planner has published units 0 and 1
  NpU d = read_desc(0);
  if ((d.flags & 3) == 3) issue_dma(d, 0);
  int k = 0;
  while (true) {
    NP_MARK(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of unit k's DMA has landed (and its output stores)
    __syncthreads();                                       // T: descriptors, tables and the image of unit k are visible
    NP_MARK(1);
    flush();                                               // the last round of unit k-1
    if (!(d.flags & 1)) break;
    const int ks = k % kNpSlots, kn = ks == kNpSlots - 1 ? 0 : ks + 1;
    const NpU dn = read_desc(kn);                          // unit k+1's window goes in flight now: it lands while unit k is pooled
    if ((dn.flags & 3) == 3) issue_dma(dn, (k + 1) & 1);
    NP_MARK(2);
    // ---- pool unit k: bins [pa PW, pb PW) in rounds of 28; this wave's four slots are the consecutive bins 28 r + 4 pwv + (0..3)
    const int pa = d.pa, pb = d.pb, staged = (d.flags >> 1) & 1, mode = (d.flags >> 3) & 3, par = (d.flags >> 5) & 3;
    const int buf = k & 1;
    const int nb = (pb - pa) * PW, b0 = pa * PW;
    float* out = reinterpret_cast<float*>(((uint64_t)d.out_hi << 32) | (uint64_t)d.out_lo);
    if (mode == 2) {
      for (int i = ptid; i < CB * nb; i += kNpThreads - 64) {
        const int c = i / nb, e = i - c * nb;
        out[(size_t)c * bins + b0 + e] = 0.f;
      }
    } else {
      const NpTab* ty = tab + (par * 2 + 0) * kNpTabEntries;
      const NpTab* tx = tab + (par * 2 + 1) * kNpTabEntries;
      // image base of this unit, moved back by the strip's first row: table offsets are relative to the item's window
      const uint32_t ib = img32 + (uint32_t)buf * img_bytes - (uint32_t)d.strip_off + (uint32_t)q * 16u;
      for (int r0 = 0; r0 + 4 * pwv < nb; r0 += kNpPoolSlots) {
        flush();                                            // the previous round
        const int bl = r0 + slot;                           // bin of the unit
        if (bl < nb) {
          const int phl = (int)(((float)bl + 0.5f) * rpwf);
          const int pw = bl - phl * PW, ph = pa + phl;
          const NpTab ex0 = tx[2 * pw], ex1 = tx[2 * pw + 1];
          nf32x2 acc[CL / 2];
#pragma unroll
          for (int c = 0; c < CL / 2; c++) acc[c] = nf32x2{0.f, 0.f};
          // reference order: for iy { for ix { acc += w1*v1 + w2*v2 + w3*v3 + w4*v4 } }   (roi_align_cpu_loop.cpp:203-214)
#pragma unroll
          for (int iy = 0; iy < 2; iy++) {
            const NpTab ey = ty[2 * ph + iy];
#pragma unroll
            for (int ix = 0; ix < 2; ix++) {
              const NpTab ex = ix ? ex1 : ex0;
              nu32x4 r1, r2, r3, r4;
              if (staged) {
                r1 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(ib + (uint32_t)(ey.off_lo + ex.off_lo));
                r2 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(ib + (uint32_t)(ey.off_lo + ex.off_hi));
                r3 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(ib + (uint32_t)(ey.off_hi + ex.off_lo));
                r4 = *reinterpret_cast<__attribute__((address_space(3))) const nu32x4*>(ib + (uint32_t)(ey.off_hi + ex.off_hi));
              } else {
                const NpRc cy = (rct + (par * 2 + 0) * kNpTabEntries)[2 * ph + iy], cx = (rct + (par * 2 + 1) * kNpTabEntries)[2 * pw + ix];
                const NpGlb g = *reinterpret_cast<const NpGlb*>(reinterpret_cast<const char*>(ig) + par * 32);
                const char* gb = reinterpret_cast<const char*>(g.fbase) + q * 16;
                auto ga = [&](int yy, int xx) { return gb + ((int64_t)yy * g.sh + (int64_t)xx * g.sw) * (int64_t)sizeof(TIn); };
                if (mode == 0) {
                  r1 = *reinterpret_cast<const nu32x4*>(ga(cy.lo, cx.lo)); r2 = *reinterpret_cast<const nu32x4*>(ga(cy.lo, cx.hi));
                  r3 = *reinterpret_cast<const nu32x4*>(ga(cy.hi, cx.lo)); r4 = *reinterpret_cast<const nu32x4*>(ga(cy.hi, cx.hi));
                } else {      // unaligned chunks: 4-byte loads
                  const uint32_t* a1 = reinterpret_cast<const uint32_t*>(ga(cy.lo, cx.lo)); const uint32_t* a2 = reinterpret_cast<const uint32_t*>(ga(cy.lo, cx.hi));
                  const uint32_t* a3 = reinterpret_cast<const uint32_t*>(ga(cy.hi, cx.lo)); const uint32_t* a4 = reinterpret_cast<const uint32_t*>(ga(cy.hi, cx.hi));
                  r1 = nu32x4{a1[0], a1[1], a1[2], a1[3]}; r2 = nu32x4{a2[0], a2[1], a2[2], a2[3]};
                  r3 = nu32x4{a3[0], a3[1], a3[2], a3[3]}; r4 = nu32x4{a4[0], a4[1], a4[2], a4[3]};
                }
              }
              float v1[CL], v2[CL], v3[CL], v4[CL];
              NlLane<TIn>::widen(r1, v1); NlLane<TIn>::widen(r2, v2); NlLane<TIn>::widen(r3, v3); NlLane<TIn>::widen(r4, v4);
              const float w1 = ey.h * ex.h, w2 = ey.h * ex.l, w3 = ey.l * ex.h, w4 = ey.l * ex.l;                     // :95
              // two channels per instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE results, half the instructions)
#pragma unroll
              for (int c = 0; c < CL; c += 2) {
                const nf32x2 a1 = {v1[c], v1[c + 1]}, a2 = {v2[c], v2[c + 1]}, a3 = {v3[c], v3[c + 1]}, a4 = {v4[c], v4[c + 1]};
                acc[c >> 1] += w1 * a1 + w2 * a2 + w3 * a3 + w4 * a4;                                                // :208-211
              }
            }
          }
          // :216  output_val /= count ; count == 4 -> x * 0.25f is the same float32.  Into the wave's scratch: [channel][bin of the four]
          float* so = scr + (size_t)(q * CL) * kNpScrPitch + sl;
#pragma unroll
          for (int c = 0; c < CL / 2; c++) {
            const nf32x2 o = acc[c] * 0.25f;
            so[(2 * c) * kNpScrPitch] = o.x; so[(2 * c + 1) * kNpScrPitch] = o.y;
          }
        }
        // the wave's four bins of every channel are 16 contiguous output bytes.  They leave at the top of the NEXT round (or iteration):
        // a store issued here would be the last thing in front of T's vmcnt(0) and its latency would be exposed
        pend_dst = out + b0 + r0 + 4 * pwv; pend_nv = min(4, nb - (r0 + 4 * pwv));
      }
    }
    NP_MARK(4);
    d = dn;
    k++;
  }
#ifdef DTC_NP_TRACE
  if (tid == 64 && blockIdx.x < 4096) {
    unsigned long long* o = g_np_trace + (size_t)blockIdx.x * 2 * kNpTraceSlots + kNpTraceSlots;
    for (int i = 0; i < 8; i++) o[i] = tt.acc[i];
  }
#endif
}

#ifdef DTC_NP_TRACE
}  // namespace dtc
extern "C" __attribute__((visibility("default"))) int dtc_debug_np_trace(void* host_dst, size_t bytes) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(dtc::g_np_trace), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
namespace dtc {
#endif

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// Which kernel (measured on MI355X, bench inputs, batch 8, float32 maps; profiles/r04_c_*):
//   <= 64 bins (box head)   roi_align_fwd_nhwc_lds (four 256-thread workgroups per CU at 40 KB).  The pipelined kernel is 7-11 % faster
//                           on the bench's RoIs (0.349 against 0.36-0.39 ms per 8000-RoI launch) but 45 % SLOWER on the harder set
//                           (log-uniform sides 16-600 px: 0.58 against 0.40 ms) -- large windows become many small units, each with
//                           the pipeline's fixed cost and a quarter of the bin slots busy -- so it is NOT the default there;
//   > 64 bins (mask head)   the pipelined kernel (two 512-thread workgroups per CU, 78 KB each: two 132-pixel images): 0.129 ms per
//                           1024-RoI launch against 0.180 for the RoI-stationary LDS kernel (roi_align.hip) that took these before.
// Float32 output only for the pipelined kernel.  Development / A-B knobs, resolved once per process: DTC_RA_NHWC_LDS=0 (neither
// kernel), DTC_RA_NHWC_LDS_KB, DTC_RA_NHWC_PIPE = 0 never / 1 by bin count (default) / 2 always, DTC_RA_NHWC_PIPE16=1 /
// DTC_RA_NHWC_LDS_16BIT=1 (16-bit maps too).
struct NlConfig { int enabled = 1, lds_kb = 0, pipe = 1, pipe16 = 0; };
static const NlConfig& nl_config() {
  static const NlConfig cfg = [] {
    NlConfig c;
    if (const char* e = getenv("DTC_RA_NHWC_LDS")) c.enabled = atoi(e) != 0;
    if (const char* e = getenv("DTC_RA_NHWC_LDS_KB")) { const int v = atoi(e); if (v >= 24 && v <= 160) c.lds_kb = v; }
    if (const char* e = getenv("DTC_RA_NHWC_PIPE")) { const int v = atoi(e); if (v >= 0 && v <= 2) c.pipe = v; }
    if (const char* e = getenv("DTC_RA_NHWC_PIPE16")) c.pipe16 = atoi(e) != 0;
    return c;
  }();
  return cfg;
}

template <typename TIn> static int nl_cb() { return 16 * NlLane<TIn>::kCh; }

// LDS per workgroup of the pipelined kernel and what it is split into (host side, once per launch)
struct NpPlan { int lds_b, img_pixels, wgs_per_cu; };
static bool np_plan(int in_dtype, NpPlan& pl) {
  const int cb = in_dtype == DTC_F32 ? 64 : 128;
  const NlConfig& cfg = nl_config();
  pl.lds_b = (cfg.lds_kb ? cfg.lds_kb : 78) * 1024;
  const int room = pl.lds_b - kNpHdrBytes - kNpPoolWaves * cb * kNpScrPitch * 4;
  pl.img_pixels = room < 0 ? 0 : ((room / 2 / kNlChunk) & ~3);       // two images; the DMA writes whole groups of 4 pixels
  if (pl.img_pixels > 8188) pl.img_pixels = 8188;                     // pixel indices stay exact in the float reciprocal
  pl.wgs_per_cu = (160 * 1024) / pl.lds_b;
  if (pl.wgs_per_cu > 4) pl.wgs_per_cu = 4;                           // 512 threads each, 2048 per CU
  if (pl.wgs_per_cu < 1) pl.wgs_per_cu = 1;
  return pl.img_pixels >= 16;
}
static bool np_takes(int bins, int in_dtype, int out_dtype) {
  const NlConfig& cfg = nl_config();
  return (cfg.pipe == 2 || (cfg.pipe == 1 && bins > kNlMaxBins)) && out_dtype == DTC_F32 && (in_dtype == DTC_F32 || cfg.pipe16);
}

bool roi_align_nhwc_lds_supported(const RoiAlignParams& p, int in_dtype, int out_dtype) {
  const NlConfig& cfg = nl_config();
  if (!cfg.enabled || p.sampling_ratio != 2) return false;
  // 16-bit maps: the direct-gather kernel (8-channel lanes, roi_align.hip) unless asked
  static const bool force16 = [] { const char* e = getenv("DTC_RA_NHWC_LDS_16BIT"); return e && atoi(e) != 0; }();
  if (in_dtype != DTC_F32 && !force16 && !cfg.pipe16) return false;
  const int bins = p.pooled_h * p.pooled_w;
  if (p.pooled_h > 16 || p.pooled_w > 16) return false;
  const int cb = in_dtype == DTC_F32 ? 64 : 128;
  if (p.channels % cb != 0) return false;
  if (np_takes(bins, in_dtype, out_dtype)) { NpPlan pl; if (!np_plan(in_dtype, pl)) return false; }
  // the round-3 kernel: <= 64 bins; the tables and the output slab must leave room for a window image (else: the direct-gather kernel)
  else if (bins > kNlMaxBins || ((cfg.lds_kb ? cfg.lds_kb : 40) * 1024 - (1024 + kNlMaxBins * kNlBinRec + cb * bins * 4)) / kNlChunk < 16 + 3) return false;
  for (int l = 0; l < p.n_levels; l++)
    if (p.lv[l].stride_c != 1 || p.lv[l].height > 65535 || p.lv[l].width > 65535) return false;
  const bool f = in_dtype == DTC_F32, h = in_dtype == DTC_F16, b = in_dtype == DTC_BF16;
  return (f && (out_dtype == DTC_F32 || out_dtype == DTC_F16 || out_dtype == DTC_BF16)) ||
         (h && (out_dtype == DTC_F32 || out_dtype == DTC_F16)) || (b && (out_dtype == DTC_F32 || out_dtype == DTC_BF16));
}

static int np_cus() {      // compute units of the device the launch goes to (MI355X: 256 = 8 XCDs x 32)
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
    return v;
  }();
  return n;
}

template <typename TIn>
static int launch_np_t(const RoiAlignParams& p, int in_dtype, hipStream_t stream) {
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_nhwc_pipe<TIn, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr_rc == hipSuccess)
      attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_nhwc_pipe<TIn, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  NpPlan pl;
  if (!np_plan(in_dtype, pl)) return DTC_EUNSUPPORTED;
  const int cb = nl_cb<TIn>();
  const long long n_items = (long long)p.n_rois * (p.channels / cb);
  // persistent workgroups: as many as are resident at once (a multiple of the 8 XCDs), fewer when there is less work
  long long per_xcd = (long long)(np_cus() / kXcds) * pl.wgs_per_cu;
  const long long need = (n_items + kXcds - 1) / kXcds;
  if (per_xcd > need) per_xcd = need;
  if (per_xcd < 1) per_xcd = 1;
  const size_t smem = (size_t)kNpHdrBytes + (size_t)kNpPoolWaves * cb * kNpScrPitch * 4 + 2 * (size_t)pl.img_pixels * kNlChunk;
  if (p.roi_desc)
    hipLaunchKernelGGL((roi_align_fwd_nhwc_pipe<TIn, true>), dim3((unsigned)(per_xcd * kXcds)), dim3(kNpThreads), smem, stream, p, pl.img_pixels);
  else
    hipLaunchKernelGGL((roi_align_fwd_nhwc_pipe<TIn, false>), dim3((unsigned)(per_xcd * kXcds)), dim3(kNpThreads), smem, stream, p, pl.img_pixels);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

template <typename TIn, typename TOut>
static int launch_nl_t(const RoiAlignParams& p, hipStream_t stream) {
  static std::once_flag once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_fwd_nhwc_lds<TIn, TOut>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_rc != hipSuccess) return DTC_ELAUNCH;
  const int bins = p.pooled_h * p.pooled_w;
  const int cb = nl_cb<TIn>();
  const int fixed = 1024 + kNlMaxBins * kNlBinRec + cb * bins * 4;
  const int lds_b = (nl_config().lds_kb ? nl_config().lds_kb : 40) * 1024;
  int img_pixels = ((lds_b - fixed) / kNlChunk) & ~3;           // the DMA writes whole groups of 4 pixels
  if (img_pixels < 16) return DTC_EUNSUPPORTED;
  if (img_pixels > 8188) img_pixels = 8188;                      // pixel indices stay exact in the float reciprocal
  const int nct = p.channels / cb;
  hipLaunchKernelGGL((roi_align_fwd_nhwc_lds<TIn, TOut>), dim3((unsigned)p.n_rois * nct), dim3(kNlThreads), fixed + img_pixels * kNlChunk, stream, p, img_pixels);
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

int launch_roi_align_nhwc_lds(const RoiAlignParams& p, int in_dtype, int out_dtype, hipStream_t stream) {
  if (p.n_rois == 0) return DTC_OK;
  if (np_takes(p.pooled_h * p.pooled_w, in_dtype, out_dtype)) {
    if (in_dtype == DTC_F32) return launch_np_t<float>(p, in_dtype, stream);
    if (in_dtype == DTC_F16) return launch_np_t<__half>(p, in_dtype, stream);
    if (in_dtype == DTC_BF16) return launch_np_t<bf16_t>(p, in_dtype, stream);
    return DTC_EUNSUPPORTED;
  }
  if (in_dtype == DTC_F32 && out_dtype == DTC_F32) return launch_nl_t<float, float>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_F16) return launch_nl_t<float, __half>(p, stream);
  if (in_dtype == DTC_F32 && out_dtype == DTC_BF16) return launch_nl_t<float, bf16_t>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F32) return launch_nl_t<__half, float>(p, stream);
  if (in_dtype == DTC_F16 && out_dtype == DTC_F16) return launch_nl_t<__half, __half>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_F32) return launch_nl_t<bf16_t, float>(p, stream);
  if (in_dtype == DTC_BF16 && out_dtype == DTC_BF16) return launch_nl_t<bf16_t, bf16_t>(p, stream);
  return DTC_EUNSUPPORTED;
}

}  // namespace dtc
