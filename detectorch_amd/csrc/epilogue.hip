// Convolution epilogue for the inference model (the call sites that carry the path): per-channel bias, an
// optional residual (same shape, or half the height / width read with nearest-neighbour x2 upsampling) and an optional ReLU,
// applied IN PLACE to the output of a bias-free convolution -- one HBM pass where the reference's graph runs three or four:
//   ResNet bottleneck   bn(conv(x)) -> relu, bn3(conv3(.)) + identity -> relu   (torchvision Bottleneck as loaded by
//                       lib/model/detector.py:153-170; the BatchNorm layers are caffe2 AffineChannel ops in eval mode,
//                       detector.py:231, i.e. a per-channel scale + shift that `detector.optimize_for_inference` folds into the
//                       convolution weights and this bias)
//   FPN top-down        F.upsample(top, scale_factor=2, mode='nearest') + lateral(c)     (detector.py:45-46)
//   RPN / mask heads    relu(conv(x) + b)                                               (detector.py:85-97, 123-127)
// HBM-bound elementwise kernel: 16-byte accesses, thread <-> 8 (16-bit) or 4 (fp32) consecutive elements of the innermost
// dimension, float32 arithmetic, one rounding to the tensor's type.  Both dense layouts of a [N, C, H, W] tensor are handled in
// place: channels_last (C innermost: the bias is a per-thread vector) and NCHW (W innermost: the bias is a per-thread scalar).
#include "dtc_common.h"

namespace dtc {

struct EpiParams {
  void* x;                 // [N, C, H, W] in the layout given by `nhwc`, updated in place
  const float* bias;       // [C] float32 or null
  const void* res;         // residual of x's type and layout: [N, C, H, W], or [N, C, H/2, W/2] when up2; or null
  long long total;         // N * C * H * W
  int c, h, w;
  int nhwc, relu, up2;
};

template <typename T> struct EpiVec;
template <> struct EpiVec<float> { static constexpr int N = 4; };
template <> struct EpiVec<__half> { static constexpr int N = 8; };
template <> struct EpiVec<bf16_t> { static constexpr int N = 8; };

template <typename T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };

// One 16-byte (VEC) or one-element run starting at element e0: channel of the run's first element c0 (channels_last) or the
// run's channel c (NCHW) is supplied by the caller, so the streaming paths below do no per-run division.
template <typename T, int N>
__device__ __forceinline__ void epi_run(const EpiParams& p, T* x, const T* res, long long e0, int c0, const T* res_run) {
  using P = Pack<T, N>;
  P v = *reinterpret_cast<const P*>(x + e0);
  float f[N];
#pragma unroll
  for (int i = 0; i < N; i++) f[i] = to_f32<T>(v.v[i]);
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < N; i++) f[i] += p.bias[p.nhwc ? c0 + i : c0];
  }
  if (res_run) {           // residual run of the same shape (or the channels_last source pixel of an upsampled one)
    const P rv = *reinterpret_cast<const P*>(res_run);
#pragma unroll
    for (int i = 0; i < N; i++) f[i] += to_f32<T>(rv.v[i]);
  }
  if (p.relu) {
#pragma unroll
    for (int i = 0; i < N; i++) f[i] = f[i] < 0.f ? 0.f : f[i];     // NaN stays NaN, like torch.relu
  }
#pragma unroll
  for (int i = 0; i < N; i++) v.v[i] = from_f32<T>(f[i]);
  *reinterpret_cast<P*>(x + e0) = v;
}

// channels_last, vector runs, no upsampling: a flat stream; the channel of a thread's run advances by a constant modulo C.
// The bias vector lives in LDS (one global read per workgroup: eight 4-byte global loads per 16-byte run would cost the
// texture unit more than the payload does), and a thread keeps UNROLL runs in flight.
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(EpiParams p) {
  constexpr int N = EpiVec<T>::N;
  using P = Pack<T, N>;
  extern __shared__ __attribute__((aligned(16))) float sbias[];
  if (p.bias)
    for (int i = threadIdx.x; i < p.c; i += blockDim.x) sbias[i] = p.bias[i];
  __syncthreads();
  T* x = reinterpret_cast<T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.res);
  const long long stride = (long long)gridDim.x * blockDim.x * N;
  long long e0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * N;
  int c0 = (int)(e0 % p.c);
  const int step = (int)(stride % p.c);
  while (e0 < p.total) {
    P v[UNROLL], rv[UNROLL];
    int cc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {                      // every load of the group first
      const long long e = e0 + u * stride;
      cc[u] = c0;
      c0 += step;
      if (c0 >= p.c) c0 -= p.c;
      if (e < p.total) {
        v[u] = *reinterpret_cast<const P*>(x + e);
        if (res) rv[u] = *reinterpret_cast<const P*>(res + e);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const long long e = e0 + u * stride;
      if (e < p.total) {
        float f[N];
#pragma unroll
        for (int i = 0; i < N; i++) f[i] = to_f32<T>(v[u].v[i]);
        if (p.bias) {
#pragma unroll
          for (int i = 0; i < N; i++) f[i] += sbias[cc[u] + i];
        }
        if (res) {
#pragma unroll
          for (int i = 0; i < N; i++) f[i] += to_f32<T>(rv[u].v[i]);
        }
        if (p.relu) {
#pragma unroll
          for (int i = 0; i < N; i++) f[i] = f[i] < 0.f ? 0.f : f[i];
        }
#pragma unroll
        for (int i = 0; i < N; i++) v[u].v[i] = from_f32<T>(f[i]);
        *reinterpret_cast<P*>(x + e) = v[u];
      }
    }
    e0 += UNROLL * stride;
  }
}

// NCHW, vector runs along W, no upsampling: workgroup <-> (plane n * C + c, chunk of the plane): the channel is uniform
template <typename T>
__global__ __launch_bounds__(256) void bias_act_nchw_kernel(EpiParams p, int chunks, int chunk_vecs) {
  constexpr int N = EpiVec<T>::N;
  T* x = reinterpret_cast<T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.res);
  const int hw = p.h * p.w, hw_vecs = hw / N;
  const long long planes = p.total / hw, items = planes * chunks;
  for (long long it = blockIdx.x; it < items; it += gridDim.x) {
    const long long plane = it / chunks;
    const int chunk = (int)(it - plane * chunks), c = (int)(plane % p.c);
    const int v_end = min(hw_vecs, (chunk + 1) * chunk_vecs);
    const float b = p.bias ? p.bias[c] : 0.f;
    const long long base = plane * hw;
    for (int v0 = chunk * chunk_vecs + threadIdx.x; v0 < v_end; v0 += 4 * blockDim.x) {     // 4 runs in flight per thread
      using P = Pack<T, N>;
      P v[4], rv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int vi = v0 + u * blockDim.x;
        if (vi < v_end) {
          v[u] = *reinterpret_cast<const P*>(x + base + (long long)vi * N);
          if (res) rv[u] = *reinterpret_cast<const P*>(res + base + (long long)vi * N);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int vi = v0 + u * blockDim.x;
        if (vi < v_end) {
#pragma unroll
          for (int i = 0; i < N; i++) {
            float f = to_f32<T>(v[u].v[i]);
            if (p.bias) f += b;
            if (res) f += to_f32<T>(rv[u].v[i]);
            if (p.relu) f = f < 0.f ? 0.f : f;
            v[u].v[i] = from_f32<T>(f);
          }
          *reinterpret_cast<P*>(x + base + (long long)vi * N) = v[u];
        }
      }
    }
  }
}

// NCHW whose planes are NOT a whole number of runs (25 x 42, 14 x 14 with 8-element runs): the tensor is still one dense
// stream, a run may cross from plane n*C + c into the next one (H * W >= N: at most once).  (plane mod C, offset in the plane)
// advance by constants per pass: no division per run.
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void bias_act_nchw_flat_kernel(EpiParams p) {
  constexpr int N = EpiVec<T>::N;
  using P = Pack<T, N>;
  T* x = reinterpret_cast<T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.res);
  const int hw = p.h * p.w;
  const long long stride = (long long)gridDim.x * blockDim.x * N;
  long long e0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * N;
  int rem = (int)(e0 % hw), c = (int)((e0 / hw) % p.c);
  const int step_rem = (int)(stride % hw), step_c = (int)((stride / hw) % p.c);
  while (e0 < p.total) {
    P v[UNROLL], rv[UNROLL];
    int cc[UNROLL], rr[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const long long e = e0 + u * stride;
      cc[u] = c; rr[u] = rem;
      rem += step_rem; c += step_c;
      if (rem >= hw) { rem -= hw; c++; }
      if (c >= p.c) c -= p.c;
      if (e < p.total) {
        v[u] = *reinterpret_cast<const P*>(x + e);
        if (res) rv[u] = *reinterpret_cast<const P*>(res + e);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const long long e = e0 + u * stride;
      if (e < p.total) {
        const int cn = cc[u] + 1 == p.c ? 0 : cc[u] + 1;
        const float b0 = p.bias ? p.bias[cc[u]] : 0.f, b1 = p.bias ? p.bias[cn] : 0.f;
#pragma unroll
        for (int i = 0; i < N; i++) {
          float f = to_f32<T>(v[u].v[i]);
          if (p.bias) f += rr[u] + i < hw ? b0 : b1;
          if (res) f += to_f32<T>(rv[u].v[i]);
          if (p.relu) f = f < 0.f ? 0.f : f;
          v[u].v[i] = from_f32<T>(f);
        }
        *reinterpret_cast<P*>(x + e) = v[u];
      }
    }
    e0 += UNROLL * stride;
  }
}

// everything else (element-wise runs on odd shapes, the x2-upsampled residual of the FPN top-down sum): index arithmetic per run
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(EpiParams p) {
  constexpr int N = VEC ? EpiVec<T>::N : 1;
  T* x = reinterpret_cast<T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.res);
  const long long nvec = p.total / N;
  const int hw = p.h * p.w;
  const int rh = p.up2 ? p.h >> 1 : p.h, rw = p.up2 ? p.w >> 1 : p.w;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (long long)gridDim.x * blockDim.x) {
    const long long e0 = t * N;                   // never straddles a row of the innermost dimension (extent % N == 0 when VEC)
    if (p.nhwc) {
      const long long pix = e0 / p.c;
      const int c0 = (int)(e0 - pix * p.c);
      const T* rr = nullptr;
      if (res) {
        long long rpix = pix;
        if (p.up2) {
          const long long n = pix / hw;
          const int r = (int)(pix - n * hw), y = r / p.w, xx = r - y * p.w;
          rpix = (n * rh + (y >> 1)) * rw + (xx >> 1);
        }
        rr = res + rpix * p.c + c0;
      }
      epi_run<T, N>(p, x, res, e0, c0, rr);
    } else {
      const long long plane = e0 / hw;            // n * C + c
      const int c = (int)(plane % p.c);
      if (res && p.up2) {                         // source columns (x0 + i) >> 1: gathered element by element
        const int r = (int)(e0 - plane * hw), y = r / p.w, x0 = r - y * p.w;
        const T* rrow = res + (plane * rh + (y >> 1)) * rw;
        Pack<T, N> rv;
#pragma unroll
        for (int i = 0; i < N; i++) rv.v[i] = rrow[(x0 + i) >> 1];
        // epi_run reads the residual run through a pointer: hand it the gathered copy
        using P = Pack<T, N>;
        P v = *reinterpret_cast<const P*>(x + e0);
        float f[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
          f[i] = to_f32<T>(v.v[i]);
          if (p.bias) f[i] += p.bias[c];
          f[i] += to_f32<T>(rv.v[i]);
          if (p.relu) f[i] = f[i] < 0.f ? 0.f : f[i];
          v.v[i] = from_f32<T>(f[i]);
        }
        *reinterpret_cast<P*>(x + e0) = v;
      } else {
        epi_run<T, N>(p, x, res, e0, c, res ? res + e0 : nullptr);
      }
    }
  }
}

template <typename T>
static int launch_bias_act(const EpiParams& p, hipStream_t stream) {
  constexpr int N = EpiVec<T>::N;
  const int inner = p.nhwc ? p.c : p.w;
  const bool aligned = (reinterpret_cast<uintptr_t>(p.x) % (sizeof(T) * N)) == 0 &&
                       (!p.res || (reinterpret_cast<uintptr_t>(p.res) % (sizeof(T) * N)) == 0);
  const bool vec = inner % N == 0 && aligned;
  constexpr long long kMaxBlocks = 256 * 16;           // 16 workgroups of 4 waves per CU, grid-stride beyond that
  if (vec && !p.up2 && p.nhwc && p.c <= 8192) {
    const long long want = (p.total / N + 255) / 256;
    const long long grp = (want + 3) / 4;                 // 4 runs per thread and pass
    hipLaunchKernelGGL((bias_act_nhwc_kernel<T, 4>), dim3((unsigned)(grp > kMaxBlocks ? kMaxBlocks : (grp < 1 ? 1 : grp))), dim3(256),
                       (size_t)p.c * sizeof(float), stream, p);
  } else if (aligned && !p.nhwc && !p.up2 && (p.h * p.w) % N == 0 && p.h * p.w >= 512 * N) {   // large planes: workgroup <-> plane chunk
    const int hw_vecs = p.h * p.w / N;
    const int chunk_vecs = 1024;                        // 4 runs per thread and chunk
    const int chunks = (hw_vecs + chunk_vecs - 1) / chunk_vecs;
    const long long items = p.total / ((long long)p.h * p.w) * chunks;
    hipLaunchKernelGGL((bias_act_nchw_kernel<T>), dim3((unsigned)(items > kMaxBlocks * 4 ? kMaxBlocks * 4 : items)), dim3(256), 0, stream, p, chunks, chunk_vecs);
  } else if (aligned && !p.nhwc && !p.up2 && p.total % N == 0 && p.h * p.w >= N) {
    const long long grp = (p.total / N + 1023) / 1024;
    hipLaunchKernelGGL((bias_act_nchw_flat_kernel<T, 4>), dim3((unsigned)(grp > kMaxBlocks ? kMaxBlocks : grp)), dim3(256), 0, stream, p);
  } else {
    const long long nvec = vec ? p.total / N : p.total;
    const long long want = (nvec + 255) / 256;
    const unsigned blocks = (unsigned)(want > kMaxBlocks * 2 ? kMaxBlocks * 2 : want);
    if (vec) hipLaunchKernelGGL((bias_act_kernel<T, true>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bias_act_kernel<T, false>), dim3(blocks), dim3(256), 0, stream, p);
  }
  DTC_CHECK_LAUNCH();
  return DTC_OK;
}

}  // namespace dtc

DTC_API int dtc_bias_act(void* x, const float* bias, const void* residual, int n, int c, int h, int w, int dtype,
                         int channels_last, int relu, int residual_up2, dtc_stream_t stream) {
  if (n < 0 || c < 0 || h < 0 || w < 0 || (!x && (long long)n * c * h * w > 0)) return DTC_EINVAL;
  if (residual_up2 && (!residual || (h & 1) || (w & 1))) return DTC_EINVAL;
  dtc::EpiParams p;
  p.x = x; p.bias = bias; p.res = residual;
  p.total = (long long)n * c * h * w;
  p.c = c; p.h = h; p.w = w;
  p.nhwc = channels_last ? 1 : 0; p.relu = relu ? 1 : 0; p.up2 = residual_up2 ? 1 : 0;
  if (p.total == 0) return DTC_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (dtype) {
    case DTC_F32: return dtc::launch_bias_act<float>(p, s);
    case DTC_F16: return dtc::launch_bias_act<__half>(p, s);
    case DTC_BF16: return dtc::launch_bias_act<dtc::bf16_t>(p, s);
    default: return DTC_EUNSUPPORTED;
  }
}
