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

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(EpiParams p) {
  constexpr int N = VEC ? EpiVec<T>::N : 1;
  using P = Pack<T, N>;
  T* x = reinterpret_cast<T*>(p.x);
  const T* res = reinterpret_cast<const T*>(p.res);
  const long long nvec = p.total / N;
  const int hw = p.h * p.w;
  const int rh = p.up2 ? p.h >> 1 : p.h, rw = p.up2 ? p.w >> 1 : p.w;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (long long)gridDim.x * blockDim.x) {
    const long long e0 = t * N;                   // first element of this thread's run (never straddles a row of the
                                                  // innermost dimension: its extent is a multiple of N when VEC)
    P v = *reinterpret_cast<const P*>(x + e0);
    float f[N];
#pragma unroll
    for (int i = 0; i < N; i++) f[i] = to_f32<T>(v.v[i]);
    if (p.nhwc) {
      const long long pix = e0 / p.c;
      const int c0 = (int)(e0 - pix * p.c);
      if (p.bias) {
#pragma unroll
        for (int i = 0; i < N; i++) f[i] += p.bias[c0 + i];
      }
      if (res) {
        long long rpix = pix;
        if (p.up2) {
          const long long n = pix / hw;
          const int r = (int)(pix - n * hw), y = r / p.w, xx = r - y * p.w;
          rpix = (n * rh + (y >> 1)) * rw + (xx >> 1);
        }
        const P rv = *reinterpret_cast<const P*>(res + rpix * p.c + c0);
#pragma unroll
        for (int i = 0; i < N; i++) f[i] += to_f32<T>(rv.v[i]);
      }
    } else {
      const long long plane = e0 / hw;            // n * C + c
      const int c = (int)(plane % p.c);
      if (p.bias) {
        const float b = p.bias[c];
#pragma unroll
        for (int i = 0; i < N; i++) f[i] += b;
      }
      if (res) {
        if (!p.up2) {
          const P rv = *reinterpret_cast<const P*>(res + e0);
#pragma unroll
          for (int i = 0; i < N; i++) f[i] += to_f32<T>(rv.v[i]);
        } else {
          const int r = (int)(e0 - plane * hw), y = r / p.w, x0 = r - y * p.w;
          const T* rrow = res + (plane * rh + (y >> 1)) * rw;
#pragma unroll
          for (int i = 0; i < N; i++) f[i] += to_f32<T>(rrow[(x0 + i) >> 1]);
        }
      }
    }
    if (p.relu) {
#pragma unroll
      for (int i = 0; i < N; i++) f[i] = f[i] < 0.f ? 0.f : f[i];     // NaN stays NaN, like torch.relu
    }
#pragma unroll
    for (int i = 0; i < N; i++) v.v[i] = from_f32<T>(f[i]);
    *reinterpret_cast<P*>(x + e0) = v;
  }
}

template <typename T>
static int launch_bias_act(const EpiParams& p, hipStream_t stream) {
  constexpr int N = EpiVec<T>::N;
  const int inner = p.nhwc ? p.c : p.w;
  const bool vec = inner % N == 0 && (reinterpret_cast<uintptr_t>(p.x) % (sizeof(T) * N)) == 0 &&
                   (!p.res || (reinterpret_cast<uintptr_t>(p.res) % (sizeof(T) * N)) == 0) &&
                   (!p.nhwc || !p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 3) == 0);
  const long long nvec = vec ? p.total / N : p.total;
  const long long want = (nvec + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 256 * 32 ? 256 * 32 : want));      // <= 32 workgroups per CU, grid-stride
  if (vec) hipLaunchKernelGGL((bias_act_kernel<T, true>), dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((bias_act_kernel<T, false>), dim3(blocks), dim3(256), 0, stream, p);
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
