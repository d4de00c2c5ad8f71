// C++ harness for the C ABI of libdetectorch_hip.so -- no Python, no torch: plain HIP runtime allocations, the public header,
// and the oracle (liboracle.so, test infrastructure) as the checker.  Built and run by tests/test_hip_cxx_harness.py:
//   hipcc --offload-arch=gfx950 -O2 tests/cxx/abi_harness.cpp -Iinclude -Ldetectorch_amd/lib -ldetectorch_hip -Loracle -loracle
// Exercises: launch_roi_align_forward_hip (the reference's own C ABI, lib/cppcuda_cffi/src/cuda/
// roi_align_forward_cuda_kernel.h:7-19), dtc_roi_align_forward (multi-level), dtc_nms, dtc_bbox_overlaps.  Bit-exact or exit 1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "detectorch_hip.h"

extern "C" {   // oracle/oracle.c
void orc_roi_align_forward(const float* features, const float* rois, int n_rois, int roi_cols, float spatial_scale, int channels,
                           int height, int width, int pooled_h, int pooled_w, int sampling_ratio, float* out);
int orc_nms(const float* dets, int n, float thresh, int max_keep, long long* keep);
void orc_bbox_overlaps(const float* boxes, int n, const float* query, int k, float* overlaps);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }

template <typename T> static T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

int main() {
  printf("library %s for %s\n", dtc_version(), dtc_target_arch());
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  int failures = 0;

  // ---- RoIAlign through the reference-shaped entry: [2,24,40,56] features, 96 rois, 7x7, sr 2, scale 1/8 ---------------------
  const int B = 2, C = 24, H = 40, W = 56, R = 96, PH = 7, PW = 7, SR = 2;
  const float scale = 0.125f;
  std::vector<float> feat((size_t)B * C * H * W), rois((size_t)R * 5);
  for (auto& v : feat) v = urand() * 2.f - 1.f;
  for (int r = 0; r < R; r++) {
    const float x1 = urand() * 400.f, y1 = urand() * 280.f, w = 8.f + urand() * 200.f, h = 8.f + urand() * 150.f;
    rois[r * 5 + 0] = (float)(r & 1); rois[r * 5 + 1] = x1; rois[r * 5 + 2] = y1; rois[r * 5 + 3] = x1 + w; rois[r * 5 + 4] = y1 + h;
  }
  std::vector<float> ref((size_t)R * C * PH * PW), got(ref.size());
  orc_roi_align_forward(feat.data(), rois.data(), R, 5, scale, C, H, W, PH, PW, SR, ref.data());
  float *d_feat = to_dev(feat), *d_rois = to_dev(rois), *d_out = nullptr;
  CK(hipMalloc(&d_out, ref.size() * sizeof(float)));
  if (launch_roi_align_forward_hip((int)ref.size(), d_feat, d_rois, scale, C, H, W, PH, PW, SR, d_out, (dtc_stream_t)stream) != 1) {
    fprintf(stderr, "launch_roi_align_forward_hip returned 0\n"); return 1;
  }
  CK(hipStreamSynchronize(stream));
  CK(hipMemcpy(got.data(), d_out, ref.size() * sizeof(float), hipMemcpyDeviceToHost));
  if (memcmp(got.data(), ref.data(), ref.size() * sizeof(float)) != 0) { fprintf(stderr, "RoIAlign (reference ABI) differs from the oracle\n"); failures++; }

  // ---- the same through dtc_roi_align_forward with an explicit level struct and 4-column rois of image 0 ------------------------
  dtc_feat_level lv;
  memset(&lv, 0, sizeof(lv));
  lv.data = d_feat; lv.height = H; lv.width = W; lv.spatial_scale = scale;
  lv.stride_n = (int64_t)C * H * W; lv.stride_c = (int64_t)H * W; lv.stride_h = W; lv.stride_w = 1;
  std::vector<float> rois4((size_t)R * 4), rois5b((size_t)R * 5);
  for (int r = 0; r < R; r++) { for (int k = 0; k < 4; k++) { rois4[r * 4 + k] = rois[r * 5 + 1 + k]; rois5b[r * 5 + 1 + k] = rois[r * 5 + 1 + k]; } rois5b[r * 5] = 0.f; }
  orc_roi_align_forward(feat.data(), rois5b.data(), R, 5, scale, C, H, W, PH, PW, SR, ref.data());
  float* d_rois4 = to_dev(rois4);
  int rc = dtc_roi_align_forward(&lv, 1, C, DTC_F32, d_rois4, 4, nullptr, R, PH, PW, SR, d_out, DTC_F32, (dtc_stream_t)stream);
  CK(hipStreamSynchronize(stream));
  CK(hipMemcpy(got.data(), d_out, ref.size() * sizeof(float), hipMemcpyDeviceToHost));
  if (rc != DTC_OK || memcmp(got.data(), ref.data(), ref.size() * sizeof(float)) != 0) { fprintf(stderr, "dtc_roi_align_forward (4-col rois) rc=%d differs\n", rc); failures++; }

  // ---- NMS: 1500 boxes, threshold 0.5, ascending kept indices ---------------------------------------------------------------------
  const int N = 1500;
  std::vector<float> dets((size_t)N * 5);
  for (int i = 0; i < N; i++) {
    const float x1 = urand() * 600.f, y1 = urand() * 400.f;
    dets[i * 5 + 0] = x1; dets[i * 5 + 1] = y1; dets[i * 5 + 2] = x1 + 20.f + urand() * 150.f; dets[i * 5 + 3] = y1 + 20.f + urand() * 150.f;
    dets[i * 5 + 4] = (float)(N - i) / (float)N * 0.999f + urand() * 1e-4f;
  }
  std::vector<long long> keep_ref(N);
  const int n_ref = orc_nms(dets.data(), N, 0.5f, 0, keep_ref.data());
  float* d_dets = to_dev(dets);
  const size_t wsb = dtc_nms_workspace_bytes(N);
  void* d_ws = nullptr; long long* d_keep = nullptr; int32_t* d_n = nullptr;
  CK(hipMalloc(&d_ws, wsb)); CK(hipMalloc(&d_keep, N * sizeof(long long))); CK(hipMalloc(&d_n, sizeof(int32_t)));
  rc = dtc_nms(d_dets, N, 0.5f, d_ws, wsb, (int64_t*)d_keep, d_n, (dtc_stream_t)stream);
  CK(hipStreamSynchronize(stream));
  int32_t n_got = -1; std::vector<long long> keep_got(N);
  CK(hipMemcpy(&n_got, d_n, sizeof(int32_t), hipMemcpyDeviceToHost));
  CK(hipMemcpy(keep_got.data(), d_keep, N * sizeof(long long), hipMemcpyDeviceToHost));
  if (rc != DTC_OK || n_got != n_ref || memcmp(keep_got.data(), keep_ref.data(), (size_t)n_ref * sizeof(long long)) != 0) {
    fprintf(stderr, "dtc_nms rc=%d kept %d vs oracle %d (or indices differ)\n", rc, n_got, n_ref); failures++;
  }

  // ---- bbox_overlaps 300 x 257 --------------------------------------------------------------------------------------------------------
  const int NB = 300, NK = 257;
  std::vector<float> bx((size_t)NB * 4), qx((size_t)NK * 4), ov_ref((size_t)NB * NK), ov_got(ov_ref.size());
  for (int i = 0; i < NB; i++) { const float x = urand() * 300.f, y = urand() * 300.f; bx[i * 4] = x; bx[i * 4 + 1] = y; bx[i * 4 + 2] = x + 5.f + urand() * 120.f; bx[i * 4 + 3] = y + 5.f + urand() * 120.f; }
  for (int i = 0; i < NK; i++) { const float x = urand() * 300.f, y = urand() * 300.f; qx[i * 4] = x; qx[i * 4 + 1] = y; qx[i * 4 + 2] = x + 5.f + urand() * 120.f; qx[i * 4 + 3] = y + 5.f + urand() * 120.f; }
  orc_bbox_overlaps(bx.data(), NB, qx.data(), NK, ov_ref.data());
  float *d_bx = to_dev(bx), *d_qx = to_dev(qx), *d_ov = nullptr;
  CK(hipMalloc(&d_ov, ov_ref.size() * sizeof(float)));
  rc = dtc_bbox_overlaps(d_bx, NB, 4, d_qx, NK, 4, d_ov, (dtc_stream_t)stream);
  CK(hipStreamSynchronize(stream));
  CK(hipMemcpy(ov_got.data(), d_ov, ov_ref.size() * sizeof(float), hipMemcpyDeviceToHost));
  if (rc != DTC_OK || memcmp(ov_got.data(), ov_ref.data(), ov_ref.size() * sizeof(float)) != 0) { fprintf(stderr, "dtc_bbox_overlaps differs\n"); failures++; }

  printf(failures ? "FAILED: %d check(s)\n" : "C ABI harness: all checks bit-exact (%d failures)\n", failures);
  return failures ? 1 : 0;
}
