/*
 * detectorch_hip.h -- C ABI of libdetectorch_hip.so: the MI355X (gfx950 / CDNA4) region-proposal hot path of detectorch.
 *
 * Plain pointers and sizes only; no torch / ATen / TH types cross this boundary.  Every entry point
 *   - takes DEVICE pointers (caller-owned; nothing is allocated or freed inside),
 *   - enqueues its kernels on the given hipStream_t and returns without synchronising,
 *   - returns DTC_OK (0) or a negative DTC_E* code (no exceptions, no printf), except the one entry that keeps the
 *     reference's own 1/0 convention (launch_roi_align_forward_hip).
 * Workspace is passed in by the caller; dtc_*_workspace_bytes() say how much.
 *
 * Each entry cites the reference interface it replaces (paths relative to the detectorch tree).
 */
#ifndef DETECTORCH_HIP_H_
#define DETECTORCH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dtc_stream_t; /* == hipStream_t */

enum { DTC_OK = 0, DTC_EINVAL = -1, DTC_ELAUNCH = -2, DTC_EWORKSPACE = -3, DTC_EUNSUPPORTED = -4 };
enum { DTC_F32 = 0, DTC_F16 = 1, DTC_U8 = 2 /* dtc_prep_images sources only */, DTC_BF16 = 3 /* RoIAlign features / output */ };

/* Library / build identification ("gfx950"); lets the host prove the native path is the one that is loaded. */
const char* dtc_version(void);
const char* dtc_target_arch(void);

/* ---------------------------------------------------------------------------------------------------------------
 * A1  RoIAlign forward
 * --------------------------------------------------------------------------------------------------------------- */

/* Drop-in for  int launch_roi_align_forward_cuda(...)  lib/cppcuda_cffi/src/cuda/roi_align_forward_cuda_kernel.h:7-19
 * (same argument list, cudaStream_t -> hipStream_t; returns 1 on success, 0 on error like the reference's launcher,
 * roi_align_forward_cuda_kernel.cu:161-201).  bottom_data float32 NCHW contiguous [B,C,H,W]; bottom_rois float32
 * [R,5] = (batch_idx, x1, y1, x2, y2); top_data float32 [R,C,PH,PW], R = outputElements / (C*PH*PW). */
int launch_roi_align_forward_hip(const int outputElements, const float* bottom_data, const float* bottom_rois,
                                 const float spatial_scale, const int channels, const int height, const int width,
                                 const int pooled_height, const int pooled_width, const int sampling_ratio,
                                 float* top_data, dtc_stream_t stream);

/* One feature map ("level").  Strides are in ELEMENTS, so NCHW-contiguous and channels_last (NHWC) tensors of the same
 * logical shape are both accepted without a copy. */
typedef struct dtc_feat_level {
  const void* data;
  int32_t height, width;
  float spatial_scale;
  int32_t _pad;
  int64_t stride_n, stride_c, stride_h, stride_w;
} dtc_feat_level;

#define DTC_MAX_LEVELS 8

/* Multi-level RoIAlign in ONE launch: replaces the per-level Python loop + torch.cat + index_select of
 * lib/model/detector.py:263-270 and lib/model/detector.py:101-106.  rois float32 [R,roi_cols] (roi_cols 5, or 4 =
 * batch 0 like lib/cppcuda/roi_align_cpu.cpp:143-147); roi_levels int32 [R] = index into levels[] (NULL: level 0);
 * out [R,C,PH,PW] contiguous, written in roi order.  in_dtype/out_dtype: DTC_F32, DTC_F16 or DTC_BF16 (always fp32 accumulate; f16 and bf16 do not mix). */
int dtc_roi_align_forward(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype, const float* rois,
                          int roi_cols, const int32_t* roi_levels, int n_rois, int pooled_h, int pooled_w,
                          int sampling_ratio, void* out, int out_dtype, dtc_stream_t stream);

/* Same, with an explicit PROCESSING order: workgroup i pools RoI roi_order[i] (int32 [R], a permutation; NULL = identity).
 * The output row of a RoI does not change.  RoIs arrive in score order (spatially random); visiting them sorted by
 * (image, level, row) makes concurrently running workgroups read the same band of the feature map, so window re-reads
 * hit the XCD's L2 instead of HBM.  dtc_fpn_collect_distribute emits such an order. */
int dtc_roi_align_forward_ordered(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                  const float* rois, int roi_cols, const int32_t* roi_levels, const int32_t* roi_order,
                                  int n_rois, int pooled_h, int pooled_w, int sampling_ratio, void* out, int out_dtype,
                                  dtc_stream_t stream);

/* Same, driven by packed descriptors: roi_desc float32 [R,8] = (batch, x1, y1, x2, y2, level, output_row, 0), one row per
 * workgroup in visiting order (level < 0: padding row, its output row is zero-filled).  Saves the three dependent global
 * loads (order -> level -> roi) at the head of every workgroup.
 * Kernel selection (all produce bit-identical results): sampling_ratio 2 on NCHW maps -> cluster-stationary kernel; ONE level
 * whose whole map of 8 channels fits LDS (H*W <= ~4900 pixels: the C4 heads) with any other sampling ratio -> map-stationary
 * kernel, which stages the map of an image once per run of RoIs and therefore wants the descriptors IMAGE-MAJOR (what
 * dtc_fpn_collect_distribute emits; any order is correct, every change of image re-stages the map).  The plain entries take the
 * map-stationary kernel only for 4-column RoIs (one image). */
int dtc_roi_align_forward_packed(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                 const float* roi_desc, int n_rois, int pooled_h, int pooled_w, int sampling_ratio, void* out,
                                 int out_dtype, dtc_stream_t stream);

/* dtc_roi_align_forward_packed with a caller-owned workspace (dtc_roi_align_workspace_bytes(n_rois) bytes, 16-byte aligned):
 * lets the map-stationary kernel (the C4 heads) form everything about a RoI that does not depend on the channels -- scaled box,
 * bin sizes, adaptive grid, the axis samples of lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:36-95 -- ONCE per launch in a
 * preparation kernel instead of once per (RoI, 8-channel workgroup).  Same results bit for bit; configurations that do not
 * take that kernel ignore the workspace (NULL is allowed: then this IS dtc_roi_align_forward_packed). */
size_t dtc_roi_align_workspace_bytes(int n_rois);

/* Exactness switch of the adaptive-sampling (sampling_ratio <= 0) single-level path.  SEMANTICS: one process-wide value shared by
 * every thread and stream, read by the HOST at launch time and baked into the launch: a launch captured into a hipGraph keeps the
 * mode it was captured with (toggling later does not change replays), and two threads that want different modes must serialise
 * their launches around the switch themselves.  Leave it at 1 unless the whole process opts into the approximate mode.
 * exact = 1 (default): the reference's float32 operations in the reference's order (roi_align_cpu_loop.cpp:203-216): bit-identical.
 * exact = 0: with a workspace (dtc_roi_align_forward_packed_ws) the kernel may merge the gh x gw samples x 4 taps of a bin into
 * (gh + 1) x (gw + 1) taps with separable weight sums: the same sum in exact arithmetic, <= 1e-5 away in float32 on O(1) features
 * (BASELINE.json's tolerance for pooled features is 1e-4), about a quarter faster on the C4 heads. */
void dtc_roi_align_set_exact(int exact);
int dtc_roi_align_get_exact(void);
int dtc_roi_align_forward_packed_ws(const dtc_feat_level* levels, int n_levels, int channels, int in_dtype,
                                    const float* roi_desc, int n_rois, int pooled_h, int pooled_w, int sampling_ratio, void* out,
                                    int out_dtype, void* workspace, size_t workspace_bytes, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A5  Hard NMS
 * --------------------------------------------------------------------------------------------------------------- */

/* Drop-in for cython_nms.nms(dets, thresh)  lib/utils_cython/cython_nms.pyx:37-87 (entry lib/utils/boxes.py:332-336).
 * dets float32 [n,5] = (x1,y1,x2,y2,score) on the device; keep_out int64 [n] receives the ASCENDING ORIGINAL indices of
 * the survivors (np.where(suppressed == 0)[0], :87), keep_count int32 [1] their number.  Order inside the greedy loop is
 * (score descending, index ascending).  n <= 16384. */
size_t dtc_nms_workspace_bytes(int n);
int dtc_nms(const float* dets, int n, float thresh, void* workspace, size_t workspace_bytes, int64_t* keep_out,
            int32_t* keep_count, dtc_stream_t stream);

/* Segmented NMS over score-SORTED boxes: one launch for all (image, level) / (image, class) segments -- replaces the
 * per-level loop of lib/model/detector.py:252 + generate_proposals.py:115-117 and the 80-iteration loop of
 * lib/utils/result_utils.py:126-143.  boxes float32 [n_seg, n_stride, 4] sorted by score descending inside each
 * segment; counts int32 [n_seg] (NULL: all n_stride valid; a NEGATIVE count marks a segment that is already reduced: its
 * keep / keep_count are left untouched).  keep int32 [n_seg, keep_stride] receives the kept
 * POSITIONS in score order, at most max_keep (>0) of them (== keep[:post_nms_top_n]); keep_count int32 [n_seg]. */
size_t dtc_nms_sorted_workspace_bytes(int n_seg, int n_stride);
int dtc_nms_sorted(const float* boxes, const int32_t* counts, int n_seg, int n_stride, float thresh, int max_keep,
                   void* workspace, size_t workspace_bytes, int32_t* keep, int keep_stride, int32_t* keep_count,
                   dtc_stream_t stream);

/* Segmented (score descending, index ascending) sort -- scores.argsort()[::-1] of cython_nms.pyx:45 with the canonical
 * tie rule.  scores [n_seg, n_stride] read with element stride score_stride_elems (5 for a dets array); optional
 * gather of boxes (element stride box_stride_elems between boxes) into sorted_boxes [n_seg, n_stride, 4]. */
int dtc_segment_sort_desc(const float* scores, int score_stride_elems, const float* boxes, int box_stride_elems,
                          const int32_t* counts, int n_seg, int n_stride, int32_t* order, float* sorted_boxes,
                          float* sorted_scores, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A2 + A3 + A4  RPN proposals: anchors, pre-NMS top-k, decode, clip, filter -- all images, all FPN levels
 * --------------------------------------------------------------------------------------------------------------- */

#define DTC_RPN_MAX_LEVELS 8
#define DTC_RPN_MAX_ANCHORS 16

/* One RPN head output (one FPN level, or the single C4 map) for a whole batch, in the conv's native layout:
 * cls_prob float32 [B, A, H, W] (post-sigmoid, lib/model/detector.py:125), bbox_pred float32 [B, 4A, H, W].
 * anchors = the A base anchors of generate_anchors(stride=feat_stride, ...) (lib/utils/generate_anchors.py:54-65),
 * row-major [A,4]; feat_stride = 1/spatial_scale (generate_proposals.py:130). pre_nms_top_n <= 0 means "all". */
typedef struct dtc_rpn_level {
  const float* cls_prob;
  const float* bbox_pred;
  int32_t num_anchors, height, width;
  int32_t pre_nms_top_n;
  float feat_stride;
  int32_t score_is_logit;   /* 0: cls_prob holds probabilities (the reference contract).  1: it holds the PRE-sigmoid logits
                             * of detector.py:125; ranking, tie-breaking and the emitted scores are those of
                             * sigmoid(logit) evaluated in double and rounded once -- the probability map is never
                             * materialised.  Field was padding before: zero-initialised callers are unchanged. */
  float anchors[DTC_RPN_MAX_ANCHORS * 4];
} dtc_rpn_level;

/* Steps 1-5 of GenerateProposals.forward (lib/model/generate_proposals.py:31-109) for every (image, level) segment
 * s = b * n_levels + l.  Outputs, per segment, in DESCENDING score order (ties: ascending (h,w,a) index):
 *   out_boxes float32 [B*L, k_stride, 4], out_scores float32 [B*L, k_stride], out_counts int32 [B*L]
 * i.e. exactly the `dets` the reference hands to NMS at :115.  k_stride >= max_l min(pre_nms_top_n_l, N_l) (0: that max).
 * (im_h, im_w) is the network input size (:100), min_size_scaled = rpn_min_size * scaling_factor (:154).
 * The workspace must be EXCLUSIVE to one in-flight call: it holds the histograms, tickets and block counts the kernels of the call
 * hand to each other (calls on different streams need different workspaces; calls on one stream may share one). */
size_t dtc_rpn_topk_decode_workspace_bytes(const dtc_rpn_level* levels, int n_levels, int batch, int k_stride);
int dtc_rpn_topk_decode(const dtc_rpn_level* levels, int n_levels, int batch, float im_h, float im_w,
                        float min_size_scaled, void* workspace, size_t workspace_bytes, float* out_boxes,
                        float* out_scores, int32_t* out_counts, int k_stride, dtc_stream_t stream);

/* proposals[keep] / scores[keep] of generate_proposals.py:119-120 for every segment: out_boxes [n_seg, keep_stride, 4],
 * out_scores [n_seg, keep_stride] (rows >= keep_count[s] untouched). */
int dtc_gather_kept(const float* sorted_boxes, const float* sorted_scores, int n_seg, int k_stride, const int32_t* keep,
                    const int32_t* keep_count, int keep_stride, float* out_boxes, float* out_scores,
                    dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A7  FPN collect + distribute
 * --------------------------------------------------------------------------------------------------------------- */

/* collect() + distribute() of lib/model/collect_and_distribute_fpn_rpn_proposals.py:84-128 and
 * map_rois_to_fpn_levels of lib/utils/multilevel_rois.py:41-53, one workgroup per image.
 *   in_boxes float32 [B, n_in_levels, in_stride, 4], in_scores float32 [B, n_in_levels, in_stride] (post-NMS RPN
 *   proposals per level, rows >= in_counts[b,l] ignored).  in_scores == NULL: no sort, the rows are taken in the given
 *   order (add_multilevel_rois_for_test, lib/utils/multilevel_rois.py:19-39 -- the mask branch).
 * Outputs (post_nms_top_n rows per image; rows >= n_out[b] are padding with level -1):
 *   rois5 [B,topN,5] = (b,x1,y1,x2,y2) in collected (score) order; roi_scores [B,topN] (nullable);
 *   roi_levels int32 [B,topN] = level - k_min; n_out int32 [B];
 *   rois_by_level [B,topN,4] + level_counts int32 [B,k_max-k_min+1] = the reference's per-level lists, concatenated;
 *   idx_restore int32 [B,topN] = the reference's rois_idx_restore (:127);
 *   roi_order int32 [B,topN] (nullable) = global row ids b*topN + r sorted by (level, y centre): the visiting order for
 *   dtc_roi_align_forward_ordered (a performance hint, not part of the reference's semantics);
 *   roi_desc float32 [B,topN,8] (nullable) = the same rois packed in visiting order as (batch,x1,y1,x2,y2,level,row,0) for
 *   dtc_roi_align_forward_packed.
 * inputs_sorted != 0 promises that every input list is already in descending score order (true for NMS output): the
 * lists are then merged by rank instead of sorted. */
int dtc_fpn_collect_distribute(const float* in_boxes, const float* in_scores, const int32_t* in_counts, int batch,
                               int n_in_levels, int in_stride, int post_nms_top_n, int k_min, int k_max, float* rois5,
                               float* roi_scores, int32_t* roi_levels, int32_t* n_out, float* rois_by_level,
                               int32_t* level_counts, int32_t* idx_restore, int32_t* roi_order, float* roi_desc,
                               int inputs_sorted, dtc_stream_t stream);

/* collect + distribute straight from the NMS output: list l of image b = rows keep[(b * L + l) * keep_stride + j], j < keep_count, of the
 * score-sorted pre-NMS arrays sorted_boxes / sorted_scores [B * L, k_stride, ...] (what dtc_rpn_topk_decode and dtc_nms_sorted leave
 * behind) -- proposals[keep], scores[keep] of generate_proposals.py:119-120 read in place, i.e. dtc_gather_kept +
 * dtc_fpn_collect_distribute(inputs_sorted = 1) in one launch, the same outputs bit for bit.  2 <= n_in_levels; shapes the merge
 * kernel does not hold (post_nms_top_n > 2048, keep_stride > 1024, n_in_levels * keep_stride > 8192) return DTC_EUNSUPPORTED: use the
 * two calls. */
int dtc_fpn_collect_distribute_kept(const float* sorted_boxes, const float* sorted_scores, int k_stride, const int32_t* keep,
                                    const int32_t* keep_count, int keep_stride, int batch, int n_in_levels, int post_nms_top_n,
                                    int k_min, int k_max, float* rois5, float* roi_scores, int32_t* roi_levels, int32_t* n_out,
                                    float* rois_by_level, int32_t* level_counts, int32_t* idx_restore, int32_t* roi_order,
                                    float* roi_desc, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A8  Detection post-processing
 * --------------------------------------------------------------------------------------------------------------- */

/* postprocess_output + box_results_with_nms_and_limit (hard NMS) of lib/utils/result_utils.py:76-168 for a batch.
 *   rois5 [B,R,5] (network-scale rois, column 0 ignored), n_rois int32 [B] (NULL: R), cls_score [B,R,n_cls] (softmax),
 *   bbox_pred [B,R,4*n_cls], scaling_factor float32 [B], im_size float32 [B,2] = original (h,w).
 *   (wx,wy,ww,wh) = bbox_reg_weights (10,10,5,5); score_thresh 0.05; nms_thresh 0.5; max_det 100 (0: unlimited).
 * Outputs: dets [B,max_out,6] = (x1,y1,x2,y2,score,class) ordered by class then by roi index (== np.vstack(cls_boxes),
 * :165); det_roi int32 [B,max_out] source roi; det_rois_scaled [B,max_out,4] = boxes * scaling_factor (nullable; the
 * rois of the mask branch, eval_mask_FPN.ipynb:249); det_count int32 [B] = true number (can exceed max_det on score
 * ties like the reference, :161; rows beyond max_out are dropped -- compare det_count with max_out). R <= 4096.
 * Cost model: the per-class NMS of a (class, image) segment runs inside ONE workgroup, n^2 / 2 pair tests for n candidates above the
 * score threshold -- microseconds for the 81-class heads this serves (tens of candidates per class), ~100 us at n = 1000, ~1.5 ms at
 * n = 4096: few-class models with low thresholds should expect the largest class segment to set the launch time. */
size_t dtc_postprocess_detections_workspace_bytes(int batch, int max_rois, int n_cls);
int dtc_postprocess_detections(const float* rois5, const int32_t* n_rois, const float* cls_score, const float* bbox_pred,
                               const float* scaling_factor, const float* im_size, int batch, int max_rois, int n_cls,
                               float wx, float wy, float ww, float wh, float score_thresh, float nms_thresh, int max_det,
                               void* workspace, size_t workspace_bytes, float* dets, int32_t* det_roi,
                               float* det_rois_scaled, int32_t* det_count, int max_out, dtc_stream_t stream);

/* Same with the box head's softmax folded in (lib/model/detector.py:281 F.softmax(cls_score) -> result_utils.py:76-94):
 * cls_logits [B,R,n_cls] is the raw output of the cls_score layer.  One wave per roi reduces (max, sum exp) -- 16 bytes per
 * roi in the workspace -- and the probability of a (roi, class) is formed where the class column is scanned:
 * float(exp(double(l) - max) / sum), rounded once (rel 1e-6 of torch's float32 softmax).  The [R,n_cls] probability
 * map is never written.  Everything else as dtc_postprocess_detections; same workspace size. */
int dtc_postprocess_detections_logits(const float* rois5, const int32_t* n_rois, const float* cls_logits,
                                      const float* bbox_pred, const float* scaling_factor, const float* im_size, int batch,
                                      int max_rois, int n_cls, float wx, float wy, float ww, float wh, float score_thresh,
                                      float nms_thresh, int max_det, void* workspace, size_t workspace_bytes, float* dets,
                                      int32_t* det_roi, float* det_rois_scaled, int32_t* det_count, int max_out,
                                      dtc_stream_t stream);

/* dtc_postprocess_detections[_logits] that ALSO emits the FPN level mapping of the detection rows for the mask branch
 * (add_multilevel_rois_for_test, lib/utils/multilevel_rois.py:19-39; call site eval_mask_FPN.ipynb:249) -- what a following
 * dtc_fpn_collect_distribute(det_rois_scaled, in_scores = NULL, det_count, batch, 1, max_out, max_out, k_min, k_max, ...) would
 * write, bit for bit, without that launch (the workgroup that finalises an image's detections has its <= ~100 rows in hand).
 * All of fpn's pointers are [B, max_out, ...] buffers as documented at dtc_fpn_collect_distribute (roi_order / roi_desc nullable);
 * det_rois_scaled is required; max_out <= 512 (DTC_EUNSUPPORTED beyond: use the separate call). */
typedef struct dtc_fpn_map_out {
  float* rois5; int32_t* roi_levels; int32_t* n_out; float* rois_by_level; int32_t* level_counts; int32_t* idx_restore;
  int32_t* roi_order; float* roi_desc;
  int32_t k_min, k_max;
} dtc_fpn_map_out;
int dtc_postprocess_detections_fpn(const float* rois5, const int32_t* n_rois, const float* cls_score, int scores_are_logits,
                                   const float* bbox_pred, const float* scaling_factor, const float* im_size, int batch,
                                   int max_rois, int n_cls, float wx, float wy, float ww, float wh, float score_thresh,
                                   float nms_thresh, int max_det, void* workspace, size_t workspace_bytes, float* dets,
                                   int32_t* det_roi, float* det_rois_scaled, int32_t* det_count, int max_out,
                                   const dtc_fpn_map_out* fpn, dtc_stream_t stream);

/* box_results_with_nms_and_limit (hard NMS) of lib/utils/result_utils.py:96-168 on ALREADY decoded + clipped boxes, for a
 * batch, in one pass on the device (the reference: 80 Python iterations, each a host NMS call):
 *   scores [B,R,n_cls], boxes [B,R,4*n_cls] (class j in columns 4j..4j+3), n_rois int32 [B] (NULL: R).
 * Per class j >= 1: score > score_thresh (:127), NMS at nms_thresh (:142); if more than max_det survive in total, those
 * with score >= the max_det-th largest (:159-163; max_det 0: unlimited).  Outputs as dtc_postprocess_detections (dets
 * [B,max_out,6] ordered by class then by row, det_roi = source row, det_count = true number).  Same workspace size. */
int dtc_box_results_nms_limit(const float* scores, const float* boxes, const int32_t* n_rois, int batch, int max_rois,
                              int n_cls, float score_thresh, float nms_thresh, int max_det, void* workspace,
                              size_t workspace_bytes, float* dets, int32_t* det_roi, int32_t* det_count, int max_out,
                              dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A9  Mask resize + binarise (+ paste geometry)
 * --------------------------------------------------------------------------------------------------------------- */

/* The per-detection body of segm_results, lib/utils/result_utils.py:182-214 (everything but the RLE encode).
 *   masks float32 [n_masks,n_cls,M,M] (mask-head output); mask_index int32 [B,max_out] row of each detection in masks
 *   (NULL: b*max_out + d); dets [B,max_out,6] / det_count [B] as produced above; im_size [B,2] original (h,w).
 * Outputs: mask_boxes int32 [B,max_out,4] expanded+truncated ref box (:183-184); mask_rects int32 [B,max_out,4] paste
 * rectangle (x_0,y_0,x_1,y_1) (:204-207); crops uint8 [B,per_image_capacity]: for detection d the binarised resized mask
 * restricted to its paste rectangle, row-major, at byte mask_offsets[b,d] of image b's region; mask_bytes int64 [B] =
 * bytes image b needs (if > per_image_capacity the detections that did not fit were skipped). */
int dtc_mask_paste(const float* masks, const int32_t* mask_index, int n_cls, int M, const float* dets,
                   const int32_t* det_count, const float* im_size, int batch, int max_out, float thresh_binarize,
                   int cls_specific_mask, uint8_t* crops, long long per_image_capacity, int32_t* mask_boxes,
                   int32_t* mask_rects, long long* mask_offsets, long long* mask_bytes, dtc_stream_t stream);

/* COCO RLE of every pasted mask, on the device: what `mask_util.encode(np.array(im_mask[:, :, np.newaxis], order='F'))`
 * returns at lib/utils/result_utils.py:217-220 (pycocotools rleEncode + rleToString), computed from dtc_mask_paste's
 * crops / mask_rects / mask_offsets without materialising the (im_h, im_w) frame.  Per detection (b, d):
 *   rle_counts uint32 [B,max_out,runs_stride]  run lengths (column-major, first run = zeros), rle_n_runs int32 [B,max_out]
 *   rle_str    uint8  [B,max_out,str_stride]   the compressed "counts" string (ASCII, no terminator), rle_str_len int32
 * A detection whose runs (string) do not fit gets rle_n_runs = -(runs needed) (rle_str_len = -(bytes needed)) and no
 * valid data: re-run with larger strides or encode that one on the host.  A detection whose crop did not fit
 * per_image_capacity (dtc_mask_paste skipped it: mask_bytes[b] > capacity) gets -1 / -1.  d >= det_count[b]: 0 / 0. */
int dtc_mask_rle(const uint8_t* crops, long long per_image_capacity, const int32_t* mask_rects,
                 const long long* mask_offsets, const int32_t* det_count, const float* im_size, int batch, int max_out,
                 uint32_t* rle_counts, int runs_stride, int32_t* rle_n_runs, uint8_t* rle_str, int str_stride,
                 int32_t* rle_str_len, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (f)-4  bbox_overlaps / box_voting (the optional bbox-vote branch of the detection post-processing)
 * --------------------------------------------------------------------------------------------------------------- */

/* Drop-in for cython_bbox.bbox_overlaps(boxes, query_boxes)  lib/utils_cython/cython_bbox.pyx:32-72:
 * boxes float32 [n, box_cols>=4], query_boxes [k, query_cols>=4] (first four columns x1,y1,x2,y2) -> overlaps [n,k]. */
int dtc_bbox_overlaps(const float* boxes, int n, int box_cols, const float* query_boxes, int k, int query_cols,
                      float* overlaps, dtc_stream_t stream);

/* box_voting(top_dets, all_dets, thresh, scoring_method='ID')  lib/utils/boxes.py:280-329: top_dets [n_top,5],
 * all_dets [n_all,5] (x1,y1,x2,y2,score), n_all <= 8192 -> top_dets_out [n_top,5] with the boxes replaced by the
 * score-weighted average of the all_dets whose IoU with the top det is >= thresh (numpy's float32 evaluation order);
 * n_voters int32 [n_top] (optional) = number of voters (0: row copied unchanged; cannot happen for the reference's
 * call, where every top det is one of all_dets). */
int dtc_box_voting(const float* top_dets, int n_top, const float* all_dets, int n_all, float thresh,
                   float* top_dets_out, int32_t* n_voters, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * OUTSIDE SURVEY section 8 (the region-proposal hot path): convolution epilogue of the backbone's inference form.
 * SURVEY 2 row 7 marks the backbone out of scope; this entry exists because detector.optimize_for_inference (round 3)
 * uses it.  It is frozen: no parity / roofline claim of this library rests on it, and a binding of the hot path does
 * not need it.
 * --------------------------------------------------------------------------------------------------------------- */

/* x [n,c,h,w] (dense NCHW, or dense channels_last when channels_last != 0; DTC_F32 / DTC_F16 / DTC_BF16), in place:
 *     x = act( x + bias[c] + residual )        float32 arithmetic, one rounding to x's type
 * bias float32 [c] or NULL; residual NULL, or a tensor of x's type and layout -- [n,c,h,w], or [n,c,h/2,w/2] read with
 * nearest-neighbour x2 upsampling when residual_up2 != 0 (h, w even); relu != 0: act = max(., 0) (NaN propagates).
 * One pass replaces what the reference's graph runs as separate ops after a convolution: the eval-mode BatchNorm /
 * AffineChannel of the ResNet body (detector.py:231; its scale folded into the weights, its shift = bias) + ReLU, the
 * bottleneck's `+ identity`, the FPN top-down `upsample(top) + lateral` (detector.py:45-46), the heads' bias + ReLU. */
int dtc_bias_act(void* x, const float* bias, const void* residual, int n, int c, int h, int w, int dtype,
                 int channels_last, int relu, int residual_up2, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * (f)-3  Network-input preparation: prep_im_for_blob + im_list_to_blob  (lib/utils/blob.py:62-87, :27-59)
 * --------------------------------------------------------------------------------------------------------------- */

/* One source image: interleaved HWC, 3 channels in BGR order (as cv2.imread delivers), DTC_U8 or DTC_F32, on the device;
 * row_stride in elements (>= 3 * width). */
typedef struct dtc_image {
  const void* data;
  int32_t height, width, dtype, row_stride;
} dtc_image;

/* Host-side plan (no GPU work): per image the scale of blob.py:75-82 (target_size on the short side, capped so that the
 * long side stays <= max_size), the resized (h, w) cv2.resize(fx=fy=scale) produces, and the blob (H, W) of
 * im_list_to_blob: max over the batch, rounded up to pad_stride when pad_stride > 1 (fpn_on, :41-44).
 * im_scales double [batch], out_hw int32 [batch,2], blob_hw int32 [2] -- all host memory. */
int dtc_prep_plan(const int32_t* heights, const int32_t* widths, int batch, int target_size, int max_size, int pad_stride,
                  double* im_scales, int32_t* out_hw, int32_t* blob_hw);

/* blob float32 [batch,3,blob_h,blob_w] (device) = for every image: (image - pixel_means) resized bilinearly by
 * im_scales[b] to out_hw[b], zero padded, channels first.  images / pixel_means (3 doubles, BGR) / im_scales / out_hw are
 * HOST arrays (the plan above); batch <= 32 per call. */
int dtc_prep_images(const dtc_image* images, int batch, const double* pixel_means, const double* im_scales,
                    const int32_t* out_hw, float* blob, int blob_h, int blob_w, dtc_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * A6  Soft-NMS  and  A4 (numpy flavour) box decode
 * --------------------------------------------------------------------------------------------------------------- */

/* Drop-in for cython_nms.soft_nms(boxes_in, sigma, Nt, threshold, method)  lib/utils_cython/cython_nms.pyx:98-203 (entry
 * lib/utils/boxes.py:339-356; method 0 hard / 1 linear / 2 gaussian).  dets float32 [n,5] on the device (not modified);
 * dets_out [n,5] and inds_out int64 [n] receive the N' surviving rows in SELECTION order, n_out int32 [1] = N'. n <= 6000. */
int dtc_soft_nms(const float* dets, int n, float sigma, float overlap_thresh, float score_thresh, int method,
                 float* dets_out, int64_t* inds_out, int32_t* n_out, dtc_stream_t stream);

/* bbox_transform (lib/utils/boxes.py:168-208) optionally followed by clip_tiled_boxes (:150-165): boxes [n,4],
 * deltas [n,4*n_cls] -> out [n,4*n_cls]. */
int dtc_bbox_transform(const float* boxes, const float* deltas, int n, int n_cls, float wx, float wy, float ww, float wh,
                       int do_clip, float im_h, float im_w, float* out, dtc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DETECTORCH_HIP_H_ */
