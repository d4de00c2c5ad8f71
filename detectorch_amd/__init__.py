"""detectorch_amd -- MI355X-native (gfx950 / CDNA4) region-proposal hot path for detectorch.

Drop-in surface (same names / argument meaning as /root/reference/lib):
    detectorch_amd.model.roi_align          RoIAlignFunction, RoIAlign, preprocess_rois
    detectorch_amd.model.generate_proposals GenerateProposals
    detectorch_amd.model.collect_and_distribute_fpn_rpn_proposals  CollectAndDistributeFpnRpnProposals
    detectorch_amd.model.detector           detector
    detectorch_amd.utils.boxes              nms, soft_nms, bbox_transform, clip_tiled_boxes, ...
    detectorch_amd.utils.result_utils       postprocess_output, box_results_with_nms_and_limit, segm_results
A notebook that did `sys.path.insert(0, "lib/")` switches by inserting this package directory instead.

The compute lives in detectorch_amd/csrc (hand-written HIP, C ABI declared in include/detectorch_hip.h) and is reached
through detectorch_amd.hip (ctypes).  There is no CPU fallback: importing detectorch_amd.hip without the built
library raises.
"""
__version__ = "0.1.0"
