"""detector -- same constructor / forward / mask_head call surface as the reference's lib/model/detector.py:129-286, with
the region-proposal hot path (GenerateProposals, collect/distribute, multi-level RoIAlign) running on the HIP kernels.

What stays plain PyTorch-ROCm (MIOpen / hipBLASLt, "the real dense contractions"): the ResNet-50/101 body, FPN lateral /
output convs, the RPN head convs, fc6/fc7/cls/bbox linears, the mask-head convs.  torchvision is not required: the
ResNet is re-declared here with torchvision-compatible parameter names (so the caffe2 blob-name mapping of
lib/utils/utils.py:44-71 still applies) and the caffe2 stride placement of detector.py:174-179 (stride 2 on the first
1x1 conv of layer2/3/4).

Differences from the reference:
  * the FPN path issues ONE batched GenerateProposals call for all 5 levels and ONE multi-level RoIAlign launch whose
    output is already in collected (score) order -- no per-level Python loop, no torch.cat + index_select
    (detector.py:251-270);
  * `channels_last=True` keeps the backbone features NHWC (same logical shape), the layout the RoIAlign kernel reads
    with full cache lines;
  * `forward_batched(images[B], ...)` (not in the reference, which is batch-1: eval_mask_FPN.ipynb:93) runs backbone ->
    region path -> heads -> detections -> mask branch for B images with no host round trip, on the same fused batched
    kernels bench.py measures (detectorch_amd.pipeline.FpnRegionPath); `forward` stays the reference's batch-1 call;
  * `head_dtype=torch.bfloat16`: RoIAlign writes the pooled features in bf16 and fc6/fc7 run as bf16 MFMA GEMMs;
  * inference only (the reference's README.md:3 scope).
"""
import contextlib
import pickle
from math import log2

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from .collect_and_distribute_fpn_rpn_proposals import CollectAndDistributeFpnRpnProposals
from .generate_proposals import GenerateProposals
from .roi_align import RoIAlignFunction, preprocess_rois


# ---- ResNet body (torchvision-compatible names) ------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # caffe2 / Detectron put the stride on the first 1x1 conv (detector.py:174-179)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.fused = False          # optimize_for_inference(): BatchNorms folded into the convs, shifts in eb1 / eb2 / eb3

    def fold_(self):
        """Eval-mode BatchNorm (= caffe2 AffineChannel) folded into the preceding bias-free conv: w' = w * g / sqrt(var + eps)
        per output channel, shift = b - mean * g / sqrt(var + eps) -> float32 epilogue bias (the downsample branch's shift
        joins bn3's: both are added before the last ReLU)."""
        b1 = _fold_bn_(self.conv1, self.bn1)
        b2 = _fold_bn_(self.conv2, self.bn2)
        b3 = _fold_bn_(self.conv3, self.bn3)
        if self.downsample is not None:
            b3 = b3 + _fold_bn_(self.downsample[0], self.downsample[1])
        self.register_buffer("eb1", b1); self.register_buffer("eb2", b2); self.register_buffer("eb3", b3)
        self.fused = True

    def forward(self, x):
        if self.fused:      # conv -> one in-place pass (bias, residual, ReLU) instead of BatchNorm, add and ReLU kernels
            out = hip.bias_act_(self.conv1(x), self.eb1)
            out = hip.bias_act_(self.conv2(out), self.eb2)
            idt = x if self.downsample is None else self.downsample[0](x)
            return hip.bias_act_(self.conv3(out), self.eb3, residual=idt)
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


@torch.no_grad()
def _fold_bn_(conv, bn):
    """Scale conv.weight by the BatchNorm's per-channel factor in place; returns the float32 shift."""
    g = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    w = conv.weight.data
    fmt = torch.channels_last if w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous() else torch.contiguous_format
    conv.weight.data = (w.float() * g.reshape(-1, 1, 1, 1)).to(w.dtype).contiguous(memory_format=fmt)
    shift = bn.bias.float() - bn.running_mean.float() * g
    if conv.bias is not None:
        shift = shift + conv.bias.float() * g
        conv.bias = None
    return shift.contiguous()


class _Epilogue(nn.Module):
    """bias (+ ReLU) of the preceding bias-free conv as one in-place pass (hip.bias_act_)."""

    def __init__(self, bias, relu):
        super().__init__()
        self.register_buffer("ebias", bias)
        self.relu = relu

    def forward(self, x):
        return hip.bias_act_(x, self.ebias, relu=self.relu)


def _conv_nobias(m, x):
    """m's convolution without its bias (the bias goes into the fused epilogue)."""
    if isinstance(m, nn.ConvTranspose2d):
        return F.conv_transpose2d(x, m.weight, None, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    return F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)


def _ebias(m):
    """float32 copy of m.bias for the epilogue (kept as a buffer: survives .to(dtype) of the parameters only)."""
    if not hasattr(m, "ebias"):
        m.register_buffer("ebias", m.bias.detach().float().contiguous())
    return m.ebias


class ResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, layers[0], 1)
        self.layer2 = self._make(128, layers[1], 2)
        self.layer3 = self._make(256, layers[2], 2)
        self.layer4 = self._make(512, layers[3], 2)
        self.avgpool = nn.AvgPool2d(7, stride=1)

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


_ARCH = {'resnet50': [3, 4, 6, 3], 'resnet101': [3, 4, 23, 3], 'resnet152': [3, 8, 36, 3]}


# ---- heads (detector.py:12-127) ---------------------------------------------------------------------------------------
class fpn_body(nn.Module):
    def __init__(self, conv_body, conv_body_layers, fpn_layers):
        super().__init__()
        self.conv_body = conv_body
        chans = [conv_body[conv_body_layers.index(l)][-1].bn3.num_features for l in fpn_layers]
        self.fpn_lateral = nn.ModuleList([nn.Conv2d(c, 256, 1) for c in chans])
        self.fpn_output = nn.ModuleList([nn.Conv2d(256, 256, 3, padding=1) for _ in chans])
        self.fpn_indices = [conv_body_layers.index(l) for l in fpn_layers]
        self.fpn_layers = fpn_layers
        self.fused = False

    def forward(self, x):
        lateral = []
        for i in range(len(self.conv_body)):
            x = self.conv_body[i](x)
            if i in self.fpn_indices:
                lateral.append(x)
        if self.fused and all(lateral[i].shape[2] == 2 * lateral[i + 1].shape[2] and lateral[i].shape[3] == 2 * lateral[i + 1].shape[3]
                              for i in range(len(lateral) - 1)):
            # lateral 1x1 conv | one pass: + bias + nearest-x2 upsampled level above (detector.py:45-46) | 3x3 conv | + bias
            n = len(lateral)
            lat = [None] * n
            for i in range(n - 1, -1, -1):
                lat[i] = hip.bias_act_(_conv_nobias(self.fpn_lateral[i], lateral[i]), _ebias(self.fpn_lateral[i]),
                                       residual=lat[i + 1] if i < n - 1 else None, relu=False, residual_up2=i < n - 1)
            return [hip.bias_act_(_conv_nobias(self.fpn_output[i], lat[i]), _ebias(self.fpn_output[i]), relu=False) for i in range(n)]
        lateral = [self.fpn_lateral[i](lateral[i]) for i in range(len(lateral))]
        for i in range(len(lateral) - 2, -1, -1):                         # top-down, nearest x2 (detector.py:45-46)
            lateral[i] = F.interpolate(lateral[i + 1], scale_factor=2, mode='nearest') + lateral[i]
        return [self.fpn_output[i](lateral[i]) for i in range(len(lateral))]


class two_layer_mlp_head(nn.Module):
    def __init__(self):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.fc6 = nn.Linear(256 * 7 * 7, 1024)
        self.fc7 = nn.Linear(1024, 1024)

    def forward(self, x):
        x = x.reshape(x.size(0), -1)
        return self.relu(self.fc7(self.relu(self.fc6(x))))


class four_layer_conv(nn.Module):
    def __init__(self):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.fcn1 = nn.Conv2d(256, 256, 3, padding=1)
        self.fcn2 = nn.Conv2d(256, 256, 3, padding=1)
        self.fcn3 = nn.Conv2d(256, 256, 3, padding=1)
        self.fcn4 = nn.Conv2d(256, 256, 3, padding=1)

    fused = False

    def forward(self, x):
        for m in (self.fcn1, self.fcn2, self.fcn3, self.fcn4):
            x = hip.bias_act_(_conv_nobias(m, x), _ebias(m)) if self.fused else self.relu(m(x))
        return x


class mask_head(nn.Module):
    def __init__(self, conv_head, roi_spatial_scale, roi_sampling_ratio, output_prob):
        super().__init__()
        self.output_prob = output_prob
        self.conv_head = conv_head
        self.transposed_conv = nn.ConvTranspose2d(256 if isinstance(conv_head, four_layer_conv) else 2048, 256, 2, stride=2)
        self.classif_logits = nn.Conv2d(256, 81, 1)
        self.relu = nn.ReLU(inplace=True)
        self.use_fpn = isinstance(roi_spatial_scale, list)
        self.roi_spatial_scale = roi_spatial_scale
        self.roi_sampling_ratio = roi_sampling_ratio
        self.roi_height = 14
        self.roi_width = 14

    @torch.no_grad()
    def forward(self, x, rois, roi_original_idx=None):
        # detector.py:99-112
        if not self.use_fpn:
            x = RoIAlignFunction.apply(x, preprocess_rois(rois), self.roi_height, self.roi_width,
                                       self.roi_spatial_scale, self.roi_sampling_ratio)
        else:
            # one multi-level launch over the per-level lists; output rows follow the concatenation order, then the
            # caller's restore permutation is applied exactly like :105-106
            lv, rs = [], []
            for i, r in enumerate(rois):
                if r is None or r.shape[0] == 0:
                    continue
                r = preprocess_rois(r)
                rs.append(r)
                lv.append(torch.full((r.shape[0],), i, dtype=torch.int32, device=r.device))
            x = hip.roi_align_forward(list(x)[:len(self.roi_spatial_scale)], self.roi_spatial_scale, torch.cat(rs, 0),
                                      self.roi_height, self.roi_width, self.roi_sampling_ratio, roi_levels=torch.cat(lv, 0))
            x = x[roi_original_idx, :]
        x = self.conv_head(x)
        x = self.relu(self.transposed_conv(x))
        x = self.classif_logits(x)
        return torch.sigmoid(x) if self.output_prob else x


class rpn_head(nn.Module):
    def __init__(self, in_channels=1024, out_channels=1024, n_anchors=15):
        super().__init__()
        self.conv_rpn = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.rpn_cls_prob = nn.Conv2d(out_channels, n_anchors, 1)
        self.rpn_bbox_pred = nn.Conv2d(out_channels, 4 * n_anchors, 1)

    def forward(self, x, logits=False):
        """(rpn_cls_probs, rpn_bbox_pred) as detector.py:123-127; logits=True skips the sigmoid (it is folded into the
        top-k kernel: GenerateProposals(..., scores_are_logits=True))."""
        if getattr(self, "fused", False):
            c = hip.bias_act_(_conv_nobias(self.conv_rpn, x), _ebias(self.conv_rpn))
        else:
            c = F.relu(self.conv_rpn(x), inplace=True)
        if getattr(self, "fused", False) and logits:
            # 1x1 heads: bias-free conv, then ONE kernel that adds the float32 bias and writes the float32 NCHW tensor the proposal
            # kernels read (instead of bias add, widening copy and layout copy)
            def head(m):
                y = _conv_nobias(m, c)
                return torch.add(y, _ebias(m).view(1, -1, 1, 1), out=torch.empty(y.shape, dtype=torch.float32, device=y.device))
            return head(self.rpn_cls_prob), head(self.rpn_bbox_pred)
        s = self.rpn_cls_prob(c)
        return (s if logits else torch.sigmoid(s)), self.rpn_bbox_pred(c)


def _caffe2_name(key):
    """torchvision-style ResNet parameter name -> caffe2 blob name (same mapping as lib/utils/utils.py:44-71)."""
    t = key.split('.')
    if t[0] == 'conv1':
        return 'conv1_w'
    if t[0] == 'bn1':
        return 'res_conv1_bn_' + ('s' if t[1] == 'weight' else 'b')
    stage = 'res%d_%s' % (int(t[0][-1]) + 1, t[1])
    if t[2] == 'downsample':
        branch = '_branch1'
        kind = 'conv' if t[3] == '0' else 'bn'
    else:
        branch = '_branch2' + 'abc'[int(t[2][-1]) - 1]
        kind = 'conv' if t[2].startswith('conv') else 'bn'
    if kind == 'conv':
        return stage + branch + '_w'
    return stage + branch + ('_bn_s' if t[-1] == 'weight' else '_bn_b')


class detector(nn.Module):
    def __init__(self, train=False, arch='resnet50',
                 conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3'],
                 conv_head_layers=['layer4', 'avgpool'], fpn_layers=[], fpn_extra_lvl=True, use_rpn_head=False,
                 use_mask_head=False, mask_head_type='upshare', roi_feature_channels=2048, N_classes=81,
                 detector_pkl_file=None, base_cnn_pkl_file=None, output_prob=True, roi_height=14, roi_width=14,
                 roi_spatial_scale=0.0625, roi_sampling_ratio=0, channels_last=False, fuse_rpn_sigmoid=True,
                 head_dtype=None, backbone_dtype=None):
        super().__init__()
        self.fuse_rpn_sigmoid = bool(fuse_rpn_sigmoid)   # extension: RPN sigmoid folded into the top-k kernel (same outputs)
        self.head_dtype = head_dtype    # extension: torch.bfloat16 / float16 -> pooled features + fc6/fc7 in that dtype (MFMA GEMMs)
        # extension (forward_batched): torch.bfloat16 / float16 -> ResNet / FPN / RPN-head convs under autocast in that dtype;
        # the feature maps reach RoIAlign as 16-bit tensors (half the bytes per staged line) and, with head_dtype set to the same
        # type, the pooled features reach fc6 without a cast.  The RPN outputs are widened to float32 for the proposal kernels.
        self.backbone_dtype = backbone_dtype
        self.N_classes = N_classes
        self._paths = {}               # (B, padded h, padded w, device) -> FpnRegionPath, least recently used first
        self.max_cached_paths = 4
        # forward_batched: detection rows per image of the fixed-shape results (>= detections_per_im = 100: ties at the image
        # threshold may exceed it, result_utils.py:161; det_count > max_out signals overflow).  The mask head convolves this many
        # rows per image whatever the detection count: optimize_for_inference() lowers it to 104.
        self.max_out = 128
        if train:
            raise NotImplementedError("detectorch_amd.detector is inference-only")
        self.roi_height, self.roi_width = int(roi_height), int(roi_width)
        self.roi_spatial_scale = [float(i) for i in roi_spatial_scale] if isinstance(roi_spatial_scale, list) else float(roi_spatial_scale)
        self.roi_sampling_ratio = int(roi_sampling_ratio)
        self.train_mode = train
        self.mask_head_type = mask_head_type
        self.use_fpn_body = len(fpn_layers) > 0
        self.fpn_extra_lvl = fpn_extra_lvl
        self.use_rpn_head, self.use_mask_head = use_rpn_head, use_mask_head
        self.use_two_layer_mlp_head = conv_head_layers == 'two_layer_mlp'
        self.output_prob = output_prob
        self.channels_last = channels_last
        if arch not in _ARCH:
            raise ValueError('Only resnet implemented so far!')
        self.model = ResNet(_ARCH[arch])
        self.conv_body = nn.Sequential(*[getattr(self.model, l) for l in conv_body_layers])
        if self.use_fpn_body:
            self.conv_body = fpn_body(self.conv_body, conv_body_layers, fpn_layers)
        if self.use_two_layer_mlp_head:
            self.conv_head = two_layer_mlp_head()
            roi_feature_channels = 1024
        else:
            self.conv_head = nn.Sequential(*[getattr(self.model, l) for l in conv_head_layers])
        if self.use_rpn_head and not self.use_fpn_body:
            self.rpn = rpn_head()
            self.proposal_generator = GenerateProposals(train=False)
        if self.use_rpn_head and self.use_fpn_body:
            self.rpn = rpn_head(in_channels=256, out_channels=256, n_anchors=3)
            scales = list(self.roi_spatial_scale)
            if self.fpn_extra_lvl:
                scales = scales + [scales[-1] / 2.]
            self.rpn_scales = scales
            self.proposal_generator = nn.ModuleList([GenerateProposals(train=False, spatial_scale=scales[i],
                                                                       anchor_sizes=(32 * 2 ** i,), rpn_pre_nms_top_n=1000,
                                                                       rpn_post_nms_top_n=1000) for i in range(len(scales))])
            self.collect_and_distr_rois = CollectAndDistributeFpnRpnProposals(spatial_scales=self.roi_spatial_scale)
        self.bbox_head = nn.Linear(roi_feature_channels, 4 * N_classes)
        self.classif_head = nn.Linear(roi_feature_channels, N_classes)
        if self.use_mask_head:
            mh_conv = self.conv_head[0] if mask_head_type == 'upshare' else four_layer_conv()
            self.mask_head = mask_head(mh_conv, self.roi_spatial_scale, self.roi_sampling_ratio, output_prob)
        if detector_pkl_file is not None:
            self.load_pretrained_weights(detector_pkl_file, model='detector')
        elif base_cnn_pkl_file is not None:
            self.load_pretrained_weights(base_cnn_pkl_file, model='base_cnn')
        self.eval()   # BN layers are caffe2 AffineChannel ops: always eval (detector.py:231)

    # ---- forward (detector.py:233-286) -------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, image, rois=None, scaling_factor=None, roi_original_idx=None):
        h, w = image.size(2), image.size(3)
        if getattr(self, "_opt_dtype", None) is not None:
            raise NotImplementedError("a model optimised to a 16-bit type serves through forward_batched()")
        if self.channels_last:
            image = image.contiguous(memory_format=torch.channels_last)
        img_features = self.conv_body(image)
        if self.use_rpn_head and not self.use_fpn_body:
            rpn_cls_prob, rpn_bbox_pred = self.rpn(img_features, logits=self.fuse_rpn_sigmoid)
            rois, _ = self.proposal_generator(rpn_cls_prob, rpn_bbox_pred, h, w, scaling_factor,
                                              scores_are_logits=self.fuse_rpn_sigmoid)
        fused = None
        if self.use_rpn_head and self.use_fpn_body:
            feats = list(img_features)
            if self.fpn_extra_lvl:
                feats = feats + [F.max_pool2d(feats[-1], 1, stride=2)]                 # detector.py:250
            cls_bbox = [self.rpn(f, logits=self.fuse_rpn_sigmoid) for f in feats]
            gens = self.proposal_generator
            boxes, scores, counts, _, _, _ = hip.generate_proposals(                     # all levels, one call
                [c for c, _ in cls_bbox], [b for _, b in cls_bbox], [g._anchors for g in gens],
                [1. / s for s in self.rpn_scales], h, w, [g.rpn_pre_nms_top_n for g in gens],
                gens[0].rpn_post_nms_top_n, gens[0].rpn_nms_thresh, scores_are_logits=self.fuse_rpn_sigmoid)
            lv = [int(log2(1 / s)) for s in self.roi_spatial_scale]
            fused = hip.fpn_collect_distribute(boxes, scores, counts, 1000, lv[0], lv[-1],   # collect...py:86
                                               inputs_sorted=True)
        if not self.use_fpn_body:
            roi_features = RoIAlignFunction.apply(img_features, preprocess_rois(rois), self.roi_height, self.roi_width,
                                                  self.roi_spatial_scale, self.roi_sampling_ratio)
        elif fused is not None:
            # the packed descriptors carry the visiting order (level, row band, x) the cluster-stationary RoIAlign kernel wants;
            # output rows stay in collected (score) order = the "restored" order of :269-270.  One host sync (the proposal
            # count) is what the reference-shaped return value costs; forward_batched() has none.
            T = fused["rois5"].shape[1]
            odt = self.head_dtype or torch.float32
            roi_features = torch.empty((T, img_features[0].shape[1], self.roi_height, self.roi_width), dtype=odt, device=image.device)
            lvs, ch, dt = hip.make_levels(list(img_features), self.roi_spatial_scale)
            hip.check(hip.lib().dtc_roi_align_forward_packed(lvs, len(self.roi_spatial_scale), ch, hip._dtype_code(dt),
                                                             fused["roi_desc"].data_ptr(), T, self.roi_height, self.roi_width,
                                                             self.roi_sampling_ratio, roi_features.data_ptr(), hip._dtype_code(odt),
                                                             hip.stream_ptr(image.device)), "roi_align(packed)")
            n = int(fused["n_out"][0].item())
            roi_features = roi_features[:n]
            rois = fused["rois5"][0, :n, 1:]
        else:
            # FPN with precomputed per-level rois (eval_fast_FPN flow): per-level lists + restore index from the caller
            lvs = [torch.full((r.shape[0],), i, dtype=torch.int32, device=r.device) for i, r in enumerate(rois)]
            cat = torch.cat([preprocess_rois(r) for r in rois], 0)
            roi_features = hip.roi_align_forward(list(img_features), self.roi_spatial_scale, cat, self.roi_height,
                                                 self.roi_width, self.roi_sampling_ratio, roi_levels=torch.cat(lvs, 0))
            roi_features = roi_features[roi_original_idx, :]
            rois = cat[roi_original_idx, 1:]
        roi_features = self._head(roi_features)
        cls_score = self.classif_head(roi_features)
        if self.output_prob:
            cls_score = F.softmax(cls_score, dim=1)
        bbox_pred = self.bbox_head(roi_features)
        return cls_score, bbox_pred, rois, img_features

    def _head(self, roi_features):
        """conv_head (fc6/fc7 or res5) on pooled features; head_dtype runs it as a low-precision MFMA GEMM, fp32 out."""
        if getattr(self, "_opt_dtype", None) is not None:          # weights already live in the 16-bit type
            x = self.conv_head(roi_features.to(self._opt_dtype)).float()
        elif self.head_dtype is not None and self.head_dtype != torch.float32:
            with torch.autocast("cuda", dtype=self.head_dtype):
                x = self.conv_head(roi_features.to(self.head_dtype))
            x = x.float()
        else:
            x = self.conv_head(roi_features.float() if roi_features.dtype != torch.float32 else roi_features)
        return x.reshape(x.size(0), -1)

    # ---- inference form of the FPN model -----------------------------------------------------------------------------
    @torch.no_grad()
    def optimize_for_inference(self, dtype=None):
        """One-way conversion of an FPN model for serving (forward_batched; forward() too while dtype is float32):
          * every eval-mode BatchNorm of the ResNet body -- a caffe2 AffineChannel, detector.py:231 -- is folded into its conv
            (scale into the weights, shift into a float32 epilogue bias);
          * what follows a convolution -- bias, ReLU, the bottleneck's `+ identity`, the FPN's `upsample(top) + lateral`
            (detector.py:45-46), the RPN / mask heads' bias + ReLU -- runs as ONE in-place pass (hip.bias_act_) instead of the
            two to four elementwise kernels of the eager graph;
          * dtype torch.bfloat16 / float16: the conv / fc6 / fc7 weights are stored once in that type and the body runs in it
            directly (no autocast: no per-call weight casts); classif_head / bbox_head stay float32 as under autocast;
          * the fixed-shape results carry 104 instead of 128 detection rows per image (the mask head convolves them all).
        Same function as the unoptimised model up to rounding (float32: the BatchNorm scale is applied to the weights instead
        of the conv output; 16-bit: one rounding per epilogue instead of one per eager op).  Load weights BEFORE calling this."""
        if not (self.use_fpn_body and self.use_rpn_head and self.use_two_layer_mlp_head):
            raise NotImplementedError("optimize_for_inference covers the FPN configurations (e2e_faster/mask_rcnn_R-*-FPN)")
        if getattr(self, "_optimized", False):
            raise RuntimeError("model is already in inference form")
        if dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("dtype must be float32, bfloat16 or float16")
        if self.training:
            raise RuntimeError("eval() first: BatchNorm statistics are folded as constants")
        body = self.conv_body.conv_body                      # nn.Sequential over conv_body_layers
        mods = list(body)
        for i, m in enumerate(mods):
            if isinstance(m, nn.BatchNorm2d):                # stem: conv1 | bn1 | relu  ->  conv1 | bias + ReLU | identity
                if not (i > 0 and isinstance(mods[i - 1], nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)):
                    raise NotImplementedError("stem must be conv, bn, relu")
                body[i] = _Epilogue(_fold_bn_(mods[i - 1], m), relu=True)
                body[i + 1] = nn.Identity()
            elif isinstance(m, nn.Sequential):
                for blk in m:
                    blk.fold_()
        self.conv_body.fused = True
        self.rpn.fused = True
        heads = [self.conv_body, self.rpn, self.conv_head]
        if self.use_mask_head:
            if not isinstance(self.mask_head.conv_head, four_layer_conv):
                raise NotImplementedError("mask head must be 1up4convs")
            self.mask_head.conv_head.fused = True
            heads.append(self.mask_head)
        for mod in heads:                                    # float32 epilogue biases BEFORE the parameters change type
            for m in mod.modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.bias is not None:
                    _ebias(m)
        low = dtype if dtype in (torch.bfloat16, torch.float16) else None
        if low is not None:
            for mod in heads:
                for prm in mod.parameters():                 # parameters only: the epilogue biases (buffers) stay float32
                    fmt = torch.channels_last if prm.dim() == 4 and self.channels_last else torch.contiguous_format
                    prm.data = prm.data.to(low).contiguous(memory_format=fmt)
            self.backbone_dtype = self.head_dtype = low
        self._optimized, self._opt_dtype = True, low
        self.max_out = 104                                   # 100 detections + room for ties; 19 % fewer mask-head rows than 128
        self._paths.clear()
        return self

    def load_state_dict(self, *args, **kw):
        """Weights are loaded BEFORE optimize_for_inference(): afterwards the conv weights hold the folded BatchNorm scale and the
        BatchNorm modules are bypassed, so loading into the inference form would silently give a different function."""
        if getattr(self, "_optimized", False):
            raise RuntimeError("model is in inference form (optimize_for_inference): load weights into a fresh model, then optimise it")
        return super().load_state_dict(*args, **kw)

    # ---- batched entry: B images, zero host round trips ------------------------------------------------------------------
    def _region_path(self, B, h, w, dev):
        from ..pipeline import FpnRegionPath
        key = (B, h, w, str(dev), self.max_out)
        if key in self._paths:
            self._paths[key] = self._paths.pop(key)          # most recently used last
        else:
            # a path owns ~100 MB of workspaces per image of batch: keep the few most recently used padded sizes only (a COCO
            # evaluation sees dozens).  NOTE the returned path is the cached object: its result tensors are overwritten by the
            # next forward_batched call with the same (B, h, w) -- copy what has to outlive that call.
            while len(self._paths) >= self.max_cached_paths:
                self._paths.pop(next(iter(self._paths)))
            self._paths[key] = FpnRegionPath(B, dev, channels=256, n_cls=self.N_classes, pad_h=h, pad_w=w, cls_logits=True,
                                             with_rle=self.use_mask_head, box_pooled=self.roi_height, mask_pooled=14,
                                             sampling_ratio=self.roi_sampling_ratio, max_out=self.max_out,
                                             feat_dtype=self.head_dtype or torch.float32)
        return self._paths[key]

    @torch.no_grad()
    def forward_batched(self, images, scaling_factor, im_size):
        """Mask / Faster R-CNN FPN forward for a BATCH (lib/model/detector.py:233-286 + :99-112 + eval_mask_FPN.ipynb:231-262,
        which the reference runs image by image with 21 synchronising copies each).
          images [B,3,H,W] prepared blobs of one padded size (utils.blob.im_list_to_blob); scaling_factor [B]; im_size [B,2]
          = original (h, w) per image.
        Everything stays on the device:  backbone(B) -> RPN heads -> FpnRegionPath.launch_proposals (top-k, NMS, collect,
        distribute, RoIAlign 7x7 in visiting order) -> fc6/fc7/cls/bbox -> launch_detections (softmax folded in, class decode,
        80-class NMS, top-100, mask-branch RoIAlign 14x14) -> mask head convs -> launch_masks (paste, binarise, COCO RLE).
        Returns the FpnRegionPath holding the fixed-shape device results (dets [B,128,6], det_count [B], rois5 [B,1000,5],
        n_rois [B], crops / RLE strings ...) plus the head outputs; `per_image(b)` gives the reference's forward() tuple."""
        if not (self.use_rpn_head and self.use_fpn_body and self.use_two_layer_mlp_head):
            raise NotImplementedError("forward_batched covers the FPN configurations (e2e_faster/mask_rcnn_R-*-FPN)")
        B, h, w = images.size(0), images.size(2), images.size(3)
        dev = images.device
        opt = getattr(self, "_opt_dtype", None)                 # optimize_for_inference(16-bit): weights live in that type
        # one copy at most: the blob in the body's type and layout
        images = images.to(dtype=opt or images.dtype, memory_format=torch.channels_last if self.channels_last else torch.preserve_format)
        low = self.backbone_dtype is not None and self.backbone_dtype != torch.float32
        if low and self.head_dtype not in (None, torch.float32, self.backbone_dtype):
            raise ValueError("backbone_dtype and head_dtype must be the same 16-bit type (RoIAlign does not mix fp16 and bf16)")
        with (torch.autocast("cuda", dtype=self.backbone_dtype) if low and opt is None else contextlib.nullcontext()):
            img_features = self.conv_body(images)
            feats = list(img_features)
            rpn_in = feats + ([F.max_pool2d(feats[-1], 1, stride=2)] if self.fpn_extra_lvl else [])
            cls_bbox = [self.rpn(f, logits=self.fuse_rpn_sigmoid) for f in rpn_in]
        if low:      # every level in the autocast type (a level left in float32 by an op outside autocast's list would mix dtypes)
            feats = [f.to(self.backbone_dtype) for f in feats]
            img_features = feats
        path = self._region_path(B, h, w, dev)
        path.bind_rpn([c.float().contiguous() for c, _ in cls_bbox], [b.float().contiguous() for _, b in cls_bbox], feats,
                      scores_are_logits=self.fuse_rpn_sigmoid)
        path.launch_proposals()
        x = self._head(path.box_feats)                                              # [B*1000, 1024]
        T = path.top_n
        cls_logits = self.classif_head(x).reshape(B, T, -1).contiguous()
        bbox_pred = self.bbox_head(x).reshape(B, T, -1).contiguous()
        sf = torch.as_tensor(scaling_factor, dtype=torch.float32, device=dev).reshape(B).contiguous()
        sz = torch.as_tensor(im_size, dtype=torch.float32, device=dev).reshape(B, 2).contiguous()
        path.bind_heads(cls_logits, bbox_pred, sf, sz)
        path.launch_detections()
        path.img_features, path.cls_logits_out, path.bbox_pred_out = img_features, cls_logits, bbox_pred
        if self.use_mask_head:
            mh = self.mask_head
            probs = None
            if opt is not None:
                m = hip.bias_act_(_conv_nobias(mh.transposed_conv, mh.conv_head(path.mask_feats.to(opt))), _ebias(mh.transposed_conv))
                # class logits: bias-free 1x1 conv, then bias + widening + NCHW layout in ONE pass, sigmoid in place
                y = _conv_nobias(mh.classif_logits, m)
                probs = torch.add(y, _ebias(mh.classif_logits).view(1, -1, 1, 1),
                                  out=torch.empty(y.shape, dtype=torch.float32, device=dev)).sigmoid_()
            elif self.head_dtype is not None and self.head_dtype != torch.float32:    # mask-head convs in the pooled features' type
                with torch.autocast("cuda", dtype=self.head_dtype):
                    m = mh.classif_logits(mh.relu(mh.transposed_conv(mh.conv_head(path.mask_feats.to(self.head_dtype)))))
                m = m.float()
            else:
                m = mh.conv_head(path.mask_feats.float() if path.mask_feats.dtype != torch.float32 else path.mask_feats)
                m = mh.classif_logits(mh.relu(mh.transposed_conv(m)))
            path.bind_masks(probs if probs is not None else torch.sigmoid(m).contiguous())     # [B*128, 81, 28, 28]
            path.launch_masks()
        return path

    @staticmethod
    def per_image(path, b):
        """The reference forward()'s tuple for image b of a forward_batched() result: (cls_score [n,81] probabilities,
        bbox_pred [n,324], rois [n,4], img_features of that image).  Syncs (one count) -- for callers that want the
        reference's shapes; the fixed-shape tensors on `path` need no sync."""
        n = int(path.n_rois[b].item())
        return (F.softmax(path.cls_logits_out[b, :n], dim=1), path.bbox_pred_out[b, :n], path.rois5[b, :n, 1:],
                [f[b:b + 1] for f in path.img_features])

    # ---- caffe2 / Detectron pickle import (detector.py:289-374) -------------------------------------------------------
    def load_pretrained_weights(self, caffe_pkl_file, model='detector'):
        with open(caffe_pkl_file, 'rb') as f:
            blobs = pickle.load(f, encoding='latin1')
        if model == 'detector':
            blobs = blobs['blobs']
        T = lambda name: torch.as_tensor(np.asarray(blobs[name]), dtype=torch.float32)
        sd = self.model.state_dict()
        for k in list(sd.keys()):
            if 'running' in k or 'num_batches' in k or 'fc' in k:
                continue
            v = T(_caffe2_name(k))
            assert sd[k].size() == v.size(), k
            sd[k] = v[:, (2, 1, 0), :, :] if k == 'conv1.weight' else v        # BGR -> RGB
        self.model.load_state_dict(sd)
        if model != 'detector':
            return

        def put(mod, wname, bname):
            mod.weight.data = T(wname)
            mod.bias.data = T(bname)
        put(self.bbox_head, 'bbox_pred_w', 'bbox_pred_b')
        put(self.classif_head, 'cls_score_w', 'cls_score_b')
        if self.use_rpn_head:
            sfx = '_fpn2' if self.use_fpn_body else ''
            put(self.rpn.conv_rpn, 'conv_rpn%s_w' % sfx, 'conv_rpn%s_b' % sfx)
            put(self.rpn.rpn_cls_prob, 'rpn_cls_logits%s_w' % sfx, 'rpn_cls_logits%s_b' % sfx)
            put(self.rpn.rpn_bbox_pred, 'rpn_bbox_pred%s_w' % sfx, 'rpn_bbox_pred%s_b' % sfx)
        if self.use_mask_head:
            put(self.mask_head.transposed_conv, 'conv5_mask_w', 'conv5_mask_b')
            put(self.mask_head.classif_logits, 'mask_fcn_logits_w', 'mask_fcn_logits_b')
            if self.mask_head_type == '1up4convs':
                for i in range(1, 5):
                    put(getattr(self.mask_head.conv_head, 'fcn%d' % i), '_[mask]_fcn%d_w' % i, '_[mask]_fcn%d_b' % i)
        if self.use_fpn_body:
            nl = len(self.conv_body.fpn_layers)
            for i, l in enumerate(self.conv_body.fpn_layers):
                last = list(getattr(self.model, l).state_dict().keys())[-1]
                stem = _caffe2_name(l + '.' + last)
                stem = stem[:stem.rfind('_branch')]                    # e.g. res2_2
                suffix = '_sum_lateral' if i < nl - 1 else '_sum'
                put(self.conv_body.fpn_lateral[i], 'fpn_inner_%s%s_w' % (stem, suffix), 'fpn_inner_%s%s_b' % (stem, suffix))
                put(self.conv_body.fpn_output[i], 'fpn_%s_sum_w' % stem, 'fpn_%s_sum_b' % stem)
        if self.use_two_layer_mlp_head:
            put(self.conv_head.fc6, 'fc6_w', 'fc6_b')
            put(self.conv_head.fc7, 'fc7_w', 'fc7_b')
