import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_hip_detector as T
from detectorch_amd.model.detector import detector
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "det":
    torch.backends.cudnn.deterministic = True
model = T._boost(T._fpn_model())
g = torch.Generator(device="cuda"); g.manual_seed(5)
image = torch.randn(1, 3, 320, 448, generator=g, device="cuda")
sf, im_size = torch.tensor([1.6], device="cuda"), torch.tensor([[200.0, 280.0]], device="cuda")
if mode == "warm":
    model(image, scaling_factor=sf); model.forward_batched(image, sf, im_size)
for rep in range(3):
    path = model.forward_batched(image, sf, im_size)
    torch.cuda.synchronize()
    cls_b, bbox_b, rois_b, feats_b = detector.per_image(path, 0)
    cls_score, bbox_pred, rois, feats = model(image, scaling_factor=sf)
    d = (rois - rois_b).abs().amax(1)
    fd = [float((a - b).abs().max()) for a, b in zip(feats, feats_b)]
    print(mode, rep, "rows differing > 1e-2:", int((d > 1e-2).sum()), "max", float(d.max()), "feature max diff per level", fd)
