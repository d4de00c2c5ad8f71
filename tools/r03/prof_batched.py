"""torch.profiler view of detector.forward_batched (bf16, channels_last): which ATen ops the elementwise / copy kernels belong to."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd.model.detector import detector
from torch.profiler import profile, ProfilerActivity
torch.backends.cudnn.benchmark = "bench" in sys.argv
torch.manual_seed(0)
m = detector(arch='resnet50', conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
             conv_head_layers='two_layer_mlp', fpn_layers=['layer1', 'layer2', 'layer3', 'layer4'], fpn_extra_lvl=True,
             roi_height=7, roi_width=7, roi_spatial_scale=[0.25, 0.125, 0.0625, 0.03125], roi_sampling_ratio=2,
             use_rpn_head=True, use_mask_head=True, mask_head_type='1up4convs', channels_last=True).cuda()
m = m.to(memory_format=torch.channels_last)
if len(sys.argv) > 1 and sys.argv[1] == "opt":
    m.optimize_for_inference(torch.bfloat16)
else:
    m.backbone_dtype = m.head_dtype = torch.bfloat16
m.classif_head.weight.data *= 60.0
B = 8
images = torch.randn(B, 3, 800, 1344, device="cuda")
sfb = torch.full((B,), 1.6, device="cuda"); szb = torch.tensor([[500.0, 833.0]] * B, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m.forward_batched(images, sfb, szb)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m.forward_batched(images, sfb, szb)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=50, max_shapes_column_width=90))
