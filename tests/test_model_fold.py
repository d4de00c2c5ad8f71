"""detector.optimize_for_inference's BatchNorm folding, checked on the CPU with plain torch (no kernels involved): a bias-free conv
followed by an eval-mode BatchNorm equals the conv with scaled weights plus the returned shift."""
import torch
import torch.nn as nn


def _randomise(bn, g):
    bn.weight.data = 0.5 + torch.rand(bn.num_features, generator=g)
    bn.bias.data = torch.randn(bn.num_features, generator=g) * 0.1
    bn.running_mean.data = torch.randn(bn.num_features, generator=g) * 0.1
    bn.running_var.data = 0.5 + torch.rand(bn.num_features, generator=g)


def test_fold_bn_equals_conv_then_batchnorm():
    from detectorch_amd.model.detector import _fold_bn_
    g = torch.Generator(); g.manual_seed(0)
    for cin, cout, k, stride, bias in [(8, 16, 1, 1, False), (16, 16, 3, 1, False), (8, 32, 1, 2, False), (4, 8, 3, 1, True)]:
        conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=bias)
        bn = nn.BatchNorm2d(cout).eval()
        _randomise(bn, g)
        x = torch.randn(2, cin, 12, 10, generator=g)
        with torch.no_grad():
            want = bn(conv(x))
            shift = _fold_bn_(conv, bn)
            assert conv.bias is None and shift.dtype == torch.float32 and shift.shape == (cout,)
            got = conv(x) + shift.view(1, -1, 1, 1)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_bottleneck_fold_keeps_the_function_and_merges_the_downsample_shift():
    """Bottleneck.fold_(): eb3 carries bn3's shift PLUS the downsample BatchNorm's (both are added before the last ReLU); the fused
    forward's arithmetic written out with torch ops equals the eager block."""
    from detectorch_amd.model.detector import Bottleneck
    g = torch.Generator(); g.manual_seed(1)
    down = nn.Sequential(nn.Conv2d(16, 32, 1, stride=2, bias=False), nn.BatchNorm2d(32))
    for blk in (Bottleneck(16, 8, stride=2, downsample=down), Bottleneck(32, 8)):
        blk.eval()
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                _randomise(m, g)
        x = torch.randn(2, blk.conv1.in_channels, 10, 8, generator=g)
        with torch.no_grad():
            want = blk(x)
            blk.fold_()
            bias = lambda t, b: t + b.view(1, -1, 1, 1)
            out = torch.relu(bias(blk.conv1(x), blk.eb1))
            out = torch.relu(bias(blk.conv2(out), blk.eb2))
            idt = x if blk.downsample is None else blk.downsample[0](x)
            got = torch.relu(bias(blk.conv3(out), blk.eb3) + idt)
        assert blk.fused and torch.allclose(got, want, rtol=1e-4, atol=1e-5)
