"""GPU check of models/_common._ConvBiasActFn on single layers: output and gradients vs a float32 PyTorch reference of the same op."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch import nn  # noqa: E402

from ssd_keras_amd.models._common import SSDModel, _ConvBiasActFn  # noqa: E402

torch.manual_seed(0)
m = SSDModel.__new__(SSDModel)
nn.Module.__init__(m)
m.fused_training = True
m.fused_inference = True
rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-20))
for (cin, cout, k, s, p, d, hw) in ((3, 64, 3, 1, 1, 1, 64), (64, 64, 3, 1, 1, 1, 64), (64, 128, 3, 1, 1, 1, 40), (128, 256, 3, 1, 1, 1, 38),
                                    (512, 1024, 3, 1, 6, 6, 19), (1024, 256, 1, 1, 0, 1, 19), (256, 512, 3, 2, 1, 1, 19), (128, 256, 3, 1, 0, 1, 5)):
    conv = nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(4, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation))
    g = torch.randn_like(yr)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, conv.weight, conv.bias], g)
    xn = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yn = m.conv_act(conv, xn, relu=True)
        ya = F.relu(conv(xn))
    gxn, gwn, gbn = torch.autograd.grad(yn, [xn, conv.weight, conv.bias], g.to(yn.dtype))
    gxa, gwa, gba = torch.autograd.grad(ya, [xn, conv.weight, conv.bias], g.to(ya.dtype))
    print("cin %4d cout %4d k%d s%d p%d d%d  %s | y %.4f (autocast %.4f)  gx %.4f (%.4f)  gw %.4f (%.4f)  gb %.4f (%.4f)" % (
        cin, cout, k, s, p, d, type(yn.grad_fn).__name__, rel(yn, yr), rel(ya, yr), rel(gxn, gxr), rel(gxa, gxr), rel(gwn, gwr), rel(gwa, gwr),
        rel(gbn, gbr), rel(gba, gbr)), flush=True)
