"""Driver for rocprofv3 --pmc passes over the slab kernel (variant 7) and the implicit-GEMM kernel (variant 4).  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

for (B, H, W, Cin, Cout) in ((32, 38, 38, 512, 512), (32, 75, 75, 256, 256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    for v in (4, 7):
        for _ in range(int(os.environ.get("REPS", "3"))):
            nat.conv2d_same(x, wt, bias, relu=True, variant=v)
    torch.cuda.synchronize()
