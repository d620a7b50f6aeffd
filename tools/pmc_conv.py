#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the convolution kernels: conv4_2- and conv1_2-shaped problems, both variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_keras_amd import _native as nat          # noqa: E402

SHAPES = {"head1": (19, 1024, 192, 3, 1), "conv4_2": (38, 512, 512, 3, 1), "conv1_2": (300, 64, 64, 3, 1), "conv3_2": (75, 256, 256, 3, 1)}


def main():
    B = 32
    for name in os.environ.get("LAYERS", "conv4_2,conv1_2").split(","):
        hw, cin, cout, k, dil = SHAPES[name]
        x = torch.randn((B, hw, hw, cin), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.randn((cout, k, k, cin), device="cuda") / (k * k * cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn((cout,), device="cuda").to(torch.bfloat16)
        for variant in [int(v) for v in os.environ.get("VARIANTS", "1,3").split(",")]:
            for _ in range(3):
                if variant == 64:
                    nat.conv3x3_c64(x, w, b, relu=True, pool=True)
                elif variant == 40:
                    nat.conv2d_same_pool2(x, w, b, dilation=dil, relu=True)
                else:
                    nat.conv2d_same(x, w, b, dilation=dil, relu=True, variant=variant)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
