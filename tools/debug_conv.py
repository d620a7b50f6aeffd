#!/usr/bin/env python3
"""Error pattern of one convolution variant against the float32 reference (GPU box only)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_keras_amd import _native as nat          # noqa: E402


def main():
    variant = int(os.environ.get("VARIANT", "4"))
    for case in [(2, 19, 19, 64, 64, 3, 1), (1, 38, 38, 128, 128, 3, 1), (3, 5, 5, 256, 128, 1, 1)]:
        B, H, W, Cin, Cout, k, dil = case
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        got = nat.conv2d_same(x, wt, None, dilation=dil, relu=False, variant=variant).float()
        want = F.conv2d(x.float(), wt.float(), None, 1, dil * (k // 2), dil)
        err = (got - want).abs()
        rms = want.pow(2).mean().sqrt().item()
        bad = err > (want.abs() * 2.0 ** -7 + 1e-2 * rms)
        print("case", case, "bad", int(bad.sum()), "of", bad.numel(), "max err", float(err.max()), "rms", rms, "nan", int(torch.isnan(got).sum()))
        if bad.any():
            per_px = bad.any(dim=1)                   # (B, H, W)
            for b in range(B):
                print(" image", b)
                for h in range(H):
                    print("  ", "".join("X" if per_px[b, h, w] else "." for w in range(W)))
            per_ch = bad.sum(dim=(0, 2, 3))
            print(" bad per channel (first 16):", per_ch[:16].tolist())
            # does an interior bad pixel look like a sum with missing / extra taps?  try each single-tap removal
            idx = bad.nonzero()[0].tolist()
            print(" first bad index (b, co, h, w):", idx, "got", float(got[tuple(idx)]), "want", float(want[tuple(idx)]))


if __name__ == "__main__":
    main()
