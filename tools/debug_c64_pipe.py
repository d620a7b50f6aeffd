"""Where do the two forms of the un-pooled Cin = 64 kernel differ?  GPU box.  (Needs profiles/r06zp_c64_pipelined_epilogue_not_adopted.patch.)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

for (B, H, W, Cout) in ((1, 8, 16, 64), (40, 16, 16, 64), (2, 150, 150, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, 64), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Cout, 3, 3, 64), generator=g, device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    os.environ["SSDHIP_C64_PIPE"] = "0"
    base = nat.conv3x3_c64(x, w, b, relu=False, pool=False).permute(0, 2, 3, 1).float()
    os.environ["SSDHIP_C64_PIPE"] = "1"
    for rep in range(3):
        got = nat.conv3x3_c64(x, w, b, relu=False, pool=False).permute(0, 2, 3, 1).float()
        d = (got != base)
        idx = d.nonzero()
        print("case", (B, H, W, Cout), "rep", rep, "differ", int(d.sum()), "of", d.numel())
        if len(idx):
            print("  rows (h) histogram:", torch.bincount(idx[:, 1] % 8, minlength=8).tolist())
            print("  cols (w %16) histogram:", torch.bincount(idx[:, 2] % 16, minlength=16).tolist())
            print("  channel histogram (c % 32):", torch.bincount(idx[:, 3] % 32, minlength=32).tolist())
            k = idx[:8]
            for r in k.tolist():
                print("   ", r, float(base[tuple(r)]), float(got[tuple(r)]))
            rel = ((got - base).abs() / (base.abs() + 1e-6))[d]
            print("  rel diff: max %.3g median %.3g" % (float(rel.max()), float(rel.median())))
