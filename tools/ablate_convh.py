"""Ablations of the slab kernel's K loop (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only.

Each mode removes ONE ingredient of the loop (results are wrong by construction) so the time it costs can be read off:
    128 shipped (persistent workgroups)     64 one workgroup per tile, second wave of every SIMD reading two slots later     0 neither
    65 no loads     66 no fragment reads     68 no waits / barrier     71 MFMAs only     32 no MFMAs
    python tools/ablate_convh.py [out.json]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("SSDHIP_LIB", os.path.join(HERE, "libssdhip_prof.so"))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

LAYERS = [("conv3_2", 32, 75, 75, 256, 256), ("conv4_2", 32, 38, 38, 512, 512), ("conv5_1", 32, 19, 19, 512, 512)]
MODES = [128, 64, 0, 65, 66, 68, 71, 32]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        t = a.elapsed_time(e) / reps
        best = t if best is None else min(best, t)
    return best * 1e3


rows = []
for name, B, H, W, Cin, Cout in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    flop = 2.0 * 9 * Cin * Cout * B * H * W
    row = {"layer": name}
    os.environ["SSDHIP_CONVH_MODE"] = "0"
    base = nat.conv2d_same(x, wt, bias, relu=True, variant=4).view(torch.int16)
    for m in MODES:
        os.environ["SSDHIP_CONVH_MODE"] = str(m)
        if m in (128, 64, 0):
            got = nat.conv2d_same(x, wt, bias, relu=True, variant=7).view(torch.int16)
            row["mode%d_differs" % m] = int((got != base).sum().item())
        us = timed(lambda: nat.conv2d_same(x, wt, bias, relu=True, variant=7))
        row["mode%d_us" % m] = round(us, 1)
    row["mfma_floor_us_at_2.4GHz"] = round(flop / 2.5e15 * 1e6, 1)
    print(json.dumps(row), flush=True)
    rows.append(row)
os.environ["SSDHIP_CONVH_MODE"] = "0"
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
