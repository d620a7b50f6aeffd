"""Where does the slab kernel's per-tile fixed cost go?  (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only.

Modes on top of the shipped schedule (128 = persistent workgroups):
    384 = 128 + 256  the epilogue without its global stores (bias / ReLU / rounding / LDS transpose stay)
    640 = 128 + 512  no epilogue at all
Both give wrong results by construction; the difference to mode 128 is what the removed part costs, INCLUDING what it costs the
next tile's K loop (a wave's vmcnt counts loads and stores in issue order: waiting for a load issued after a store waits for
the store's acknowledgement too).
    python tools/ablate_convh2.py [out.json]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("SSDHIP_LIB", os.path.join(HERE, "libssdhip_prof.so"))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

# name, B, H, W, Cin, Cout, pool
LAYERS = [("conv2_2+pool", 32, 150, 150, 128, 128, True), ("conv3_1", 32, 75, 75, 128, 256, False),
          ("conv3_2", 32, 75, 75, 256, 256, False), ("conv3_3+pool", 32, 75, 75, 256, 256, True),
          ("conv4_1", 32, 38, 38, 256, 512, False), ("conv4_2", 32, 38, 38, 512, 512, False),
          ("conv5_1", 32, 19, 19, 512, 512, False)]
MODES = [int(m) for m in os.environ.get("ABLATE_MODES", "128,384,640").split(",")]


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
for name, B, H, W, Cin, Cout, pool in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    flop = 2.0 * 9 * Cin * Cout * B * H * W
    row = {"layer": name, "tiles_per_cu": None}
    base = (nat.conv2d_same_pool2(x, wt, bias, relu=True) if pool else nat.conv2d_same(x, wt, bias, relu=True, variant=4)).view(torch.int16)
    for m in MODES:
        os.environ["SSDHIP_CONVH_MODE"] = str(m)
        if m in (128, 1152, 64):
            diffs = []
            for _ in range(3):                            # races show up as run-to-run differences: check more than once
                got = nat.conv3x3_halo(x, wt, bias, relu=True, pool=pool).view(torch.int16)
                diffs.append(int((got != base).sum().item()))
            row["mode%d_differs" % m] = diffs
        us = timed(lambda: nat.conv3x3_halo(x, wt, bias, relu=True, pool=pool))
        row["mode%d_us" % m] = round(us, 1)
    row["mfma_floor_us_at_2.4GHz"] = round(flop / 2.5e15 * 1e6, 1)
    print(json.dumps(row), flush=True)
    rows.append(row)
os.environ["SSDHIP_CONVH_MODE"] = "128"
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
