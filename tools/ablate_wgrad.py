"""Where the time of conv_wgrad_kernel goes: the profiling build's ablation modes (SSDHIP_WGRAD_ABL, wrong results by construction)
on conv4_2 / conv3_2 / conv2_2 at batch 32.   SSDHIP_LIB=tools/libssdhip_prof.so python tools/ablate_wgrad.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

MODES = {0: "product", 1: "no requests in the loop", 2: "no fragment reads", 3: "MFMAs + barrier only", 8: "no MFMAs", 16: "no wait / barrier",
         23: "MFMAs only"}


def timed(fn, reps=10):
    for _ in range(2):
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


for name, B, H, W, Cin, Cout in (("conv4_2", 32, 38, 38, 512, 512), ("conv3_2", 32, 75, 75, 256, 256), ("conv2_2", 32, 150, 150, 128, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    row = {"layer": name}
    for m, what in MODES.items():
        os.environ["SSDHIP_WGRAD_ABL"] = str(m)
        row["%d: %s" % (m, what)] = round(timed(lambda: nat.conv3x3_wgrad(x, dy)), 1)
    os.environ.pop("SSDHIP_WGRAD_ABL", None)
    print(json.dumps(row), flush=True)
