"""The SSD300 extra layers (conv6_1 ... conv9_2, batch 32) one by one: the one-pass kernels (variants None / 5 / 6) against the split-K
form (variant 8).  GPU box.   python tools/time_extras.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

LAYERS = [("conv6_1", 19, 1024, 256, 1, 1, 0), ("conv6_2", 19, 256, 512, 3, 2, 1), ("conv7_1", 10, 512, 128, 1, 1, 0),
          ("conv7_2", 10, 128, 256, 3, 2, 1), ("conv8_1", 5, 256, 128, 1, 1, 0), ("conv8_2", 5, 128, 256, 3, 1, 0),
          ("conv9_1", 3, 256, 128, 1, 1, 0), ("conv9_2", 3, 128, 256, 3, 1, 0), ("fc7", 19, 1024, 1024, 1, 1, 0)]


def timed(fn, reps=30):
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
    return round(best * 1e3, 1)


rows = []
B = 32
for name, H, Cin, Cout, k, stride, pad in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, H, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    row = {"layer": name}
    for nm, v in (("igemm", None), ("ring4", 5), ("ring3", 6), ("splitk", 8)):
        row[nm + "_us"] = timed(lambda v=v: nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True, variant=v))
    print(json.dumps(row), flush=True)
    rows.append(row)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
