"""Slab kernel (csrc/ssdhip_convh.hip, variant 7) against the implicit-GEMM kernels (variants 4, 6) on the SSD300 / SSD512 layer
shapes it covers, batch 32: bit-equality with variant 4 and event timing of back-to-back launches.  GPU box.

    python tools/time_convh.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

MODES = [int(m) for m in os.environ.get("CONVH_MODES", "128,1152").split(",")]   # + 4096: four waves per workgroup
LAYERS = [  # name, B, H, W, Cin, Cout
    ("conv3_1", 32, 75, 75, 128, 256), ("conv3_2", 32, 75, 75, 256, 256), ("conv4_1", 32, 38, 38, 256, 512),
    ("conv4_2", 32, 38, 38, 512, 512), ("conv5_1", 32, 19, 19, 512, 512),
    ("ssd512_conv4_2", 16, 64, 64, 512, 512),
]


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
    row = {"layer": name, "shape": [B, H, W, Cin, Cout], "gflop": flop / 1e9}
    base = nat.conv2d_same(x, wt, bias, relu=True, variant=4).view(torch.int16)
    try:
        got = nat.conv2d_same(x, wt, bias, relu=True, variant=7).view(torch.int16)
        row["differs_from_variant4"] = int((got != base).sum().item())
    except Exception as exc:                               # noqa: BLE001
        row["error"] = repr(exc)[:200]
    for v in (4,):
        us = timed(lambda v=v: nat.conv2d_same(x, wt, bias, relu=True, variant=v))
        row["v%d_us" % v] = round(us, 1)
    if "error" not in row:
        for mode in MODES:                                 # SSDHIP_CONVH_MODE is read at every launch
            os.environ["SSDHIP_CONVH_MODE"] = str(mode)
            bad = 0
            for _ in range(5):
                got = nat.conv2d_same(x, wt, bias, relu=True, variant=7).view(torch.int16)
                bad += int((got != base).sum().item())
            us = timed(lambda: nat.conv2d_same(x, wt, bias, relu=True, variant=7))
            row["mode%d" % mode] = {"us": round(us, 1), "tflops": round(flop / us / 1e6, 1), "differs": bad}
        os.environ.pop("SSDHIP_CONVH_MODE", None)
    print(json.dumps(row), flush=True)
    rows.append(row)
# conv2_2 -> pool2 (and its SSD512 twin): the fused-pool kernels
for name, B, H, W, Cin, Cout in (("conv2_2_pool", 32, 150, 150, 128, 128), ("conv3_3_pool", 32, 75, 75, 256, 256),
                                 ):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    flop = 2.0 * 9 * Cin * Cout * B * H * W
    row = {"layer": name, "gflop": flop / 1e9}
    base = nat.conv2d_same_pool2(x, wt, bias, relu=True).view(torch.int16)
    row["igemm_pool_us"] = round(timed(lambda: nat.conv2d_same_pool2(x, wt, bias, relu=True)), 1)
    for mode in MODES:
        os.environ["SSDHIP_CONVH_MODE"] = str(mode)
        got = nat.conv3x3_halo(x, wt, bias, relu=True, pool=True).view(torch.int16)
        us = timed(lambda: nat.conv3x3_halo(x, wt, bias, relu=True, pool=True))
        row["mode%d" % mode] = {"us": round(us, 1), "tflops": round(flop / us / 1e6, 1), "differs": int((got != base).sum().item())}
    os.environ.pop("SSDHIP_CONVH_MODE", None)
    print(json.dumps(row), flush=True)
    rows.append(row)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
