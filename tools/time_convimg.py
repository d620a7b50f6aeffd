"""csrc/ssdhip_convimg.hip (one image per tile, dilated taps as per-lane addresses) against the implicit-GEMM kernels on fc6 and its
small-map relatives, batch 32: bit-equality and event timing of back-to-back launches.  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

CASES = [("fc6", 32, 19, 19, 512, 1024, 6), ("conv5_1", 32, 19, 19, 512, 512, 1), ("ssd512_tail", 16, 16, 16, 512, 256, 1), ("small", 2, 5, 7, 64, 128, 2), ("10x10", 8, 10, 10, 256, 256, 3)]


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
    return round(best * 1e3, 1)


for name, B, H, W, Cin, Cout, d in CASES:
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    row = {"case": name, "shape": [B, H, W, Cin, Cout, d], "gflop": round(2.0 * 9 * Cin * Cout * B * H * W / 1e9, 2)}
    base = nat.conv2d_same(x, wt, bias, dilation=d, relu=True, variant=4).view(torch.int16)
    try:
        got = nat.conv3x3_image(x, wt, bias, dilation=d, relu=True).view(torch.int16)
        torch.cuda.synchronize()
        row["differs_from_variant4"] = int((got != base).sum().item())
        row["image_us"] = timed(lambda: nat.conv3x3_image(x, wt, bias, dilation=d, relu=True))
    except Exception as exc:                               # noqa: BLE001
        row["error"] = repr(exc)[:300]
    for v in (4, 5, 6):
        row["igemm_v%d_us" % v] = timed(lambda v=v: nat.conv2d_same(x, wt, bias, dilation=d, relu=True, variant=v))
    print(json.dumps(row), flush=True)
