"""Weight-gradient kernel (csrc/ssdhip_wgrad.hip) against the framework's convolution_backward (MIOpen) on the SSD300 trunk layers at
batch 32: event timing of back-to-back launches + the error against the float32 framework gradient.  GPU box.

    python tools/time_wgrad.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

LAYERS = [  # name, B, H, W, Cin, Cout
    ("conv1_2", 32, 300, 300, 64, 64), ("conv2_1", 32, 150, 150, 64, 128), ("conv2_2", 32, 150, 150, 128, 128),
    ("conv3_1", 32, 75, 75, 128, 256), ("conv3_2", 32, 75, 75, 256, 256), ("conv4_1", 32, 38, 38, 256, 512),
    ("conv4_2", 32, 38, 38, 512, 512), ("conv5_1", 32, 19, 19, 512, 512),
]
if os.environ.get("WGRAD_LAYERS"):
    LAYERS = [l for l in LAYERS if l[0] in os.environ["WGRAD_LAYERS"].split(",")]


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


rows = []
torch.backends.cudnn.benchmark = True
for name, B, H, W, Cin, Cout in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wb = torch.zeros((Cout, 3, 3, Cin), device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
    flop = 2.0 * 9 * Cin * Cout * B * H * W
    row = {"layer": name, "shape": [B, H, W, Cin, Cout], "gflop": round(flop / 1e9, 1)}

    def lib():
        return torch.ops.aten.convolution_backward(dy, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]

    got = nat.conv3x3_wgrad(x, dy)
    ref = lib().float()
    rms = ref.pow(2).mean().sqrt().item()
    row["max_err_over_rms_vs_framework_bf16_out"] = round(float((got - ref).abs().max()) / rms, 5)
    us = timed(lambda: nat.conv3x3_wgrad(x, dy))
    row["ssdhip_us"] = round(us, 1)
    row["ssdhip_tflops"] = round(flop / us / 1e6, 1)
    us = timed(lib)
    row["framework_us"] = round(us, 1)
    row["framework_tflops"] = round(flop / us / 1e6, 1)
    print(json.dumps(row), flush=True)
    rows.append(row)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
