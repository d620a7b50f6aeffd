"""Timing of the Cin = 64 kernels under one library build (SSDHIP_LIB selects it; the c64a* builds of tools/prof_build.sh remove
one ingredient each: 1 global stores, 2 the epilogue, 4 the fragment reads, 8 the producers' work).  GPU box.
    SSDHIP_LIB=tools/libssdhip_prof_c64a1.so python tools/ablate_c64.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


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


x3 = torch.randn((32, 300, 300, 3), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w1 = (torch.randn((64, 3, 3, 3), device="cuda") / 5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b1 = torch.randn((64,), device="cuda").to(torch.bfloat16)
x64 = torch.randn((32, 300, 300, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w2 = (torch.randn((64, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b2 = torch.randn((64,), device="cuda").to(torch.bfloat16)
x21 = torch.randn((32, 150, 150, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w21 = (torch.randn((128, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b21 = torch.randn((128,), device="cuda").to(torch.bfloat16)
row = {"lib": os.path.basename(nat.lib_path()),
       "conv1_block_us": timed(lambda: nat.conv1_block(x3, w1, b1, w2, b2, relu=True, pool=True)),
       "conv1_2_pool_us": timed(lambda: nat.conv3x3_c64(x64, w2, b2, relu=True, pool=True)),
       "conv2_1_us": timed(lambda: nat.conv3x3_c64(x21, w21, b21, relu=True, pool=False))}
print(json.dumps(row), flush=True)
