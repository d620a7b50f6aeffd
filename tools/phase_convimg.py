#!/usr/bin/env python3
"""In-kernel phase timers of csrc/ssdhip_convimg.hip (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only.
fc6 at batch 32: shader cycles per workgroup of wave 0 in prologue | wait + barrier | steps | epilogue (72 steps; MFMA floor of a step:
2 waves x 24 MFMAs x 32 cycles = 1536)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("SSDHIP_LIB", os.path.join(HERE, "libssdhip_prof.so"))
sys.path.insert(0, os.path.dirname(HERE))
import torch            # noqa: E402

from ssd_keras_amd import _native as nat       # noqa: E402

lib = nat.load()


def read():
    buf = (ctypes.c_ulonglong * 8)()
    assert lib.ssdhip_profile_read_convimg(buf, 1) == 0
    return list(buf)


for name, B, H, W, Cin, Cout, d, k, st, pd in (("fc6", 32, 19, 19, 512, 1024, 6, 3, 1, 6), ("conv5_1", 32, 19, 19, 512, 512, 1, 3, 1, 1),
                                               ("fc7", 32, 19, 19, 1024, 1024, 1, 1, 1, 0), ("conv6_1", 32, 19, 19, 1024, 256, 1, 1, 1, 0),
                                               ("conv6_2", 32, 19, 19, 256, 512, 1, 3, 2, 1)):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    fn = lambda: nat.conv2d_image(x, wt, bias, stride=st, padding=pd, dilation=d, relu=True)
    fn(); fn()
    torch.cuda.synchronize()
    read()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    v = read()
    n = max(v[4], 1)
    steps = k * k * Cin // 64
    print("%-8s %.1f us | per workgroup: prologue %.0f  wait+barrier %.0f (%.0f per step)  steps %.0f (%.0f per step)  epilogue %.0f cycles"
          % (name, a.elapsed_time(e) * 1e3, v[0] / n, v[1] / n, v[1] / n / steps, v[2] / n, v[2] / n / steps, v[3] / n), flush=True)
