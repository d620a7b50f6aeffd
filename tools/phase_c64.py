#!/usr/bin/env python3
"""In-kernel phase timers of the Cin = 64 kernels (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only.
Prints shader cycles per tile of multiplier wave 0 (K loop | barrier | epilogue) and of wave 4 (loader / producer 0)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("SSDHIP_LIB", os.path.join(HERE, "libssdhip_prof.so"))
sys.path.insert(0, os.path.dirname(HERE))
import torch            # noqa: E402

from ssd_keras_amd import _native as nat       # noqa: E402


def read(lib):
    buf = (ctypes.c_ulonglong * 32)()
    assert lib.ssdhip_profile_read_c64(buf, 1) == 0
    return list(buf)


def report(name, fn):
    lib = nat.load()
    fn()
    torch.cuda.synchronize()
    read(lib)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    g = read(lib)
    n0, n4 = max(g[8], 1), max(g[24], 1)
    print("%-16s %.1f us | multiplier tiles %d: K loop %.0f  barrier %.0f  epilogue %.0f cycles/tile | wave 4 tiles %d: %s"
          % (name, a.elapsed_time(e) * 1e3, g[8], g[0] / n0, g[1] / n0, g[2] / n0, g[24], "  ".join("%.0f" % (v / n4) for v in g[16:24])), flush=True)
    # wave 4 columns: loader = wait for halo | barrier | issue; producer = request | wait older request | MFMAs | epilogue | LDS drain | barrier | patch stores | gathers


x3 = torch.randn((32, 300, 300, 3), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w1 = (torch.randn((64, 3, 3, 3), device="cuda") / 5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b1 = torch.randn((64,), device="cuda").to(torch.bfloat16)
x64 = torch.randn((32, 300, 300, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w2 = (torch.randn((64, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b2 = torch.randn((64,), device="cuda").to(torch.bfloat16)
x21 = torch.randn((32, 150, 150, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w21 = (torch.randn((128, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b21 = torch.randn((128,), device="cuda").to(torch.bfloat16)
report("conv1_block", lambda: nat.conv1_block(x3, w1, b1, w2, b2, relu=True, pool=True))
report("conv1_2+pool", lambda: nat.conv3x3_c64(x64, w2, b2, relu=True, pool=True))
report("conv2_1", lambda: nat.conv3x3_c64(x21, w21, b21, relu=True, pool=False))
