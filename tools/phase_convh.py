#!/usr/bin/env python3
"""In-kernel phase timers of the slab kernel (needs tools/libssdhip_prof.so: tools/prof_build.sh).  GPU box only.
Per layer: shader cycles per tile of waves 0 and 4 (SIMD 0's pair) in K loop | epilogue | end-of-tile barrier, against the MFMA floor of
the K loop (a SIMD's two waves issue 2 x MFMAs-per-wave x 32 cycles)."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("SSDHIP_LIB", os.path.join(HERE, "libssdhip_prof.so"))
sys.path.insert(0, os.path.dirname(HERE))
import torch            # noqa: E402

from ssd_keras_amd import _native as nat       # noqa: E402

LAYERS = [("conv2_2", 32, 150, 150, 128, 128), ("conv3_1", 32, 75, 75, 128, 256), ("conv3_2", 32, 75, 75, 256, 256), ("conv4_1", 32, 38, 38, 256, 512),
          ("conv4_2", 32, 38, 38, 512, 512), ("conv5_1", 32, 19, 19, 512, 512)]


def read(lib):
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.ssdhip_profile_read_convh(buf, 1) == 0
    return list(buf)


lib = nat.load()
for name, B, H, W, Cin, Cout in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    fn = lambda: nat.conv2d_same(x, wt, bias, relu=True, variant=7)
    fn()
    torch.cuda.synchronize()
    read(lib)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    g_ = read(lib)
    # a wave's tile is 64 channels x 64 positions: per (tap, 16 input channels) 4 MFMAs; two waves per SIMD
    floor = 2 * 4 * 9 * (Cin // 16) * 32
    out = "%-8s %.1f us  K-loop floor %d cycles/tile |" % (name, a.elapsed_time(e) * 1e3, floor)
    for w, base in ((0, 0), (4, 8)):
        n = max(g_[base + 4], 1)
        out += " wave %d (%d tiles): K loop %.0f  epilogue %.0f  barrier %.0f |" % (w, g_[base + 4], g_[base] / n, g_[base + 1] / n, g_[base + 2] / n)
    print(out, flush=True)
