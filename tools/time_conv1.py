"""Timing of the first-layer convolution (3 -> 64 channels, 300 x 300, batch 32).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402

x = torch.randn((32, 300, 300, 3), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w = (torch.randn((64, 3, 3, 3), device="cuda") / 5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b = torch.randn((64,), device="cuda").to(torch.bfloat16)
for _ in range(3):
    y = nat.conv3x3_cin3(x, w, b, relu=True)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    y = nat.conv3x3_cin3(x, w, b, relu=True)
e.record()
e.synchronize()
ms = a.elapsed_time(e) / 50
print("conv1_1: %.1f us, output write %.2f TB/s" % (1e3 * ms, y.numel() * 2 / ms / 1e9))

# conv1_1 -> conv1_2 -> pool1: two kernels against the fused one
w2 = (torch.randn((64, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b2 = torch.randn((64,), device="cuda").to(torch.bfloat16)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / reps * 1e3


two = timed(lambda: nat.conv3x3_c64(nat.conv3x3_cin3(x, w, b, relu=True), w2, b2, relu=True, pool=True))
c64 = timed(lambda: nat.conv3x3_c64(y, w2, b2, relu=True, pool=True))
one = timed(lambda: nat.conv1_block(x, w, b, w2, b2, relu=True, pool=True))
same = torch.equal(nat.conv1_block(x, w, b, w2, b2, relu=True, pool=True),
                   nat.conv3x3_c64(nat.conv3x3_cin3(x, w, b, relu=True), w2, b2, relu=True, pool=True))
print("conv1_1 + conv1_2 + pool1: two kernels %.1f us (conv1_2 + pool alone %.1f), fused %.1f us, identical %s" % (two, c64, one, same))

import os
for prio in ("0", "1"):
    os.environ["SSDHIP_C64_PRIO"] = prio
    print("SSDHIP_C64_PRIO=%s: fused %.1f us" % (prio, timed(lambda: nat.conv1_block(x, w, b, w2, b2, relu=True, pool=True))))
os.environ.pop("SSDHIP_C64_PRIO", None)
