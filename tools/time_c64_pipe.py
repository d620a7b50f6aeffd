"""conv2_1 (64 -> 128 channels, 150 x 150, batch 32) on the Cin = 64 kernel: the two-block form (SSDHIP_C64_PIPE=0) against the
pipelined-epilogue form, alternating in one process; bit-identity checked.  GPU box.  (The form is NOT in the product: apply
profiles/r06zp_c64_pipelined_epilogue_not_adopted.patch first -- see profiles/r06zp_c64_pipelined_epilogue_negative.txt.)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for (B, H, W, Cout) in ((32, 150, 150, 128), (32, 300, 300, 64), (32, 75, 75, 64)):
    x = torch.randn((B, H, W, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Cout, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((Cout,), device="cuda").to(torch.bfloat16)
    os.environ["SSDHIP_C64_PIPE"] = "0"
    base = nat.conv3x3_c64(x, w, b, relu=True, pool=False)
    for mode in ("0", "1", "0", "1", "0", "1"):
        os.environ["SSDHIP_C64_PIPE"] = mode
        t = timed(lambda: nat.conv3x3_c64(x, w, b, relu=True, pool=False))
        same = torch.equal(nat.conv3x3_c64(x, w, b, relu=True, pool=False), base)
        print("c64 %dx%dx%d -> %d  PIPE=%s  %.1f us  identical %s" % (B, H, W, Cout, mode, t, same), flush=True)
os.environ.pop("SSDHIP_C64_PIPE", None)
