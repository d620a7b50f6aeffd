"""The pooled slab kernel on tiles of the STACKED batch (default) against tiles per image (SSDHIP_CONVH_STACK=0), alternating in one
process, bit-identity checked: conv3_3 + pool3 of SSD300 at batch 32 (760 against 800 position tiles x 2 channel tiles: six rounds of
256 CUs against seven), its float16 x 3 form, and conv2_2 + pool2 (a tie: the plan does not change).  GPU box."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


def timed(fn, reps=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for (B, H, W, Cin, Cout, x3) in ((32, 75, 75, 256, 256, False), (32, 150, 150, 128, 128, False), (32, 75, 75, 256, 256, True), (16, 75, 75, 256, 256, False)):
    g = torch.Generator(device="cuda").manual_seed(7)
    if x3:
        xf = (torch.randn((B, Cin, H, W), generator=g, device="cuda") * 30).relu().contiguous(memory_format=torch.channels_last)
        wf = torch.randn((Cout, Cin, 3, 3), generator=g, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
        bias = torch.randn((Cout,), generator=g, device="cuda")
        pw, oscale = nat.x3_pack_weight(wf)
        xs = nat.x3_split(xf)
        fn = lambda: nat.conv2d_x3(xs, pw, bias, oscale, stride=1, padding=1, dilation=1, relu=True, pool=True)
    else:
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
        fn = lambda: nat.conv3x3_halo(x, w, b, relu=True, pool=True)
    os.environ["SSDHIP_CONVH_STACK"] = "0"
    base = fn().clone()
    for mode in ("0", "1", "0", "1", "0", "1"):
        os.environ["SSDHIP_CONVH_STACK"] = mode
        plan = nat.conv3x3_halo_plan(B, H, W, True)
        t = timed(fn)
        same = torch.equal(fn().view(torch.int16), base.view(torch.int16))
        print("%s %dx%dx%dx%d -> %d + pool  STACK=%s plan %s  %.1f us  identical %s" % ("x3  " if x3 else "bf16", B, H, W, Cin, Cout, mode, plan, t, same),
              flush=True)
os.environ.pop("SSDHIP_CONVH_STACK", None)
