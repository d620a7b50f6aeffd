"""conv6_1 / fc7 (1x1 layers of the image-resident kernel) on whole-image tiles (SSDHIP_CONVIMG_PXT=0) against the default tile pick
(conv6_1: 128 channels x 128 pixels), alternating in one process, bit-identity checked; bf16 and float16 x 3.  GPU box."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


def timed(fn, reps=100):
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


for (B, Cin, Cout, x3, name) in ((32, 1024, 256, False, "conv6_1"), (32, 1024, 1024, False, "fc7"), (32, 1024, 256, True, "conv6_1 x3"),
                                 (32, 256, 1024, False, "conv6_1 data gradient"), (8, 1024, 256, False, "conv6_1 batch 8")):
    g = torch.Generator(device="cuda").manual_seed(3)
    if x3:
        xf = (torch.randn((B, Cin, 19, 19), generator=g, device="cuda") * 30).relu().contiguous(memory_format=torch.channels_last)
        wf = torch.randn((Cout, Cin, 1, 1), generator=g, device="cuda") * (2.0 / Cin) ** 0.5
        bias = torch.randn((Cout,), generator=g, device="cuda")
        pw, oscale = nat.x3_pack_weight(wf)
        xs = nat.x3_split(xf)
        os.environ["SSDHIP_X3_NO_HALO"] = "1"
        fn = lambda: nat.conv2d_x3(xs, pw, bias, oscale, stride=1, padding=0, dilation=1, relu=True)
    else:
        x = torch.randn((B, 19, 19, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        w = (torch.randn((Cout, 1, 1, Cin), generator=g, device="cuda") / Cin ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
        fn = lambda: nat.conv2d_image(x, w, b, relu=True)
    os.environ["SSDHIP_CONVIMG_PXT"] = "0"
    base = fn().clone()
    for mode in ("0", "1", "0", "1", "0", "1"):
        os.environ["SSDHIP_CONVIMG_PXT"] = mode
        t = timed(fn)
        same = torch.equal(fn().view(torch.int16), base.view(torch.int16))
        print("%-22s batch %2d  PXT=%s  %.1f us  identical %s" % (name, B, mode, t, same), flush=True)
os.environ.pop("SSDHIP_CONVIMG_PXT", None)
