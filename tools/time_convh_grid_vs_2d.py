"""Un-pooled slab convolutions of SSD512 at batch 16 / 8 (conv4_x: 64 x 64 x 512, conv5_x: 32 x 32 x 512) and of SSD300 at batch 32 on the
padded position grid (SSDHIP_CONVH_GRID=1) against the default pick (2-D tiles where they take fewer rounds of one workgroup per CU),
alternating in one process, bit-identity checked.  GPU box."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402


def timed(fn, reps=50):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for (name, B, H, Cin, Cout) in (("ssd512 conv4_2", 16, 64, 512, 512), ("ssd512 conv5_x", 16, 32, 512, 512), ("ssd512 conv4_1", 16, 64, 256, 512),
                                ("ssd512 conv4_2 b8", 8, 64, 512, 512), ("ssd300 conv4_2", 32, 38, 512, 512)):
    g = torch.Generator(device="cuda").manual_seed(H)
    x = torch.randn((B, H, H, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    fn = lambda: nat.conv3x3_halo(x, w, b, relu=True, pool=False)
    os.environ["SSDHIP_CONVH_GRID"] = "1"
    base = fn().clone()
    for mode in ("1", "0", "1", "0", "1", "0"):
        os.environ["SSDHIP_CONVH_GRID"] = mode
        plan = nat.conv3x3_halo_plan(B, H, H, False, Cout)
        t = timed(fn)
        same = torch.equal(fn().view(torch.int16), base.view(torch.int16))
        print("%-18s GRID=%s plan %s  %.1f us  identical %s" % (name, mode, plan, t, same), flush=True)
os.environ.pop("SSDHIP_CONVH_GRID", None)
