"""The Cin = 64 kernel with its loader wave (SSDHIP_C64_SELF=0) against the two-workgroups-per-CU form whose multiplying waves request
their own halos (SSDHIP_C64_SELF=1), alternating in one process, bit-identity checked: conv2_1, conv1_2 un-pooled / pooled / pool-keep.
(NOT in the product: apply profiles/r06zu_c64_two_workgroups_per_cu_not_adopted.patch first.)
GPU box."""
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


for (B, H, W, Cout, what) in ((32, 150, 150, 128, "plain"), (32, 300, 300, 64, "pool"), (32, 300, 300, 64, "plain"), (32, 300, 300, 64, "keep"),
                              (32, 75, 75, 64, "plain"), (3, 5, 7, 64, "plain"), (40, 16, 16, 64, "pool")):
    x = torch.randn((B, H, W, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn((Cout, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
    b = torch.randn((Cout,), device="cuda").to(torch.bfloat16)
    if what == "keep":
        fn = lambda: nat.conv3x3_c64_pool_keep(x, w, b, relu=True)
    else:
        fn = lambda: (nat.conv3x3_c64(x, w, b, relu=True, pool=(what == "pool")),)
    os.environ["SSDHIP_C64_SELF"] = "0"
    base = [t.clone() for t in fn()]
    for mode in ("0", "1", "0", "1", "0", "1"):
        os.environ["SSDHIP_C64_SELF"] = mode
        t = timed(fn)
        same = all(torch.equal(a_, b_) for a_, b_ in zip(fn(), base))
        print("c64 %dx%dx%d -> %d %-5s SELF=%s  %.1f us  identical %s" % (B, H, W, Cout, what, mode, t, same), flush=True)
os.environ.pop("SSDHIP_C64_SELF", None)
