import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ssd_keras_amd import _native as nat
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return a.elapsed_time(e) / reps * 1e3
x = torch.randn((32, 150, 150, 64), device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
w = (torch.randn((128, 3, 3, 64), device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
b = torch.randn((128,), device="cuda").to(torch.bfloat16)
base = nat.conv3x3_c64(x, w, b, relu=True, pool=False)
for prio in ("0", "1", "0", "1"):
    os.environ["SSDHIP_C64_PRIO"] = prio
    t = timed(lambda: nat.conv3x3_c64(x, w, b, relu=True, pool=False))
    print("conv2_1 c64 prio", prio, round(t, 1), torch.equal(nat.conv3x3_c64(x, w, b, relu=True, pool=False), base))
