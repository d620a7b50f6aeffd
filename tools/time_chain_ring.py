"""conv7_1 ... conv9_2 of SSD300 at batch 32 as ONE launch (csrc/ssdhip_chain.hip) with 8 / 16 / 32 filter fragments in flight per wave
(SSDHIP_CHAIN_RING), alternating in one process, bit-identity checked; and the float16 x 3 chain with 4 / 8 K-steps of twins
(SSDHIP_CHAIN_X3_RING) through models/precise.py.  GPU box."""
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


B = 32
spec = [(1, 1, 0, 128, 1, 0), (3, 2, 1, 256, 1, 1), (1, 1, 0, 128, 1, 0), (3, 1, 0, 256, 1, 1), (1, 1, 0, 128, 1, 0), (3, 1, 0, 256, 1, 1)]
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((B, 10, 10, 512), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
layers, cin = [], 512
for (k, s, pd, cout, relu, keep) in spec:
    w = (torch.randn((cout, k, k, cin), generator=g, device="cuda") / (k * k * cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((cout,), generator=g, device="cuda").to(torch.bfloat16)
    layers.append(dict(packed=nat.conv_chain_pack(w), bias=bias, k=k, stride=s, pad=pd, cout=cout, relu=relu, keep=bool(keep)))
    cin = cout
os.environ["SSDHIP_CHAIN_RING"] = "8"
base = [t.clone() for t in nat.conv_chain(x, layers)]
for ring in ("8", "16", "32", "8", "16", "32", "8", "16", "32"):
    os.environ["SSDHIP_CHAIN_RING"] = ring
    t = timed(lambda: nat.conv_chain(x, layers))
    same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(nat.conv_chain(x, layers), base))
    print("bf16 chain  ring %2s  %.1f us  identical %s" % (ring, t, same), flush=True)
os.environ.pop("SSDHIP_CHAIN_RING", None)

# the float16 x 3 chain through the reference-precision model's own packing
from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.models.precise import PreciseForward  # noqa: E402

cfg = syn.SSD300_VOC
torch.manual_seed(3)
m32 = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
              steps=cfg["steps"], offsets=cfg["offsets"]).cuda().eval()
for prm in m32.parameters():
    prm.requires_grad_(False)
pf = PreciseForward(m32)
pf._scale, pf._packed, pf._calibrated = {}, {}, True
xx = (torch.rand((B, 512, 10, 10), device="cuda", generator=g) * 50.0).contiguous(memory_format=torch.channels_last)
act = (nat.x3_split(xx / 2.0), 2.0)
convs = [m32.conv7_1, m32.conv7_2, m32.conv8_1, m32.conv8_2, m32.conv9_1, m32.conv9_2]
os.environ["SSDHIP_CHAIN_X3_RING"] = "4"
base3 = [t[0].clone() for t in pf._extras_chain(act, convs)]
for ring in ("4", "8", "4", "8", "4", "8"):
    os.environ["SSDHIP_CHAIN_X3_RING"] = ring
    t = timed(lambda: pf._extras_chain(act, convs), reps=50)
    same = all(torch.equal(a[0].view(torch.int16), b.view(torch.int16)) for a, b in zip(pf._extras_chain(act, convs), base3))
    print("x3 chain    ring %2s  %.1f us  identical %s" % (ring, t, same), flush=True)
os.environ.pop("SSDHIP_CHAIN_X3_RING", None)
