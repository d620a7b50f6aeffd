"""Does the reference-precision forward's three-stream schedule depend on WHICH streams it gets?  In a process that has already created
other streams (bench.py: graph capture, decode, head overlap), ordinary-priority side streams can land on the hardware queue of the
default stream or of each other and the overlap is lost (8.3 ms inside bench.py against 7.1 ms alone, r03zz).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.models.precise import PreciseForward  # noqa: E402

cfg = syn.SSD300_VOC
torch.manual_seed(1234)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"]).cuda().to(memory_format=torch.channels_last).eval()
images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(32, 300, 300, 3)).astype(np.float32)).cuda()


def timed(pf):
    for _ in range(3):
        pf(images)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        pf(images)
    b.record()
    b.synchronize()
    return round(a.elapsed_time(b) / 5, 3)


keep = []
for prio in (0, -1):
    fixed, picked = [], []
    for n in range(6):
        keep.append(torch.cuda.Stream())                 # one more ordinary stream taken from the pool before each trial
        pf = PreciseForward(model, stream_priority=prio)
        pf._side[str(images.device)] = (torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio))     # the first pair, unpicked
        fixed.append(timed(pf))
        picked.append(timed(PreciseForward(model, stream_priority=prio)))                                         # the picked pair
    print("side-stream priority %2d: forward ms, first pair %s" % (prio, fixed), flush=True)
    print("side-stream priority %2d: forward ms, picked pair %s" % (prio, picked), flush=True)
