"""One reference-precision forward (models/precise.py) under rocprofv3: which kernels take the 10 ms.  GPU box."""
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
pf = PreciseForward(model)
for _ in range(3):
    pf(images)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    pf(images)
b.record()
b.synchronize()
print("x3 forward: %.3f ms" % (a.elapsed_time(b) / 5))
