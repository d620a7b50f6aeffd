"""Which predictor-head path the graphed SSD300 step takes (one stream / two streams), by logging the grouped launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402
from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402

cfg = syn.SSD300_VOC
dev = torch.device("cuda:0")
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"], confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400).to(dev)
model = model.to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(32, 300, 300, 3)).astype(np.float32)).to(dev)
orig = nat.conv3x3_halo_group


def logged(xs, ws, biases=None, relu=False, max_workgroups=0):
    print("halo_group: %d problems, max_workgroups %d, stream %x, overlap attr %r" % (
        len(xs), max_workgroups, torch.cuda.current_stream().cuda_stream, model.__dict__.get("_head_overlap")), flush=True)
    return orig(xs, ws, biases, relu=relu, max_workgroups=max_workgroups)


nat.conv3x3_halo_group = logged
with torch.no_grad():
    model(images)
    print("--- graphed()", flush=True)
    runner = model.graphed(images)
    print("--- replay", flush=True)
    runner(images)
torch.cuda.synchronize()
