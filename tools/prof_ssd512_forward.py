"""SSD512 (21 classes) forward + DecodeDetections at batch 16 (BASELINE configs[4]'s batch) on the model's own fused bf16 path, eager: a few
steps for `rocprofv3 --kernel-trace --stats` (tools/gpu.sh stats=prof_ssd512_forward).  GPU box."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd512 import ssd_512  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c5 = syn.SSD512_COCO
torch.manual_seed(5)
model = ssd_512((512, 512, 3), 20, mode="inference", scales=[0.07, 0.15, 0.3, 0.45, 0.6, 0.75, 0.9, 1.05],
                aspect_ratios_per_layer=c5["aspect_ratios_per_layer"], steps=c5["steps"], offsets=c5["offsets"],
                confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400)
model = model.cuda().to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
images = torch.from_numpy(np.random.RandomState(B).randint(0, 256, size=(B, 512, 512, 3)).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(4):
        y = model(images)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = model(images)
    e.record()
    e.synchronize()
print("ssd512 batch %d: %.3f ms per step, output %s" % (B, a.elapsed_time(e) / 10, tuple(y.shape)))
