"""Reproducer / regression aid: two HIP graphs captured from ONE model, the older one replayed on a stream that is not its capture stream.
On ROCm 7.2 that segfaults inside hipGraphLaunch when the graph object is launched there directly (argv[1] = "raw"); GraphedInference
routes such a replay through its capture stream (argv[1] = "guarded", the default).  Prints OK <mode> when the replays survive and equal
the eager step.  GPU box; run in its own process (tests/test_end_to_end_gpu.py does)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "guarded"
cfg = syn.SSD300_VOC
torch.manual_seed(1234)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                nms_max_output_size=400).cuda().to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(4, 300, 300, 3)).astype(np.float32)).cuda()
with torch.no_grad():
    want = model(images).clone()
    first = model.graphed(images.clone())
    second = model.graphed(images.clone())
    foreign = torch.cuda.Stream()
    with torch.cuda.stream(foreign):
        if mode == "raw":
            first.graph.replay()
            out1 = first.static_out
        else:
            out1 = first(None)
        out2 = second(None)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out1, want)) and bool(torch.equal(out2, want))
print(("OK " if ok else "MISMATCH ") + mode, flush=True)
