"""The reference-precision step (model.precise(): float32 model on the float16 x 3 MFMA path + DecodeDetections), eager and as ONE HIP
graph (model.graphed), with the parity of the graphed output against the eager one.  GPU box.
  python tools/time_precise.py            timings as one JSON line
  TRACE=1 python tools/time_precise.py    only a few steps (for rocprofv3 --kernel-trace --stats)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402

B = int(os.environ.get("B", "32"))
cfg = syn.SSD300_VOC
torch.manual_seed(1234)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                nms_max_output_size=400).cuda().to(memory_format=torch.channels_last).eval()
with torch.no_grad():
    for head in model.conf_heads:
        head.weight.mul_(1e-3)
        head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
    for head in model.loc_heads:
        head.weight.mul_(1e-3)
images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).cuda()
model.precise()


def ev_ms(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


reps = 3 if os.environ.get("TRACE") else 10
res = {"B": B}
with torch.no_grad():
    out_eager = model(images).clone()
    res["eager_step_ms"] = round(ev_ms(lambda: model(images), reps), 4)
    res["eager_forward_ms"] = round(ev_ms(lambda: model.raw_predictions(images), reps), 4)
    try:
        runner = model.graphed(images)
        out_graph = runner(images).clone()
        res["graph_step_ms"] = round(ev_ms(lambda: runner(images), reps), 4)
        res["graph_equals_eager"] = bool(torch.equal(out_graph, out_eager))
        res["images_per_sec_graph"] = round(B / (res["graph_step_ms"] * 1e-3), 1)
    except Exception as exc:                                  # noqa: BLE001
        res["graph_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
res["detections_per_image"] = float((out_eager[:, :, 1] > 0).sum(dim=1).float().mean())
print(json.dumps(res), flush=True)
