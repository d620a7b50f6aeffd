"""Event timings of the phases of one SSD300 training step (GPU box): encoder / forward / loss / backward / optimizer, libssdhip
forward (fused_training) on and off."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("MIOPEN_BENCH", "1") == "1"
cfg = syn.SSD300_VOC
B = 32
torch.manual_seed(4321)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"]).to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-7, momentum=0.9)
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
images = torch.from_numpy(np.random.RandomState(100).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
lf = SSDLoss()
res = {}
for fused in (True, False, True):
    model.fused_training = fused
    acc = {k: 0.0 for k in ("encode", "forward", "loss", "backward", "optimizer", "wall")}
    n = 0
    import time
    for it in range(8):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        y_true, _, _ = enc.encode_to_device(gt, device=dev)
        ev[1].record()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y_pred = model(images)
        ev[2].record()
        loss = lf.compute_loss(y_true, y_pred.float()).mean()
        ev[3].record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        ev[4].record()
        opt.step()
        ev[5].record()
        t1 = time.perf_counter()                                   # host time to ISSUE the step
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 3:
            n += 1
            for i, k in enumerate(("encode", "forward", "loss", "backward", "optimizer")):
                acc[k] += ev[i].elapsed_time(ev[i + 1])
            acc["wall"] += 1e3 * (t2 - t0)
            acc["host_issue"] = acc.get("host_issue", 0.0) + 1e3 * (t1 - t0)
    res["fused" if fused else "framework"] = {k: round(v / n, 3) for k, v in acc.items()}
    print(("fused" if fused else "framework"), json.dumps(res["fused" if fused else "framework"]), flush=True)
