#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: the decode path (K3 scan / K4 nms / K5 topk) on the SAME
predictions bench.py decodes (SSD300 VGG-16, 21 classes, batch 32, random-init weights, seed as bench.py rank 0),
launched REPS times.  Run under `rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...` in separate passes
(MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_keras_amd import _native as nat          # noqa: E402
from ssd_keras_amd import synthetic as syn         # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300   # noqa: E402

REPS = int(os.environ.get("REPS", "5"))
B = int(os.environ.get("B", "32"))


def main():
    nat.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    cfg = syn.SSD300_VOC
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"],
                    confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400).to(dev)
    model = model.to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
    images = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
    with torch.no_grad():
        pred = model.raw_predictions(images)
        torch.cuda.synchronize()
        for _ in range(REPS):
            out = model.decoder(pred)
    torch.cuda.synchronize()
    print("decoded", tuple(out.shape), "reps", REPS)


if __name__ == "__main__":
    main()
