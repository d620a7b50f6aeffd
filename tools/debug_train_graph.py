"""Bisect a non-finite final loss of bench_extra.train_leg: graph replay vs eager, L2Normalization kernels vs framework ops.  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402
from ssd_keras_amd import _native as nat  # noqa: E402

dev = torch.device("cuda:0")
os.environ["SSD_TRAIN_RAW"] = "0"
keep = nat.l2_normalize_supported
for graph, l2k in (("1", True), ("1", False), ("0", True)):
    os.environ["SSD_TRAIN_GRAPH"] = graph
    nat.l2_normalize_supported = keep if l2k else (lambda x: False)
    r = bx.train_leg(dev, 0, 1, 32, steps=6, warmup=3, tame=True)
    print("graph", graph, "l2norm kernels", l2k, json.dumps({k: r.get(k) for k in ("first_loss", "final_loss", "ms_per_step", "launch", "error")}), flush=True)
