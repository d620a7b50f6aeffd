"""The synthetic SSD300 training leg (bench_extra.train_leg, eager launches) under rocprofv3 --kernel-trace --stats: which kernels make
up the step.  GPU box.   rocprofv3 --kernel-trace --stats -d OUT -o train -- python tools/prof_train.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SSD_TRAIN_GRAPH"] = "0"
os.environ["SSD_TRAIN_RAW"] = "0"
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402

r = bx.train_leg(torch.device("cuda:0"), 0, 1, 32, steps=6, warmup=3, tame=True)
print(json.dumps({k: r.get(k) for k in ("ms_per_step", "eager_ms_per_step", "first_loss", "final_loss", "error")}))
