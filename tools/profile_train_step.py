"""Kernel-trace target for the training step (BASELINE configs[2]): bench_extra.train_leg on one GPU.
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o train -- python tools/profile_train_step.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as be  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("MIOPEN_BENCH", "1") == "1"
print(json.dumps(be.train_leg(dev, 0, 1, 32, steps=int(os.environ.get("STEPS", "4")), warmup=3)))
