"""Kernel-trace target for the training-side kernels (encoder E1-E3, loss L1-L4 + backward): runs the two bench_extra
legs without their CPU baselines.  Usage (on the GPU box):
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o train -- python tools/profile_train_path.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as be  # noqa: E402

dev = torch.device("cuda:0")
print(json.dumps({"encoder": be.encoder_leg(dev, 32, with_cpu=False), "loss": be.loss_leg(dev, 32, with_cpu=False)}))
