"""The SSD300 training step of bench_extra.train_leg (BASELINE configs[2]) alone: ms per step graph-replayed and eager, first / final
loss.  Environment switches are read by the product code: SSDHIP_NO_OWN_WGRAD=1 (framework weight gradients), SSDHIP_NO_OWN_DGRAD=1.

    python tools/time_train_leg.py [batch]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
os.environ.setdefault("SSD_TRAIN_RAW", "0")
r = bx.train_leg(torch.device("cuda:0"), 0, 1, B, steps=10, warmup=3, tame=True)
print(json.dumps({k: r.get(k) for k in ("ms_per_step", "eager_ms_per_step", "images_per_sec", "first_loss", "final_loss", "launch", "error")}))
