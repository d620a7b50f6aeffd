"""bench_extra.augmentation_leg alone (GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402

leg = bx.augmentation_leg(torch.device("cuda", 0), 32, with_cpu=False)
print(json.dumps({k: v for k, v in leg.items() if k.startswith("augment_batch")}), flush=True)
