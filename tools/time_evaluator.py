"""bench_extra.evaluator_leg alone (GPU box): Evaluator.match_predictions on the VOC2007-test-sized synthetic problem."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402

print(json.dumps(bx.evaluator_leg(torch.device("cuda", 0), with_cpu=os.environ.get("CPU", "1") == "1")), flush=True)
