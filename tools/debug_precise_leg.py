"""bench_extra.fp32x3_forward_leg in a fresh process, alone (argv: a) or behind the MIOpen float32 leg as in bench.py (argv: b):
does the graph replay of the reference-precision step depend on what ran before it?  GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_extra as bx  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if "m" in mode:
    torch.backends.cudnn.benchmark = True
keep = []
if "h" in mode:                                           # the headline's graphed bf16 step first, its runner kept alive (as bench.py does)
    import numpy as np
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    cfg = syn.SSD300_VOC
    m = ssd_300((300, 300, 3), cfg["n_classes"], mode="inference", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
                nms_max_output_size=400).to(dev).to(memory_format=torch.channels_last).eval().to(torch.bfloat16)
    im = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(32, 300, 300, 3)).astype(np.float32)).to(dev)
    with torch.no_grad():
        r = m.graphed(im)
        for _ in range(30):
            r(im)
    torch.cuda.synchronize()
    if "k" in mode:
        keep += [m, r, im]
    else:
        del m, r, im
        torch.cuda.empty_cache()
n_dummy = sum(ch == "d" for ch in mode)                  # "d" x n: n streams created (and used once) between the two graphs
for _ in range(n_dummy):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        torch.zeros(8, device=dev).add_(1)
    keep.append(st)
torch.cuda.synchronize()
if "e" in mode:
    keep.append(bx.encoder_leg(dev, 32, False))
if "l" in mode:
    keep.append(bx.loss_leg(dev, 32, False))
if "s" in mode:
    keep.append(bx.sparse_decode_leg(dev, 32, False))
if "5" in mode:
    keep.append(bx.ssd512_decode_leg(dev, False))
first = bx.fp32_forward_leg(dev, 32) if "b" in mode else None
if "p" in mode:                                           # the same leg twice: is it "any second graph of the process"?
    pre = bx.fp32x3_forward_leg(dev, 32, first)
    print(json.dumps({"first_run_graph": pre.get("graph")}), flush=True)
    del pre
    torch.cuda.empty_cache()
leg = bx.fp32x3_forward_leg(dev, 32, first)
print(json.dumps({"mode": mode, "launch": leg.get("launch"), "graph": leg.get("graph"), "step_ms": leg.get("step_ms_fwd_plus_decode"),
                  "error": leg.get("error")}), flush=True)
