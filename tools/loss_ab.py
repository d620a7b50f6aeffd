"""Loss kernels A/B (GPU box): the streaming L1 / backward kernels against the tiled ones in ONE process (csrc/ssdhip_loss.hip reads
SSDHIP_LOSS_STREAM per call), on the SSD300 / batch 32 tensors of tools/time_loss.py.  Under `rocprofv3 --kernel-trace --stats` the
kernel names tell the variants apart.
    python tools/loss_ab.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import np_oracle as orc  # noqa: E402
from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402


def events_ms(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        t = a.elapsed_time(b) / reps
        best = t if best is None else min(best, t)
    return best


dev = torch.device("cuda:0")
cfg = syn.SSD300_VOC
B = 32
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
ora = orc.EncoderOracle(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
y_true, _, _ = enc.encode_to_device(gt, device=dev)
av = ora.generate_encoding_template(1)[0, :, -8:]
C = enc.n_classes
sparse = syn.make_y_pred(av, B, C, bias=7.0, seed=1234)
ties = sparse.copy()
ties[:, :, :C] = 1.0 / C
lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
rows = []
for name, host in (("sparse", sparse), ("all_ties", ties)):
    want = orc.ssd_loss(y_true.cpu().numpy(), host)
    for stream in ("1", "0", "1", "0"):
        os.environ["SSDHIP_LOSS_STREAM"] = stream
        yp = torch.from_numpy(host).to(dev).requires_grad_(True)
        with torch.no_grad():
            fwd = events_ms(lambda: lf.compute_loss(y_true, yp.detach()))

        def fb():
            yp.grad = None
            lf.compute_loss(y_true, yp).sum().backward()
        both = events_ms(fb)
        got = lf.compute_loss(y_true, yp).detach().cpu().numpy()
        row = {"case": name, "stream": stream, "fwd_us": round(1e3 * fwd, 1), "fwd_bwd_us": round(1e3 * both, 1),
               "loss_within_1e-4_of_oracle": bool(np.allclose(got, want, rtol=1e-4, atol=1e-6))}
        print(json.dumps(row), flush=True)
        rows.append(row)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
