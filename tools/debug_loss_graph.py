"""SSDLoss forward (+ backward) alone inside a HIP graph, replayed on constant inputs: every replay must return the eager value.
(Round 4: the graph-replayed training step computed a wrong LOSS from correct predictions from the second-to-fourth replay on.)
  DBG_B (32), DBG_REPLAYS (8), DBG_SYNC=1 synchronize between replays, DBG_TOUCH=1 an unrelated eager kernel between replays"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

E = os.environ.get
dev = torch.device("cuda:0")
cfg = syn.SSD300_VOC
B = int(E("DBG_B", "32"))
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
y_true, _, _ = enc.encode_to_device(gt, device=dev)
g = torch.Generator(device="cuda").manual_seed(3)
logits = torch.randn((B, 8732, 21), generator=g, device=dev)
logits[:, :, 0] += 4.0
lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
lin = torch.nn.Linear(21, 21).to(dev)                     # something with a gradient in front of the loss


def run():
    conf = torch.softmax(lin(logits), dim=-1)
    y_pred = torch.cat([conf, y_true[:, :, 21:25] + 0.1, y_true[:, :, 25:]], dim=2)
    loss = lf.compute_loss(y_true, y_pred).mean()
    loss.backward()
    return loss


with torch.cuda.device(dev):
    eager = []
    for _ in range(3):
        lin.zero_grad(set_to_none=True)
        eager.append(float(run().detach()))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            lin.zero_grad(set_to_none=True)
            run()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    lin.zero_grad(set_to_none=True)
    with torch.cuda.graph(gr):
        loss_static = run()
    torch.cuda.synchronize()
    out = []
    junk = torch.zeros((64, 1024, 1024), device=dev)
    for i in range(int(E("DBG_REPLAYS", "8"))):
        gr.replay()
        out.append(float(loss_static.detach()))
        if E("DBG_TOUCH", "0") == "1":
            junk.add_(1.0)
        if E("DBG_SYNC", "1") == "1":
            torch.cuda.synchronize()
print("LOSSGRAPH", " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DBG_") or k.startswith("DEBUG_CLR")),
      "| eager", ["%.5f" % v for v in eager], "| replays", ["%.5f" % v for v in out])
