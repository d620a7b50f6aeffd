"""Which stage of the training step refuses HIP graph capture?  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

dev = torch.device("cuda:0")
cfg = syn.SSD300_VOC
B = 8
torch.manual_seed(0)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                steps=cfg["steps"], offsets=cfg["offsets"]).to(dev).to(memory_format=torch.channels_last).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-7, momentum=0.9)
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
images = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
lf = SSDLoss()
y_true, _, _ = enc.encode_to_device(gt, device=dev)


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model(images)


def fwd_loss():
    return lf.compute_loss(y_true, fwd().float()).mean()


def fwd_loss_bwd():
    loss = fwd_loss()
    loss.backward()
    return loss


def full():
    loss = fwd_loss_bwd()
    opt.step()
    return loss


def fwd_plain_bwd():
    y = fwd()
    y[:, :, :25].float().sum().backward()


for name, fn in (("forward", fwd), ("forward+loss", fwd_loss), ("forward+plain backward", fwd_plain_bwd), ("forward+loss+backward", fwd_loss_bwd),
                 ("full step", full)):
    for fused in (True, False):
        model.fused_training = fused
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    opt.zero_grad(set_to_none=True)
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                fn()
            g.replay()
            torch.cuda.synchronize()
            print("%-28s fused_training=%s: captured and replayed" % (name, fused), flush=True)
        except Exception as exc:                                    # noqa: BLE001
            print("%-28s fused_training=%s: FAILED %s: %s" % (name, fused, type(exc).__name__, str(exc).splitlines()[0][:150]), flush=True)
            torch.cuda.synchronize()
