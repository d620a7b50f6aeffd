"""bench_extra.train_leg's flow (E eager steps, capture after two side-stream steps, replays) with switches, to find what makes the
replayed step leave the eager trajectory.  GPU box.
  DBG_EAGER=9  DBG_ENC=1 (encoder call + copy before every replay)  DBG_FUSED=1 (libssdhip autograd functions)  DBG_L2K=1 (L2Norm kernels)
  DBG_HIPLOSS=1 (HIP SSDLoss; 0: a plain PyTorch restatement)  DBG_SYNC=1 (synchronize before the third replay as the leg does)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import _native as nat  # noqa: E402
from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

E = int(os.environ.get("DBG_EAGER", "9"))
ENC = os.environ.get("DBG_ENC", "1") == "1"
dev = torch.device("cuda:0")
cfg = syn.SSD300_VOC
B = 32
if os.environ.get("DBG_L2K", "1") == "0":
    nat.l2_normalize_supported = lambda x: False
torch.manual_seed(4321)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).to(dev)
model = model.to(memory_format=torch.channels_last).train()
if os.environ.get("DBG_FUSED", "1") == "0":
    model.fused_training = False
if os.environ.get("DBG_RELU", "1") == "0":
    nat.relu_bwd_bias = lambda *a: None
    nat.maxpool2_relu_bwd_bias = None
if os.environ.get("DBG_SHADOW", "1") == "0":
    model._bf16_shadow = lambda conv: (None, None)
with torch.no_grad():
    for head in model.conf_heads:
        head.weight.mul_(1e-2)
        head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
    for head in model.loc_heads:
        head.weight.mul_(1e-2)
decay = [p for p in model.parameters() if p.dim() > 1]
plain = [p for p in model.parameters() if p.dim() <= 1]
opt = torch.optim.SGD([{"params": decay, "weight_decay": 1e-3}, {"params": plain, "weight_decay": 0.0}], lr=1e-7, momentum=0.9)
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
images = torch.from_numpy(np.random.RandomState(100).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)


def loss_fn(y_true, y_pred):
    if os.environ.get("DBG_HIPLOSS", "1") == "1":
        return lf.compute_loss(y_true, y_pred.float()).mean()
    # plain PyTorch: softmax log loss + smooth L1 over positives, no mining (a different loss, only to take the HIP loss out)
    C = cfg["n_classes"] + 1
    cls = -(y_true[:, :, :C] * torch.log(y_pred[:, :, :C].float().clamp_min(1e-15))).sum(-1)
    d = (y_true[:, :, C:C + 4] - y_pred[:, :, C:C + 4].float()).abs()
    loc = torch.where(d < 1, 0.5 * d * d, d - 0.5).sum(-1)
    pos = y_true[:, :, 1:C].amax(-1)
    return ((cls * pos).sum(-1) + (loc * pos).sum(-1)).mean() / 10.0


def eager_step():
    y_true, _, _ = enc.encode_to_device(gt, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_pred = model(images)
    loss = loss_fn(y_true, y_pred)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return round(float(loss), 4)


trace = [("eager", eager_step()) for _ in range(E)]
y_static, _, _ = enc.encode_to_device(gt, device=dev)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y_pred = model(images)
        loss = loss_fn(y_static, y_pred)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
if os.environ.get("DBG_KEEP_OLD_GRAPH", "0") != "1":      # 1: the warm-up's `loss` / `y_pred` stay alive through the capture (the bug)
    del y_pred, loss
    import gc
    gc.collect()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_pred = model(images)
    loss_static = loss_fn(y_static, y_pred)
    loss_static.backward()
    if os.environ.get("DBG_OPT_OUTSIDE", "0") != "1":
        opt.step()
if os.environ.get("DBG_SYNC_AFTER_CAPTURE", "0") == "1":
    torch.cuda.synchronize()
for i in range(8):
    if ENC:
        y_true, _, _ = enc.encode_to_device(gt, device=dev)
        y_static.copy_(y_true)
    g.replay()
    if os.environ.get("DBG_OPT_OUTSIDE", "0") == "1":
        opt.step()                                       # the captured backward wrote the .grad tensors the optimizer holds
    if os.environ.get("DBG_SYNC", "1") == "1" and i == 1:
        torch.cuda.synchronize()
    trace.append(("graph", round(float(loss_static), 4)))
print("TRACE", " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DBG_") or k.startswith("SSDHIP_NO")), [v for _, v in trace[E - 2:]])
