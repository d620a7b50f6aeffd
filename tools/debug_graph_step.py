"""Is the HIP-graph-replayed training step the same computation as the eager one?  From one saved state (parameters + momentum), K eager
steps vs K replays of a step captured at that state: losses and parameter deltas.  GPU box."""
import copy
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
B = int(os.environ.get("DBG_B", "32"))
K = 4
torch.manual_seed(4321)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).to(dev)
model = model.to(memory_format=torch.channels_last).train()
if os.environ.get("DBG_FUSED", "1") == "0":
    model.fused_training = False
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
y_true, _, _ = enc.encode_to_device(gt, device=dev)
y_true = y_true.clone()


def eager_step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_pred = model(images)
    loss = lf.compute_loss(y_true, y_pred.float()).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return float(loss)


print("warm-up (eager):", [round(eager_step(), 4) for _ in range(3)])
torch.cuda.synchronize()
state_m = copy.deepcopy(model.state_dict())
state_o = copy.deepcopy(opt.state_dict())


def restore():
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(state_m[k])
        for g_now, g_saved in zip(opt.state_dict()["state"].values(), state_o["state"].values()):
            g_now["momentum_buffer"].copy_(g_saved["momentum_buffer"])
    torch.cuda.synchronize()


eager_losses = [round(eager_step(), 4) for _ in range(K)]
torch.cuda.synchronize()
eager_params = [p.detach().clone() for p in model.parameters()]
restore()

side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2):
        eager_step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
restore()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_pred = model(images)
    loss_static = lf.compute_loss(y_true, y_pred.float()).mean()
    loss_static.backward()
    opt.step()
torch.cuda.synchronize()
restore()                                                # capture executes nothing, but be explicit
graph_losses = []
for _ in range(K):
    g.replay()
    torch.cuda.synchronize()
    graph_losses.append(round(float(loss_static), 4))
graph_params = [p.detach().clone() for p in model.parameters()]
print("eager losses ", eager_losses)
print("graph losses ", graph_losses)
names = [n for n, _ in model.named_parameters()]
init = [state_m[n] for n in names]
worst = []
for n, a, b_, i in zip(names, eager_params, graph_params, init):
    da, db = (a - i).float(), (b_ - i).float()
    den = float(da.norm()) + 1e-30
    worst.append((float((da - db).norm()) / den, n, float(da.norm()), float(db.norm())))
worst.sort(reverse=True)
for r in worst[:12]:
    print("rel diff of the parameter update %.3e  %-28s |eager update| %.3e |graph update| %.3e" % r)
