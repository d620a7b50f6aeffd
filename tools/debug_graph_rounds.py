"""Reproducer / bisect aid for the graph-replayed training step (tests/test_train_graph_gpu.py): SSD300, batch 2, forward + loss + backward
captured once; per round an EAGER forward + backward on the same weights, then a replay, the eager optimizer step and a device-wide
synchronize.  Prints per round: eager loss, replayed loss, worst relative gradient distance.  Switches (environment):
  DBG_SYNC=0        no synchronize between rounds          DBG_EAGER_BETWEEN=0  no eager forward / backward between the replays
  DBG_FUSED=0       framework autograd only                DBG_TORCHLOSS=1      a plain tensor loss instead of SSDLoss
  DBG_ZERO_WS=1     re-zero every libssdhip workspace before each replay        SSDHIP_NO_OWN_WGRAD / _DGRAD = 1
  DBG_FUSED_SGD=1   ssd_keras_amd.optimizers.SGD (one launch)                   DBG_OPT_IN_GRAPH=1   the optimizer step is captured too
                                                                                (round 6: the whole step as ONE graph)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ssd_keras_amd import synthetic as syn  # noqa: E402
from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss  # noqa: E402
from ssd_keras_amd.models.keras_ssd300 import ssd_300  # noqa: E402
from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder  # noqa: E402

E = os.environ.get
dev = torch.device("cuda:0")
cfg = syn.SSD300_VOC
B = int(E("DBG_B", "2"))
torch.manual_seed(4321)
model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).to(dev)
model = model.to(memory_format=torch.channels_last).train()
if E("DBG_FUSED", "1") == "0":
    model.fused_training = False
with torch.no_grad():
    for head in model.conf_heads:
        head.weight.mul_(1e-2)
        head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
    for head in model.loc_heads:
        head.weight.mul_(1e-2)
if E("DBG_FUSED_SGD", "0") == "1":
    from ssd_keras_amd.optimizers import SGD as _SGD
else:
    _SGD = torch.optim.SGD
opt = _SGD(model.parameters(), lr=float(E("DBG_LR", "1e-7")), momentum=0.9)
OPT_IN_GRAPH = E("DBG_OPT_IN_GRAPH", "0") == "1"
enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
images = torch.from_numpy(np.random.RandomState(100).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
y_static, _, _ = enc.encode_to_device(gt, device=dev)


TRACE = []                                            # (name, output) of every traced model call of the current forward
if E("DBG_TRACE", "0") == "1":
    for name in ("conv_act", "conv_act_pool", "conv1_block_pool", "max_pool", "preprocess"):
        def wrap(fn, name=name):
            def inner(*a, **k):
                out = fn(*a, **k)
                TRACE.append((name, out))
                return out
            return inner
        setattr(model, name, wrap(getattr(model, name)))


def shadow_error():
    st = model.__dict__.get("_shadow_state")
    if st is None:
        return float("nan")
    src = [c.weight for c in st["convs"]] + [c.bias for c in st["convs"]]
    return max(float((d.float() - t.detach().to(torch.bfloat16).float()).abs().max()) for d, t in zip(st["dst"], src))


def fwd_bwd():
    del TRACE[:]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_pred = model(images)
    TRACE.append(("y_pred", y_pred))
    if E("DBG_TORCHLOSS", "0") == "1":
        loss = (y_pred.float() - y_static).pow(2).mean() * 100.0
    else:
        loss = lf.compute_loss(y_static, y_pred.float()).mean()
    loss.backward()
    return loss


out = []
with torch.cuda.device(dev):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            fwd_bwd()
            opt.step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        loss_static = fwd_bwd()
        if OPT_IN_GRAPH:
            opt.step()
    torch.cuda.synchronize()
    watched = [p for p in model.parameters() if p.grad is not None]
    graph_grads = [p.grad for p in watched]
    graph_trace = list(TRACE)
    for rnd in range(int(E("DBG_ROUNDS", "6"))):
        le, eager_grads = float("nan"), None
        if E("DBG_EAGER_BETWEEN", "1") == "1":
            for p in watched:
                p.grad = None
            le = float(fwd_bwd().detach())
            eager_trace = [(n, t.detach().float().clone()) for n, t in TRACE]
            eager_grads = [p.grad.detach().clone() for p in watched]
            for p, gg in zip(watched, graph_grads):
                p.grad = gg
        if E("DBG_ZERO_WS", "0") == "1":
            from ssd_keras_amd import _native as nat
            for buf in nat.workspaces._bufs.values():
                buf.zero_()
        g.replay()
        lg = float(loss_static.detach())
        worst = float("nan")
        if eager_grads is not None:
            worst = 0.0
            names = {id(p): n for n, p in model.named_parameters()}
            for p, ge in zip(watched, eager_grads):
                rel = float((p.grad.float() - ge.float()).norm()) / (float(ge.float().norm()) + 1e-20)
                if E("DBG_WORST", "0") == "1" and not rel <= 5e-2:
                    print("ROUND", rnd, "gradient of", names.get(id(p)), tuple(p.shape), "rel %.3g  |eager| %.3g  |graph| %.3g" % (
                        rel, float(ge.float().norm()), float(p.grad.float().norm())), flush=True)
                worst = max(worst, rel)
        if E("DBG_TRACE", "0") == "1" and eager_grads is not None:
            first = None
            for i, ((n, tg_), (_, te)) in enumerate(zip(graph_trace, eager_trace)):
                d = float((tg_.detach().float() - te).abs().max())
                if not (d <= 1e-3 * (float(te.abs().max()) + 1e-30)):
                    first = "%d:%s diff %.3g of max %.3g" % (i, n, d, float(te.abs().max()))
                    break
            print("ROUND", rnd, "first differing traced output:", first, "| shadow error after the replay %.3g" % shadow_error(), flush=True)
        if not OPT_IN_GRAPH:
            opt.step()
        else:
            # the replayed update does not pass through Python: tell the version-keyed caches (the bf16 weight shadows of the EAGER
            # comparison forward; the graph's own forward holds its refresh launch) that the parameters moved
            from ssd_keras_amd.optimizers import _bump_versions
            _bump_versions([p for p in model.parameters()])
        if E("DBG_SYNC", "1") == "1":
            torch.cuda.synchronize()
        out.append("%.4f/%.4f/%.3g" % (le, lg, worst))
print("ROUNDS", " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DBG_") or k.startswith("SSDHIP_NO")), "|", "  ".join(out))
