"""Regression test for the training leg's HIP-graph replay: forward + loss + backward captured once, the SGD step
issued eagerly after every replay (bench_extra.train_leg).  In round 3 the replays with the optimizer INSIDE the capture left the eager
trajectory after the first device-wide synchronize (30.5 -> 96, profiles/r03zk_train_graph_bisect.txt); the workaround had no test.

Whole trajectories cannot be compared tightly: at a small batch the optimisation is chaotic (losses 35 -> 25 -> 36 -> 29 -> 52) and the
framework's remaining backward kernels are not bit-reproducible (two eager runs of the same seed part by 1e-4 at the third step), so
the test compares STEP BY STEP on the same weights: before every update the loss and the gradients of an eager forward + backward must
equal those of a graph replay; then the eager optimizer step, then a device-wide synchronize -- five rounds.  A replay that reads stale
or corrupted state after the synchronize shows up in the next round.  Needs an MI355X.
Reference: the optimisation loop behind model.fit_generator, ssd300_training.ipynb:171-173, cell 14 (SGD, momentum 0.9)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager_step_on_the_same_weights_across_synchronizes():
    import torch
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    dev = torch.device("cuda:0")
    cfg = syn.SSD300_VOC
    B = 2
    torch.manual_seed(4321)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).to(dev)
    model = model.to(memory_format=torch.channels_last).train()
    with torch.no_grad():                                 # tamed heads (bench_extra.train_leg): a softmax neither saturated nor uniform
        for head in model.conf_heads:
            head.weight.mul_(1e-2)
            head.bias.view(-1, cfg["n_classes"] + 1)[:, 0] = 4.0
        for head in model.loc_heads:
            head.weight.mul_(1e-2)
    # lr 1e-8: at batch 2 the leg's 1e-7 is past the edge of stability (36 -> 30 -> 28 -> 53 -> inf gradients, in eager mode and with the
    # framework's own autograd alike: tools/debug_graph_rounds.py, profiles/r04i_graph_rounds.txt) and the comparison would measure
    # chaos, not the replay.  The weights still move every round, and the loss with them.
    opt = torch.optim.SGD(model.parameters(), lr=1e-8, momentum=0.9)
    enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
    gt = syn.make_ground_truth(B, cfg["n_classes"], 300, 300, max_boxes=8, seed=7)
    images = torch.from_numpy(np.random.RandomState(100).randint(0, 256, size=(B, 300, 300, 3)).astype(np.float32)).to(dev)
    lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
    y_static, _, _ = enc.encode_to_device(gt, device=dev)

    def fwd_bwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y_pred = model(images)
        loss = lf.compute_loss(y_static, y_pred.float()).mean()
        loss.backward()
        return loss

    with torch.cuda.device(dev):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                     # warm-up off the default stream: autotune, workspaces, allocator
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
        torch.cuda.synchronize()
        watched = [p for p in model.parameters() if p.grad is not None]
        assert len(watched) > 30
        graph_grads = [p.grad for p in watched]           # the tensors the replay writes and the eager optimizer reads
        losses = []
        for rnd in range(5):
            # eager forward + backward on the current weights, into FRESH gradient tensors
            for p in watched:
                p.grad = None
            le = float(fwd_bwd().detach())
            eager_grads = [p.grad.detach().clone() for p in watched]
            for p, gg in zip(watched, graph_grads):
                p.grad = gg
            g.replay()
            lg = float(loss_static.detach())
            assert abs(lg - le) <= 2e-3 * abs(le), "round %d: replayed loss %.6f, eager loss %.6f" % (rnd, lg, le)
            worst = 0.0
            for p, ge in zip(watched, eager_grads):
                den = float(ge.float().norm()) + 1e-20
                worst = max(worst, float((p.grad.float() - ge.float()).norm()) / den)
            assert worst <= 5e-2, "round %d: a replayed gradient is %.3g of its norm away from the eager one" % (rnd, worst)
            print("round %d: loss eager %.5f replay %.5f, worst gradient distance %.3g of its norm" % (rnd, le, lg, worst))
            opt.step()                                    # eager, on the replay's gradients
            torch.cuda.synchronize()                      # the round-3 failure needed a device-wide synchronize between replays
            losses.append(lg)
        assert all(np.isfinite(losses)), losses
        assert len(set(losses)) == len(losses), "the weights move every round, so must the loss: %s" % losses
