"""SSDLoss under data parallelism, on the real kernels.  Needs an MI355X.

`ssd_keras_amd/distributed.py` states what a data-parallel step IS: every rank runs the reference's loss on ITS shard (mining and the
1 / n_positives normalisation per rank, `keras_ssd_loss.py:143-209`), DDP averages the ranks' gradients -- so the step's gradient is
the MEAN over ranks of the reference's per-shard gradients, not the gradient of the reference's loss on the concatenated batch.
Here two ranks (both on GPU 0, talking over gloo: RCCL needs a second GPU) each encode their half of the labels with the HIP
`SSDInputEncoder`, run the HIP `SSDLoss` on their half of the predictions through a DDP-wrapped module, and the all-reduced gradient
is compared with mean_r oracle.ssd_loss_grad(shard r) (1e-4, north_star's tolerance), the targets with the oracle encoder bit for bit.
"""
import os
import socket

import numpy as np
import pytest

from oracle import np_oracle as orc
from ssd_keras_amd import synthetic as syn

pytestmark = pytest.mark.gpu

CFG = syn.SSD7_300
B_GLOBAL = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    """The global batch: labels, predictions of a 'model' whose only parameters are an additive correction of every prediction row."""
    ora = orc.EncoderOracle(**CFG)
    gt = syn.make_ground_truth(B_GLOBAL, CFG["n_classes"], CFG["img_height"], CFG["img_width"], max_boxes=6, seed=11)
    anchors = ora.generate_encoding_template(1)[0, :, -8:]
    y_pred = syn.make_y_pred(anchors, B_GLOBAL, ora.n_classes, bias=3.0, seed=12)
    return ora, gt, y_pred


def _worker(rank, world, port, tmp):
    import torch
    from ssd_keras_amd import distributed as dp
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = dp.init_from_env("gloo")                      # both ranks on GPU 0; the collective itself runs over gloo
    assert (r, w) == (rank, world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _, gt, y_pred = _problem()
    lo, hi = dp.shard_range(B_GLOBAL, rank, world)

    class Correction(torch.nn.Module):                      # predictions = this rank's rows + a shared, trainable correction
        def __init__(self, shape):
            super().__init__()
            self.delta = torch.nn.Parameter(torch.zeros(shape))

        def forward(self, base):
            return base + self.delta

    net = dp.data_parallel(Correction(y_pred[lo:hi].shape).to(dev), device=dev, bucket_cap_mb=1)
    enc = SSDInputEncoder(**CFG)
    y_true = enc.encode_to_device(gt[lo:hi], want_f32=True)[0]          # HIP encoder on the shard's labels
    pred = net(torch.from_numpy(y_pred[lo:hi]).to(dev))
    loss = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0).compute_loss(y_true, pred)      # HIP loss, (b,)
    loss.mean().backward()                                  # what Keras does with the per-item losses; DDP averages over ranks
    torch.cuda.synchronize()
    module = net.module if hasattr(net, "module") else net
    np.savez(os.path.join(tmp, "r%d.npz" % rank), grad=module.delta.grad.cpu().numpy(), y_true=y_true.cpu().numpy(),
             loss=loss.detach().cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_gradient_is_the_mean_of_the_reference_per_shard_gradients(tmp_path):
    import torch.multiprocessing as mp
    from ssd_keras_amd import distributed as dp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(world)]
    assert np.array_equal(outs[0]["grad"], outs[1]["grad"])            # both ranks hold the all-reduced gradient
    ora, gt, y_pred = _problem()
    want = np.zeros_like(outs[0]["grad"], dtype=np.float64)
    for r in range(world):
        lo, hi = dp.shard_range(B_GLOBAL, r, world)
        with np.errstate(invalid="ignore", divide="ignore"):
            y_true = ora(gt[lo:hi]).astype(np.float32)
        got_true = outs[r]["y_true"]
        C = ora.n_classes
        assert np.array_equal(got_true[:, :, :C], y_true[:, :, :C]), "rank %d: class targets differ from the oracle encoder" % r
        np.testing.assert_allclose(got_true, y_true, rtol=1e-6, atol=1e-7)
        b = hi - lo
        np.testing.assert_allclose(outs[r]["loss"], orc.ssd_loss(y_true, y_pred[lo:hi]), rtol=1e-4, atol=1e-6)
        # the reference on THIS shard (mining and n_positives local to it), loss.mean() -> grad_out = 1 / b
        want += orc.ssd_loss_grad(y_true, y_pred[lo:hi], np.full((b,), 1.0 / b)).astype(np.float64)
    want /= world
    np.testing.assert_allclose(outs[0]["grad"], want.astype(np.float32), rtol=1e-4, atol=1e-6)
