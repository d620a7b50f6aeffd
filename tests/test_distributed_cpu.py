"""The N > 1 path on CPU: two gloo ranks on 127.0.0.1.  Checks the sharding arithmetic, the max-over-ranks
timing reduction bench.py uses, and that the bucketed gradient all-reduce of two half-batches reproduces the
single-process gradient of the full batch (the data-parallel training step of BASELINE configs[3])."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ssd_keras_amd import distributed as dp


def test_shard_range_partitions():
    for n in (1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build():
    from ssd_keras_amd.models.keras_ssd7 import build_model
    torch.manual_seed(0)
    m = build_model((64, 64, 3), 3, scales=[0.1, 0.3, 0.5, 0.7, 0.9], normalize_coords=True)
    return m.eval()                       # BatchNorm on running statistics: shards and full batch see the same function


def _loss(pred):
    c = pred.shape[2] - 12
    return (pred[:, :, :c + 4] ** 2).mean()


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = dp.data_parallel(_build(), bucket_cap_mb=1)
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    lo, hi = dp.shard_range(4, rank, world)
    _loss(model(x[lo:hi])).backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    t = dp.max_over_ranks(1.0 + rank)
    s = dp.sum_over_ranks(1.0)
    torch.save({"grads": grads, "t": t, "s": s}, os.path.join(tmp, "r%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]
    assert torch.equal(outs[0]["grads"], outs[1]["grads"])              # both ranks hold the averaged gradient
    assert outs[0]["t"] == 2.0 and outs[1]["t"] == 2.0 and outs[0]["s"] == 2.0
    model = _build()
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    _loss(model(x)).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(outs[0]["grads"], ref, rtol=1e-4, atol=1e-6)


@pytest.mark.timeout(300)
def test_four_rank_gradient_allreduce_matches_single_process(tmp_path):
    """world_size 4 (one image per rank): the averaged gradient of four shards is the full batch's, the max / sum reductions see
    every rank."""
    port = _free_port()
    mp.spawn(_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(4)]
    for o in outs[1:]:
        assert torch.equal(outs[0]["grads"], o["grads"])
    assert all(o["t"] == 4.0 and o["s"] == 4.0 for o in outs)
    model = _build()
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    _loss(model(x)).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # batch-1 shards take other float32 convolution algorithms on the CPU than the batch-4 reference: compare in the norm
    err = float((outs[0]["grads"] - ref).norm() / ref.norm())
    assert err < 1e-3, err
    torch.testing.assert_close(outs[0]["grads"], ref, rtol=0.0, atol=1e-3 * float(ref.abs().max()))


def _worker_ssd300(rank, world, port, tmp):
    """The model bench.py's training leg wraps in DDP (SSD300/VOC, mode='training'), two steps on two gloo ranks: every
    parameter must take part in the reduction (an unused one makes DDP raise on the second forward)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dp.init_from_env("gloo")
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    cfg = syn.SSD300_VOC
    torch.manual_seed(4321)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).train()
    ddp = dp.data_parallel(model)
    opt = torch.optim.SGD(model.parameters(), lr=1e-9, momentum=0.9)
    x = torch.from_numpy(np.random.RandomState(10 + rank).randint(0, 256, size=(1, 300, 300, 3)).astype(np.float32))
    for _ in range(2):
        pred = ddp(x)
        assert pred.shape == (1, 8732, 33)
        loss = _loss(pred)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    grads = torch.cat([p.grad.reshape(-1)[:64] for p in model.parameters()])
    torch.save({"missing": missing, "grads": grads, "n": sum(p.numel() for p in model.parameters())}, os.path.join(tmp, "s%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_ddp_over_the_ssd300_training_model(tmp_path):
    port = _free_port()
    mp.spawn(_worker_ssd300, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r)) for r in range(2)]
    assert outs[0]["missing"] == [] and outs[1]["missing"] == []
    assert outs[0]["n"] == 26285486                                      # SSD300 / 21 classes
    assert torch.equal(outs[0]["grads"], outs[1]["grads"])              # averaged gradients are identical on both ranks


def _worker_bn(rank, world, port, tmp):
    """SSD7 in train mode: BatchNorm running statistics must stay identical on both ranks (rank 0's are broadcast each forward),
    and the averaged gradient must be the mean of the two per-shard gradients (the documented data-parallel semantics)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    model = _build().train()
    ddp = dp.data_parallel(model, bucket_cap_mb=1)
    x = torch.from_numpy(np.random.RandomState(5).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    lo, hi = dp.shard_range(4, rank, world)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        _loss(ddp(x[lo:hi])).backward()
    model.eval()
    ddp(x[lo:hi])                                                        # buffers are synchronised at the start of a forward
    bufs = torch.cat([b.detach().float().reshape(-1) for n, b in model.named_buffers() if "num_batches" not in n])
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    torch.save({"bufs": bufs, "grads": grads}, os.path.join(tmp, "b%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_batchnorm_buffers_stay_in_step(tmp_path):
    port = _free_port()
    mp.spawn(_worker_bn, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "b%d.pt" % r)) for r in range(2)]
    assert outs[0]["bufs"].numel() > 0 and torch.equal(outs[0]["bufs"], outs[1]["bufs"])       # rank 0's running statistics everywhere
    assert torch.equal(outs[0]["grads"], outs[1]["grads"])


def _worker_sgd(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dp.init_from_env("gloo")
    from ssd_keras_amd.optimizers import SGD
    model = dp.data_parallel(_build(), bucket_cap_mb=1)
    decay = [p for p in model.parameters() if p.dim() == 4]
    plain = [p for p in model.parameters() if p.dim() != 4]
    opt = SGD([{"params": decay, "weight_decay": 1e-3}, {"params": plain, "weight_decay": 0.0}], lr=1e-8, momentum=0.9)
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    lo, hi = dp.shard_range(4, rank, world)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        _loss(model(x[lo:hi])).backward()
        opt.step()
    torch.save(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), os.path.join(tmp, "p%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_training_steps_with_the_package_sgd_match_single_process(tmp_path):
    """Round 5: `ssd_keras_amd.optimizers.SGD` (the reference's keras.optimizers.SGD(lr, momentum=0.9), ssd300_training.ipynb:169; on a GPU
    one libssdhip launch, here its tensor-expression path) under DDP on two gloo ranks: three steps leave both ranks with the
    parameters three single-process torch.optim.SGD steps on the whole batch produce."""
    port = _free_port()
    mp.spawn(_worker_sgd, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "p%d.pt" % r)) for r in range(2))
    assert torch.equal(a, b)
    model = _build()
    decay = [p for p in model.parameters() if p.dim() == 4]
    plain = [p for p in model.parameters() if p.dim() != 4]
    opt = torch.optim.SGD([{"params": decay, "weight_decay": 1e-3}, {"params": plain, "weight_decay": 0.0}], lr=1e-8, momentum=0.9)
    x = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 64, 64, 3)).astype(np.float32))
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        _loss(model(x)).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    init = torch.cat([p.detach().reshape(-1) for p in _build().parameters()])
    moved = (ref - init).abs()
    assert float(moved.max()) > 1e-6                                      # the steps did something (tiny lr: raw 0..255 inputs)
    d = ((a - init) - (ref - init)).abs()
    assert float(d.max()) <= 1e-3 * float(moved.max()) + 1e-9, (float(d.max()), float(moved.max()))


def _run_bench(argv, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, cwd=root, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.timeout(400)
def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it creates two ranks itself (the flag was parsed and
    never read); the dry launch brings the process group up and counts the ranks with an all-reduce of ones."""
    import json
    res = _run_bench(["--gpus", "2", "--dry-launch"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                                   # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["dry_launch"] is True and rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["ranks_seen"] == 2


@pytest.mark.timeout(200)
def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Under a launcher that created a different number of ranks than --gpus says, the bench stops instead of printing a line whose
    n_gpus lies (world 1 pretending to be 2)."""
    res = _run_bench(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0
    assert "--gpus 2" in res.stderr and "WORLD_SIZE=1" in res.stderr
