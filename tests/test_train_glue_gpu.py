"""Training-step glue kernels (csrc/ssdhip_train.hip) and the convolution data gradient through the forward's MFMA kernel, against
PyTorch-ROCm's own backward ops on the same tensors.  Needs an MI355X.

Bars: ReLU mask bit exact, bias gradient within float32 summation-order noise (rtol 1e-5); max-pool gradient bit exact (same arg-max
rule, bf16 sums of at most nine bf16 terms accumulated in float32); data gradient within one bf16 rounding of a float32-accumulated
sum (the tests/test_conv_gpu.py bar)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from ssd_keras_amd import _native as nat
    return torch, nat


@pytest.mark.parametrize("shape", [(4, 64, 37, 41), (2, 512, 19, 19), (3, 1024, 5, 5), (1, 128, 1, 1)])
def test_relu_backward_and_bias_gradient(shape):
    torch, nat = _t()
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.randn(shape, device="cuda", generator=g).clamp_min(0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.permute(0, 2, 3, 1).reshape(-1)[::97] = float("nan")              # (a view: the memory is NHWC)
    gy = torch.randn(shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out, gb = nat.relu_bwd_bias(gy, y)
    want = torch.ops.aten.threshold_backward(gy, y, 0)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    torch.testing.assert_close(gb, want.float().sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-4)
    out2, gb2 = nat.relu_bwd_bias(gy, y)
    assert torch.equal(gb, gb2)                                           # fixed summation order: reproducible


def test_relu_backward_unsupported_channel_count_falls_back():
    torch, nat = _t()
    y = torch.zeros((1, 24, 3, 3), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert nat.relu_bwd_bias(y, y) is None                                # C / 8 = 3 does not divide 256: the caller uses the framework ops


@pytest.mark.parametrize("shape,k,s,p,ceil", [((2, 64, 75, 75), 2, 2, 0, True), ((2, 64, 38, 38), 2, 2, 0, True), ((3, 128, 19, 19), 3, 1, 1, False),
                                              ((1, 8, 7, 5), 3, 2, 1, True), ((2, 16, 300, 300), 2, 2, 0, True),
                                              # 3 x 3 / 1: channel counts and map sizes outside the per-image form (the neighbourhood kernel)
                                              ((2, 24, 19, 19), 3, 1, 1, False), ((1, 32, 60, 60), 3, 1, 1, False), ((2, 512, 5, 3), 3, 1, 1, False)])
def test_maxpool_backward_matches_pytorch(shape, k, s, p, ceil):
    torch, nat = _t()
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(shape, device="cuda", generator=g).clamp_min(0)       # post-ReLU map: many exact ties at 0
    x = (x * 4).round() / 4                                               # and ties among positive values
    x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
    gy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    (want,) = torch.autograd.grad(y, x, gy)
    got = nat.maxpool_bwd(x.detach(), gy, k, s, p)
    assert got.shape == want.shape
    torch.testing.assert_close(got.float(), want.float(), rtol=2 ** -7, atol=1e-6)
    if s == k:                                                            # disjoint windows: one term per element, no rounding at all
        assert torch.equal(got.view(torch.int16), want.contiguous(memory_format=torch.channels_last).view(torch.int16))


@pytest.mark.parametrize("cin,cout,k,d,hw", [(64, 64, 3, 1, 40), (128, 256, 3, 1, 38), (512, 1024, 3, 6, 19), (1024, 256, 1, 1, 19)])
def test_data_gradient_through_the_forward_kernel(cin, cout, k, d, hw):
    torch, nat = _t()
    g = torch.Generator(device="cuda").manual_seed(3)
    w = (torch.randn((cout, cin, k, k), device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
    w = w.contiguous(memory_format=torch.channels_last)
    gy = torch.randn((2, cout, hw, hw), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
    got = nat.conv2d_same(gy, wt, None, dilation=d, relu=False).float()
    want = torch.nn.grad.conv2d_input((2, cin, hw, hw), w.float(), gy.float(), stride=1, padding=d * (k // 2), dilation=d)
    err = (got - want).abs()
    bound = 2.0 ** -7 * want.abs() + 1e-2 * want.pow(2).mean().sqrt()
    assert bool((err <= bound).all()), float((err - bound).max())
    if k == 3 and d == 1 and cout == 64:
        # a 64-channel dL/dy (conv1_2) takes the resident-filter kernel (csrc/ssdhip_conv64.hip): same bits
        alt = nat.conv3x3_c64(gy, wt, None, relu=False, pool=False)
        assert torch.equal(alt.view(torch.int16), nat.conv2d_same(gy, wt, None, dilation=d, relu=False).view(torch.int16))
    if k == 3 and nat.conv3x3_image_supported(gy, wt, d):
        # the form the training step takes on small maps at batch 32 (models/_common.py _conv_input_weight_grads): same bits
        alt = nat.conv3x3_image(gy, wt, None, dilation=d, relu=False)
        assert torch.equal(alt.view(torch.int16), nat.conv2d_same(gy, wt, None, dilation=d, relu=False).view(torch.int16))


@pytest.mark.parametrize("shape", [(3, 64, 37, 41), (2, 128, 30, 30), (2, 256, 75, 75), (1, 64, 1, 1), (2, 64, 2, 3)])
def test_fused_pool_relu_backward_equals_the_two_kernels(shape):
    """Conv2D(relu) -> MaxPooling2D(2, 2, 'same') backward in one pass (maxpool2_relu_bwd_bias_kernel) against maxpool_bwd followed by
    relu_bwd_bias on the same tensors: masked gradient bit-identical, bias gradient equal up to the order of the float32 sums (round 5:
    one thread per window, not per pixel); and against PyTorch's max_pool2d backward + threshold_backward.  Ties (ReLU zeros fill whole
    windows), NaNs and odd map sizes."""
    torch, nat = _t()
    import torch.nn.functional as F
    g = torch.Generator(device="cuda").manual_seed(3)
    b, c, h, w = shape
    y = torch.randn(shape, device="cuda", generator=g).clamp_min(0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.permute(0, 2, 3, 1).reshape(-1)[::211] = float("nan")
    gp = torch.randn((b, c, (h + 1) // 2, (w + 1) // 2), device="cuda", generator=g).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    gp.permute(0, 2, 3, 1).reshape(-1)[::53] = -0.0
    got, gb = nat.maxpool2_relu_bwd_bias(y, gp)
    gx = nat.maxpool_bwd(y, gp, 2, 2, 0)
    want, wb = nat.relu_bwd_bias(gx, y)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    fin = torch.isfinite(wb)
    assert torch.equal(fin, torch.isfinite(gb))
    assert torch.allclose(gb[fin], wb[fin], rtol=1e-5, atol=1e-5 * float(wb[fin].abs().max().clamp_min(1.0)))
    # and the framework's own backward ops on a NaN-free copy (bf16 gradient of a 2x2 window = one term: exact)
    y2 = torch.nan_to_num(y, nan=0.5)
    yf = y2.float().requires_grad_(True)
    F.max_pool2d(yf, 2, 2, 0, ceil_mode=True).backward(gp.float())
    ref = torch.ops.aten.threshold_backward(yf.grad.to(torch.bfloat16), y2, 0)
    got2, _ = nat.maxpool2_relu_bwd_bias(y2, gp)
    assert torch.equal(got2.float(), ref.float())                           # (-0 and +0 compare equal)


@pytest.mark.parametrize("channels_last_master", [False, True])
def test_shadow_refresh_builds_every_layout_in_one_launch(channels_last_master):
    """csrc/ssdhip_optim.hip: the bf16 channels_last filters, the transposed / tap-flipped filters of the data gradient and the bf16
    biases of many tensors from ONE launch == the framework expressions round 4 ran per layer (cast, contiguous(channels_last),
    flip + permute), bit for bit -- including ragged channel counts (3, 84, 16), 1 x 1 and 4 x 4 filters and the conf / loc heads of a
    source map as rows of one packed tensor."""
    torch, nat = _t()
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(64, 3, 3, 3), (128, 64, 3, 3), (1024, 1024, 1, 1), (256, 128, 4, 4), (96, 160, 3, 3), (84, 512, 3, 3), (16, 512, 3, 3)]
    ws = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    ws[1].view(-1)[::101] = float("nan")
    ws[2].view(-1)[5] = float("inf")
    if channels_last_master:
        ws = [w.contiguous(memory_format=torch.channels_last) for w in ws]
    bs = [torch.randn((s[0],), device="cuda", generator=g) for s in shapes]
    cl = [torch.empty(s, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last) for s in shapes]
    cl = [c if c.permute(0, 2, 3, 1).is_contiguous() else torch.empty((s[0], s[2], s[3], s[1]), dtype=torch.bfloat16, device="cuda").permute(0, 3, 1, 2)
          for c, s in zip(cl, shapes)]
    tr = [torch.empty((s[1], s[2], s[3], s[0]), dtype=torch.bfloat16, device="cuda").permute(0, 3, 1, 2) for s in shapes]
    # the last two are the conf / loc heads of one map: rows 0..83 and 84..99 of a packed (128, 512, 3, 3) tensor
    pw = torch.zeros((128, 512, 3, 3), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    pwt = torch.zeros((512, 128, 3, 3), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    cl[5], cl[6] = pw[:84], pw[84:100]
    weights = [(w, c, t, s[0], 0) for w, c, t, s in zip(ws[:5], cl[:5], tr[:5], shapes[:5])]
    weights += [(ws[5], cl[5], pwt, 128, 0), (ws[6], cl[6], pwt, 128, 84)]
    bdst = [torch.empty((s[0],), dtype=torch.bfloat16, device="cuda") for s in shapes]
    table = nat.shadow_table(weights, list(zip(bs, bdst)), torch.device("cuda", 0))
    nat.shadow_refresh(table)
    torch.cuda.synchronize()
    bits = lambda t: t.contiguous().view(torch.int16)
    for i, w in enumerate(ws):
        wb = w.to(torch.bfloat16)
        assert torch.equal(bits(cl[i]), bits(wb)), shapes[i]
        assert torch.equal(bits(bdst[i]), bits(bs[i].to(torch.bfloat16)))
        if i < 5:
            assert torch.equal(bits(tr[i]), bits(wb.flip(2, 3).permute(1, 0, 2, 3))), shapes[i]
    packed = torch.cat([ws[5], ws[6], torch.zeros((28, 512, 3, 3), device="cuda")], dim=0).to(torch.bfloat16)
    assert torch.equal(bits(pw), bits(packed))
    assert torch.equal(bits(pwt), bits(packed.flip(2, 3).permute(1, 0, 2, 3)))


def test_fused_sgd_momentum_step_follows_torch_sgd():
    """ssd_keras_amd.optimizers.SGD (one launch over all parameters) against torch.optim.SGD on the same parameters and gradients
    over five steps: within one float32 rounding per step (the framework's kernel contracts p + (-lr) buf into an FMA), ragged sizes,
    channels_last parameters, weight decay in one group only; `_version` of every parameter moves (the bf16 shadows key on it)."""
    torch, _ = _t()
    from ssd_keras_amd.optimizers import SGD
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(64, 3, 3, 3), (7,), (512, 256, 3, 3), (1000003,), (33, 5)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11 + i)))
                  for i, s in enumerate(shapes)]
    a, b = mk(), mk()
    with torch.no_grad():
        for ps in (a, b):
            ps[2].data = ps[2].data.contiguous(memory_format=torch.channels_last)
    ours = SGD([{"params": a[:3], "weight_decay": 1e-3}, {"params": a[3:], "weight_decay": 0.0}], lr=1e-2, momentum=0.9)
    ref = torch.optim.SGD([{"params": b[:3], "weight_decay": 1e-3}, {"params": b[3:], "weight_decay": 0.0}], lr=1e-2, momentum=0.9)
    for step in range(5):
        v0 = [p._version for p in a]
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device="cuda", generator=g)
            if pa.dim() == 4 and not pa.is_contiguous():
                gr = gr.contiguous(memory_format=torch.channels_last)
            pa.grad, pb.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
        ours.step()
        ref.step()
        assert all(p._version > v for p, v in zip(a, v0))
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pa, pb, rtol=3e-7 * (step + 1), atol=1e-7 * (step + 1))
            torch.testing.assert_close(ours.state[pa]["momentum_buffer"], ref.state[pb]["momentum_buffer"], rtol=1e-6, atol=1e-6)


def test_fused_sgd_survives_load_state_dict_and_copies():
    """the fused step caches raw pointers; `load_state_dict` after a step replaces the momentum buffers, a deep copy /
    unpickled optimizer has no cache at all.  step, load_state_dict (a checkpoint taken after step 1), step -- against
    torch.optim.SGD doing the same; the restored momentum is the one used and the live buffers are the ones updated."""
    import copy
    import pickle
    torch, _ = _t()
    from ssd_keras_amd.optimizers import SGD
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(64, 3, 3, 3), (129,), (256, 128, 3, 3)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3 + i)))
                  for i, s in enumerate(shapes)]
    a, b = mk(), mk()
    ours, ref = SGD(a, lr=1e-2, momentum=0.9, weight_decay=1e-4), torch.optim.SGD(b, lr=1e-2, momentum=0.9, weight_decay=1e-4)

    def both_step():
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, device="cuda", generator=g)
            if pa.grad is None:
                pa.grad, pb.grad = gr.clone(), gr.clone()
            else:                                                  # gradients stay where they are: the old key would not change
                pa.grad.copy_(gr)
                pb.grad.copy_(gr)
        ours.step()
        ref.step()

    both_step()
    ck_a, ck_b = copy.deepcopy(ours.state_dict()), copy.deepcopy(ref.state_dict())
    both_step()
    both_step()
    ours.load_state_dict(ck_a)                                     # back to the momentum of step 1 (new buffer tensors)
    ref.load_state_dict(ck_b)
    both_step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(ours.state[pa]["momentum_buffer"], ref.state[pb]["momentum_buffer"], rtol=1e-6, atol=1e-6)
    # copies: no stale table travels, the first step of the copy rebuilds it against ITS tensors
    clone = pickle.loads(pickle.dumps(ours))
    assert clone._tables == {}
    twin = copy.deepcopy(ours)
    before = [p.detach().clone() for p in a]
    tp = [p for grp in twin.param_groups for p in grp["params"]]
    for p, q in zip(tp, a):
        p.grad = q.grad.clone()
    twin.step()
    for q, was in zip(a, before):
        assert torch.equal(q, was)                                 # the original's parameters were not touched by the copy's step
    assert all(not torch.equal(p, was) for p, was in zip(tp, before))


@pytest.mark.parametrize("shape", [(2, 300, 300), (3, 37, 41), (1, 1, 1), (2, 5, 130), (1, 64, 64)])
def test_first_layer_backward_in_one_pass(shape):
    """conv1_1's backward (csrc/ssdhip_train.hip, conv1_1_bwd_kernel: ReLU mask + bias gradient + weight gradient from one read of the
    gradient, the activation and the image) against the three-step form on the same tensors: relu_bwd_bias, then the float64
    weight gradient of the masked gradient.  bf16 operands, float32 accumulation: 1e-3 of the gradient's scale; bias sums to 1e-5."""
    torch, nat = _t()
    b, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((b, 3, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.randn((b, 64, h, w), device="cuda", generator=g).clamp_min(0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((b, 64, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if h * w > 4:
        y.permute(0, 2, 3, 1).reshape(-1)[::97] = float("nan")             # a NaN activation lets the gradient through
    gw, gb = nat.conv1_1_backward(gy, y, x)
    masked, wb = nat.relu_bwd_bias(gy, y)
    want_w = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 3, 3), masked.double(), stride=1, padding=1)
    assert gw.shape == (64, 3, 3, 3) and gw.dtype == torch.float32 and gb.shape == (64,)
    scale = float(want_w.abs().max().clamp_min(1e-6))
    assert float((gw.double() - want_w).abs().max()) <= 1e-3 * scale
    assert torch.allclose(gb, wb, rtol=1e-5, atol=1e-5 * float(wb.abs().max().clamp_min(1.0)))


@pytest.mark.parametrize("C", [21, 36, 40, 81])
def test_training_assembly_node_equals_the_framework_expression(C):
    """_AssembleTrainFn (forward: the one-launch assembly; backward: ssdhip_assemble_predictions_backward_bf16) against the framework
    expression it replaces -- slices of the packed maps, Reshape, Concatenate, softmax, Concatenate with the anchors -- forward to 2e-6,
    the packed gradients to one bf16 rounding (padding channels exactly zero).  n_boxes 4 and 6, maps whose anchors do not fill a tile.
    Class counts at and beyond the old 64 KB LDS limit of the backward kernel (36-40 classes passed the model's gate and
    failed in backward); the gate is now the kernel's own formula, checked here for every case."""
    torch, nat = _t()
    from ssd_keras_amd.models._common import _AssembleTrainFn
    g = torch.Generator(device="cuda").manual_seed(17)
    B = 3
    pad = lambda nb: -(-(nb * (C + 4)) // 128) * 128
    geo = [(7, 9, 4, pad(4)), (5, 5, 6, pad(6)), (3, 2, 6, pad(6)), (1, 1, 4, pad(4))]        # (h, w, n_boxes, packed channels)
    assert nat.assemble_backward_supported(C, [nb for _, _, nb, _ in geo], [cp for _, _, _, cp in geo])
    ys = [(torch.randn((B, h, w, cp), device="cuda", generator=g) * 3).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True) for h, w, nb, cp in geo]
    N = sum(h * w * nb for h, w, nb, cp in geo)
    anchors = torch.rand((N, 8), device="cuda", generator=g)
    n_boxes = [nb for _, _, nb, _ in geo]
    pred = _AssembleTrainFn.apply(anchors, C, n_boxes, *ys)
    wgt = torch.randn(pred.shape, device="cuda", generator=g)
    got = torch.autograd.grad((pred * wgt).sum(), ys)
    refs = [y.detach().clone().requires_grad_(True) for y in ys]
    confs, locs = [], []
    for y, (h, w, nb, cp) in zip(refs, geo):
        t = y.permute(0, 2, 3, 1)
        confs.append(t[..., :nb * C].reshape(B, -1, C))
        locs.append(t[..., nb * C:nb * (C + 4)].reshape(B, -1, 4))
    want = torch.cat([torch.softmax(torch.cat(confs, 1).float(), -1), torch.cat(locs, 1).float(), anchors.unsqueeze(0).expand(B, -1, -1)], 2)
    assert pred.shape == want.shape and torch.allclose(pred, want, rtol=2e-6, atol=2e-7)
    ref = torch.autograd.grad((want * wgt).sum(), refs)
    for a, b_, (h, w, nb, cp) in zip(got, ref, geo):
        assert a.shape == b_.shape and a.dtype == torch.bfloat16
        assert bool((a[:, nb * (C + 4):] == 0).all())
        err = (a.float() - b_.float()).abs()
        assert bool((err <= 2.0 ** -7 * b_.float().abs() + 1e-6).all()), float(err.max())


@pytest.mark.parametrize("which", ["ssd300_80_classes", "ssd512_coco"])
def test_training_step_of_the_other_builders_runs_on_the_own_backward_kernels(which):
    """models/_common.py, _conv_input_weight_grads (round 6): two training steps (HIP encoder, HIP SSDLoss, bf16 autocast) of SSD300 with
    80 classes and of SSD512 -- finite losses and gradients, and aten.convolution_backward only where no kernel applies (SSD300:
    nowhere; SSD512: its 4 x 4 conv10_2 and the two layers around it)."""
    import torch
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.keras_loss_function.keras_ssd_loss import SSDLoss
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    from ssd_keras_amd.models.keras_ssd512 import ssd_512
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    torch.manual_seed(1)
    if which == "ssd512_coco":
        cfg, size, allowed = syn.SSD512_COCO, 512, 3
        model = ssd_512((512, 512, 3), cfg["n_classes"], mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                        aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"])
    else:
        cfg, size, allowed = dict(syn.SSD300_VOC, n_classes=80), 300, 0
        model = ssd_300((300, 300, 3), 80, mode="training", l2_regularization=0.0005, scales=cfg["scales"],
                        aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"])
    model = model.cuda().to(memory_format=torch.channels_last).train()
    B = 2
    enc = SSDInputEncoder(matching_type='multi', pos_iou_threshold=0.5, neg_iou_limit=0.5, **cfg)
    gt = syn.make_ground_truth(B, cfg["n_classes"], size, size, max_boxes=6, seed=3)
    images = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(B, size, size, 3)).astype(np.float32)).cuda()
    lf = SSDLoss(neg_pos_ratio=3, n_neg_min=0, alpha=1.0)
    opt = torch.optim.SGD(model.parameters(), lr=1e-9, momentum=0.9)
    losses, calls = [], []
    for _ in range(2):
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            y_true, _, _ = enc.encode_to_device(gt, device=images.device)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y_pred = model(images)
            loss = lf.compute_loss(y_true, y_pred.float()).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        losses.append(float(loss.detach()))
        calls.append(sum(e.count for e in prof.key_averages() if "convolution_backward" in e.key))
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    assert np.isfinite(losses).all() and losses[1] != losses[0]
    assert max(calls) <= allowed, calls


@pytest.mark.parametrize("shape", [(3, 128, 128, 19, 19), (2, 256, 128, 38, 38), (2, 128, 256, 75, 75), (1, 128, 128, 150, 150),
                                   (2, 128, 128, 9, 100), (1, 256, 128, 40, 130), (16, 128, 512, 32, 32)])
def test_data_gradient_with_the_relu_mask_of_the_layer_below_in_its_epilogue(shape):
    """ssdhip_conv3x3_halo_masked_nhwc_bf16 (round 6, fourth session): the slab kernel's result zeroed where the activation of the
    layer below is <= 0 == the slab kernel followed by relu_bwd_bias_kernel's mask (threshold_backward's rule: -0 and negative values
    block, NaN of either sign lets the gradient through), bit for bit, on every tiling of the kernel (position grid with five / six /
    seven slab pieces, 16 x 16 and 8 x 32 pixel tiles), repeated launches equal."""
    torch, nat = _t()
    B, Cy, Cx, H, W = shape                               # dL/dy channels, dL/dx channels
    g = torch.Generator(device="cuda").manual_seed(H * W + Cx)
    gy = torch.randn((B, Cy, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((Cx, Cy, 3, 3), device="cuda", generator=g) * (2.0 / (9 * Cy)) ** 0.5).to(torch.bfloat16)
    wt = wt.contiguous(memory_format=torch.channels_last)
    act = torch.randn((B, Cx, H, W), device="cuda", generator=g).clamp_min(0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    flat = act.permute(0, 2, 3, 1).reshape(-1)           # (a view: the memory is NHWC)
    flat[::97] = float("nan")
    flat[5::193] = -0.0
    flat[7::211] = -1.5                                  # not a ReLU output, but the rule is threshold_backward's
    flat.view(torch.int16)[11::223] = -64                # 0xffc0: a NaN with the sign bit set
    plain = nat.conv2d_same(gy, wt, None, dilation=1, relu=False, variant=7)
    want = torch.ops.aten.threshold_backward(plain, act, 0)
    fused = nat.relu_bwd_bias(plain, act, reduce=False)
    assert fused is not None and torch.equal(fused[0].view(torch.int16), want.view(torch.int16))
    got = nat.conv3x3_halo_masked(gy, wt, act)
    assert got is not None and got.shape == want.shape
    diff = int((got.view(torch.int16) != want.view(torch.int16)).sum())
    assert diff == 0, "%d of %d outputs differ" % (diff, got.numel())
    for _ in range(5):
        assert torch.equal(nat.conv3x3_halo_masked(gy, wt, act).view(torch.int16), got.view(torch.int16))
    # the channel sums the layer below still needs: the same rows as the one-pass kernel's from a pass over the masked map ...
    assert torch.equal(nat.channel_sums_partial(got), fused[1])
    # ... or -- what the training step takes -- from the masked kernel's own epilogue: every row written, sums in a fixed order
    got2, part = nat.conv3x3_halo_masked(gy, wt, act, sums=True)
    assert torch.equal(got2.view(torch.int16), got.view(torch.int16)) and part.dtype == torch.float32 and part.shape[1] == Cx
    sums = torch.nan_to_num(got.float()).sum(dim=(0, 2, 3))
    torch.testing.assert_close(torch.nan_to_num(part).sum(0), sums, rtol=1e-4, atol=2e-2)
    part.fill_(float("nan"))                              # (the allocator hands the same block back: a row the kernel skips would show)
    del part
    for _ in range(3):
        again = nat.conv3x3_halo_masked(gy, wt, act, sums=True)[1]
        assert bool(torch.isfinite(again).all()) or bool(torch.isnan(got.float()).any())
        torch.testing.assert_close(torch.nan_to_num(again).sum(0), sums, rtol=1e-4, atol=2e-2)
    a, b = nat.conv3x3_halo_masked(gy, wt, act, sums=True)[1], nat.conv3x3_halo_masked(gy, wt, act, sums=True)[1]
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))                           # reproducible
    assert nat.conv3x3_halo_masked(gy[:, :64].contiguous(memory_format=torch.channels_last), wt[:, :64].contiguous(memory_format=torch.channels_last),
                                   act) is None         # not the slab kernel's geometry: the caller keeps its own path


def test_relu_links_of_the_training_step_change_no_gradient(monkeypatch):
    """models/_common.py _ReluLink: conv2_2 / conv3_2 / conv3_3 / conv4_2 / conv4_3 hand the layer below a data gradient that is already
    masked, and that layer skips its mask pass -- every parameter gradient of an SSD300 training step equals, BIT FOR BIT, the step with
    the links off (SSDHIP_NO_MASKED_DGRAD=1), five masked launches were made and five mask passes were not."""
    import torch
    from ssd_keras_amd import _native as nat
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    cfg = syn.SSD300_VOC
    torch.manual_seed(11)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", scales=cfg["scales"],
                    aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"], steps=cfg["steps"], offsets=cfg["offsets"]).cuda()
    model = model.to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        for head in list(model.conf_heads) + list(model.loc_heads):
            head.weight.mul_(1e-3)
    images = torch.from_numpy(np.random.RandomState(5).randint(0, 256, size=(3, 300, 300, 3)).astype(np.float32)).cuda()
    w = torch.randn(3, 8732, 25, device="cuda")
    counts = {"masked": 0, "relu": 0}
    real_masked, real_relu = nat.conv3x3_halo_masked, nat.relu_bwd_bias

    def counting_masked(*a, **k):
        out = real_masked(*a, **k)
        counts["masked"] += out is not None
        return out

    def counting_relu(*a, **k):
        counts["relu"] += 1
        return real_relu(*a, **k)

    monkeypatch.setattr(nat, "conv3x3_halo_masked", counting_masked)
    monkeypatch.setattr(nat, "relu_bwd_bias", counting_relu)

    def run():
        counts["masked"] = counts["relu"] = 0
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(images)
        (pred[:, :, :25].float() * w).sum().backward()
        return [p.grad.detach().clone() for p in model.parameters()], dict(counts)

    monkeypatch.setenv("SSDHIP_NO_MASKED_DGRAD", "1")
    g_off, c_off = run()
    monkeypatch.delenv("SSDHIP_NO_MASKED_DGRAD")
    g_on, c_on = run()
    assert c_off["masked"] == 0 and c_on["masked"] == 5, (c_off, c_on)
    assert c_on["relu"] == c_off["relu"] - 5, (c_off, c_on)
    names = [n for n, _ in model.named_parameters()]
    # the five lower layers' bias gradients are the same numbers added in another order (the masked kernel's epilogue sums instead of a
    # pass over the map): float32 summation-order noise; everything else bit for bit
    reordered = {"conv2_1.bias", "conv3_1.bias", "conv3_2.bias", "conv4_1.bias", "conv4_2.bias"}
    bad = [n for n, a, b in zip(names, g_on, g_off) if n not in reordered and not torch.equal(a, b)]
    assert not bad, bad
    for n, a, b in zip(names, g_on, g_off):
        if n in reordered:
            torch.testing.assert_close(a.float(), b.float(), rtol=1e-4, atol=1e-4 * float(b.float().abs().max()) + 1e-6)
    assert all(bool(torch.isfinite(g).all()) for g in g_on)
    monkeypatch.setenv("SSDHIP_NO_MASKED_SUMS", "1")      # the sums from a pass over the masked map: every gradient bit for bit
    g_pass, _ = run()
    bad = [n for n, a, b in zip(names, g_pass, g_off) if not torch.equal(a, b)]
    assert not bad, bad
