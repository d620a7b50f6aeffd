"""ssdhip_conv3x3_wgrad_nhwc_bf16 (csrc/ssdhip_wgrad.hip): the weight gradient of the 3x3 'same' convolutions of the training graph
(models/keras_ssd300.py:274-296 under model.fit_generator, ssd300_training.ipynb:171-173) against a float32 / float64 PyTorch reference
of the same op on the same bf16 operands.  Needs an MI355X.

Bar: the kernel multiplies exact bf16 products and accumulates in float32 (MFMA), so against the float64 sum of the same products the
error of one output is bounded by float32 summation noise: |got - want| <= 2^-18 * sum|dy||x| per element is comfortable for up to
3 M positions split over >= 8 ordered partial sums; and two runs are BIT-identical (fixed summation order, no atomics)."""
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, Cin, Cout
    (2, 19, 19, 128, 128),       # conv5-like map, one tile pair
    (3, 10, 7, 64, 128),         # Cin = 64 (conv2_1's channel shape), narrow map: several row wraps per 64-position block
    (2, 38, 38, 256, 256),       # four (co, ci) tiles x 8 ci tiles
    (1, 75, 75, 128, 256),       # conv3_1: two halo blocks
    (1, 150, 150, 64, 128),      # conv2_1: three halo blocks, the ring wraps many times
    (1, 40, 150, 128, 128),      # conv2_2's width
    (1, 60, 300, 64, 64),        # conv1_2's width: the 64-output-channel form (two K halves per block), five halo blocks
    (2, 1, 1, 64, 128),          # one pixel per image: eight of the nine taps see only padding
    (1, 5, 190, 64, 128),        # the widest map of the 128-channel form
    (4, 19, 19, 512, 512),       # conv5_x at a small batch: 32 tiles
]


def _ref(x, dy):
    import torch
    # d/dw of sum(conv(x, w) * dy): exact bf16 operands, float64 accumulation
    return torch.nn.grad.conv2d_weight(x.double(), (dy.shape[1], x.shape[1], 3, 3), dy.double(), stride=1, padding=1)


@pytest.mark.parametrize("case", CASES)
def test_wgrad_matches_float64_reference(case):
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = dy * (torch.rand((B, Cout, H, W), generator=g, device="cuda") > 0.4).to(torch.bfloat16)      # a ReLU-masked gradient has zeros
    got = nat.conv3x3_wgrad(x, dy)
    assert got is not None, "geometry must be supported"
    assert got.shape == (Cout, Cin, 3, 3) and got.dtype == torch.float32
    want = _ref(x, dy)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), (Cout, Cin, 3, 3), dy.double().abs(), stride=1, padding=1)
    err = (got.double() - want).abs()
    bad = int((err > mag * 2.0 ** -18 + 1e-30).sum().item())
    assert bad == 0, "%d of %d weight gradients off (worst %.3g of bound)" % (bad, got.numel(), float((err / (mag * 2.0 ** -18 + 1e-30)).max()))
    again = nat.conv3x3_wgrad(x, dy)
    assert torch.equal(again, got), "two runs must be bit-identical"


def test_wgrad_full_batch_conv4_and_race_screen():
    """BASELINE configs[2] sizes: conv4_2 at batch 32 (32 tiles x 8 splits = 256 workgroups), ten launches bit-identical, and within the
    convolution bar of the framework's own float32 weight gradient."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout = 32, 38, 38, 512, 512
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    dy = torch.randn((B, H, W, Cout), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    got = nat.conv3x3_wgrad(x, dy)
    want = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin, 3, 3), dy.float(), stride=1, padding=1)
    rms = want.pow(2).mean().sqrt().item()
    assert float((got - want).abs().max()) <= 1e-3 * rms
    for _ in range(10):
        assert torch.equal(nat.conv3x3_wgrad(x, dy), got)


def test_wgrad_unsupported_geometries_return_none():
    import torch
    from ssd_keras_amd import _native as nat
    x = torch.zeros((1, 3, 8, 8), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(nat.SsdHipError):
        nat.conv3x3_wgrad(x, torch.zeros((1, 64, 8, 8), device="cuda", dtype=torch.bfloat16))          # 3 channels: not a multiple of 8
    x = torch.zeros((1, 32, 8, 8), device="cuda", dtype=torch.bfloat16)
    assert nat.conv3x3_wgrad(x, torch.zeros((1, 64, 8, 8), device="cuda", dtype=torch.bfloat16)) is None   # Cin % 64 != 0
    x = torch.zeros((1, 64, 4, 200), device="cuda", dtype=torch.bfloat16)
    assert nat.conv3x3_wgrad(x, torch.zeros((1, 128, 4, 200), device="cuda", dtype=torch.bfloat16)) is None  # too wide for the 128-channel form


@pytest.mark.parametrize("shape", [(3, 19, 19, 1024, 126, 24), (2, 38, 38, 512, 84, 16), (4, 1, 1, 256, 84, 16), (2, 5, 5, 256, 126, 24)])
def test_packed_head_node_matches_two_framework_convolutions(shape):
    """models/_common.py _PackedHeadFn (the two predictor heads of a source map in the training step: slab-kernel forward and data
    gradient, MFMA weight gradient, one bias reduction; reference models/keras_ssd300.py:322-335) against the two float32 framework
    convolutions on the same bf16-rounded operands: outputs within one bf16 rounding, every gradient within 1e-2 of its norm."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd.models._common import _PackedHeadFn
    B, H, W, Cin, nc, nl = shape
    g = torch.Generator(device="cuda").manual_seed(B * H + nc)
    x = torch.randn((B, Cin, H, W), generator=g, device="cuda").to(memory_format=torch.channels_last).to(torch.bfloat16).float().requires_grad_(True)
    mk = lambda n: (torch.randn((n, Cin, 3, 3), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).float().requires_grad_(True)
    wc, wl = mk(nc), mk(nl)
    bc = torch.randn((nc,), generator=g, device="cuda").to(torch.bfloat16).float().requires_grad_(True)
    bl = torch.randn((nl,), generator=g, device="cuda").to(torch.bfloat16).float().requires_grad_(True)
    go = torch.randn((B, nc + nl, H, W), generator=g, device="cuda").to(torch.bfloat16).float()
    want = torch.cat([F.conv2d(x, wc, bc, 1, 1), F.conv2d(x, wl, bl, 1, 1)], dim=1)
    (want * go).sum().backward()
    ref = [t.grad.clone() for t in (x, wc, bc, wl, bl)]
    for t in (x, wc, bc, wl, bl):
        t.grad = None
    # the packed bf16 filters / biases / transposed-flipped filters as the model's shadow set builds them (csrc/ssdhip_optim.hip)
    from ssd_keras_amd import _native as nat
    cp = -(-(nc + nl) // 128) * 128
    pw = torch.zeros((cp, Cin, 3, 3), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    pwt = torch.zeros((Cin, cp, 3, 3), dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)
    pb = torch.zeros((cp,), dtype=torch.bfloat16, device="cuda")
    table = nat.shadow_table([(wc.detach(), pw[:nc], pwt, cp, 0), (wl.detach(), pw[nc:nc + nl], pwt, cp, nc)],
                             [(bc.detach(), pb[:nc]), (bl.detach(), pb[nc:nc + nl])], x.device)
    nat.shadow_refresh(table)
    y = _PackedHeadFn.apply(x, wc, bc, wl, bl, pw, pb, pwt)
    assert y.shape[0] == B and y.shape[1] % 128 == 0 and y.dtype == torch.bfloat16
    got = y[:, :nc + nl].float()
    rms = want.pow(2).mean().sqrt().item()
    assert int(((got - want).abs() > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0
    (got * go).sum().backward()
    for name, t, r in zip(("x", "w conf", "b conf", "w loc", "b loc"), (x, wc, bc, wl, bl), ref):
        d = float((t.grad - r).norm() / (r.norm() + 1e-20))
        assert d <= 1e-2, "%s gradient %.3g of its norm away" % (name, d)


@pytest.mark.parametrize("shape", [(4, 19, 19, 512, 512), (2, 38, 38, 256, 128), (1, 7, 9, 64, 64)])
def test_weight_gradient_launch_also_finishes_the_bias_gradient(shape):
    """ssdhip_conv3x3_wgrad_bias_nhwc_bf16: the per-workgroup channel sums the ReLU backward leaves behind are added by extra
    workgroups of the weight gradient's reduction launch -- db within float32 summation-order noise of the framework's sum, the same
    bits on every run, and dw bit-identical to the call without them."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(H * Cout)
    x = torch.randn((B, Cin, H, W), generator=g, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.randn((B, Cout, H, W), generator=g, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, Cout, H, W), generator=g, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    masked, partial = nat.relu_bwd_bias(gy, y, reduce=False)
    assert partial.dim() == 2 and partial.shape[1] == Cout
    dw0 = nat.conv3x3_wgrad(x, masked)
    dw1, db = nat.conv3x3_wgrad(x, masked, bias_partial=partial)
    assert torch.equal(dw0, dw1)
    torch.testing.assert_close(db, masked.float().sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(db, partial.sum(dim=0), rtol=1e-5, atol=1e-4)
    assert torch.equal(db, nat.conv3x3_wgrad(x, masked, bias_partial=partial)[1])
    plain = nat.channel_sums_partial(gy)                                  # no activation: the predictor heads' form
    torch.testing.assert_close(plain.sum(dim=0), gy.float().sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("case", [(32, 19, 19, 1024, 1024), (32, 19, 19, 1024, 256), (32, 10, 10, 512, 128), (2, 3, 3, 256, 128), (1, 1, 1, 128, 128)])
def test_1x1_weight_gradient_as_a_gemm_over_the_pixels(case):
    """csrc/ssdhip_wgrad.hip, conv1x1_wgrad_kernel (fc7, conv6_1 ... conv9_1) against the float64 weight gradient of the same bf16
    tensors, with the bias partials riding in the reduction launch; bit-reproducible run to run; channel counts off 128 -> None."""
    import torch
    from ssd_keras_amd import _native as nat
    b, h, w, cin, cout = case
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.randn((b, cin, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((b, cout, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    part = nat.channel_sums_partial(dy)
    gw, gb = nat.conv1x1_wgrad(x, dy, bias_partial=part)
    want = torch.einsum("bohw,bihw->oi", dy.double(), x.double())
    assert gw.shape == (cout, cin, 1, 1) and gw.dtype == torch.float32
    scale = float(want.abs().max().clamp_min(1e-6))
    assert float((gw.view(cout, cin).double() - want).abs().max()) <= 1e-3 * scale
    assert torch.allclose(gb.double(), dy.double().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    gw2, gb2 = nat.conv1x1_wgrad(x, dy, bias_partial=part)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    assert nat.conv1x1_wgrad(x[:, :64], dy) is None


def _double_conv_grads(x, w, gy, stride, padding, dilation):
    """d/dx, d/dw of conv2d(x, w) . gy in float64 through unfold (no library convolution involved)."""
    import torch
    import torch.nn.functional as F
    x64 = x.double().contiguous().requires_grad_(True)
    w64 = w.double().contiguous().requires_grad_(True)
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    cols = F.unfold(x64, 3, dilation=dilation, padding=padding, stride=stride)            # [B, Cin 9, Ho Wo]
    y = (w64.view(cout, -1) @ cols).view(b, cout, gy.shape[2], gy.shape[3])
    (y * gy.double()).sum().backward()
    return x64.grad, w64.grad


TAP_CASES = [  # B, H, W, Cin, Cout, stride, padding, dilation
    (32, 19, 19, 512, 1024, 1, 6, 6),      # fc6
    (32, 19, 19, 256, 512, 2, 1, 1),       # conv6_2
    (32, 10, 10, 128, 256, 2, 1, 1),       # conv7_2
    (32, 5, 5, 128, 256, 1, 0, 1),         # conv8_2
    (32, 3, 3, 128, 256, 1, 0, 1),         # conv9_2
    (3, 7, 9, 128, 128, 2, 1, 1),
    (2, 8, 6, 128, 128, 3, 0, 1),
    (1, 9, 9, 128, 128, 1, 2, 2),
    (5, 6, 7, 256, 128, 2, 0, 1),
]


@pytest.mark.parametrize("case", TAP_CASES + [TAP_CASES[0] + ("gather",), (2, 19, 19, 128, 128, 1, 6, 6), (3, 5, 4, 128, 256, 1, 6, 6)])
def test_3x3_weight_gradient_with_gathered_taps(case, monkeypatch):
    """csrc/ssdhip_wgrad.hip, conv_taps_wgrad_kernel (fc6's dilation, the strided and the 'valid' extras) against the float64 weight
    gradient of the same bf16 tensors, bias partials in the reduction launch; bit-reproducible; unsupported geometry -> None."""
    import torch
    from ssd_keras_amd import _native as nat
    if len(case) == 9:                                  # fc6's geometry goes to the dilated position-grid kernel: keep the gather kernel covered on it too
        monkeypatch.setenv("SSDHIP_WGRAD_GATHER_ONLY", "1")
        case = case[:8]
    b, h, w, cin, cout, s, p, d = case
    ho, wo = (h + 2 * p - 2 * d - 1) // s + 1, (w + 2 * p - 2 * d - 1) // s + 1
    g = torch.Generator(device="cuda").manual_seed(29)
    x = torch.randn((b, cin, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((b, cout, ho, wo), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    part = nat.channel_sums_partial(dy)
    gw, gb = nat.conv3x3_taps_wgrad(x, dy, s, p, d, bias_partial=part)
    wdummy = torch.zeros((cout, cin, 3, 3), device="cuda")
    _, want = _double_conv_grads(x, wdummy, dy, s, p, d)
    assert gw.shape == (cout, cin, 3, 3) and gw.dtype == torch.float32 and gw.permute(0, 2, 3, 1).is_contiguous()
    scale = float(want.abs().max().clamp_min(1e-6))
    assert float((gw.double() - want).abs().max()) <= 1e-3 * scale
    assert torch.allclose(gb.double(), dy.double().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    gw2, gb2 = nat.conv3x3_taps_wgrad(x, dy, s, p, d, bias_partial=part)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    assert nat.conv3x3_taps_wgrad(x[:, :64], dy, s, p, d) is None
    if ho > 1:
        assert nat.conv3x3_taps_wgrad(x, dy[:, :, :-1], s, p, d) is None            # not this convolution's output size


@pytest.mark.parametrize("case", [c for c in TAP_CASES if c[7] == 1 and c[6] in (0, 1)])
def test_strided_and_valid_layers_backward_without_the_framework(case):
    """models/_common.py, _conv_input_weight_grads on the stride-2 / 'valid' 3 x 3 layers: data gradient = embed_strided + the forward's
    'same' kernel on the transposed, flipped filters; weight gradient = the tap-gathered kernel -- against float64, and no
    aten.convolution_backward in the trace."""
    import torch
    from ssd_keras_amd import _native as nat
    from ssd_keras_amd.models import _common as cm
    b, h, w, cin, cout, s, p, d = case
    ho, wo = (h + 2 * p - 3) // s + 1, (w + 2 * p - 3) // s + 1
    g = torch.Generator(device="cuda").manual_seed(31)
    x = torch.randn((b, cin, h, w), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((b, cout, ho, wo), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    z = nat.embed_strided(dy, h, w, s, 1 - p)
    want_z = torch.zeros((b, cout, h, w), device="cuda", dtype=torch.bfloat16)
    want_z[:, :, 1 - p::s, 1 - p::s][:, :, :ho, :wo] = dy
    assert torch.equal(z, want_z)
    part = nat.channel_sums_partial(dy)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        gx, gw, gb = cm._conv_input_weight_grads(dy, x, wt, (s, s), (p, p), (1, 1), True, None, part)
    assert not any("convolution_backward" in e.key for e in prof.key_averages())
    want_x, want_w = _double_conv_grads(x, wt, dy, s, p, d)
    assert gx.shape == x.shape and gb is not None
    sx = float(want_x.abs().max().clamp_min(1e-6))
    assert float((gx.double() - want_x).abs().max()) <= 2.0 ** -7 * sx                  # one bf16 rounding of the result
    sw = float(want_w.abs().max().clamp_min(1e-6))
    assert float((gw.double() - want_w).abs().max()) <= 1e-3 * sw
