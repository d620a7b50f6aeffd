"""Implicit-GEMM MFMA convolution (csrc/ssdhip_conv.hip, through the C ABI) vs a plain PyTorch float32 reference of the
same op on the same bf16-valued inputs.  Needs an MI355X.  Bar: |got - want| <= 2^-7 |want| + 1e-2 * rms (one bf16
rounding of a float32-accumulated sum; the reference accumulates in a different order)."""
import os

import pytest

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, Cin, Cout, k, dil, bias, relu
    (2, 19, 19, 64, 64, 3, 1, True, True),        # BC=64 path, M = 722 (partial last tile)
    (1, 38, 38, 128, 128, 3, 1, True, True),      # BC=128 path
    (2, 10, 7, 64, 256, 3, 1, True, False),       # W < 32: several image rows inside one row group, no ReLU
    (1, 19, 19, 128, 128, 3, 6, True, True),      # fc6-style dilation 6
    (3, 5, 5, 256, 128, 1, 1, False, True),       # 1x1, no bias
    (1, 1, 1, 64, 64, 3, 1, True, True),          # single pixel: only the centre tap is inside the image
    (1, 75, 75, 256, 256, 3, 1, True, True),      # conv3_2 shape, K = 2304
    (2, 300, 300, 64, 64, 3, 1, True, True),      # conv1_2 shape
    (2, 19, 19, 128, 128, 1, 1, True, True),      # 1x1 with two channel slices (strip needed by the very next step)
    (1, 19, 19, 192, 64, 3, 6, True, True),       # dilation 6, three channel slices, BC = 64
    (4, 3, 3, 64, 128, 3, 2, True, True),         # map smaller than the dilated kernel reach
]


@pytest.mark.parametrize("variant", [None, 1, 4, 5, 6])
@pytest.mark.parametrize("case", CASES)
def test_conv_vs_float32_reference(case, variant):
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, k, dil, has_bias, relu = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    # asymmetric weights (a transposed operand or tap order would not survive this)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if has_bias else None
    got = nat.conv2d_same(x, wt, bias, dilation=dil, relu=relu, variant=variant).float()
    want = F.conv2d(x.float(), wt.float(), bias.float() if has_bias else None, 1, dil * (k // 2), dil)
    if relu:
        want = torch.relu(want)
    assert got.shape == want.shape
    rms = want.pow(2).mean().sqrt().item()
    err = (got - want).abs()
    tol = want.abs() * 2.0 ** -7 + 1e-2 * rms
    bad = int((err > tol).sum().item())
    assert bad == 0, "%d of %d outputs off; max err %g (rms %g)" % (bad, err.numel(), err.max().item(), rms)


GENERAL_CASES = [  # B, H, W, Cin, Cout, k, stride, pad, dil, bias, relu
    (32, 19, 19, 256, 512, 3, 2, 1, 1, True, True),   # conv6_2 (ZeroPadding2D(1) + stride 2): 19 -> 10
    (32, 10, 10, 128, 256, 3, 2, 1, 1, True, True),   # conv7_2: 10 -> 5
    (32, 5, 5, 128, 256, 3, 1, 0, 1, True, True),     # conv8_2 ('valid'): 5 -> 3
    (32, 3, 3, 128, 256, 3, 1, 0, 1, True, True),     # conv9_2: 3 -> 1
    (2, 32, 32, 128, 256, 3, 2, 1, 1, True, True),    # SSD512 conv7_2-like, even size: 32 -> 16
    (3, 11, 7, 64, 64, 3, 3, 0, 1, False, False),     # stride 3, no padding, rectangular, no bias / ReLU, BC = 64
    (1, 20, 20, 64, 128, 3, 2, 2, 2, True, True),     # dilation 2 with full padding and stride 2
    (2, 9, 9, 64, 64, 1, 2, 0, 1, True, True),        # 1x1 stride 2 (sub-sampling)
    (1, 16, 16, 64, 64, 3, 1, 1, 1, True, True),      # the 'same' special case through the general entry
    (1, 13, 13, 64, 64, 3, 1, 0, 2, True, True),      # 'valid' with dilation 2: 13 -> 9
]


@pytest.mark.parametrize("variant", [None, 5, 6, 8])
@pytest.mark.parametrize("case", GENERAL_CASES)
def test_general_conv_vs_float32_reference(case, variant):
    """Strided / partially padded convolutions (the SSD extra layers) through ssdhip_conv2d_nhwc_bf16[_variant]; variant 8 is the
    split-K form (ssdhip_conv2d_splitk_nhwc_bf16: K ranges side by side, float32 partial tiles added in a fixed order)."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, k, stride, pad, dil, has_bias, relu = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if has_bias else None
    got = nat.conv2d(x, wt, bias, stride=stride, padding=pad, dilation=dil, relu=relu, variant=variant).float()
    want = F.conv2d(x.float(), wt.float(), bias.float() if has_bias else None, stride, pad, dil)
    if relu:
        want = torch.relu(want)
    assert got.shape == want.shape
    rms = want.pow(2).mean().sqrt().item()
    err = (got - want).abs()
    tol = want.abs() * 2.0 ** -7 + 1e-2 * rms
    bad = int((err > tol).sum().item())
    assert bad == 0, "%d of %d outputs off; max err %g (rms %g)" % (bad, err.numel(), err.max().item(), rms)


@pytest.mark.parametrize("case", [(32, 19, 19, 256, 512, 3, 2, 1), (32, 10, 10, 512, 128, 1, 1, 0), (32, 10, 10, 128, 256, 3, 2, 1),
                                  (32, 5, 5, 128, 256, 3, 1, 0), (32, 3, 3, 128, 256, 3, 1, 0), (32, 19, 19, 1024, 256, 1, 1, 0),
                                  (3, 3, 3, 256, 128, 1, 1, 0)])
def test_splitk_conv_extra_layers(case):
    """The SSD300 extra layers at batch 32 through the split-K form: within one bf16 rounding of the float32 reference, within two
    of the one-pass kernel (another float32 summation order), and bit-identical from launch to launch (the ranges are added in
    order, no atomics)."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    got = nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True, variant=8)
    one = nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True)
    want = torch.relu(F.conv2d(x.float(), wt.float(), bias.float(), stride, pad))
    assert got.shape == want.shape == one.shape
    rms = want.pow(2).mean().sqrt().item()
    assert int(((got.float() - want).abs() > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0
    assert int(((got.float() - one.float()).abs() > want.abs() * 2.0 ** -6 + 1e-2 * rms).sum().item()) == 0
    for _ in range(5):
        assert torch.equal(nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True, variant=8).view(torch.int16), got.view(torch.int16))


def test_general_conv_rejects_unsupported_arguments():
    import torch
    from ssd_keras_amd import _native as nat
    x = torch.zeros((1, 8, 8, 64), device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
    w = torch.zeros((64, 64, 3, 3), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(nat.SsdHipError):
        nat.conv2d(x, w, None, stride=1, padding=2)          # more padding than the kernel reaches
    with pytest.raises(nat.SsdHipError):
        nat.conv2d(x, w, None, stride=5, padding=0)
    with pytest.raises(nat.SsdHipError):
        nat.conv2d(x[:, :, :2, :2].contiguous(memory_format=torch.channels_last), w, None, stride=1, padding=0)   # empty output


def test_conv_rejects_unsupported_shapes():
    import torch
    from ssd_keras_amd import _native as nat
    x = torch.zeros((1, 8, 8, 3), device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
    w = torch.zeros((64, 3, 3, 3), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(nat.SsdHipError):
        nat.conv2d_same(x, w, None)


@pytest.mark.parametrize("shape", [(2, 300, 300), (3, 7, 5), (1, 1, 1), (2, 33, 64)])
def test_first_layer_conv_vs_float32_reference(shape):
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + H)
    x = (torch.randn((B, H, W, 3), generator=g, device="cuda") * 60).to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((64, 3, 3, 3), generator=g, device="cuda") / 27 ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((64,), generator=g, device="cuda").to(torch.bfloat16)
    for relu in (True, False):
        got = nat.conv3x3_cin3(x, wt, bias, relu=relu).float()
        want = F.conv2d(x.float(), wt.float(), bias.float(), 1, 1)
        if relu:
            want = torch.relu(want)
        rms = want.pow(2).mean().sqrt().item()
        err = (got - want).abs()
        bad = int((err > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item())
        assert bad == 0, "%d outputs off, max err %g (rms %g)" % (bad, err.max().item(), rms)


POOL_CASES = [  # B, H, W, Cin, Cout, k, dil
    (2, 300, 300, 64, 64, 3, 1),      # conv1_2 -> pool1
    (1, 150, 150, 128, 128, 3, 1),    # conv2_2 -> pool2
    (2, 75, 75, 256, 256, 3, 1),      # conv3_3 -> pool3: odd size, clipped last row / column
    (3, 5, 7, 64, 64, 3, 1),          # tiny odd map, one partial column tile
    (1, 1, 1, 64, 128, 3, 1),         # single pixel
    (2, 8, 130, 64, 64, 1, 1),        # 1x1 kernel, three column tiles (the last with 2 columns)
    (1, 19, 19, 128, 64, 3, 2),       # dilation 2
]


@pytest.mark.parametrize("case", POOL_CASES)
def test_conv_pool_fused(case):
    """Convolution with the 2x2 max-pool in its epilogue == the unfused libssdhip path bit for bit (same accumulation order;
    max commutes with bias / ReLU / rounding), and matches the float32 reference within the convolution tolerance."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, k, dil = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    for relu in (True, False):
        got = nat.conv2d_same_pool2(x, wt, bias, dilation=dil, relu=relu)
        assert got.shape == (B, Cout, (H + 1) // 2, (W + 1) // 2)
        unfused = nat.bias_act_maxpool(nat.conv2d_same(x, wt, bias, dilation=dil, relu=relu, variant=4), None, 2, 2, 0, True, relu=False)
        assert torch.equal(got, unfused), "%d of %d outputs differ from the unfused path" % (int((got != unfused).sum()), got.numel())
        ref = F.conv2d(x.float(), wt.float(), bias.float(), 1, dil * (k // 2), dil)
        if relu:
            ref = torch.relu(ref)
        want = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
        rms = ref.pow(2).mean().sqrt().item()
        err = (got.float() - want).abs()
        bad = int((err > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item())
        assert bad == 0, "%d outputs off, max err %g" % (bad, err.max().item())


def test_grouped_launch_equals_single_launches():
    """ssdhip_conv2d_same_group_nhwc_bf16: six head-shaped problems in one launch == the same problems launched one by one
    (identical K order per output element -> identical bytes)."""
    import torch
    from ssd_keras_amd import _native as nat
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(2, 38, 38, 512, 128), (2, 19, 19, 1024, 192), (2, 10, 10, 512, 192), (2, 5, 5, 256, 192), (2, 3, 3, 256, 128),
              (2, 1, 1, 256, 128)]
    xs, ws = [], []
    for B, H, W, Cin, Cout in shapes:
        xs.append(torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2))
        ws.append((torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2))
    got = nat.conv2d_same_group(xs, ws, None, relu=False)
    for x, w, y in zip(xs, ws, got):
        want = nat.conv2d_same(x, w, None, dilation=1, relu=False, variant=4)
        assert y.shape == want.shape and torch.equal(y, want)
    # with biases and ReLU, and a 1x1 member
    bs = [torch.randn((w.shape[0],), generator=g, device="cuda").to(torch.bfloat16) for w in ws[:2]]
    w11 = (torch.randn((64, 1, 1, 512), generator=g, device="cuda") / 512 ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    got = nat.conv2d_same_group([xs[0], xs[1], xs[2]], [ws[0], ws[1], w11], [bs[0], bs[1], None], relu=True)
    assert torch.equal(got[0], nat.conv2d_same(xs[0], ws[0], bs[0], relu=True, variant=4))
    assert torch.equal(got[1], nat.conv2d_same(xs[1], ws[1], bs[1], relu=True, variant=4))
    assert torch.equal(got[2], nat.conv2d_same(xs[2], w11, None, relu=True, variant=4))


C64_CASES = [  # B, H, W, Cout
    (2, 300, 300, 64),        # conv1_2
    (2, 150, 150, 128),       # conv2_1: two 64-channel output slices
    (1, 75, 75, 64),          # odd size
    (3, 5, 7, 64),            # tiny: a single partial tile per image
    (1, 1, 1, 128),           # single pixel
    (2, 9, 130, 64),          # wide and short
    (40, 16, 16, 64),         # more tiles than one round of persistent workgroups would need per image
]


@pytest.mark.parametrize("case", C64_CASES)
def test_resident_weight_conv_c64(case):
    """csrc/ssdhip_conv64.hip (persistent workgroups, filters resident in registers, activation halos from a loader wave) == the
    implicit-GEMM kernel bit for bit, with and without the fused pool, and within the convolution tolerance of the float32 reference."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cout = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, 64), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, 64), generator=g, device="cuda") / 24.0).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    for relu in (True, False):
        for b_ in (bias, None):
            got = nat.conv3x3_c64(x, wt, b_, relu=relu, pool=False)
            want = nat.conv2d_same(x, wt, b_, dilation=1, relu=relu, variant=4)
            assert got.shape == want.shape
            assert torch.equal(got, want), "%d of %d outputs differ from the implicit-GEMM kernel" % (int((got != want).sum()), got.numel())
            gotp = nat.conv3x3_c64(x, wt, b_, relu=relu, pool=True)
            wantp = nat.bias_act_maxpool(want, None, 2, 2, 0, True, relu=False)
            assert gotp.shape == wantp.shape
            assert torch.equal(gotp, wantp), "%d of %d pooled outputs differ" % (int((gotp != wantp).sum()), gotp.numel())
            # round 6, the training step's form: ONE launch writes the activation and the pooled map
            full, pooled = nat.conv3x3_c64_pool_keep(x, wt, b_, relu=relu)
            assert torch.equal(full, want) and torch.equal(pooled, wantp)
    ref = torch.relu(F.conv2d(x.float(), wt.float(), bias.float(), 1, 1))
    got = nat.conv3x3_c64(x, wt, bias, relu=True, pool=False).float()
    rms = ref.pow(2).mean().sqrt().item()
    bad = int(((got - ref).abs() > ref.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item())
    assert bad == 0


HALO_CASES = [  # B, H, W, Cin, Cout, bias, relu      (csrc/ssdhip_convh.hip: 3x3, dilation 1, Cin % 128 == 0, Cout % 128 == 0)
    (2, 38, 38, 512, 512, True, True),       # conv4_2: four weight stages, six slab pieces per wave
    (1, 75, 75, 256, 256, True, True),       # conv3_2: three weight stages, seven slab pieces (all 160 KB of LDS)
    (2, 19, 19, 512, 512, True, True),       # conv5_x: four stages, five pieces
    (3, 10, 10, 128, 256, True, False),      # two slices only (first + last), no ReLU
    (1, 64, 64, 256, 512, False, True),      # SSD512 conv4_1, no bias
    (2, 5, 94, 128, 128, True, True),        # the widest map the slab supports
    (1, 1, 1, 128, 128, True, True),         # one pixel: eight of the nine taps are padding
    (5, 7, 31, 256, 128, True, True),        # W + 1 = 32: a tile row boundary on every image row
    (1, 3, 63, 128, 128, True, True),        # W + 1 = 64
    (64, 38, 38, 128, 256, True, True),      # 3044 tiles: every persistent workgroup walks over ~12 of them
    (48, 19, 19, 256, 384, True, True),      # three channel tiles per position tile; tiles change channel tile between hops
    # fourth session: an un-pooled map up to 94 wide leaves the position grid when 2-D tiles take fewer rounds of 256 workgroups
    (16, 32, 32, 128, 512, True, True),      # SSD512 conv5_x at batch 16: 64 tiles of 16 x 16 x 4 channel tiles = one round (grid: 276 units = two)
    (16, 64, 64, 128, 512, False, True),     # SSD512 conv4_x: 1 024 units = four rounds (grid: 1 060 = five)
]


@pytest.fixture(params=["128", "1152"] + (os.environ.get("SSDHIP_TEST_CONVH_MODES", "").split(",") if os.environ.get("SSDHIP_TEST_CONVH_MODES") else []))
def slab_mode(request):
    """SSDHIP_CONVH_MODE: 128 = persistent workgroups prefetching across tiles, 1152 = the same with the tolerant waits after an epilogue
    (read at every launch).  The profiling build (tools/prof_build.sh) also has the round-4 experiments + 4096 (four waves per workgroup)
    and + 8192 (fragment reads in flight across the barrier): SSDHIP_TEST_CONVH_MODES=4224,5248,8320,9344,12416 with SSDHIP_LIB set."""
    import os
    old = os.environ.get("SSDHIP_CONVH_MODE")
    os.environ["SSDHIP_CONVH_MODE"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("SSDHIP_CONVH_MODE", None)
    else:
        os.environ["SSDHIP_CONVH_MODE"] = old


@pytest.mark.parametrize("case", HALO_CASES)
def test_slab_conv_equals_implicit_gemm(case, slab_mode):
    """ssdhip_conv3x3_halo_nhwc_bf16 promises the implicit-GEMM kernel's accumulation order: BIT-identical outputs; and the
    float32 reference within the usual bar.  Repeated launches must agree with each other (LDS-DMA / barrier races show up as rare
    differing tiles)."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, has_bias, relu = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if has_bias else None
    base = nat.conv2d_same(x, wt, bias, dilation=1, relu=relu, variant=4)
    got = nat.conv2d_same(x, wt, bias, dilation=1, relu=relu, variant=7)
    assert got.shape == base.shape and got.dtype == base.dtype
    diff = int((got.view(torch.int16) != base.view(torch.int16)).sum().item())
    assert diff == 0, "%d of %d outputs differ from the implicit-GEMM kernel" % (diff, got.numel())
    want = F.conv2d(x.float(), wt.float(), bias.float() if has_bias else None, 1, 1, 1)
    if relu:
        want = torch.relu(want)
    rms = want.pow(2).mean().sqrt().item()
    err = (got.float() - want).abs()
    assert int((err > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0
    for _ in range(10):
        again = nat.conv2d_same(x, wt, bias, dilation=1, relu=relu, variant=7)
        assert torch.equal(again.view(torch.int16), got.view(torch.int16))


def test_slab_conv_full_batch_race_screen(slab_mode):
    """BASELINE configs[1] sizes (batch 32): conv3_2, conv4_2, conv5_1 -- every CU busy, 20 launches each, all bit-identical to
    the implicit-GEMM kernel."""
    import torch
    from ssd_keras_amd import _native as nat
    for (B, H, W, Cin, Cout) in ((32, 75, 75, 256, 256), (32, 38, 38, 512, 512), (32, 19, 19, 512, 512), (32, 75, 75, 128, 256)):
        g = torch.Generator(device="cuda").manual_seed(B * H + Cin)
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
        base = nat.conv2d_same(x, wt, bias, dilation=1, relu=True, variant=4).view(torch.int16)
        for _ in range(20):
            got = nat.conv2d_same(x, wt, bias, dilation=1, relu=True, variant=7).view(torch.int16)
            assert torch.equal(got, base), (B, H, W, Cin, Cout)


HALO_2D_CASES = [  # B, H, W, Cin, Cout, bias, relu, pool      (2-D tiles: maps wider than 94 and every pooled call)
    (2, 150, 150, 128, 128, True, True, True),     # conv2_2 -> pool2: 8 x 32 tiles (150 = 4.7 x 32: one ragged tile column)
    (2, 150, 150, 128, 128, True, True, False),
    (1, 75, 75, 256, 256, True, True, True),       # conv3_3 -> pool3: odd map, 'same' pooling clips the last window
    (3, 37, 21, 128, 256, False, True, True),      # odd in both directions, no bias, two channel tiles
    (1, 128, 128, 128, 256, True, True, False),    # SSD512 conv3_1: 16 x 16 tiles, no padding waste
    (2, 9, 100, 128, 128, True, False, False),     # wide and flat, no ReLU
    (5, 2, 2, 128, 128, True, True, True),         # one pooling window per image
    (2, 16, 16, 256, 128, True, True, True),       # exactly one tile per image
    # round 6, fourth session: pooled calls tile the STACKED batch when that takes fewer tiles (ssdhip_conv3x3_halo_plan)
    (5, 75, 75, 128, 256, True, True, True),       # pitch 76: 24 x 5 tiles for 125; tiles straddle two images, the last row of tiles leaves the stack
    (4, 75, 75, 256, 128, True, True, True),       # pitch 76: the stack is a whole number of tile rows
    (6, 20, 20, 128, 128, True, True, True),       # even map: pitch 22 (two rows of zeros between images), 8 x 32 tiles
    (5, 21, 21, 128, 128, False, False, True),     # the float pooling path (no ReLU), no bias
    (9, 17, 40, 128, 128, True, True, True),       # pitch 18 = rows of a tile + 2: the smallest gap the stacked form takes
]


@pytest.mark.parametrize("case", HALO_2D_CASES)
def test_slab_conv_2d_tiles_and_fused_pool(case, slab_mode):
    """ssdhip_conv3x3_halo_nhwc_bf16 on 2-D tiles, plain and with MaxPooling2D(2, 2, 'same') fused: BIT-identical to the
    implicit-GEMM kernels (conv2d_same / conv2d_same_pool2), stable across repeated launches."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, has_bias, relu, pool = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if has_bias else None
    base = (nat.conv2d_same_pool2(x, wt, bias, relu=relu) if pool else nat.conv2d_same(x, wt, bias, relu=relu, variant=4))
    got = nat.conv3x3_halo(x, wt, bias, relu=relu, pool=pool)
    assert got.shape == base.shape
    diff = int((got.view(torch.int16) != base.view(torch.int16)).sum().item())
    assert diff == 0, "%d of %d outputs differ from the implicit-GEMM kernel" % (diff, got.numel())
    for _ in range(8):
        assert torch.equal(nat.conv3x3_halo(x, wt, bias, relu=relu, pool=pool).view(torch.int16), got.view(torch.int16))


@pytest.mark.parametrize("shape", [(32, 75, 75, 256, 256), (5, 75, 75, 128, 128), (6, 20, 20, 128, 128), (7, 33, 47, 128, 256)])
def test_slab_conv_pool_on_the_stacked_batch_equals_tiles_per_image(shape, monkeypatch):
    """conv3_3 -> pool3 of BASELINE configs[1] (and smaller stacks): the pooled slab kernel on tiles of the stacked batch (760 position
    tiles at batch 32 where tiling every image takes 800) == the same kernel on tiles per image (SSDHIP_CONVH_STACK=0) == the
    implicit-GEMM kernel, bit for bit, over repeated launches."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout = shape
    monkeypatch.delenv("SSDHIP_CONVH_STACK", raising=False)
    monkeypatch.delenv("SSDHIP_CONVH_MODE", raising=False)
    geom, tiles, pitch, rows = nat.conv3x3_halo_plan(B, H, W, True)
    monkeypatch.setenv("SSDHIP_CONVH_STACK", "0")
    geom0, tiles0, pitch0, _ = nat.conv3x3_halo_plan(B, H, W, True)
    assert pitch > H and pitch % 2 == 0 and pitch0 == 0 and tiles < tiles0, "the case is meant to run on the stacked batch"
    g = torch.Generator(device="cuda").manual_seed(B * H + W)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    per_image = nat.conv3x3_halo(x, wt, bias, relu=True, pool=True).view(torch.int16)
    assert torch.equal(per_image, nat.conv2d_same_pool2(x, wt, bias, relu=True).view(torch.int16))
    monkeypatch.delenv("SSDHIP_CONVH_STACK")
    for _ in range(10):
        got = nat.conv3x3_halo(x, wt, bias, relu=True, pool=True).view(torch.int16)
        assert torch.equal(got, per_image), "%d of %d outputs differ" % (int((got != per_image).sum()), got.numel())


@pytest.mark.parametrize("shape", [(2, 150, 150, 128, 128), (5, 75, 75, 256, 256), (3, 75, 75, 128, 128), (6, 20, 20, 128, 256),
                                   (2, 16, 16, 256, 128), (5, 2, 2, 128, 128), (3, 37, 21, 128, 128)])
@pytest.mark.parametrize("has_bias", [True, False])
def test_slab_conv_pool_keep_writes_both_maps(shape, has_bias, monkeypatch):
    """ssdhip_conv3x3_halo_pool_keep_nhwc_bf16 (the training step's conv2_2 -> pool2, conv3_3 -> pool3): ONE launch, the full-resolution
    activation == the un-pooled slab kernel's and the pooled map == the pooled slab kernel's, bit for bit, on tiles per image and on
    tiles of the stacked batch, over repeated launches."""
    import torch
    from ssd_keras_amd import _native as nat
    monkeypatch.delenv("SSDHIP_CONVH_MODE", raising=False)
    B, H, W, Cin, Cout = shape
    g = torch.Generator(device="cuda").manual_seed(B * H + Cin)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if has_bias else None
    want_full = nat.conv3x3_halo(x, wt, bias, relu=True, pool=False).view(torch.int16)
    want_pool = nat.conv3x3_halo(x, wt, bias, relu=True, pool=True).view(torch.int16)
    for _ in range(6):
        full, pooled = nat.conv3x3_halo_pool_keep(x, wt, bias, relu=True)
        assert full.shape == (B, Cout, H, W) and pooled.shape == (B, Cout, (H + 1) // 2, (W + 1) // 2)
        assert torch.equal(full.view(torch.int16), want_full), "%d full-resolution outputs differ" % int((full.view(torch.int16) != want_full).sum())
        assert torch.equal(pooled.view(torch.int16), want_pool), "%d pooled outputs differ" % int((pooled.view(torch.int16) != want_pool).sum())
    assert nat.conv3x3_halo_pool_keep(x[:, :64].contiguous(memory_format=torch.channels_last),
                                      wt[:, :64].contiguous(memory_format=torch.channels_last), bias) is None


def test_slab_conv_pool_full_batch_race_screen(slab_mode):
    """conv2_2 -> pool2 at batch 32 (BASELINE configs[1]): 15 launches, all bit-identical to conv_igemm4_pool_kernel."""
    import torch
    from ssd_keras_amd import _native as nat
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((32, 150, 150, 128), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((128, 3, 3, 128), generator=g, device="cuda") / 34.0).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((128,), generator=g, device="cuda").to(torch.bfloat16)
    base = nat.conv2d_same_pool2(x, wt, bias, relu=True).view(torch.int16)
    for _ in range(15):
        assert torch.equal(nat.conv3x3_halo(x, wt, bias, relu=True, pool=True).view(torch.int16), base)


def test_slab_conv_rejects_unsupported_shapes():
    import torch
    from ssd_keras_amd import _native as nat
    for (H, W, Cin, Cout, dil) in ((8, 8, 64, 128, 1), (8, 8, 128, 64, 1), (8, 8, 128, 128, 2)):
        x = torch.zeros((1, H, W, Cin), device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
        wt = torch.zeros((Cout, 3, 3, Cin), device="cuda", dtype=torch.bfloat16).permute(0, 3, 1, 2)
        with pytest.raises(nat.SsdHipError):
            nat.conv2d_same(x, wt, None, dilation=dil, relu=True, variant=7)


@pytest.mark.parametrize("shape", [(2, 300, 300), (1, 37, 53), (3, 16, 16), (2, 7, 150), (1, 1, 1), (4, 150, 2)])
@pytest.mark.parametrize("pool", [True, False])
def test_conv1_block_equals_the_two_kernels(shape, pool):
    """conv1_1 -> conv1_2 [-> pool1] in one kernel (the 64-channel map between them recomputed per tile from the image): BIT-identical
    to conv3x3_cin3 followed by conv3x3_c64, repeatable."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H * 1000 + W)
    x = (torch.randn((B, H, W, 3), generator=g, device="cuda") * 60).to(torch.bfloat16).permute(0, 3, 1, 2)
    w1 = (torch.randn((64, 3, 3, 3), generator=g, device="cuda") / 5).to(torch.bfloat16).permute(0, 3, 1, 2)
    b1 = torch.randn((64,), generator=g, device="cuda").to(torch.bfloat16)
    w2 = (torch.randn((64, 3, 3, 64), generator=g, device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
    b2 = torch.randn((64,), generator=g, device="cuda").to(torch.bfloat16)
    mid = nat.conv3x3_cin3(x, w1, b1, relu=True)
    base = nat.conv3x3_c64(mid, w2, b2, relu=True, pool=pool)
    got = nat.conv1_block(x, w1, b1, w2, b2, relu=True, pool=pool)
    assert got.shape == base.shape
    diff = int((got.view(torch.int16) != base.view(torch.int16)).sum().item())
    assert diff == 0, "%d of %d outputs differ from the two-kernel form" % (diff, got.numel())
    for _ in range(5):
        assert torch.equal(nat.conv1_block(x, w1, b1, w2, b2, relu=True, pool=pool).view(torch.int16), got.view(torch.int16))


def test_conv1_block_full_batch():
    """Batch 32, 300 x 300 (BASELINE configs[1]) and a 128-channel second layer: 10 launches, all bit-identical to the two kernels."""
    import torch
    from ssd_keras_amd import _native as nat
    for cout in (64, 128):
        g = torch.Generator(device="cuda").manual_seed(cout)
        x = (torch.randn((32, 300, 300, 3), generator=g, device="cuda") * 60).to(torch.bfloat16).permute(0, 3, 1, 2)
        w1 = (torch.randn((64, 3, 3, 3), generator=g, device="cuda") / 5).to(torch.bfloat16).permute(0, 3, 1, 2)
        b1 = torch.randn((64,), generator=g, device="cuda").to(torch.bfloat16)
        w2 = (torch.randn((cout, 3, 3, 64), generator=g, device="cuda") / 24).to(torch.bfloat16).permute(0, 3, 1, 2)
        b2 = torch.randn((cout,), generator=g, device="cuda").to(torch.bfloat16)
        base = nat.conv3x3_c64(nat.conv3x3_cin3(x, w1, b1, relu=True), w2, b2, relu=True, pool=True).view(torch.int16)
        for _ in range(10):
            assert torch.equal(nat.conv1_block(x, w1, b1, w2, b2, relu=True, pool=True).view(torch.int16), base)


def test_slab_group_equals_single_launches():
    """The packed predictor heads of SSD300 (batch 32, filters padded to 128 channels) in ONE slab-kernel launch: every output
    BIT-identical to the implicit-GEMM kernel on the same problem; 5 launches agree."""
    import torch
    from ssd_keras_amd import _native as nat
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(32, 38, 38, 512, 128), (32, 19, 19, 1024, 256), (32, 10, 10, 512, 256), (32, 5, 5, 256, 256), (32, 3, 3, 256, 128),
              (32, 1, 1, 256, 128)]
    xs, ws = [], []
    for (B, H, W, Cin, Cout) in shapes:
        xs.append(torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2))
        ws.append((torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2))
    base = [nat.conv2d_same(x, w, None, relu=False, variant=4).view(torch.int16) for x, w in zip(xs, ws)]
    for _ in range(5):
        got = nat.conv3x3_halo_group(xs, ws, None, relu=False)
        for k, (a, b) in enumerate(zip(got, base)):
            assert torch.equal(a.view(torch.int16), b), "problem %d differs" % k
    with pytest.raises(nat.SsdHipError):                   # a 64-channel-multiple filter bank is not the slab kernel's
        nat.conv3x3_halo_group(xs[:1], [ws[0][:64]], None)


@pytest.mark.parametrize("case", [(32, 19, 19, 256, 512, 2, 1), (32, 10, 10, 128, 256, 2, 1), (32, 5, 5, 128, 256, 1, 0),
                                  (32, 3, 3, 128, 256, 1, 0), (2, 32, 32, 128, 256, 2, 1), (3, 11, 7, 128, 128, 1, 1),
                                  (1, 20, 20, 256, 128, 2, 0), (2, 9, 94, 128, 128, 1, 0)])
def test_slab_conv_strided_and_valid_forms(case):
    """conv6_2 ... conv9_2 through the slab kernel (stride-1 'same' result, strided / cropped positions kept): BIT-identical to the
    general implicit-GEMM kernel."""
    import torch
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(hash(case) & 0xffff)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
    base = nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True)
    for _ in range(4):
        got = nat.conv2d(x, wt, bias, stride=stride, padding=pad, relu=True, variant=7)
        assert got.shape == base.shape
        assert torch.equal(got.view(torch.int16), base.view(torch.int16))


CHAIN_CASES = [  # B, H, W, C0, [(k, stride, pad, cout, relu, keep), ...]
    (3, 10, 10, 512, [(1, 1, 0, 128, 1, 0), (3, 2, 1, 256, 1, 1), (1, 1, 0, 128, 1, 0), (3, 1, 0, 256, 1, 1), (1, 1, 0, 128, 1, 0),
                      (3, 1, 0, 256, 1, 1)]),                               # the SSD300 tail: conv7_1 ... conv9_2
    (2, 7, 5, 128, [(3, 1, 1, 128, 0, 1), (1, 1, 0, 96, 1, 1)]),             # 'same' padding, no ReLU, 96 output channels, every map kept
    (1, 4, 4, 256, [(3, 2, 1, 32, 1, 1)]),                                   # one layer
]


@pytest.mark.parametrize("case", CHAIN_CASES)
def test_conv_chain_matches_layer_by_layer_reference(case):
    """ssdhip_conv_chain_nhwc_bf16 (csrc/ssdhip_chain.hip: the extra-layer tail of models/keras_ssd300.py:304-313 in one launch, the
    intermediate maps in LDS) against a float32 PyTorch chain that rounds every layer's output to bf16 like the kernel does: within one
    bf16 rounding of the float32-accumulated sums per layer (2^-7 |want| + 1e-2 rms on every kept map), and bit-identical run to run."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, C0, spec = case
    g = torch.Generator(device="cuda").manual_seed(hash(str(case)) & 0xffff)
    x = torch.randn((B, H, W, C0), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    layers, ref, cin, want = [], x.float(), C0, []
    for (k, s, pd, cout, relu, keep) in spec:
        w = (torch.randn((cout, k, k, cin), generator=g, device="cuda") / (k * k * cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        bias = torch.randn((cout,), generator=g, device="cuda").to(torch.bfloat16)
        packed = nat.conv_chain_pack(w)
        assert packed is not None
        layers.append(dict(packed=packed, bias=bias, k=k, stride=s, pad=pd, cout=cout, relu=relu, keep=bool(keep)))
        ref = F.conv2d(ref, w.float(), bias.float(), s, pd)
        if relu:
            ref = torch.relu(ref)
        if keep:
            want.append(ref)
        ref = ref.to(torch.bfloat16).float()              # the next layer reads the rounded map
        cin = cout
    got = nat.conv_chain(x, layers)
    assert got is not None and len(got) == len(want)
    for y, wnt in zip(got, want):
        assert y.shape == wnt.shape and y.dtype == torch.bfloat16
        rms = wnt.pow(2).mean().sqrt().item()
        err = (y.float() - wnt).abs()
        assert int((err > wnt.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0
    again = nat.conv_chain(x, layers)
    for a, y in zip(again, got):
        assert torch.equal(a.view(torch.int16), y.view(torch.int16))
    # the depth of the filter ring (fragments in flight per wave: 8 by default; 16 / 32 measured slower) changes when a fragment is requested, not what is
    # multiplied in which order: every depth returns the same bits
    import os
    old = os.environ.get("SSDHIP_CHAIN_RING")
    try:
        for ring in ("8", "16", "32"):
            os.environ["SSDHIP_CHAIN_RING"] = ring
            for a, y in zip(nat.conv_chain(x, layers), got):
                assert torch.equal(a.view(torch.int16), y.view(torch.int16)), ring
    finally:
        if old is None:
            os.environ.pop("SSDHIP_CHAIN_RING", None)
        else:
            os.environ["SSDHIP_CHAIN_RING"] = old


def test_conv_chain_in_the_ssd300_model_equals_the_layer_by_layer_path():
    """the graphed inference model with and without the chain kernel (SSDHIP_NO_CHAIN=1): same detections up to bf16 rounding of the
    extra-layer maps (the chain sums a tile's K loop in two interleaved accumulators, the split-K kernels in ranges)."""
    import os
    import numpy as np
    import torch
    from ssd_keras_amd import synthetic as syn
    from ssd_keras_amd.models.keras_ssd300 import ssd_300
    cfg = syn.SSD300_VOC
    torch.manual_seed(5)
    model = ssd_300((300, 300, 3), cfg["n_classes"], mode="training", scales=cfg["scales"], aspect_ratios_per_layer=cfg["aspect_ratios_per_layer"],
                    steps=cfg["steps"], offsets=cfg["offsets"]).cuda().to(torch.bfloat16).to(memory_format=torch.channels_last).eval()
    images = torch.from_numpy(np.random.RandomState(1).randint(0, 256, size=(4, 300, 300, 3)).astype(np.float32)).cuda()
    with torch.no_grad():
        x = model.preprocess(images)
        feats = model.features(x)
        os.environ["SSDHIP_NO_CHAIN"] = "1"
        try:
            base = model.features(x)
        finally:
            os.environ.pop("SSDHIP_NO_CHAIN", None)
    assert len(feats) == len(base) == 6
    for a, b in zip(feats[:3], base[:3]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for a, b in zip(feats[3:], base[3:]):
        rms = b.float().pow(2).mean().sqrt().item()
        assert float((a.float() - b.float()).abs().max()) <= 3e-2 * rms + 2.0 ** -6 * float(b.float().abs().max())


IMAGE_CASES = [  # B, H, W, Cin, Cout, dilation, bias, relu   (csrc/ssdhip_convimg.hip: one image per tile, H W <= 384, any dilation)
    (32, 19, 19, 512, 1024, 6, True, True),      # fc6 (models/keras_ssd300.py:298): 256 tiles, the XCD-sharing tile map
    (3, 19, 19, 128, 256, 6, True, True),        # the same geometry, a batch that is no multiple of 8 (plain tile map)
    (2, 19, 19, 64, 128, 1, True, False),        # one slice, dilation 1, no ReLU
    (8, 10, 10, 256, 256, 3, True, True),        # conv6-sized map
    (2, 5, 7, 64, 128, 2, False, True),          # a non-square map narrower than the dilated taps reach, no bias
    (1, 1, 1, 128, 128, 6, True, True),          # one pixel: eight of the nine taps are padding
    (2, 16, 24, 192, 384, 4, True, True),        # exactly 384 pixels (54 slab pieces), three slices, three channel tiles
    (16, 19, 19, 512, 512, 1, True, True),       # conv5_x shape, batch a multiple of 8: the 64-channel tile form (8 tiles per image)
    (3, 7, 9, 128, 192, 2, True, True),          # Cout = 192: only the 64-channel form applies
]


@pytest.mark.parametrize("case", IMAGE_CASES)
def test_image_conv_equals_implicit_gemm(case):
    """ssdhip_conv3x3_image_nhwc_bf16 against the implicit-GEMM kernel (variant 4) on the same operands: bit-identical (same K order:
    slices, taps, 16-channel blocks), five launches each (a race between the LDS-DMA ring and the fragment reads would show up as a
    run that differs), and within the convolution bar of the float32 reference."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, d, use_bias, relu = case
    g = torch.Generator(device="cuda").manual_seed(B * 131 + H * 7 + d)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if use_bias else None
    assert nat.conv3x3_image_supported(x, wt, d)
    base = nat.conv2d_same(x, wt, bias, dilation=d, relu=relu, variant=4)
    for _ in range(5):
        got = nat.conv3x3_image(x, wt, bias, dilation=d, relu=relu)
        assert got.shape == base.shape
        assert torch.equal(got.view(torch.int16), base.view(torch.int16))
    want = F.conv2d(x.float(), wt.float(), bias.float() if use_bias else None, 1, d, d)
    if relu:
        want = want.clamp_min(0)
    rms = want.pow(2).mean().sqrt().item()
    assert int(((got.float() - want).abs() > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0


IMAGE2_CASES = [  # B, H, W, Cin, Cout, k, stride, padding, dilation, bias, relu   (round 6: ssdhip_conv2d_image_nhwc_bf16)
    (32, 19, 19, 1024, 1024, 1, 1, 0, 1, True, True),    # fc7 (models/keras_ssd300.py:299): 256 tiles of 128 channels, 16 one-tap slices
    (32, 19, 19, 1024, 256, 1, 1, 0, 1, True, True),     # conv6_1 (:301): 128 tiles of 64 channels
    (32, 19, 19, 256, 512, 3, 2, 1, 1, True, True),      # conv6_2 (:302-303): 19 x 19 -> 10 x 10, the 128-pixel tile form
    (3, 19, 19, 128, 192, 1, 1, 0, 1, True, False),      # 1x1, two slices (fewer than the ring has stages), Cout = 192, no ReLU, plain tile map
    (2, 7, 9, 64, 64, 1, 1, 0, 1, False, True),          # 1x1, ONE slice, no bias, a 63-pixel map on the one-block tile
    (8, 16, 24, 192, 128, 1, 1, 0, 1, True, True),       # 1x1 on exactly 384 pixels (54 slab pieces per slice: all seven rounds)
    (5, 10, 10, 512, 128, 1, 1, 0, 1, True, True),       # conv7_1 geometry (1x1 on 10 x 10), eight slices on the one-block tile
    (4, 10, 10, 128, 256, 3, 2, 1, 1, True, True),       # conv7_2: 10 x 10 -> 5 x 5
    (4, 5, 5, 128, 256, 3, 1, 0, 1, True, True),         # conv8_2: 'valid', 5 x 5 -> 3 x 3
    (2, 3, 3, 128, 256, 3, 1, 0, 1, True, True),         # conv9_2: 'valid', 3 x 3 -> 1 x 1
    (2, 19, 19, 64, 128, 3, 1, 2, 2, True, True),        # 'same' through the general entry (dilation 2, padding 2)
    (2, 18, 20, 128, 64, 3, 3, 1, 1, True, True),        # stride 3 on a non-square map
    (3, 19, 19, 64, 128, 1, 2, 0, 1, True, True),        # a strided 1x1
    (2, 17, 17, 128, 128, 3, 1, 1, 6, True, True),       # padding smaller than the dilated reach: 17 x 17 -> 7 x 7
    # round 6, fourth session: 1x1 layers on 128-pixel tiles where that moves fewer bytes per CU (ConvImgParams::n_pxt)
    (5, 13, 10, 128, 128, 1, 1, 0, 1, True, True),       # 130 pixels: the second pixel tile holds two of them
    (16, 19, 19, 256, 1024, 1, 1, 0, 1, False, True),    # conv6_1's data gradient geometry: 8 x 3 tiles per image
    (7, 15, 20, 192, 64, 1, 1, 0, 1, True, False),       # 300 pixels, a 64-channel layer, the plain tile map (batch not a multiple of 8)
]


@pytest.mark.parametrize("case", IMAGE2_CASES)
def test_general_image_conv_equals_implicit_gemm(case):
    """ssdhip_conv2d_image_nhwc_bf16 (1x1 layers: one step per slice with the next slice's seven slab rounds inside it; strided /
    partially padded 3x3 layers: taps as addresses, 128- and 384-pixel tiles) against the implicit-GEMM kernel on the same operands:
    bit-identical over five launches, and within the convolution bar of the float32 reference."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd import _native as nat
    B, H, W, Cin, Cout, k, stride, pad, d, use_bias, relu = case
    g = torch.Generator(device="cuda").manual_seed(B * 131 + H * 7 + d + 17 * k + stride)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
    bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16) if use_bias else None
    assert nat.conv2d_image_supported(x, wt, stride, pad, d)
    base = nat.conv2d(x, wt, bias, stride=stride, padding=pad, dilation=d, relu=relu)
    for _ in range(5):
        got = nat.conv2d_image(x, wt, bias, stride=stride, padding=pad, dilation=d, relu=relu)
        assert got.shape == base.shape
        assert torch.equal(got.view(torch.int16), base.view(torch.int16))
    want = F.conv2d(x.float(), wt.float(), bias.float() if use_bias else None, stride, pad, d)
    if relu:
        want = want.clamp_min(0)
    rms = want.pow(2).mean().sqrt().item()
    assert int(((got.float() - want).abs() > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item()) == 0


def test_general_image_conv_race_screen():
    """fc7, conv6_1 and conv6_2 at batch 32: thirty launches each, every one bit-identical to the implicit-GEMM result (the 1x1 form
    waits for a slab requested ONE step earlier: a wrong count shows up as stale rows)."""
    import torch
    from ssd_keras_amd import _native as nat
    for B, H, W, Cin, Cout, k, stride, pad in ((32, 19, 19, 1024, 1024, 1, 1, 0), (32, 19, 19, 1024, 256, 1, 1, 0), (32, 19, 19, 256, 512, 3, 2, 1)):
        g = torch.Generator(device="cuda").manual_seed(13)
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        wt = (torch.randn((Cout, k, k, Cin), generator=g, device="cuda") / (k * k * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
        base = nat.conv2d(x, wt, bias, stride=stride, padding=pad, dilation=1, relu=True).view(torch.int16)
        bad = 0
        for _ in range(30):
            bad += int((nat.conv2d_image(x, wt, bias, stride=stride, padding=pad, dilation=1, relu=True).view(torch.int16) != base).sum().item())
        assert bad == 0, (B, H, W, Cin, Cout, k, bad)


def test_pixel_tiles_of_the_1x1_image_kernel_equal_whole_image_tiles(monkeypatch):
    """conv6_1 at batch 32 (BASELINE configs[1]) and smaller batches of it: 128-channel x 128-pixel tiles (the default where they
    move fewer bytes per CU) == whole-image tiles (SSDHIP_CONVIMG_PXT=0) == the implicit-GEMM kernel, bit for bit, twenty launches."""
    import torch
    from ssd_keras_amd import _native as nat
    for B in (32, 3):
        g = torch.Generator(device="cuda").manual_seed(B)
        x = torch.randn((B, 19, 19, 1024), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        wt = (torch.randn((256, 1, 1, 1024), generator=g, device="cuda") / 32.0).to(torch.bfloat16).permute(0, 3, 1, 2)
        bias = torch.randn((256,), generator=g, device="cuda").to(torch.bfloat16)
        base = nat.conv2d(x, wt, bias, stride=1, padding=0, dilation=1, relu=True).view(torch.int16)
        monkeypatch.setenv("SSDHIP_CONVIMG_PXT", "0")
        assert torch.equal(nat.conv2d_image(x, wt, bias, relu=True).view(torch.int16), base)
        monkeypatch.delenv("SSDHIP_CONVIMG_PXT")
        for _ in range(20):
            assert torch.equal(nat.conv2d_image(x, wt, bias, relu=True).view(torch.int16), base)


def test_general_image_conv_rejects_other_geometries():
    import torch
    from ssd_keras_amd import _native as nat
    mk = lambda *shape: torch.zeros(shape, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not nat.conv2d_image_supported(mk(1, 64, 20, 20), mk(64, 64, 1, 1))                # 400 pixels
    assert not nat.conv2d_image_supported(mk(1, 64, 8, 8), mk(64, 64, 5, 5), 1, 2, 1)         # 5 x 5 filters
    assert not nat.conv2d_image_supported(mk(1, 64, 8, 8), mk(64, 64, 1, 1), 1, 1, 1)         # padding on a 1 x 1 layer
    assert not nat.conv2d_image_supported(mk(1, 64, 8, 8), mk(64, 64, 3, 3), 5, 1, 1)         # stride 5
    assert not nat.conv2d_image_supported(mk(1, 64, 2, 2), mk(64, 64, 3, 3), 1, 0, 1)         # the filter does not fit
    with pytest.raises(nat.SsdHipError):
        nat.conv2d_image(mk(1, 64, 20, 20), mk(64, 64, 1, 1), None)
    with pytest.raises(nat.SsdHipError):
        nat.conv2d_image(mk(1, 64, 8, 8), mk(64, 64, 1, 1), None, padding=1)
    with pytest.raises(nat.SsdHipError):
        nat.conv2d_image(mk(1, 64, 8, 8), mk(32, 64, 1, 1), None)                            # Cout % 64 != 0


def test_image_conv_race_screen():
    """fc6 at batch 32 and conv5-sized tiles at batch 16: thirty launches each, every one bit-identical to the implicit-GEMM result (the
    kernel's LDS-DMA rings are guarded by counted waits and one barrier per step; a missing guard shows up as an occasional stale
    fragment -- one did during development, on exactly these shapes)."""
    import torch
    from ssd_keras_amd import _native as nat
    for B, H, W, Cin, Cout, d in ((32, 19, 19, 512, 1024, 6), (16, 19, 19, 512, 512, 1), (8, 10, 10, 256, 256, 3)):
        g = torch.Generator(device="cuda").manual_seed(11)
        x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
        wt = (torch.randn((Cout, 3, 3, Cin), generator=g, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16).permute(0, 3, 1, 2)
        bias = torch.randn((Cout,), generator=g, device="cuda").to(torch.bfloat16)
        base = nat.conv2d_same(x, wt, bias, dilation=d, relu=True, variant=4).view(torch.int16)
        bad = 0
        for _ in range(30):
            bad += int((nat.conv3x3_image(x, wt, bias, dilation=d, relu=True).view(torch.int16) != base).sum().item())
        assert bad == 0, (B, H, W, Cin, Cout, d, bad)


def test_image_conv_rejects_other_geometries():
    import torch
    from ssd_keras_amd import _native as nat
    x = torch.zeros((1, 64, 20, 20), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros((128, 64, 3, 3), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not nat.conv3x3_image_supported(x, w, 1)                       # 400 pixels
    with pytest.raises(nat.SsdHipError):
        nat.conv3x3_image(x, w, None, dilation=1)
    x = torch.zeros((1, 64, 8, 8), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w32 = torch.zeros((32, 64, 3, 3), device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with pytest.raises(nat.SsdHipError):
        nat.conv3x3_image(x, w32, None, dilation=1)                      # Cout % 64 != 0
    with pytest.raises(nat.SsdHipError):
        nat.conv3x3_image(x, w, None, dilation=17)


@pytest.mark.parametrize("case", [(32, 19, 19, 1024, 1024, True), (32, 19, 19, 1024, 256, True), (2, 5, 7, 64, 128, False)])
def test_library_gemm_form_of_1x1_layers(case):
    """A 1 x 1 layer on NHWC memory as a plain GEMM through the library (SSDModel._gemm_1x1: fc7, conv6_1 where the autotune picks it)
    vs the float32 reference, the same bar as the implicit-GEMM kernels -- and the result is a channels_last view like theirs."""
    import torch
    import torch.nn.functional as F
    from ssd_keras_amd.models._common import SSDModel
    B, H, W, Cin, Cout, relu = case
    g = torch.Generator(device="cuda").manual_seed(91)
    x = torch.randn((B, H, W, Cin), generator=g, device="cuda").to(torch.bfloat16).permute(0, 3, 1, 2)
    conv = torch.nn.Conv2d(Cin, Cout, 1).cuda().to(torch.bfloat16)
    with torch.no_grad():
        got = SSDModel._gemm_1x1(conv, x, relu)
        want = F.conv2d(x.float(), conv.weight.float(), conv.bias.float())
        if relu:
            want = torch.relu(want)
    assert got.shape == want.shape and got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    rms = want.pow(2).mean().sqrt().item()
    err = (got.float() - want).abs()
    bad = int((err > want.abs() * 2.0 ** -7 + 1e-2 * rms).sum().item())
    assert bad == 0, "%d of %d outputs off; max err %g (rms %g)" % (bad, err.numel(), err.max().item(), rms)
