"""Image half of the augmentation (drop-in for data_generator/object_detection_2d_photometric_ops.py, the resize / flip ops of
object_detection_2d_geometric_ops.py, SSDPhotometricDistortions and SSDDataAugmentation) against vectors generated from the REAL
reference (tests/golden/make_golden.py gen_image_ops; its cv2 is built on oracle/np_image.py -- OpenCV is not installed, see that
module's header: the four OpenCV primitives are restated, everything the reference does around them is pinned).

CPU: the fixture is current; the host logic (program construction, random draws, label arithmetic, tap tables, equalisation table)
reproduces the reference's outputs with the kernels replaced by the NumPy restatement of their specification.
GPU: the same cases through the HIP kernels, bit for bit; a batch through one launch equals the per-image results."""
import ast
import types

import numpy as np
import pytest

from tests import image_cases as ic
from tests import util


def _ns():
    import ssd_keras_amd.data_generator.object_detection_2d_photometric_ops as pho
    import ssd_keras_amd.data_generator.object_detection_2d_geometric_ops as geo
    import ssd_keras_amd.data_generator.object_detection_2d_image_boxes_validation_utils as val
    import ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd as chain
    ns = types.SimpleNamespace(BoxFilter=val.BoxFilter, SSDPhotometricDistortions=chain.SSDPhotometricDistortions,
                               SSDDataAugmentation=chain.SSDDataAugmentation, Resize=geo.Resize, ResizeRandomInterp=geo.ResizeRandomInterp,
                               Flip=geo.Flip, RandomFlip=geo.RandomFlip)
    for name in ("ConvertColor", "ConvertDataType", "ConvertTo3Channels", "Hue", "RandomHue", "Saturation", "RandomSaturation", "Brightness",
                 "RandomBrightness", "Contrast", "RandomContrast", "HistogramEqualization", "RandomHistogramEqualization", "ChannelSwap",
                 "RandomChannelSwap"):
        setattr(ns, name, getattr(pho, name))
    return ns


def _check_all(needs_gpu_boxes):
    z = util.load("image_ops")
    assert int(z["n_cases"]) == len(ic.CASES), "fixture is stale: rerun tests/golden/make_golden.py image_ops"
    ns = _ns()
    for i, case in enumerate(ic.CASES):
        assert ast.literal_eval(str(z["i%03d_case" % i])) == case, "fixture is stale: rerun tests/golden/make_golden.py image_ops"
        if (case["op"] == "ssd_augmentation" or case.get("box_filter")) and not needs_gpu_boxes:
            continue                                           # (crop validation / BoxFilter are GPU kernels: GPU test only)
        res = ic.run(ns, case)
        for k, v in res.items():
            want = z["i%03d_%s" % (i, k)]
            assert v.dtype == want.dtype and v.shape == want.shape, (case, k, v.dtype, want.dtype, v.shape, want.shape)
            assert np.array_equal(v, want, equal_nan=True), (case, k, float(np.abs(v.astype(np.float64) - want.astype(np.float64)).max()))


def test_fixture_matches_the_case_list():
    z = util.load("image_ops")
    assert int(z["n_cases"]) == len(ic.CASES)
    for i, case in enumerate(ic.CASES):
        assert ast.literal_eval(str(z["i%03d_case" % i])) == case


def test_call_surface_equals_the_reference():
    """Constructor and __call__ signatures (names, order, defaults) of every mirrored class, as recorded from the reference."""
    import inspect
    want = ast.literal_eval(str(util.load("image_ops")["signatures"]))
    ns = _ns()
    for name, (init_sig, call_sig) in want.items():
        cls = getattr(ns, name)
        assert str(inspect.signature(cls.__init__)) == init_sig, (name, str(inspect.signature(cls.__init__)), init_sig)
        assert str(inspect.signature(cls.__call__)) == call_sig, (name, str(inspect.signature(cls.__call__)), call_sig)


def test_host_tables_against_hand_computed_cases():
    """cv2.resize's 8-bit arithmetic (round 6) against cases worked out BY HAND from imgproc/resize.cpp
    (tests/resize_hand_cases.py: every expected pixel and coefficient derived in a comment there) -- the oracle's `resize`, and the
    PRODUCT's plans (`_image_ops.resize_plan`: dispatch, tap indices, fixed-point shorts, area tables) evaluated by the pixel rule.
    Then the product's equalizeHist table against a hand-computed histogram."""
    from oracle import np_image as npi
    from ssd_keras_amd.data_generator import _image_ops as iop
    from tests import resize_hand_cases as hc
    for name, src, dsize, interp, want in hc.CASES:
        assert np.array_equal(npi.resize(src, dsize, interp), want), name
        plan = iop.resize_plan(src.shape[0], src.shape[1], dsize[1], dsize[0], interp)
        assert plan[1].dtype == np.int32 and plan[3].dtype == np.int32
        assert np.array_equal(npi.cv_apply_plan(src[:, :, None], plan)[:, :, 0], want), name
    for n_src, n_dst, interp, horizontal, idx, coef in hc.TABLES:
        i, c = iop.fixed_axis(n_src, n_dst, interp, False, horizontal)
        if idx is not None:
            assert i.tolist() == idx, (n_src, n_dst, interp, horizontal)
        if isinstance(coef, dict):
            for row, values in coef.items():
                assert c[row].tolist() == values
        else:
            assert c.tolist() == coef, (n_src, n_dst, interp, horizontal)
    # the dispatch: which of cv::resize's paths a geometry takes
    kinds = lambda sh, sw, dh, dw, m: iop.resize_plan(sh, sw, dh, dw, m)[0]
    assert kinds(10, 10, 10, 10, 2) == iop.KIND_COPY and kinds(10, 10, 5, 5, 1) == iop.KIND_AREA_FAST2
    assert kinds(12, 12, 4, 4, 3) == iop.KIND_AREA_FAST and iop.resize_plan(12, 12, 4, 4, 3)[5] == 9
    assert kinds(12, 12, 4, 4, 1) == iop.KIND_LINEAR                      # INTER_LINEAR is the area path only at exactly 2 x 2
    assert kinds(375, 500, 300, 300, 3) == iop.KIND_AREA and kinds(375, 500, 300, 600, 3) == iop.KIND_LINEAR     # one axis grows: area-mode bilinear
    assert kinds(20, 20, 30, 30, 4) == iop.KIND_KERNEL and iop.resize_plan(20, 20, 30, 30, 4)[1].shape[1] == 8
    assert kinds(49, 49, 1, 1, 3) == iop.KIND_AREA                        # 1 / (1 / 49) is not 49 in float64: the `fast` test fails as in OpenCV
    for interp in range(5):                                               # structural: indices inside the image; fixed-point rows sum to ~2048
        for sh, sw, dh, dw in ((20, 20, 30, 30), (48, 48, 13, 13), (300, 300, 1, 1), (5, 5, 64, 64), (1000, 1000, 300, 300)):
            kind, ix, wx, iy, wy, area = iop.resize_plan(sh, sw, dh, dw, interp)
            assert np.all(ix >= 0) and np.all(ix < sw) and np.all(iy >= 0) and np.all(iy < sh)
            if kind in (iop.KIND_LINEAR, iop.KIND_KERNEL):
                assert np.all(np.abs(wx.sum(axis=1) - 2048) <= 4) and np.all(np.abs(wy.sum(axis=1) - 2048) <= 4)
            if kind == iop.KIND_AREA:
                assert np.allclose(wx.sum(axis=1), 1.0, atol=1e-6) and np.allclose(wy.sum(axis=1), 1.0, atol=1e-6)
    # Lanczos' sine / cosine chain (shared with the device build of the tables): within an ulp or two of the C library's
    y = np.linspace(-np.pi, -0.75 * np.pi, 2001)
    sn, cs = iop.sincos_near_minus_pi(y)
    assert np.abs(sn - np.sin(y)).max() <= 4e-16 and np.abs(cs - np.cos(y)).max() <= 4e-16
    on, oc = npi.det_sincos(y)
    assert np.array_equal(sn, on) and np.array_equal(cs, oc)
    # equalizeHist: 4 pixels of value 10, 4 of 20, 8 of 30 -> lut[10] = 0, lut[20] = round(4 * 255 / 12) = 85, lut[30] = 255, below 10 -> 0
    hist = np.zeros(256, dtype=np.int64)
    hist[10], hist[20], hist[30] = 4, 4, 8
    lut = iop.equalize_table(hist)
    assert lut.dtype == np.uint8 and lut[10] == 0 and lut[20] == 85 and lut[30] == 255 and lut[5] == 0 and lut[25] == 85
    assert np.array_equal(iop.equalize_table(np.bincount([7] * 9, minlength=256)), np.arange(256))      # a constant plane is left alone


def test_product_plans_equal_the_oracles():
    """The product's plan builder and the oracle's are written separately (vectorised / per destination): same kind, indices and table
    values on a sweep of geometries, every interpolation mode."""
    from oracle import np_image as npi
    from ssd_keras_amd.data_generator import _image_ops as iop
    rng = np.random.RandomState(9)
    geos = [(23, 37, 23, 50), (23, 37, 15, 13), (40, 48, 20, 24), (40, 48, 10, 16), (30, 30, 10, 10), (23, 37, 46, 74), (375, 500, 300, 300),
            (12, 46, 36, 46), (33, 45, 20, 20), (7, 3, 2, 9), (301, 299, 300, 300), (1, 1, 5, 5), (9, 1, 3, 4)]
    geos += [tuple(int(v) for v in rng.randint(1, 400, size=4)) for _ in range(40)]
    for sh, sw, dh, dw in geos:
        for interp in range(5):
            a, b = iop.resize_plan(sh, sw, dh, dw, interp), npi.cv_resize_plan(sh, sw, dh, dw, interp)
            assert a[0] == b[0] and a[5] == b[5], (sh, sw, dh, dw, interp)
            for u, v in zip(a[1:5], b[1:5]):
                assert u.shape == v.shape and np.array_equal(u, v), (sh, sw, dh, dw, interp)


def test_oracle_colour_conversions_have_the_documented_properties():
    from oracle import np_image as npi
    rng = np.random.RandomState(1)
    rgb = rng.randint(0, 256, size=(64, 64, 3)).astype(np.uint8)
    hsv = npi.rgb2hsv_u8(rgb)
    assert hsv[..., 0].max() < 180 and np.array_equal(hsv[..., 2], rgb.max(axis=-1))
    back = npi.hsv2rgb_u8(hsv).astype(np.int64)
    assert np.abs(back - rgb).max() <= 4                      # 8-bit HSV quantises hue to 2 degrees
    grey = np.repeat(rng.randint(0, 256, size=(8, 8, 1)), 3, axis=-1).astype(np.uint8)
    g = npi.rgb2hsv_u8(grey)
    assert not g[..., 0].any() and not g[..., 1].any() and np.array_equal(npi.hsv2rgb_u8(g), grey)
    for (r, g_, b), h in (((255, 0, 0), 0), ((0, 255, 0), 60), ((0, 0, 255), 120), ((255, 255, 0), 30)):
        assert tuple(npi.rgb2hsv_u8(np.array([[[r, g_, b]]], dtype=np.uint8))[0, 0]) == (h, 255, 255)
    f = rng.uniform(0, 255, size=(32, 32, 3)).astype(np.float32)
    hf = npi.rgb2hsv_f32(f)
    assert hf[..., 0].min() >= 0 and hf[..., 0].max() < 360.001 and np.abs(npi.hsv2rgb_f32(hf) - f).max() < 1e-2
    assert tuple(npi.rgb2gray(np.array([[[255, 255, 255]]], dtype=np.uint8))[0]) == (255,)
    # resize: same size is the identity for every mode; a constant image stays constant
    img = rng.randint(0, 256, size=(13, 17, 3)).astype(np.uint8)
    for interp in range(5):
        assert np.array_equal(npi.resize(img, (17, 13), interp), img)
        assert np.array_equal(npi.resize(np.full((9, 9, 3), 99, np.uint8), (20, 5), interp), np.full((5, 20, 3), 99, np.uint8))


def test_host_logic_matches_reference(monkeypatch):
    """No GPU: the kernels are replaced by the NumPy restatement of their specification (oracle/np_image.py run_program / resize /
    LUT); everything else -- the product's program construction, draws, dtype bookkeeping, labels -- is the product's."""
    import torch
    from oracle import np_image as npi
    from ssd_keras_amd import _native as nat

    def fake_program(images, ops, args, out_dtype):
        ops, args = np.asarray(ops), np.asarray(args)
        outs = [npi.run_program(images[b].numpy(), ops[b], args[b]) for b in range(images.shape[0])]
        out = torch.from_numpy(np.stack(outs))
        assert out.dtype == out_dtype, (out.dtype, out_dtype)
        return out

    def fake_resize(images, out_h, out_w, kind, area, ix, wx, iy, wy):
        plan = (kind, ix, wx, iy, wy, area)
        return torch.from_numpy(np.stack([npi.cv_apply_plan(img, plan) for img in images.numpy()]))

    monkeypatch.setattr(nat, "to_device", lambda a, device=None, dtype=None: torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a)
    monkeypatch.setattr(nat, "image_program", fake_program)
    monkeypatch.setattr(nat, "image_resize_cv_u8", fake_resize)
    monkeypatch.setattr(nat, "image_hist_u8", lambda image, channel: torch.from_numpy(
        np.bincount(image.numpy()[..., channel].reshape(-1), minlength=256).astype(np.int64)))
    monkeypatch.setattr(nat, "image_lut_u8", lambda image, table, mask: torch.from_numpy(np.where(
        (np.array([(mask >> c) & 1 for c in range(image.shape[-1])]) == 1), np.asarray(table)[image.numpy()], image.numpy()).astype(np.uint8)))
    _check_all(needs_gpu_boxes=False)


@pytest.mark.gpu
def test_image_ops_match_reference_on_gpu():
    _check_all(needs_gpu_boxes=True)


@pytest.mark.gpu
def test_batch_distortion_is_one_launch_of_the_per_image_programs():
    import torch
    from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDPhotometricDistortions
    rng = np.random.RandomState(3)
    batch = rng.randint(0, 256, size=(9, 33, 47, 3)).astype(np.uint8)
    d = SSDPhotometricDistortions()
    np.random.seed(11)
    lab = np.array([[1, 2, 3, 10, 12]])
    want = np.stack([d(batch[i], lab)[0] for i in range(batch.shape[0])])
    np.random.seed(11)
    got = d.distort_batch(torch.from_numpy(batch).cuda())
    assert got.dtype == torch.uint8 and got.is_cuda and np.array_equal(got.cpu().numpy(), want)
    assert not np.array_equal(want, batch)


@pytest.mark.gpu
def test_gamma_and_batched_resize():
    import torch
    from oracle import np_image as npi
    from ssd_keras_amd.data_generator import _image_ops as iop
    from ssd_keras_amd.data_generator.object_detection_2d_photometric_ops import Gamma, RandomGamma
    rng = np.random.RandomState(4)
    img = rng.randint(0, 256, size=(21, 19, 3)).astype(np.uint8)
    g = Gamma(gamma=0.6)
    assert np.array_equal(g(img), g.table[img])               # (the reference's Gamma cannot run: NameError, see the module docstring)
    np.random.seed(2)
    out = RandomGamma(prob=1.0)(img)
    assert out.shape == img.shape and out.dtype == np.uint8
    batch = rng.randint(0, 256, size=(5, 40, 52, 3)).astype(np.uint8)
    for interp in range(5):
        got = iop.resize(torch.from_numpy(batch).cuda(), 30, 30, interp).cpu().numpy()
        want = np.stack([npi.resize(batch[i], (30, 30), interp) for i in range(batch.shape[0])])
        assert np.array_equal(got, want), interp
    big = rng.randint(0, 256, size=(2, 375, 500, 3)).astype(np.uint8)          # a VOC-sized image down to the network input
    got = iop.resize(torch.from_numpy(big).cuda(), 300, 300, 3).cpu().numpy()
    assert np.array_equal(got[1], npi.resize(big[1], (300, 300), 3))
    # every path of cv::resize's dispatch through the kernel: fast area (2 x 2, 3 x 3, 4 x 1), true area, area-mode bilinear, the
    # fixed-point kernels shrinking and enlarging, nearest, copy -- and the hand-computed cases of tests/resize_hand_cases.py
    from tests import resize_hand_cases as hc
    src = rng.randint(0, 256, size=(3, 48, 60, 3)).astype(np.uint8)
    for out_h, out_w in ((24, 30), (16, 20), (48, 15), (31, 47), (48, 90), (96, 120), (100, 33), (48, 60), (7, 5)):
        for interp in range(5):
            got = iop.resize(torch.from_numpy(src).cuda(), out_h, out_w, interp).cpu().numpy()
            want = np.stack([npi.resize(src[i], (out_w, out_h), interp) for i in range(src.shape[0])])
            assert np.array_equal(got, want), (out_h, out_w, interp, int((got != want).sum()))
    for name, img, dsize, interp, want in hc.CASES:
        assert np.array_equal(iop.resize(img, dsize[1], dsize[0], interp), want), name


@pytest.mark.gpu
def test_augment_batch_equals_the_per_image_chain():
    """`SSDDataAugmentation.augment_batch` (device pipeline: one photometric launch + one gather launch per batch; reference
    data_generator/data_augmentation_chain_original_ssd.py:208-280) == the per-image chain called on image 0, 1, 2, ... with the same
    NumPy random state: images and labels bit for bit, and the random stream ends in the same place."""
    import torch
    from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation
    rng = np.random.RandomState(42)
    B, H, W = 12, 120, 160
    batch = rng.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    labels = []
    for _ in range(B):
        n = rng.randint(1, 5)
        x0, y0 = rng.randint(0, W - 40, size=n), rng.randint(0, H - 40, size=n)
        labels.append(np.stack([rng.randint(1, 21, size=n), x0, y0, x0 + rng.randint(12, 40, size=n), y0 + rng.randint(12, 40, size=n)], axis=1))
    for seed in (0, 1, 2):
        aug = SSDDataAugmentation(img_height=64, img_width=64)
        np.random.seed(seed)
        want = [aug(batch[i], labels[i]) for i in range(B)]
        after_want = np.random.uniform()
        np.random.seed(seed)
        got_img, got_lab = aug.augment_batch(torch.from_numpy(batch).cuda(), labels)
        after_got = np.random.uniform()
        assert after_got == after_want, "the random stream must end where the per-image chain leaves it"
        assert got_img.is_cuda and got_img.dtype == torch.uint8 and tuple(got_img.shape) == (B, 64, 64, 3)
        g = got_img.cpu().numpy()
        for i in range(B):
            assert np.array_equal(g[i], want[i][0]), "seed %d image %d: %d pixels differ" % (seed, i, int((g[i] != want[i][0]).sum()))
            assert np.array_equal(got_lab[i], want[i][1]), (seed, i)


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", [(12, 120, 160, 64, "int64"), (32, 375, 500, 300, "int64"), (9, 96, 128, 48, "float64")])
def test_augment_batch_with_seeds_equals_the_chain_under_each_seed(geometry):
    """Round 5: `augment_batch(images, labels, seeds=s)` -- the host makes each image's photometric draws, ONE launch
    (`ssdhip_ssd_augment_decide`, a wave per image consuming that image's NumPy MT19937 stream on the device) takes every other decision of
    the chain and does the label arithmetic -- == `np.random.seed(s[i]); chain(image_i, labels_i)` of the per-image chain (reference
    data_generator/data_augmentation_chain_original_ssd.py:208-280): pixels and labels bit for bit, and each image's generator ends in
    the state the per-image chain leaves it in (the next draw is the same).  The global generator is not touched."""
    import torch
    from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation
    B, H, W, out, dt = geometry
    rng = np.random.RandomState(7 + B)
    batch = rng.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    labels = []
    for i in range(B):
        n = rng.randint(0 if i % 5 == 4 else 1, 7)
        x0, y0 = rng.randint(0, W - W // 4, size=n), rng.randint(0, H - H // 4, size=n)
        lab = np.stack([rng.randint(1, 21, size=n), x0, y0, x0 + rng.randint(W // 12, W // 4, size=n), y0 + rng.randint(H // 12, H // 4, size=n)],
                       axis=1).reshape(n, 5)
        labels.append(lab.astype(dt) + (0.25 if dt == "float64" else 0))
    for trial in range(3):
        seeds = rng.randint(0, 2 ** 31 - 1, size=B)
        aug = SSDDataAugmentation(img_height=out, img_width=out)
        want, nxt = [], []
        for i in range(B):
            np.random.seed(int(seeds[i]))
            want.append(aug(batch[i], labels[i]))
            nxt.append(np.random.uniform())
        np.random.seed(12345)
        before = np.random.get_state()
        got_img, got_lab = aug.augment_batch(torch.from_numpy(batch).cuda(), labels, seeds=seeds)
        after = np.random.get_state()
        assert before[2] == after[2] and np.array_equal(before[1], after[1]), "the global generator must be left alone"
        g = got_img.cpu().numpy()
        states = aug._last_generator_states
        for i in range(B):
            assert got_lab[i].dtype == labels[i].dtype and got_lab[i].shape == want[i][1].shape, (trial, i, got_lab[i], want[i][1])
            assert np.array_equal(got_lab[i], want[i][1]), (trial, i, got_lab[i], want[i][1])
            assert np.array_equal(g[i], want[i][0]), "trial %d image %d: %d pixels differ" % (trial, i, int((g[i] != want[i][0]).sum()))
            rs = np.random.RandomState()
            rs.set_state(("MT19937", states[i, :624].copy(), int(states[i, 624]), 0, 0.0))
            assert rs.uniform() == nxt[i], "image %d: the device left the stream elsewhere" % i


@pytest.mark.gpu
def test_augment_batch_global_stream_on_the_device_equals_the_reference_chain():
    """Round 6: `augment_batch(images, labels)` WITHOUT seeds keeps the reference's contract -- one global np.random
    stream across the batch (object_detection_2d_data_generator.py:1050-1089 -> data_augmentation_chain_original_ssd.py:208-280) -- with
    every decision, photometric ones included, taken by one wave on the device (ssdhip_ssd_augment_decide_stream).
    (a) the ten `ssd_augmentation` cases of the reference-generated fixture, each as a batch of one under the case's seed: pixels,
        labels and the three random numbers that follow, bit for bit against the REAL reference's outputs;
    (b) a VOC-sized batch of 32 and a second batch behind it: == the per-image host loop (SSDHIP_AUG_HOST_STREAM=1) and == the
        per-image chain, pixels, labels and the FULL generator state (np.random.get_state()) after each batch; int64 and float64 labels;
    (c) a generator position on the twist boundary (624) and right behind a twist."""
    import os
    import torch
    from ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd import SSDDataAugmentation
    z = util.load("image_ops")
    n = 0
    for i, case in enumerate(ic.CASES):
        if case["op"] != "ssd_augmentation":
            continue
        img, labels = ic.make_inputs(case["seed"], case["n_boxes"], size=(40, 52))
        aug = SSDDataAugmentation(img_height=case["out"][0], img_width=case["out"][1])
        np.random.seed(case["seed"])
        got_img, got_lab = aug.augment_batch(torch.from_numpy(np.ascontiguousarray(img)[None]).cuda(), [labels])
        probe = np.random.uniform(0, 1, size=3)
        pre = "i%03d_" % i
        assert np.array_equal(got_img[0].cpu().numpy(), z[pre + "image"]), case
        assert np.array_equal(got_lab[0], z[pre + "labels"]) and got_lab[0].dtype == z[pre + "labels"].dtype, case
        assert np.array_equal(probe, z[pre + "probe"]), case
        n += 1
    assert n == 10

    def states_equal(a, b):
        return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]

    for dt in (np.int64, np.float64):
        rng = np.random.RandomState(77)
        B, H, W = 32, 375, 500
        batches = [rng.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8) for _ in range(2)]
        labels = []
        for _ in range(2):
            cur = []
            for k in range(B):
                g = rng.randint(0 if k % 7 == 6 else 1, 8)
                x0, y0 = rng.randint(0, W - 60, size=g), rng.randint(0, H - 60, size=g)
                cur.append(np.stack([rng.randint(1, 21, size=g), x0, y0, x0 + rng.randint(10, 60, size=g), y0 + rng.randint(10, 60, size=g)],
                                    axis=1).astype(dt))
            labels.append(cur)
        aug = SSDDataAugmentation(img_height=300, img_width=300)
        np.random.seed(5)
        want, want_states = [], []
        for bi in range(2):
            want.append([aug(batches[bi][k], labels[bi][k]) for k in range(B)])
            want_states.append(np.random.get_state())
        for host in ("0", "1"):
            os.environ["SSDHIP_AUG_HOST_STREAM"] = host
            try:
                np.random.seed(5)
                for bi in range(2):
                    got_img, got_lab = aug.augment_batch(torch.from_numpy(batches[bi]).cuda(), labels[bi])
                    assert states_equal(np.random.get_state(), want_states[bi]), (dt, host, bi)
                    g = got_img.cpu().numpy()
                    for k in range(B):
                        assert np.array_equal(g[k], want[bi][k][0]), (dt, host, bi, k, int((g[k] != want[bi][k][0]).sum()))
                        assert np.array_equal(got_lab[k], want[bi][k][1]) and got_lab[k].dtype == want[bi][k][1].dtype, (dt, host, bi, k)
            finally:
                os.environ.pop("SSDHIP_AUG_HOST_STREAM", None)

    # (c) the generator's position at 624 (every word of the block consumed: the next draw twists) and at 2
    small = np.random.RandomState(3).randint(0, 256, size=(5, 60, 80, 3)).astype(np.uint8)
    small_lab = [np.array([[3, 10, 12, 50, 44], [7, 30, 20, 70, 55]], dtype=np.int64) for _ in range(5)]
    aug = SSDDataAugmentation(img_height=48, img_width=48)
    for burn in (312, 313):                                 # 312 doubles = 624 words -> pos 624; one more double -> pos 2 of the next block
        np.random.seed(11)
        np.random.random_sample(burn)
        assert np.random.get_state()[2] == (624 if burn == 312 else 2)
        want = [aug(small[k], small_lab[k]) for k in range(5)]
        want_state = np.random.get_state()
        np.random.seed(11)
        np.random.random_sample(burn)
        got_img, got_lab = aug.augment_batch(torch.from_numpy(small).cuda(), small_lab)
        assert states_equal(np.random.get_state(), want_state), burn
        assert all(np.array_equal(got_img[k].cpu().numpy(), want[k][0]) and np.array_equal(got_lab[k], want[k][1]) for k in range(5)), burn
