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
    """The product's tap / table builders (`_image_ops.axis_taps`, `equalize_table`) against values worked out BY HAND from the
    published definitions (OpenCV's sampling geometry src = (dst + 0.5) scale - 0.5 with replicated borders, the a = -0.75 cubic
    kernel, the Lanczos-4 kernel, the box filter, equalizeHist's scale) -- not against oracle/np_image.py, whose builders are their twins
    (comparing the two with each other pinned nothing)."""
    from ssd_keras_amd.data_generator import _image_ops as iop
    # nearest, 6 -> 4: floor(dst 1.5) = 0, 1, 3, 4
    i, w = iop.axis_taps(6, 4, iop.INTER_NEAREST)
    assert i.dtype == np.int32 and i[:, 0].tolist() == [0, 1, 3, 4] and np.array_equal(w, np.ones((4, 1)))
    # linear, 4 -> 2: centres 0.5 and 2.5 -> taps (0, 1) and (2, 3) with weights (0.5, 0.5)
    i, w = iop.axis_taps(4, 2, iop.INTER_LINEAR)
    assert i.tolist() == [[0, 1], [2, 3]] and np.array_equal(w, [[0.5, 0.5], [0.5, 0.5]])
    # linear, 2 -> 4 (enlarging): centres -0.25, 0.25, 0.75, 1.25; the first / last taps are clamped to the border pixel
    i, w = iop.axis_taps(2, 4, iop.INTER_LINEAR)
    assert i.tolist() == [[0, 0], [0, 1], [0, 1], [1, 1]]
    assert np.allclose(w, [[0.25, 0.75], [0.75, 0.25], [0.25, 0.75], [0.75, 0.25]], rtol=0, atol=1e-15)
    # cubic (a = -0.75), 8 -> 4: every centre falls half way between two pixels: weights (-3/32, 19/32, 19/32, -3/32)
    i, w = iop.axis_taps(8, 4, iop.INTER_CUBIC)
    assert i.tolist() == [[0, 0, 1, 2], [1, 2, 3, 4], [3, 4, 5, 6], [5, 6, 7, 7]]
    assert np.allclose(w, np.tile([-0.09375, 0.59375, 0.59375, -0.09375], (4, 1)), rtol=0, atol=1e-15)
    # cubic, same size: fraction 0 -> the pixel itself
    i, w = iop.axis_taps(5, 5, iop.INTER_CUBIC)
    assert np.allclose(w, np.tile([0.0, 1.0, 0.0, 0.0], (5, 1)), rtol=0, atol=1e-15) and i[:, 1].tolist() == [0, 1, 2, 3, 4]
    # Lanczos-4, same size: the delta; 8 -> 4: symmetric in the two centre taps, weights sum to one, outer lobes as the closed form gives
    i, w = iop.axis_taps(7, 7, iop.INTER_LANCZOS4)
    assert np.allclose(w[:, 3], 1.0) and np.allclose(np.delete(w, 3, axis=1), 0.0)
    i, w = iop.axis_taps(16, 8, iop.INTER_LANCZOS4)
    lz = lambda t: 4 * np.sin(np.pi * t) * np.sin(np.pi * t / 4) / (np.pi ** 2 * t ** 2)
    row = np.array([lz(t) for t in (3.5, 2.5, 1.5, 0.5, 0.5, 1.5, 2.5, 3.5)])
    assert np.allclose(w[3], row / row.sum(), rtol=1e-12) and i[3].tolist() == [3, 4, 5, 6, 7, 8, 9, 10]
    # area, 4 -> 2: the box filter over two pixels; 3 -> 2: cells [0, 1.5) and [1.5, 3): weights (1, 0.5) / 1.5 and (0.5, 1) / 1.5
    i, w = iop.axis_taps(4, 2, iop.INTER_AREA)
    assert np.allclose(w[:, :2], 0.5) and np.allclose(w[:, 2:], 0.0) and i[:, :2].tolist() == [[0, 1], [2, 3]]
    i, w = iop.axis_taps(3, 2, iop.INTER_AREA)
    assert np.allclose(w[0, :2], [2 / 3, 1 / 3]) and np.allclose(w[1, :2], [1 / 3, 2 / 3]) and i[0, :2].tolist() == [0, 1] and i[1, :2].tolist() == [1, 2]
    # area when NOT both axes shrink (cv2's `area_mode` bilinear variant): an integer enlargement replicates pixels, 2 -> 4: a a b b;
    # 3 -> 4: sx = 0, 0, 1, 2 and fx = frac(1 - 4/3 <= 0 -> 0), 2 - 4/3 = 2/3, 3 - 8/3 = 1/3, last pixel -> 0
    i, w = iop.axis_taps(2, 4, iop.INTER_AREA, area_linear=True)
    assert (i[:, 0] * (w[:, 0] == 1)).tolist() == [0, 0, 1, 1] and np.array_equal(w, [[1, 0]] * 4)
    i, w = iop.axis_taps(3, 4, iop.INTER_AREA, area_linear=True)
    assert i[:, 0].tolist() == [0, 0, 1, 2] and np.allclose(w[:, 1], [0.0, 2 / 3, 1 / 3, 0.0], rtol=0, atol=1e-6)
    for interp in range(5):                               # structural: indices inside the image, weights sum to one
        for n_src, n_dst in ((20, 30), (48, 13), (300, 1), (5, 64), (1000, 300)):
            i, w = iop.axis_taps(n_src, n_dst, interp)
            assert i.dtype == np.int32 and np.all(i >= 0) and np.all(i < n_src) and np.allclose(w.sum(axis=1), 1.0)
    # equalizeHist: 4 pixels of value 10, 4 of 20, 8 of 30 -> lut[10] = 0, lut[20] = round(4 * 255 / 12) = 85, lut[30] = 255, below 10 -> 0
    hist = np.zeros(256, dtype=np.int64)
    hist[10], hist[20], hist[30] = 4, 4, 8
    lut = iop.equalize_table(hist)
    assert lut.dtype == np.uint8 and lut[10] == 0 and lut[20] == 85 and lut[30] == 255 and lut[5] == 0 and lut[25] == 85
    assert np.array_equal(iop.equalize_table(np.bincount([7] * 9, minlength=256)), np.arange(256))      # a constant plane is left alone


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

    def fake_resize(images, out_h, out_w, ix, wx, iy, wy):
        src = images.numpy().astype(np.float64)
        acc = np.zeros((src.shape[0], out_h, out_w, src.shape[3]))
        for j in range(iy.shape[1]):
            rows = src[:, iy[:, j]]
            racc = np.zeros_like(acc)
            for t in range(ix.shape[1]):
                racc = racc + wx[None, None, :, t, None] * rows[:, :, ix[:, t]]
            acc = acc + wy[None, :, j, None, None] * racc
        return torch.from_numpy(np.clip(np.rint(acc), 0, 255).astype(np.uint8))

    monkeypatch.setattr(nat, "to_device", lambda a, device=None, dtype=None: torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a)
    monkeypatch.setattr(nat, "image_program", fake_program)
    monkeypatch.setattr(nat, "image_resize_u8", fake_resize)
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
