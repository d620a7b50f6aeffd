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


def test_host_tables_equal_the_oracle_restatement():
    from oracle import np_image as npi
    from ssd_keras_amd.data_generator import _image_ops as iop
    for interp in range(5):
        for n_src, n_dst in ((20, 30), (48, 13), (37, 37), (300, 1), (5, 64), (1000, 300)):
            a, b = iop.axis_taps(n_src, n_dst, interp), npi.resize_taps(n_src, n_dst, interp)
            assert a[0].dtype == np.int32 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            assert np.all(a[0] >= 0) and np.all(a[0] < n_src) and np.allclose(a[1].sum(axis=1), 1.0)
    rng = np.random.RandomState(0)
    for _ in range(6):
        plane = rng.randint(0, 256, size=(31, 17)).astype(np.uint8) // rng.randint(1, 9)
        hist = np.bincount(plane.reshape(-1), minlength=256)
        assert np.array_equal(iop.equalize_table(hist)[plane], npi.equalize_hist(plane))
    assert np.array_equal(iop.equalize_table(np.bincount([7] * 9, minlength=256))[np.full((3, 3), 7)], np.full((3, 3), 7))


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
