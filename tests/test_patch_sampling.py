"""Patch sampling ops (drop-in for data_generator/object_detection_2d_patch_sampling_ops.py + SSDRandomCrop / SSDExpand) against
golden outputs of the real reference on the seeded cases of tests/patch_cases.py: same patch, same labels, same inverter, and the
same position of NumPy's global random stream afterwards.

Two runs of the same comparison: on the GPU the (patch, box) validity tests go through ssdhip_box_filter; without a GPU the kernel
call alone is replaced by the oracle's restatement of the same test (test infrastructure), which exercises all the host logic --
trial batching, stream replay, cropping, label arithmetic."""
import ast
import types

import numpy as np
import pytest

from tests import patch_cases as pc
from tests import util


def _namespace():
    import ssd_keras_amd.data_generator.data_augmentation_chain_original_ssd as chain
    import ssd_keras_amd.data_generator.object_detection_2d_image_boxes_validation_utils as val
    import ssd_keras_amd.data_generator.object_detection_2d_patch_sampling_ops as ops
    ns = types.SimpleNamespace(SSDRandomCrop=chain.SSDRandomCrop, SSDExpand=chain.SSDExpand, BoundGenerator=val.BoundGenerator,
                               BoxFilter=val.BoxFilter, ImageValidator=val.ImageValidator)
    for name in ("PatchCoordinateGenerator", "CropPad", "Crop", "Pad", "RandomPatch", "RandomPatchInf", "RandomMaxCropFixedAR",
                 "RandomPadFixedAR"):
        setattr(ns, name, getattr(ops, name))
    return ns


def _compare_all():
    z = util.load("patch_sampling")
    ns = _namespace()
    assert int(z["n_cases"]) == len(pc.CASES)
    state = np.random.get_state()
    try:
        for i, case in enumerate(pc.CASES):
            assert ast.literal_eval(str(z["p%03d_case" % i])) == case, "fixture is stale: rerun tests/golden/make_golden.py patch_sampling"
            got = pc.run(ns, case)
            want = {k[5:]: z[k] for k in z.files if k.startswith("p%03d_" % i) and not k.endswith("_case")}
            assert set(got) == set(want), (case, sorted(got), sorted(want))
            for k in want:
                g, w = np.asarray(got[k]), want[k]
                assert g.shape == w.shape, (case, k, g.shape, w.shape)
                assert g.dtype == w.dtype, (case, k, g.dtype, w.dtype)
                assert np.array_equal(g, w), (case, k)
    finally:
        np.random.set_state(state)


def test_patch_sampling_host_logic_matches_reference(monkeypatch):
    import torch
    from oracle import np_oracle as orc
    from ssd_keras_amd import _native as nat
    launches = []

    def box_filter(boxes, box_image, image_hw, check_overlap, check_min_area, check_degenerate, criterion, lower, upper, min_area,
                   border_pixels):
        boxes, box_image, image_hw = np.asarray(boxes), np.asarray(box_image), np.asarray(image_hw)
        launches.append(len(image_hw))
        keep = np.zeros(len(boxes), dtype=bool)
        for i in range(len(image_hw)):
            sel = box_image == i
            lab = np.concatenate([np.zeros((int(sel.sum()), 1)), boxes[sel]], axis=1)
            with np.errstate(divide="ignore", invalid="ignore"):
                keep[sel] = orc.box_filter_mask(lab, image_hw[i, 0], image_hw[i, 1], check_overlap, check_min_area, check_degenerate,
                                                criterion, (lower, upper), min_area, border_pixels)
        return torch.from_numpy(keep.astype(np.uint8))

    monkeypatch.setattr(nat, "box_filter", box_filter)
    _compare_all()
    assert max(launches) == 50                        # a whole SSDRandomCrop round (50 candidate patches) went out as one call


@pytest.mark.gpu
def test_patch_sampling_matches_reference_on_gpu():
    _compare_all()


def test_crop_pad_without_labels_returns_the_image():
    """The reference raises IndexError here (np.copy(None) is not None, :247-253); the drop-in does what the docstring says."""
    from ssd_keras_amd.data_generator.object_detection_2d_patch_sampling_ops import CropPad, Pad
    image = np.arange(6 * 8 * 3, dtype=np.uint8).reshape(6, 8, 3)
    patch = CropPad(1, 2, 4, 5)(image)
    assert np.array_equal(patch, image[1:5, 2:7])
    patch, inv = Pad(1, 1, 2, 2, background=(9, 9, 9))(image, return_inverter=True)
    assert patch.shape == (8, 12, 3) and np.array_equal(patch[1:7, 2:10], image) and np.all(patch[0] == 9)
    assert np.array_equal(inv(np.zeros((1, 6)))[0], [0, 0, -2, -1, -2, -1])
    with pytest.raises(ValueError):
        CropPad(7, 0, 2, 2)(image)


def test_apply_inverse_transforms():
    """object_detection_2d_misc_utils.apply_inverse_transforms (:22-73) with CropPad's inverters: list and array containers,
    `None` inverters, batch items without predictions."""
    from ssd_keras_amd.data_generator.object_detection_2d_misc_utils import apply_inverse_transforms
    from ssd_keras_amd.data_generator.object_detection_2d_patch_sampling_ops import CropPad, Pad
    image = np.zeros((20, 30, 3), dtype=np.uint8)
    _, inv_a = CropPad(2, 3, 10, 12)(image, return_inverter=True)
    _, inv_b = Pad(4, 0, 5, 0)(image, return_inverter=True)
    pred = np.array([[1, 0.9, 1.0, 2.0, 5.0, 6.0], [2, 0.8, 0.0, 0.0, 3.0, 3.0]])
    empty = np.array([])
    out = apply_inverse_transforms([pred, empty, pred], [[inv_a, None, inv_b], [inv_a], []])
    assert np.array_equal(out[0][:, 2:], pred[:, 2:] + np.array([3 - 5, 2 - 4, 3 - 5, 2 - 4])) and np.array_equal(out[0][:, :2], pred[:, :2])
    assert out[1].shape == (0,) and np.array_equal(out[2], pred) and out[2] is not pred
    arr = np.stack([pred, pred])
    out = apply_inverse_transforms(arr, [[inv_a], [None]])
    assert np.array_equal(out[0][:, 2:], pred[:, 2:] + np.array([3, 2, 3, 2])) and np.array_equal(out[1], pred) and out is not arr
    with pytest.raises(ValueError):
        apply_inverse_transforms((pred,), [[inv_a]])
