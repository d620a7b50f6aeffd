"""HIP encoder (through the C ABI) vs the oracle and the reference's golden outputs.  Needs an MI355X.
Bars: match assignment (which GT each anchor got / background / neutral) and class vectors bit exact;
float64 offsets within 1e-12 relative (device log vs NumPy log), float32 output within 1e-6."""
import ast

import numpy as np
import pytest

from oracle import np_oracle as orc
from ssd_keras_amd import synthetic as syn
from tests import util

pytestmark = pytest.mark.gpu


def _pair(cfg, **over):
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import SSDInputEncoder
    kw = dict(cfg)
    kw.update(over)
    return SSDInputEncoder(**kw), orc.EncoderOracle(**kw)


def _check(enc, ora, gt, pre=""):
    with np.errstate(invalid="ignore", divide="ignore"):
        want, want_mm = ora(gt, return_matches=True)
    y32, y64, mm = enc.encode_to_device(gt, want_f32=True, want_f64=True, want_matches=True)
    got, got32, got_mm = y64.cpu().numpy(), y32.cpu().numpy(), mm.cpu().numpy()
    C = enc.n_classes
    assert np.array_equal(got_mm, want_mm), pre + "match map differs"
    assert np.array_equal(got[:, :, :C], want[:, :, :C]), pre + "class vectors differ"
    assert np.array_equal(got[:, :, -8:], want[:, :, -8:]), pre + "anchor/variance columns differ"
    np.testing.assert_allclose(got[:, :, C:C + 4], want[:, :, C:C + 4], rtol=1e-12, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(got32, want.astype(np.float32), rtol=1e-6, atol=1e-7, equal_nan=True)
    return got


def test_golden_encoder_cases():
    z = util.load("encoder")
    for ci in range(int(z["n_cases"])):
        pre = "c%02d_" % ci
        case = ast.literal_eval(str(z[pre + "params"]))
        over = {k: v for k, v in case.items() if k not in ("cfg", "seed", "B", "max_boxes", "min_boxes")}
        enc, ora = _pair(util.CFGS[case["cfg"]], **over)
        gt = util.unragged(z[pre + "gt"], z[pre + "gt_off"])
        got = _check(enc, ora, gt, pre)
        # straight against the reference's stored rows
        idx, rows = z[pre + "idx"], z[pre + "rows"]
        C = enc.n_classes
        sel = got[idx[:, 0], idx[:, 1], :]
        assert np.array_equal(sel[:, :C], rows[:, :C]), pre
        np.testing.assert_allclose(sel[:, C:], rows[:, C:], rtol=1e-12, atol=1e-13, equal_nan=True)
        mask = np.ones(got.shape[:2], bool)
        mask[idx[:, 0], idx[:, 1]] = False
        assert np.all(got[mask][:, enc.background_id] == 1) and np.all(got[mask][:, C:C + 4] == 0), pre


def test_call_returns_reference_container():
    enc, ora = _pair(syn.TINY)
    gt = syn.make_ground_truth(3, 5, 96, 128, max_boxes=5, seed=2)
    y = enc(gt)
    assert isinstance(y, np.ndarray) and y.dtype == np.float64 and y.shape == (3, enc.n_anchors, enc.n_classes + 12)
    y2, ym = enc(gt, diagnostics=True)
    assert np.array_equal(y, y2) and np.all(ym[:, :, -12:-8] == 0)
    assert np.array_equal(enc.generate_encoding_template(2), ora.generate_encoding_template(2))


@pytest.mark.parametrize("cfg,B,max_boxes,seed", [("ssd300", 32, 8, 7), ("ssd300", 8, 16, 9), ("ssd512", 16, 16, 11), ("ssd7", 4, 8, 13)])
def test_full_size(cfg, B, max_boxes, seed):
    c = util.CFGS[cfg]
    over = dict(pos_iou_threshold=0.5, neg_iou_limit=0.5 if cfg == "ssd300" else 0.3)
    enc, ora = _pair(c, **over)
    gt = syn.make_ground_truth(B, c["n_classes"], c["img_height"], c["img_width"], max_boxes=max_boxes, seed=seed)
    gt[0] = np.zeros((0, 5))
    _check(enc, ora, gt)


def test_many_boxes_and_quirks():
    enc, ora = _pair(syn.TINY, neg_iou_limit=0.2)
    rng = np.random.RandomState(0)
    gt = syn.make_ground_truth(2, 5, 96, 128, max_boxes=60, min_boxes=60, seed=5)
    # slivers that overlap nothing -> the "GT 0 gets anchor 0" quirk; duplicates -> last-write-wins
    gt[1] = np.concatenate([gt[1][:3], [[2, 0.0, 0.0, 0.3, 0.3]], [[3, 127.0, 95.0, 127.5, 95.5]], gt[1][:3]], axis=0)
    _check(enc, ora, gt)
    for mt in ("bipartite", "multi"):
        for thr in (0.0, 0.5):
            enc2, ora2 = _pair(syn.TINY, matching_type=mt, pos_iou_threshold=thr, neg_iou_limit=0.0 if thr == 0 else 0.3)
            _check(enc2, ora2, gt)


def test_degenerate_box_raises():
    from ssd_keras_amd.ssd_encoder_decoder.ssd_input_encoder import DegenerateBoxError
    enc, _ = _pair(syn.TINY)
    with pytest.raises(DegenerateBoxError):
        enc([np.array([[1, 10, 10, 10, 20.]])])
    with pytest.raises(ValueError):
        _pair(syn.TINY, variances=[0.1, 0.1, 0.2])
