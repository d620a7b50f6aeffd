"""HIP decoder (through the C ABI) vs the oracle and vs the reference's golden outputs.  Needs an MI355X.

Bars: selection (anchor, class, confidence) bit exact; float64 boxes bit exact against the oracle
run with the same deterministic exp; within rtol=atol=1e-4 (pixels) against the reference's own
outputs, whose np.exp(float32) is host dependent (north_star tolerance: 1e-4).
"""
import numpy as np
import pytest

from oracle import np_oracle as orc
from ssd_keras_amd import synthetic as syn
from tests import util
from tests.test_oracle_golden import _encoder, _y_pred_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from ssd_keras_amd.ssd_encoder_decoder import ssd_output_decoder as d
    return d


def _golden_cases(z):
    for name in [str(s) for s in z["cases"]]:
        yield name, str(z[name + "_fn"]), util.kw_of(z, name)


def test_golden_cases_vs_oracle_and_reference(dec):
    z = util.load("decoder")
    n = n64 = 0
    for name, fn_name, kw in _golden_cases(z):
        y = _y_pred_for(z, name)
        n64 += int(y.dtype == np.float64)            # float64 predictions: the reference's all-float64 flow (ssdhip_decode64.hip)
        want_ref = util.unragged(z[name + "_out"], z[name + "_off"])
        if fn_name == "decode_detections":
            got = dec.decode_detections(y, **kw)
            want = orc.decode_detections(y, exp_mode="det", **kw)
            n_meta = 2
        elif fn_name == "decode_detections_fast":
            got = dec.decode_detections_fast(y, **kw)
            want = orc.decode_detections_fast(y, exp_mode="det", **kw)
            n_meta = 2
        else:
            got = dec.decode_detections_debug(y, **kw)
            want = orc.decode_detections(y, exp_mode="det", with_anchor_index=True, decode_order="debug", **kw)
            n_meta = 3
        util.dets_equal(got, want, exact=True, n_meta=n_meta)                      # vs oracle: bit exact
        util.dets_equal(got, want_ref, exact=False, rtol=1e-4, atol=1e-4, n_meta=n_meta)   # vs the reference itself
        if fn_name == "decode_detections":
            for g, w in zip(got, want_ref):
                if w.shape[0] == 0:
                    assert g.shape == (0,)
        n += 1
    assert n >= 100 and n64 >= 40


def test_row_order_matches_reference_when_nothing_is_cut(dec):
    enc = _encoder(syn.TINY)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], 3, enc.n_classes, bias=2.0, seed=42)
    kw = dict(confidence_thresh=0.05, iou_threshold=0.45, top_k=5000, normalize_coords=True, img_height=96, img_width=128)
    got = dec.decode_detections(y, **kw)
    want = orc.decode_detections(y, exp_mode="det", **kw)
    for g, w in zip(got, want):
        assert 0 < w.shape[0] < 5000
        assert np.array_equal(g, w)             # same order: classes ascending, confidence descending


@pytest.mark.parametrize("cfg,B,bias,thr", [("ssd300", 4, 7.0, 0.01), ("ssd300", 2, 4.0, 0.01), ("ssd7", 2, 0.0, 0.01),
                                            ("ssd300", 2, 0.0, 0.5)])
def test_full_size_vs_oracle(dec, cfg, B, bias, thr):
    c = util.CFGS[cfg]
    enc = _encoder(c)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], B, enc.n_classes, bias=bias, seed=99)
    kw = dict(confidence_thresh=thr, iou_threshold=0.45, top_k=200, normalize_coords=True,
              img_height=c["img_height"], img_width=c["img_width"])
    got = dec.decode_detections(y, **kw)
    want = orc.decode_detections(y, exp_mode="det", **kw)
    # top-k ties at the k-th confidence are ambiguous in the reference itself (np.argpartition)
    util.dets_equal(got, want, exact=True)


def test_torch_tensor_input_stays_on_device(dec):
    import torch
    enc = _encoder(syn.TINY)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], 2, enc.n_classes, bias=1.0, seed=3)
    kw = dict(confidence_thresh=0.05, top_k=20, img_height=96, img_width=128)
    a = dec.decode_detections(torch.from_numpy(y).cuda(), **kw)
    b = dec.decode_detections(y, **kw)
    util.dets_equal(a, b, exact=True)


def test_errors_match_reference(dec):
    enc = _encoder(syn.TINY)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], 1, enc.n_classes)
    with pytest.raises(ValueError):
        dec.decode_detections(y, normalize_coords=True)            # image size missing (:164-165)
    with pytest.raises(ValueError):
        dec.decode_detections(y, input_coords="polar", img_height=1, img_width=1)
