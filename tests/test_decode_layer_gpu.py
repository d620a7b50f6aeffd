"""The in-graph decoders (`DecodeDetections`, `DecodeDetectionsFast`: libssdhip SSDHIP_SEM_KERAS) vs the oracle's
restatement of keras_layer_DecodeDetections.py:109-265 / keras_layer_DecodeDetectionsFast.py:111-248, plus the
degenerate-box regime a randomly initialised SSD300 produces (what bench.py decodes).  Needs an MI355X.

Bar: the `(B, top_k, 6)` float32 output is bit exact (rows, order, padding).  Parity of the layer itself is
UNPINNED (TensorFlow absent): the oracle restates it from source and is cross-checked against the pinned NumPy
decoder in `test_layer_selection_equals_numpy_decoder`.
"""
import numpy as np
import pytest

from oracle import np_oracle as orc
from ssd_keras_amd import synthetic as syn
from tests import util
from tests.test_oracle_golden import _encoder

pytestmark = pytest.mark.gpu


def _layers():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from ssd_keras_amd.keras_layers.keras_layer_DecodeDetections import DecodeDetections
    from ssd_keras_amd.keras_layers.keras_layer_DecodeDetectionsFast import DecodeDetectionsFast
    return torch, DecodeDetections, DecodeDetectionsFast


def _wild(cfg, B, seed, sigma=150.0, bias=0.0, special=True):
    """Predictions like a random-init network's: offsets of order +-sigma (boxes from 1e-15 px to inf), plus
    hand-placed inf / NaN / zero-size rows."""
    enc = _encoder(cfg)
    av = enc.generate_encoding_template(1)[0, :, -8:]
    y = syn.make_y_pred(av, B, enc.n_classes, bias=bias, seed=seed, loc_sigma=sigma)
    if special:
        C = enc.n_classes
        rng = np.random.RandomState(seed + 1)
        n = y.shape[1]
        for b in range(B):
            rows = rng.choice(n, size=min(40, n // 4), replace=False)
            y[b, rows[0:8], C + 2] = 1e4            # exp overflows: w = inf
            y[b, rows[8:16], C + 3] = -1e4          # h = 0
            y[b, rows[16:20], C + 0] = np.inf       # cx = inf -> xmin = xmax = inf (area NaN)
            y[b, rows[20:24], C + 1] = np.nan
            y[b, rows[24:32], C:C + 4] = 0.0        # exactly the anchor
            y[b, rows[32:40], C + 2:C + 4] = 440.0  # w ~ e^88: float32 area overflows, float64 does not
    return y, enc


@pytest.mark.parametrize("cfg,B,bias,thr,top_k,cap", [("tiny", 3, 1.0, 0.05, 20, 400), ("tiny", 2, 0.0, 0.01, 200, 7),
                                                      ("ssd7", 2, 0.0, 0.01, 200, 400), ("ssd300", 2, 7.0, 0.01, 200, 400),
                                                      ("ssd300", 1, 0.0, 0.3, 200, 400)])
def test_layer_vs_oracle(cfg, B, bias, thr, top_k, cap):
    torch, DD, DDF = _layers()
    c = util.CFGS[cfg]
    enc = _encoder(c)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], B, enc.n_classes, bias=bias, seed=11)
    kw = dict(confidence_thresh=thr, iou_threshold=0.45, top_k=top_k, nms_max_output_size=cap, normalize_coords=True,
              img_height=c["img_height"], img_width=c["img_width"])
    yd = torch.from_numpy(y).cuda()
    for layer, fast in ((DD, False), (DDF, True)):
        got = layer(**kw)(yd).cpu().numpy()
        want = orc.decode_detections_layer(y, fast=fast, exp_mode="det", **kw)
        assert got.shape == (B, top_k, 6) and got.dtype == np.float32
        assert np.array_equal(got, want), "%s fast=%s: %d rows differ" % (cfg, fast, int((got != want).any(-1).sum()))


def test_layer_selection_equals_numpy_decoder():
    """keras layer == NumPy decode_detections as a set of (class, conf) with boxes within 1e-4 px when the cap cannot
    bind (nms_max_output_size >= top_k) -- ties the unpinned layer restatement to the pinned NumPy path."""
    torch, DD, _ = _layers()
    c = util.CFGS["ssd7"]
    enc = _encoder(c)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], 2, enc.n_classes, bias=2.0, seed=5)
    kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=300, img_width=300)
    got = DD(nms_max_output_size=400, **kw)(torch.from_numpy(y).cuda()).cpu().numpy()
    want = orc.decode_detections(y, exp_mode="det", **kw)
    for g, w in zip(got, want):
        g = g[g[:, 1] > 0]
        assert g.shape[0] == w.shape[0]
        util.dets_equal([g.astype(np.float64)], [w], exact=False, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg,B,bias", [("ssd7", 2, 2.0), ("tiny", 4, 1.0), ("ssd300", 2, 5.0)])
def test_fast_layer_selection_equals_numpy_fast_decoder(cfg, B, bias):
    """DecodeDetectionsFast (keras_layer_DecodeDetectionsFast.py:111-248) against the PINNED NumPy decode_detections_fast
    (ssd_output_decoder.py:228-333): class-agnostic NMS over each anchor's arg-max class.  When the layer's NMS cap cannot bind
    (nms_max_output_size >= top_k) and no confidence sits exactly on the threshold (the layer tests `>`, the NumPy function `>=`)
    the two select the same (class, confidence) rows with boxes within 1e-4 px -- this ties the unpinned layer restatement to the
    reference-pinned half of the oracle, as test_layer_selection_equals_numpy_decoder does for DecodeDetections."""
    torch, _, DDF = _layers()
    c = util.CFGS[cfg]
    enc = _encoder(c)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], B, enc.n_classes, bias=bias, seed=9)
    kw = dict(confidence_thresh=0.3, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=c["img_height"],
              img_width=c["img_width"])
    assert not np.any(y[:, :, :enc.n_classes].max(axis=-1) == np.float32(0.3))
    got = DDF(nms_max_output_size=400, **kw)(torch.from_numpy(y).cuda()).cpu().numpy()
    want = orc.decode_detections_fast(y, exp_mode="det", **kw)
    n_rows = 0
    for g, w in zip(got, want):
        g = g[g[:, 1] > 0]
        w = np.asarray(w).reshape(-1, 6)
        assert g.shape[0] == w.shape[0]
        n_rows += g.shape[0]
        util.dets_equal([g.astype(np.float64)], [w], exact=False, rtol=1e-4, atol=1e-4)
    assert n_rows > 0


@pytest.mark.parametrize("cfg,B,sigma", [("tiny", 4, 150.0), ("tiny", 2, 30.0), ("ssd7", 1, 150.0)])
def test_degenerate_boxes_layer_and_numpy(cfg, B, sigma):
    """inf / NaN / zero-area / astronomically large boxes: every pair the division-free NMS test cannot decide must
    fall through to the reference's exact IEEE evaluation."""
    torch, DD, DDF = _layers()
    from ssd_keras_amd.ssd_encoder_decoder import ssd_output_decoder as dec
    c = util.CFGS[cfg]
    y, enc = _wild(c, B, seed=21, sigma=sigma)
    kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, normalize_coords=True, img_height=c["img_height"],
              img_width=c["img_width"])
    with np.errstate(all="ignore"):
        want_np = orc.decode_detections(y, exp_mode="det", **kw)
        want_l = orc.decode_detections_layer(y, fast=False, exp_mode="det", nms_max_output_size=400, **kw)
        want_f = orc.decode_detections_layer(y, fast=True, exp_mode="det", nms_max_output_size=400, **kw)
    got_np = dec.decode_detections(y, **kw)
    for g, w in zip(got_np, want_np):
        gs, ws = util.sort_rows(g), util.sort_rows(w)
        assert gs.shape == ws.shape and np.array_equal(gs, ws, equal_nan=True)
    yd = torch.from_numpy(y).cuda()
    got_l = DD(nms_max_output_size=400, **kw)(yd).cpu().numpy()
    got_f = DDF(nms_max_output_size=400, **kw)(yd).cpu().numpy()
    assert np.array_equal(got_l, want_l, equal_nan=True)
    assert np.array_equal(got_f, want_f, equal_nan=True)


@pytest.mark.parametrize("border", ["half", "include", "exclude"])
def test_degenerate_boxes_corners_and_borders(border):
    """float32 flow ('corners' input) and the border_pixels variants on the same wild tensor."""
    _layers()
    from ssd_keras_amd.ssd_encoder_decoder import ssd_output_decoder as dec
    c = util.CFGS["tiny"]
    y, enc = _wild(c, 2, seed=33, sigma=40.0)
    for coords in ("centroids", "corners", "minmax"):
        kw = dict(confidence_thresh=0.01, iou_threshold=0.45, top_k=100, normalize_coords=True, img_height=c["img_height"],
                  img_width=c["img_width"], input_coords=coords, border_pixels=border)
        with np.errstate(all="ignore"):
            want = orc.decode_detections(y, exp_mode="det", **kw)
        got = dec.decode_detections(y, **kw)
        for g, w in zip(got, want):
            gs, ws = util.sort_rows(g), util.sort_rows(w)
            assert gs.shape == ws.shape and np.array_equal(gs, ws, equal_nan=True), (coords, border)


def test_full_batch_properties_ssd300():
    """BASELINE size (B=32, 8732 anchors, 21 classes; dense regime): size-independent properties of the layer output --
    rows sorted by confidence, zero padding contiguous at the end, every row's class/conf/box is an actual
    (class, anchor) candidate above the threshold, no two rows of one class overlap by more than the NMS threshold,
    and decoding is idempotent across calls and batch order."""
    torch, DD, _ = _layers()
    c = util.CFGS["ssd300"]
    enc = _encoder(c)
    y = syn.make_y_pred(enc.generate_encoding_template(1)[0, :, -8:], 32, enc.n_classes, bias=0.0, seed=77)
    layer = DD(confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400, normalize_coords=True,
               img_height=300, img_width=300)
    yd = torch.from_numpy(y).cuda()
    out = layer(yd).cpu().numpy()
    out2 = layer(yd.flip(0).contiguous()).cpu().numpy()[::-1]
    assert np.array_equal(out, out2)
    conf = out[:, :, 1]
    assert np.all(np.diff(conf, axis=1) <= 0)
    valid = conf > 0
    assert np.all(valid[:, :-1] >= valid[:, 1:])                   # padding only at the end
    assert np.all(out[~valid] == 0)
    for b in range(0, 32, 5):
        rows = out[b][valid[b]].astype(np.float64)
        cls = rows[:, 0].astype(int)
        assert np.all((cls >= 1) & (cls <= 20)) and np.all(rows[:, 1] > np.float32(0.01))
        for cl in np.unique(cls):
            r = rows[cls == cl]
            assert np.isin(r[:, 1].astype(np.float32), y[b, :, cl]).all()
            if r.shape[0] > 1:
                iou = orc.iou(r[:, 2:], r[:, 2:], coords="corners", mode="outer_product")
                np.fill_diagonal(iou, 0.0)
                assert iou.max() <= 0.45 + 1e-6
