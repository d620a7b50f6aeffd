"""The public box utilities on the GPU (through the C ABI: ssdhip_convert_coordinates, ssdhip_box_overlap,
ssdhip_match_bipartite_greedy, ssdhip_match_multi, ssdhip_greedy_nms) vs the reference's golden outputs and, at
sizes the reference would not finish quickly, vs the oracle.  Needs an MI355X.  Bar: bit exact (values and dtypes)."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

CONVS = ("minmax2centroids", "centroids2minmax", "corners2centroids", "centroids2corners", "minmax2corners", "corners2minmax")


def _mods():
    from ssd_keras_amd.bounding_box_utils import bounding_box_utils as bbu
    from ssd_keras_amd.ssd_encoder_decoder import matching_utils as mu
    from ssd_keras_amd.ssd_encoder_decoder import ssd_output_decoder as dec
    return bbu, mu, dec


def _same(got, want):
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want, equal_nan=True), "max abs diff %g" % np.nanmax(np.abs(got - want))


def _views(c1, c2):
    return {"corners": (c1, c2), "minmax": (c1[:, [0, 2, 1, 3]], c2[:, [0, 2, 1, 3]]),
            "centroids": (orc.convert_coordinates(c1, 0, "corners2centroids"), orc.convert_coordinates(c2, 0, "corners2centroids"))}


def test_golden_convert_and_iou():
    bbu, _, _ = _mods()
    z = util.load("box_utils")
    c1, c2 = z["corners1"], z["corners2"]
    for conv in CONVS:
        for bp in ("half", "include", "exclude"):
            _same(bbu.convert_coordinates(c1, 0, conv, bp), z["cc_%s_%s" % (conv, bp)])
        _same(bbu.convert_coordinates(c1.astype(np.float32), 0, conv, "half"), z["cc32_%s" % conv])
    for coords, (p, q) in _views(c1, c2).items():
        for bp in ("half", "include", "exclude"):
            _same(bbu.iou(p, q, coords, "outer_product", bp), z["iou_outer_%s_%s" % (coords, bp)])
            _same(bbu.iou(p, q[:7], coords, "element-wise", bp), z["iou_elem_%s_%s" % (coords, bp)])
            _same(bbu.iou(p, q[3], coords, "element-wise", bp), z["iou_bcast_%s_%s" % (coords, bp)])


def test_golden_overlap_dtypes_and_nd_convert():
    bbu, _, _ = _mods()
    z = util.load("box_utils2")
    for coords, (p, q) in _views(z["corners1"], z["corners2"]).items():
        p32, q32 = p.astype(np.float32), q.astype(np.float32)
        for bp in ("half", "include", "exclude"):
            tag = "%s_%s" % (coords, bp)
            _same(bbu.intersection_area(p, q, coords, "outer_product", bp), z["ia_outer_" + tag])
            _same(bbu.intersection_area(p, q[:13], coords, "element-wise", bp), z["ia_elem_" + tag])
            _same(bbu.iou(p32, q32, coords, "outer_product", bp), z["iou32_outer_" + tag])
            _same(bbu.iou(p32, q, coords, "outer_product", bp), z["iou3264_outer_" + tag])
            _same(bbu.iou(p, q32[:13], coords, "element-wise", bp), z["iou6432_elem_" + tag])
            _same(bbu.intersection_area(p32, q32, coords, "outer_product", bp), z["ia32_outer_" + tag])
    t = z["nd_in"]
    for conv in CONVS:
        _same(bbu.convert_coordinates(t, 3, conv, "include"), z["nd_" + conv])
        _same(bbu.convert_coordinates(t.astype(np.float32), 3, conv, "exclude"), z["nd32_" + conv])
    for conv in ("minmax2centroids", "centroids2minmax"):
        _same(bbu.convert_coordinates2(t, 3, conv), z["cc2_" + conv])
        _same(bbu.convert_coordinates2(t.astype(np.float32), 3, conv), z["cc2_32_" + conv])


def test_box_util_errors_and_device_tensors():
    import torch
    bbu, _, _ = _mods()
    a = np.zeros((3, 4))
    with pytest.raises(ValueError):
        bbu.convert_coordinates(a, 0, "corners2nothing")
    with pytest.raises(ValueError):
        bbu.iou(np.zeros((2, 3, 4)), a)
    with pytest.raises(ValueError):
        bbu.iou(np.zeros((3, 5)), a)
    with pytest.raises(ValueError):
        bbu.iou(a, a, mode="inner")
    with pytest.raises(ValueError):
        bbu.iou(a, a, coords="polar")
    with pytest.raises(ValueError):
        bbu.iou(a, np.zeros((2, 4)), coords="corners", mode="element-wise")
    with pytest.raises(ValueError):
        bbu.convert_coordinates2(a, 0, "corners2centroids")
    # CUDA tensors stay on the GPU
    rng = np.random.RandomState(0)
    p = rng.uniform(0, 1, size=(5, 4)); p[:, 2:] += p[:, :2]
    got = bbu.iou(torch.from_numpy(p).cuda(), torch.from_numpy(p).cuda(), "corners", "outer_product")
    assert torch.is_tensor(got) and got.is_cuda
    _same(got.cpu().numpy(), orc.iou(p, p, "corners", "outer_product"))
    # 0/0 -> NaN like NumPy (zero-area boxes that do not intersect)
    zbox = np.array([[1., 1, 1, 1], [2, 2, 2, 2]])
    with np.errstate(invalid="ignore", divide="ignore"):
        _same(bbu.iou(zbox, zbox, "corners", "outer_product"), orc.iou(zbox, zbox, "corners", "outer_product"))


def test_iou_at_encoder_scale_vs_oracle():
    """(g, 8732)- and (24564, 64)-shaped problems, all coords / borders: HIP == oracle bit for bit, plus symmetry and
    the unit diagonal as size-independent properties."""
    bbu, _, _ = _mods()
    rng = np.random.RandomState(5)
    for m, n in ((16, 8732), (24564, 64), (1, 1), (300, 300)):
        a = rng.uniform(0, 300, size=(m, 2)); c1 = np.concatenate([a, a + rng.uniform(1, 150, size=(m, 2))], axis=1)
        b = rng.uniform(0, 300, size=(n, 2)); c2 = np.concatenate([b, b + rng.uniform(1, 150, size=(n, 2))], axis=1)
        for coords, (p, q) in _views(c1, c2).items():
            for bp in ("half", "include"):
                _same(bbu.iou(p, q, coords, "outer_product", bp), orc.iou(p, q, coords, "outer_product", bp))
                _same(bbu.intersection_area(p, q, coords, "outer_product", bp), orc.intersection_area(p, q, coords, "outer_product", bp))
        if m == n:
            s = bbu.iou(c1, c1, "corners", "outer_product")
            assert np.array_equal(s, s.T) and np.all(np.diag(s) == 1.0)
            _same(bbu.iou(c1, c2, "corners", "element-wise"), orc.iou(c1, c2, "corners", "element-wise"))


def test_golden_matching():
    _, mu, _ = _mods()
    for name in ("box_utils", "box_utils2"):
        z = util.load(name)
        i = 0
        while "match_in_%d" % i in z:
            m = z["match_in_%d" % i]
            keep = m.copy()
            got = mu.match_bipartite_greedy(m)
            assert np.array_equal(m, keep)                                    # input not modified
            assert got.dtype == np.int64 and np.array_equal(got, z["match_bip_%d" % i]), (name, i)
            g, a = mu.match_multi(m, 0.5)
            assert np.array_equal(g, z["match_multi_gt_%d" % i]) and np.array_equal(a, z["match_multi_anchor_%d" % i]), (name, i)
            i += 1
        assert i >= 5


def test_matching_at_encoder_scale_vs_oracle():
    _, mu, _ = _mods()
    rng = np.random.RandomState(9)
    for m, n, q in ((8, 8732, None), (64, 24564, None), (40, 2000, 2), (300, 301, 1), (1, 7, None), (5, 5, None)):
        w = rng.uniform(0, 1, size=(m, n))
        w *= rng.uniform(0, 1, size=(m, n)) > 0.7                              # sparse, like IoU matrices
        if q is not None:
            w = np.round(w, q)                                                 # heavy ties
        assert np.array_equal(mu.match_bipartite_greedy(w), orc.match_bipartite_greedy(w)), (m, n, q)
        for thr in (0.5, 0.0, 0.99):
            g, a = mu.match_multi(w, thr)
            wg, wa = orc.match_multi(w, thr)
            assert np.array_equal(g, wg) and np.array_equal(a, wa), (m, n, q, thr)
    # property: a bipartite matching of a matrix with distinct positive entries is injective
    w = rng.uniform(0.1, 1, size=(50, 4000))
    got = mu.match_bipartite_greedy(w)
    assert len(set(got.tolist())) == 50


def test_golden_greedy_nms_family():
    _, _, dec = _mods()
    z = util.load("box_utils2")
    for coords in ("corners", "minmax", "centroids"):
        for bp in ("half", "include", "exclude"):
            pre = "nms_%s_%s_" % (coords, bp)
            items = util.unragged(z[pre + "in"], z[pre + "in_off"])
            want = util.unragged(z[pre + "out"], z[pre + "out_off"])
            got = dec.greedy_nms(items, iou_threshold=0.45, coords=coords, border_pixels=bp)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                if w.shape[0] == 0:
                    assert g.shape == (0,)
                else:
                    _same(g, w)
    _same(dec._greedy_nms(z["nms1_in"], iou_threshold=0.3, coords="corners", border_pixels="half"), z["nms1_out"])
    _same(dec._greedy_nms2(z["nms2_in"], iou_threshold=0.6, coords="corners", border_pixels="include"), z["nms2_out"])
    _same(dec._greedy_nms_debug(z["nms2_in"], iou_threshold=0.6, coords="corners", border_pixels="include"), z["nms2_out"])


def test_greedy_nms_at_decoder_scale_vs_oracle():
    """8732 candidate rows per image (one SSD300 image's worth), 4 images in one launch; plus idempotence."""
    _, _, dec = _mods()
    rng = np.random.RandomState(3)
    items = []
    for n in (8732, 2000, 1, 513):
        a = rng.uniform(0, 280, size=(n, 2))
        box = np.concatenate([a, a + rng.uniform(4, 120, size=(n, 2))], axis=1)
        items.append(np.concatenate([rng.randint(1, 21, size=(n, 1)).astype(np.float64), rng.uniform(0, 1, size=(n, 1)), box], axis=1))
    got = dec.greedy_nms(items, iou_threshold=0.45, coords="corners", border_pixels="half")
    want = orc.greedy_nms(items, 0.45, "corners", "half")
    for g, w in zip(got, want):
        _same(g, w)
    again = dec.greedy_nms(got, iou_threshold=0.45, coords="corners", border_pixels="half")
    for g, w in zip(again, got):
        _same(g, w)                                                            # NMS of an NMS result changes nothing


def test_golden_box_filter_and_image_validator():
    """BoxFilter / ImageValidator (ssdhip_box_filter) vs the real reference's outputs: 648 configurations (criteria x border modes x
    bounds x check flags, integer and float64 labels, degenerate and out-of-image boxes), single calls and one batched launch."""
    from ssd_keras_amd.data_generator.object_detection_2d_image_boxes_validation_utils import BoundGenerator, BoxFilter, ImageValidator
    from tests.test_oracle_golden import box_filter_cases
    cases = list(box_filter_cases())
    for cfg, lab, H, W, want, valid in cases[::7]:                            # every 7th configuration through the per-image API
        f = BoxFilter(check_overlap=cfg["flags"][0], check_min_area=cfg["flags"][1], check_degenerate=cfg["flags"][2],
                      overlap_criterion=cfg["crit"], overlap_bounds=cfg["bounds"], min_area=16, border_pixels=cfg["bp"])
        got = f(lab, H, W)
        assert got.dtype == lab.dtype and np.array_equal(got, lab[want]), cfg
        v = ImageValidator(overlap_criterion=cfg["crit"], bounds=cfg["bounds"], n_boxes_min=2, border_pixels=cfg["bp"])
        va = ImageValidator(overlap_criterion=cfg["crit"], bounds=cfg["bounds"], n_boxes_min="all", border_pixels=cfg["bp"])
        assert v(lab, H, W) == bool(valid[0]) and va(lab, H, W) == bool(valid[1]), cfg
    # all configurations: group by filter settings, one batched launch per group over the six label sets
    groups = {}
    for cfg, lab, H, W, want, valid in cases:
        groups.setdefault((cfg["crit"], cfg["bp"], cfg["bounds"], cfg["flags"]), []).append((lab, H, W, want))
    assert len(groups) == 108
    for (crit, bp, bounds, flags), items in groups.items():
        f = BoxFilter(check_overlap=flags[0], check_min_area=flags[1], check_degenerate=flags[2], overlap_criterion=crit,
                      overlap_bounds=bounds, min_area=16, border_pixels=bp)
        got = f.filter_batch([it[0].astype(np.float64) for it in items], [it[1] for it in items], [it[2] for it in items])
        for g, it in zip(got, items):
            assert np.array_equal(g, it[0].astype(np.float64)[it[3]]), (crit, bp, bounds, flags)
    with pytest.raises(ValueError):
        BoxFilter(overlap_criterion="corner")
    with pytest.raises(ValueError):
        BoxFilter(overlap_bounds=(0.8, 0.2))
    with pytest.raises(ValueError):
        ImageValidator(n_boxes_min=0)
    bg = BoundGenerator(sample_space=((0.1, None), (None, 0.9)), weights=[0.5, 0.5])
    assert bg() in ([0.1, 1.0], [0.0, 0.9])
    f = BoxFilter(overlap_criterion="iou", overlap_bounds=bg)                # bounds drawn per call
    assert f(cases[0][1], cases[0][2], cases[0][3]).shape[1] == 5
