"""Pin `oracle/np_oracle.py` against fixtures produced by the REAL reference
(`tests/golden/make_golden.py`).  CPU only.  Bit-exact unless stated otherwise."""
import ast

import numpy as np
import pytest

from oracle import np_oracle as orc
from ssd_keras_amd import synthetic as syn
from tests import util


def _encoder(cfg, **over):
    kw = dict(cfg)
    kw.update(over)
    return orc.EncoderOracle(**kw)


def test_convert_coordinates_and_iou():
    z = util.load("box_utils")
    c1, c2 = z["corners1"], z["corners2"]
    for conv in ("minmax2centroids", "centroids2minmax", "corners2centroids", "centroids2corners",
                 "minmax2corners", "corners2minmax"):
        for bp in ("half", "include", "exclude"):
            assert np.array_equal(orc.convert_coordinates(c1, 0, conv, bp), z["cc_%s_%s" % (conv, bp)])
        got = orc.convert_coordinates(c1.astype(np.float32), 0, conv, "half")
        assert got.dtype == np.float64 and np.array_equal(got, z["cc32_%s" % conv])
    for coords in ("corners", "minmax", "centroids"):
        if coords == "corners":
            p, q = c1, c2
        elif coords == "minmax":
            p, q = c1[:, [0, 2, 1, 3]], c2[:, [0, 2, 1, 3]]
        else:
            p, q = orc.convert_coordinates(c1, 0, "corners2centroids"), orc.convert_coordinates(c2, 0, "corners2centroids")
        for bp in ("half", "include", "exclude"):
            assert np.array_equal(orc.iou(p, q, coords, "outer_product", bp), z["iou_outer_%s_%s" % (coords, bp)])
            assert np.array_equal(orc.iou(p, q[:7], coords, "element-wise", bp), z["iou_elem_%s_%s" % (coords, bp)])
            assert np.array_equal(orc.iou(p, q[3], coords, "element-wise", bp), z["iou_bcast_%s_%s" % (coords, bp)])


def test_iou_border_pixel_quirk():
    # only the union sees border_pixels
    a, b = np.array([0., 0, 10, 10]), np.array([5., 5, 15, 15])
    assert abs(orc.iou(a, b, "corners", "element-wise", "half")[0] - 0.142857) < 1e-6
    assert abs(orc.iou(a, b, "corners", "element-wise", "include")[0] - 0.115207) < 1e-6
    assert abs(orc.iou(a, b, "corners", "element-wise", "exclude")[0] - 0.182482) < 1e-6


def test_matching():
    z = util.load("box_utils")
    i = 0
    while "match_in_%d" % i in z:
        m = z["match_in_%d" % i]
        assert np.array_equal(orc.match_bipartite_greedy(m), z["match_bip_%d" % i])
        g, a = orc.match_multi(m, 0.5)
        assert np.array_equal(g, z["match_multi_gt_%d" % i]) and np.array_equal(a, z["match_multi_anchor_%d" % i])
        i += 1
    assert i >= 5
    # the two quirks spelled out
    assert list(orc.match_bipartite_greedy(np.array([[0, 0, 0, 0], [0, .6, 0, 0], [0, 0, 0, 0.]]))) == [0, 1, 0]
    assert list(orc.match_bipartite_greedy(np.array([[.5, .6], [0, 0.]]))) == [0, 0]


def test_anchors():
    z = util.load("anchors")
    n = 0
    for key in z.files:
        if key == "tiny_abs_steps":
            enc = _encoder(syn.TINY, normalize_coords=False, steps=[8, (16, 20), 32, 64],
                           offsets=[0.5, (0.3, 0.7), 0.5, 0.5])
        else:
            name, coords, clip = key.split("_")
            enc = _encoder(util.CFGS[name], coords=coords, clip_boxes=clip == "clip1")
        assert np.array_equal(enc.anchors(), z[key]), key
        n += 1
    assert n >= 10
    assert _encoder(syn.SSD300_VOC).anchors().shape == (8732, 4)
    assert _encoder(syn.SSD512_COCO).anchors().shape == (24564, 4)
    assert _encoder(syn.SSD7_300).anchors().shape == (7160, 4)


def test_encoder():
    z = util.load("encoder")
    n_cases = int(z["n_cases"])
    assert n_cases >= 20
    for ci in range(n_cases):
        pre = "c%02d_" % ci
        case = ast.literal_eval(str(z[pre + "params"]))
        cfg = util.CFGS[case["cfg"]]
        over = {k: v for k, v in case.items() if k not in ("cfg", "seed", "B", "max_boxes", "min_boxes")}
        enc = _encoder(cfg, **over)
        gt = util.unragged(z[pre + "gt"], z[pre + "gt_off"])
        with np.errstate(invalid="ignore", divide="ignore"):
            y, y_diag, mm = enc(gt, diagnostics=True, return_matches=True)
        idx, rows = z[pre + "idx"], z[pre + "rows"]
        # every row the reference changed is identical ...
        assert np.array_equal(y[idx[:, 0], idx[:, 1], :], rows, equal_nan=True), pre
        # ... and every other row is still the all-background template (checksum + direct test)
        mask = np.ones(y.shape[:2], bool)
        mask[idx[:, 0], idx[:, 1]] = False
        tmpl = enc.generate_encoding_template(len(gt))
        tmpl[:, :, enc.background_id] = 1
        tmpl[:, :, -12:-8] = 0
        assert np.array_equal(y[mask], tmpl[mask]), pre
        assert np.array_equal(y.reshape(len(gt), -1).sum(axis=1), z[pre + "sums"], equal_nan=True), pre
        assert np.array_equal(y_diag.reshape(len(gt), -1).sum(axis=1), z[pre + "diag_sums"], equal_nan=True), pre
        # match map is consistent with the encoded classes
        C = enc.n_classes
        cls_sum = y[:, :, :C].sum(axis=-1)
        assert np.array_equal(mm == -2, (cls_sum == 0)), pre
        if enc.background_id == 0:      # (a GT class equal to a non-zero background_id re-sets that slot)
            assert np.array_equal(mm >= 0, (cls_sum == 1) & (y[:, :, 0] == 0)), pre


def test_encoder_degenerate_box_raises():
    enc = _encoder(syn.TINY)
    with pytest.raises(orc.DegenerateBoxError):
        enc([np.array([[1, 10, 10, 10, 20.]])])


def _y_pred_for(z, name):
    if name.startswith("ssd7") or name.startswith("ssd300"):
        cfg_name = "ssd7" if name.startswith("ssd7") else "ssd300"
        key = "ssd7_dense_conf_loc" if cfg_name == "ssd7" else "ssd300_sparse_conf_loc"
        enc = _encoder(util.CFGS[cfg_name])
        cl = z[key]
        av = enc.generate_encoding_template(1)[0, :, -8:]
        y = np.empty(cl.shape[:2] + (cl.shape[2] + 8,), dtype=cl.dtype)
        y[:, :, :-8] = cl
        y[:, :, -8:] = av
        return y
    if name + "_y_pred" in z:
        return z[name + "_y_pred"]
    return z["_".join(name.split("_")[:4]) + "_y_pred"]


def test_decoder():
    z = util.load("decoder")
    strict = util.local_exp_matches_golden()
    fns = dict(decode_detections=orc.decode_detections, decode_detections_fast=orc.decode_detections_fast)
    n = 0
    for name in [str(s) for s in z["cases"]]:
        kw = util.kw_of(z, name)
        fn_name = str(z[name + "_fn"])
        y = _y_pred_for(z, name)
        want = util.unragged(z[name + "_out"], z[name + "_off"])
        if fn_name == "decode_detections_debug":
            got = orc.decode_detections(y, with_anchor_index=True, decode_order="debug", **kw)
            n_meta = 3
        else:
            got = fns[fn_name](y, **kw)
            n_meta = 2
        for g, w in zip(got, want):                      # container quirks: empty -> shape (0,)
            if w.shape[0] == 0 and fn_name != "decode_detections_fast":
                assert g.shape == (0,), name
        exact = strict or y.dtype == np.float64 or kw["input_coords"] != "centroids"
        util.dets_equal(got, want, exact=exact, rtol=1e-6, atol=1e-6, n_meta=n_meta)
        n += 1
    assert n >= 100


def test_decoder_det_exp_selection_matches_reference():
    """With the deterministic exp the *selection* (anchor, class, conf) still equals the reference's on every
    golden case and boxes move by < 1e-5 px: no fixture sits on an NMS knife edge."""
    z = util.load("decoder")
    for name in [str(s) for s in z["cases"]]:
        kw = util.kw_of(z, name)
        if str(z[name + "_fn"]) != "decode_detections" or kw["input_coords"] != "centroids":
            continue
        y = _y_pred_for(z, name)
        if y.dtype != np.float32:
            continue
        got = orc.decode_detections(y, exp_mode="det", **kw)
        want = util.unragged(z[name + "_out"], z[name + "_off"])
        util.dets_equal(got, want, exact=False, rtol=1e-6, atol=1e-5)


def test_det_expf_accuracy():
    x = np.concatenate([np.linspace(-104, 89, 200001), np.linspace(-2, 2, 200001)]).astype(np.float32)
    got = orc.det_expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    ok = np.isfinite(ref) & (ref > 1e-37) & (ref < 3e38)
    ulp = np.spacing(ref[ok].astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got[ok] - ref[ok]) / ulp) <= 0.5000001
    assert orc.det_expf(np.float32(0.0)) == 1.0
    assert np.isnan(orc.det_expf(np.array([np.nan], np.float32))[0])
    assert orc.det_expf(np.array([200.0], np.float32))[0] == np.inf
    assert orc.det_expf(np.array([-200.0], np.float32))[0] == 0.0


def test_det_exp64_accuracy_and_float64_flow():
    """The float64 flow's exp: <= 1 ulp of exp(x) (np.exp(float64) is not correctly rounded either), exact at the ends; with it
    the oracle's float64 decodes keep the reference's selection on every float64 golden case, boxes within 1e-9 px."""
    x = np.concatenate([np.linspace(-745, 709, 400001), np.linspace(-2, 2, 200001)])
    got, ref = orc.det_exp64(x), np.exp(x)
    ok = np.isfinite(ref) & (ref > 1e-300)
    assert np.max(np.abs(got[ok] - ref[ok]) / np.spacing(ref[ok])) <= 1.0
    assert orc.det_exp64(0.0) == 1.0 and np.isnan(orc.det_exp64(np.nan))
    assert orc.det_exp64(800.0) == np.inf and orc.det_exp64(-800.0) == 0.0
    z = util.load("decoder")
    n = 0
    for name in [str(s) for s in z["cases"]]:
        kw = util.kw_of(z, name)
        y = _y_pred_for(z, name)
        if y.dtype != np.float64:
            continue
        fn = orc.decode_detections_fast if str(z[name + "_fn"]) == "decode_detections_fast" else orc.decode_detections
        got = fn(y, exp_mode="det", **kw)
        want = util.unragged(z[name + "_out"], z[name + "_off"])
        util.dets_equal(got, want, exact=False, rtol=1e-12, atol=1e-9)
        n += 1
    assert n >= 40


def test_greedy_nms_public():
    z = util.load("decoder")
    items = [it for it in util.unragged(z["gnms_in"], z["gnms_in_off"]) if it.shape[0] > 0]
    for bp in ("half", "include"):
        got = orc.greedy_nms(items, 0.45, "corners", bp)
        want = util.unragged(z["gnms_out_" + bp], z["gnms_out_off_" + bp])
        for g, w in zip(got, want):
            assert np.array_equal(g, w)


def _views(c1, c2):
    return {"corners": (c1, c2), "minmax": (c1[:, [0, 2, 1, 3]], c2[:, [0, 2, 1, 3]]),
            "centroids": (orc.convert_coordinates(c1, 0, "corners2centroids"), orc.convert_coordinates(c2, 0, "corners2centroids"))}


def test_box_utils2_overlap_and_dtypes():
    z = util.load("box_utils2")
    for coords, (p, q) in _views(z["corners1"], z["corners2"]).items():
        p32, q32 = p.astype(np.float32), q.astype(np.float32)
        for bp in ("half", "include", "exclude"):
            tag = "%s_%s" % (coords, bp)
            assert np.array_equal(orc.intersection_area(p, q, coords, "outer_product", bp), z["ia_outer_" + tag])
            assert np.array_equal(orc.intersection_area(p, q[:13], coords, "element-wise", bp), z["ia_elem_" + tag])
            r = orc.iou(p32, q32, coords, "outer_product", bp)
            assert r.dtype.name == str(z["iou32_outer_%s_dtype" % tag]) and np.array_equal(r, z["iou32_outer_" + tag])
            assert np.array_equal(orc.iou(p32, q, coords, "outer_product", bp), z["iou3264_outer_" + tag])
            assert np.array_equal(orc.iou(p, q32[:13], coords, "element-wise", bp), z["iou6432_elem_" + tag])
            r = orc.intersection_area(p32, q32, coords, "outer_product", bp)
            assert r.dtype == z["ia32_outer_" + tag].dtype and np.array_equal(r, z["ia32_outer_" + tag])


def test_box_utils2_convert_nd():
    z = util.load("box_utils2")
    t = z["nd_in"]
    for conv in ("minmax2centroids", "centroids2minmax", "corners2centroids", "centroids2corners", "minmax2corners", "corners2minmax"):
        assert np.array_equal(orc.convert_coordinates(t, 3, conv, "include"), z["nd_" + conv])
        assert np.array_equal(orc.convert_coordinates(t.astype(np.float32), 3, conv, "exclude"), z["nd32_" + conv])
    for conv in ("minmax2centroids", "centroids2minmax"):
        assert np.array_equal(orc.convert_coordinates2(t, 3, conv), z["cc2_" + conv])
        assert np.array_equal(orc.convert_coordinates2(t.astype(np.float32), 3, conv), z["cc2_32_" + conv])


def test_box_utils2_greedy_nms_family():
    z = util.load("box_utils2")
    for coords in ("corners", "minmax", "centroids"):
        for bp in ("half", "include", "exclude"):
            pre = "nms_%s_%s_" % (coords, bp)
            items = util.unragged(z[pre + "in"], z[pre + "in_off"])
            want = util.unragged(z[pre + "out"], z[pre + "out_off"])
            got = orc.greedy_nms(items, 0.45, coords, bp)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert np.array_equal(g.reshape(-1, 6), w.reshape(-1, 6))
    assert np.array_equal(orc.greedy_nms_single(z["nms1_in"], 0.3, "corners", "half"), z["nms1_out"])
    assert np.array_equal(orc.greedy_nms_single2(z["nms2_in"], 0.6, "corners", "include"), z["nms2_out"])


def test_box_utils2_matching():
    z = util.load("box_utils2")
    i = 0
    while "match_in_%d" % i in z:
        m = z["match_in_%d" % i]
        assert np.array_equal(orc.match_bipartite_greedy(m), z["match_bip_%d" % i]), i
        g, a = orc.match_multi(m, 0.5)
        assert np.array_equal(g, z["match_multi_gt_%d" % i]) and np.array_equal(a, z["match_multi_anchor_%d" % i])
        i += 1
    assert i == 6


def _eval_case(z, ci):
    pre = "e%d_" % ci
    labels = [a.astype(np.int64) for a in util.unragged(z[pre + "labels"], z[pre + "labels_off"])]
    neutral = None
    if int(z[pre + "has_neutral"]):
        flat, off = z[pre + "neutral"].astype(bool), z[pre + "labels_off"]
        neutral = [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    image_ids = [str(s) for s in z[pre + "image_ids"]]
    preds = [[]]
    for c in range(1, 5):
        img, rows = z[pre + "c%d_pred_img" % c], z[pre + "c%d_pred" % c]
        preds.append([(str(img[k]),) + tuple(float(v) for v in rows[k]) for k in range(rows.shape[0])])
    return pre, labels, neutral, image_ids, preds, ast.literal_eval(str(z[pre + "params"]))


def test_evaluator_matching_and_average_precision():
    z = util.load("evaluator")
    for ci in range(int(z["n_cases"])):
        pre, labels, neutral, image_ids, preds, case = _eval_case(z, ci)
        num_gt = orc.evaluator_num_gt_per_class(labels, neutral, 4, 0, case["ignore"])
        assert np.array_equal(num_gt, z[pre + "num_gt"])
        tp, fp, ctp, cfp = orc.evaluator_match_predictions(preds, labels, image_ids, neutral, 4, ignore_neutral_boxes=case["ignore"],
                                                           matching_iou_threshold=case["thr"], border_pixels=case["bp"])
        for c in range(1, 5):
            assert np.array_equal(tp[c], z[pre + "c%d_tp" % c]) and np.array_equal(fp[c], z[pre + "c%d_fp" % c]), (ci, c)
            if len(preds[c]):
                assert np.array_equal(ctp[c], z[pre + "c%d_ctp" % c]) and np.array_equal(cfp[c], z[pre + "c%d_cfp" % c])
        if case.get("empty_last_class"):
            continue
        prec, rec = orc.evaluator_precision_recall(ctp, cfp, num_gt)
        for c in range(1, 5):
            assert np.array_equal(prec[c], z[pre + "c%d_prec" % c]) and np.array_equal(rec[c], z[pre + "c%d_rec" % c])
        for mode in ("sample", "integrate"):
            ap = np.asarray(orc.evaluator_average_precisions(prec, rec, mode=mode), dtype=np.float64)
            assert np.array_equal(ap, z[pre + "ap_" + mode]), (ci, mode)
            assert np.mean(ap[1:]) == float(z[pre + "map_" + mode])


def box_filter_cases():
    z = util.load("box_filter")
    for i, c in enumerate(z["cfgs"]):
        cfg = ast.literal_eval(str(c))
        lab, hw = z["L%d" % cfg["set"]], z["L%d_hw" % cfg["set"]]
        want = z["kept_idx"][z["kept_off"][i]:z["kept_off"][i + 1]]
        yield cfg, lab, int(hw[0]), int(hw[1]), want, z["valid"][i]


def test_box_filter_oracle_vs_reference():
    n = 0
    for cfg, lab, H, W, want, valid in box_filter_cases():
        m = orc.box_filter_mask(lab, H, W, check_overlap=cfg["flags"][0], check_min_area=cfg["flags"][1], check_degenerate=cfg["flags"][2],
                                overlap_criterion=cfg["crit"], overlap_bounds=cfg["bounds"], min_area=16, border_pixels=cfg["bp"])
        assert np.array_equal(np.nonzero(m)[0], want), cfg
        mv = orc.box_filter_mask(lab, H, W, True, False, False, cfg["crit"], cfg["bounds"], 16, cfg["bp"])
        assert bool(mv.sum() >= 2) == bool(valid[0]) and bool(mv.sum() == len(mv)) == bool(valid[1])
        n += 1
    assert n == 648
