"""Evaluator (eval_utils/average_precision_evaluator.py) with the matching step on the GPU (ssdhip_match_predictions, through
the C ABI) vs the golden outputs of the REAL reference class and, at VOC scale, vs the oracle.  Needs an MI355X.
Bar: true / false positive flags, their cumulative sums, precisions, recalls, average precisions and mAP bit exact."""

import numpy as np
import pytest

from oracle import np_oracle as orc
from tests import util
from tests.test_oracle_golden import _eval_case

pytestmark = pytest.mark.gpu


class _Gen:
    def __init__(self, labels, neutral, image_ids):
        self.labels, self.eval_neutral, self.image_ids = labels, neutral, image_ids


def _evaluator(labels, neutral, image_ids, preds, n_classes=4):
    from ssd_keras_amd.eval_utils.average_precision_evaluator import Evaluator
    ev = Evaluator(model=None, n_classes=n_classes, data_generator=_Gen(labels, neutral, image_ids))
    ev.prediction_results = preds
    return ev


def test_golden_evaluator_cases():
    z = util.load("evaluator")
    for ci in range(int(z["n_cases"])):
        pre, labels, neutral, image_ids, preds, case = _eval_case(z, ci)
        ev = _evaluator(labels, neutral, image_ids, preds)
        num_gt = ev.get_num_gt_per_class(ignore_neutral_boxes=case["ignore"], verbose=False, ret=True)
        assert np.array_equal(num_gt, z[pre + "num_gt"])
        tp, fp, ctp, cfp = ev.match_predictions(ignore_neutral_boxes=case["ignore"], matching_iou_threshold=case["thr"],
                                                border_pixels=case["bp"], verbose=False, ret=True)
        for c in range(1, 5):
            assert np.array_equal(tp[c], z[pre + "c%d_tp" % c]) and np.array_equal(fp[c], z[pre + "c%d_fp" % c]), (ci, c)
            if len(preds[c]):
                assert np.array_equal(ctp[c], z[pre + "c%d_ctp" % c]) and np.array_equal(cfp[c], z[pre + "c%d_cfp" % c])
        prec, rec = ev.compute_precision_recall(verbose=False, ret=True)
        if case.get("empty_last_class"):
            # the reference cannot get past this point (IndexError); ours: empty arrays, AP 0 for the empty class
            assert len(prec[4]) == 0
            ap = ev.compute_average_precisions(mode="sample", verbose=False, ret=True)
            assert ap[4] == 0.0 and np.isfinite(ev.compute_mean_average_precision())
            continue
        for c in range(1, 5):
            assert np.array_equal(prec[c], z[pre + "c%d_prec" % c]) and np.array_equal(rec[c], z[pre + "c%d_rec" % c])
        for mode in ("sample", "integrate"):
            ap = np.asarray(ev.compute_average_precisions(mode=mode, verbose=False, ret=True), dtype=np.float64)
            assert np.array_equal(ap, z[pre + "ap_" + mode]), (ci, mode)
            assert ev.compute_mean_average_precision() == float(z[pre + "map_" + mode])


def test_matching_at_voc_scale_vs_oracle():
    """~4952 images (VOC2007 test size), 20 classes, ~60 k predictions: HIP == oracle; plus counting properties."""
    rng = np.random.RandomState(0)
    n_images, n_classes = 1200, 6
    labels, neutral, image_ids = [], [], []
    preds = [[] for _ in range(n_classes + 1)]
    for i in range(n_images):
        image_ids.append("%06d" % i)
        g = int(rng.randint(0, 6))
        cls = rng.randint(1, n_classes + 1, size=g)
        x0, y0 = rng.randint(0, 400, size=g), rng.randint(0, 300, size=g)
        lab = np.stack([cls, x0, y0, x0 + rng.randint(8, 120, size=g), y0 + rng.randint(8, 120, size=g)], axis=1).astype(np.int64).reshape(-1, 5)
        labels.append(lab)
        neutral.append(rng.uniform(size=g) < 0.15)
        for b in lab:
            for _ in range(int(rng.randint(0, 5))):
                j = rng.normal(0, 5, size=4)
                preds[int(b[0])].append((image_ids[-1], float(rng.uniform(0.01, 1)), float(b[1] + j[0]), float(b[2] + j[1]), float(b[3] + j[2]),
                                         float(b[4] + j[3])))
        for _ in range(int(rng.randint(0, 12))):
            c = int(rng.randint(1, n_classes + 1))
            x, y = rng.uniform(0, 450, size=2)
            preds[c].append((image_ids[-1], float(np.round(rng.uniform(0.01, 0.5), 3)), float(x), float(y), float(x + rng.uniform(4, 90)),
                             float(y + rng.uniform(4, 90))))
    ev = _evaluator(labels, neutral, image_ids, preds, n_classes)
    tp, fp, ctp, cfp = ev.match_predictions(matching_iou_threshold=0.5, border_pixels="include", verbose=False, ret=True)
    wtp, wfp, wctp, wcfp = orc.evaluator_match_predictions(preds, labels, image_ids, neutral, n_classes)
    num_gt = ev.get_num_gt_per_class(verbose=False, ret=True)
    for c in range(1, n_classes + 1):
        assert np.array_equal(tp[c], wtp[c]) and np.array_equal(fp[c], wfp[c]), c
        assert np.array_equal(ctp[c], wctp[c]) and np.array_equal(cfp[c], wcfp[c])
        assert tp[c].sum() <= num_gt[c]                                   # every box is claimed at most once
        assert np.all(tp[c] + fp[c] <= 1) and len(tp[c]) == len(preds[c])
    ev.compute_precision_recall(verbose=False)
    ev.compute_average_precisions(verbose=False)
    m = ev.compute_mean_average_precision()
    assert 0.0 < m < 1.0


def test_evaluator_errors():
    from ssd_keras_amd.eval_utils.average_precision_evaluator import Evaluator
    ev = Evaluator(model=None, n_classes=2, data_generator=_Gen([np.zeros((0, 5), dtype=np.int64)], None, ["0"]))
    with pytest.raises(ValueError):
        ev.match_predictions()
    with pytest.raises(ValueError):
        ev.compute_precision_recall()
    with pytest.raises(ValueError):
        ev.compute_average_precisions()
    with pytest.raises(ValueError):
        ev.compute_mean_average_precision()
    ev.prediction_results = [[], [], []]
    ev.match_predictions(verbose=False)
    ev.get_num_gt_per_class(verbose=False)
    ev.compute_precision_recall(verbose=False)
    with pytest.raises(ValueError):
        ev.compute_average_precisions(mode="trapezoid")


def test_packed_inputs_are_reused_and_follow_their_sources():
    """Round 6: all classes in one call, predictions / ground truth packed once per object.  A second evaluation at another threshold
    and border mode reuses the device copies and still equals the oracle; new result lists (or forget_packed_inputs after an in-place
    edit) are re-packed."""
    rng = np.random.RandomState(5)
    n_images, n_classes = 300, 5
    labels, neutral, image_ids = [], [], []
    preds = [[] for _ in range(n_classes + 1)]
    for i in range(n_images):
        image_ids.append("%06d" % i)
        g = int(rng.randint(0, 5))
        cls = rng.randint(1, n_classes + 1, size=g)
        x0, y0 = rng.randint(0, 300, size=g), rng.randint(0, 200, size=g)
        lab = np.stack([cls, x0, y0, x0 + rng.randint(8, 90, size=g), y0 + rng.randint(8, 90, size=g)], axis=1).astype(np.int64)
        labels.append(lab)
        neutral.append(rng.uniform(size=g) < 0.2)
        for b in lab:
            for _ in range(int(rng.randint(0, 3))):
                j = rng.normal(0, 4, size=4)
                preds[int(b[0])].append((image_ids[-1], float(np.round(rng.uniform(0.05, 1), 2)), float(b[1] + j[0]), float(b[2] + j[1]),
                                         float(b[3] + j[2]), float(b[4] + j[3])))
        for _ in range(int(rng.randint(0, 6))):
            c = int(rng.randint(1, n_classes))                                # the last class gets no random false positives
            x, y = rng.uniform(0, 300, size=2)
            preds[c].append((image_ids[-1], float(np.round(rng.uniform(0.05, 0.5), 2)), float(x), float(y), float(x + 30), float(y + 40)))
    ev = _evaluator(labels, neutral, image_ids, preds, n_classes=n_classes)
    for thr, bp, ignore in ((0.5, "include", True), (0.3, "half", True), (0.5, "exclude", False)):
        got = ev.match_predictions(ignore_neutral_boxes=ignore, matching_iou_threshold=thr, border_pixels=bp, verbose=False, ret=True)
        want = orc.evaluator_match_predictions(preds, labels, image_ids, neutral, n_classes, ignore_neutral_boxes=ignore,
                                               matching_iou_threshold=thr, border_pixels=bp)
        for g, w in zip(got, want):
            for c in range(1, n_classes + 1):
                assert np.array_equal(g[c], w[c]), (thr, bp, c)
    first = ev.__dict__["_packed_pred_memo"]
    ev.match_predictions(verbose=False)
    assert ev.__dict__["_packed_pred_memo"] is first                           # same results object: no re-packing
    # an in-place edit that keeps every length: invisible to the cache until told
    preds[1][0] = (preds[1][0][0], 0.999) + tuple(preds[1][0][2:])
    ev.forget_packed_inputs()
    got = ev.match_predictions(verbose=False, ret=True)
    want = orc.evaluator_match_predictions(preds, labels, image_ids, neutral, n_classes)
    assert all(np.array_equal(got[0][c], want[0][c]) and np.array_equal(got[2][c], want[2][c]) for c in range(1, n_classes + 1))
    # a new results object is packed again by itself
    ev.prediction_results = [list(rows) for rows in preds]
    ev.match_predictions(verbose=False)
    assert ev.__dict__["_packed_pred_memo"] is not first
