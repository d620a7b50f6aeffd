"""TensorFlow's own published known-answer vectors (tests/tf_known_answers.py) against
  * the oracle's restatement of `tf.image.non_max_suppression` (`oracle/np_oracle._tf_nms`) and of the layer around it -- CPU;
  * the HIP path of the `DecodeDetections` layer (K3 / K4 with the POL_TF32 pair test / K5) -- `-m gpu`, through the C ABI.
The layer is fed predictions whose decoded boxes ARE the test's boxes (offsets 0, variances 1, anchors = the boxes as centroids,
absolute coordinates), one foreground class carrying the test's scores.  Reference call sites:
keras_layers/keras_layer_DecodeDetections.py:195-199 (NMS), :238-251 (top-k + padding)."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from tests import tf_known_answers as tfk


def _as_predictions(boxes_yxyx, scores, n_pad=0):
    """(1, n, 2 + 12) float32 predictions whose layer decode yields exactly `boxes_yxyx` (as xmin, ymin, xmax, ymax) for class 1.
    `n_pad` extra anchors score below any threshold used here (they keep N > 0 for the empty-input case)."""
    n = len(scores)
    y = np.zeros((1, n + n_pad, 14), dtype=np.float32)
    y[0, :, 0] = 1.0                                    # background column (ignored by the per-class loop)
    y[0, :, 1] = -1e30                                  # class 1: below every threshold unless set below
    y[0, :n, 1] = scores
    b = np.asarray(boxes_yxyx, dtype=np.float32).reshape(-1, 4)
    y1, x1, y2, x2 = (b[:, i] for i in range(4)) if n else (np.zeros(0, np.float32),) * 4
    y[0, :n, 6] = (x1 + x2) / np.float32(2)             # anchor cx, cy, w, h (w, h negative for flipped corners)
    y[0, :n, 7] = (y1 + y2) / np.float32(2)
    y[0, :n, 8] = x2 - x1
    y[0, :n, 9] = y2 - y1
    y[0, n:, 6:8] = 1e6                                  # padding anchors: far away, unit size
    y[0, n:, 8:10] = 1.0
    y[0, :, 10:14] = 1.0                                # variances
    return y


def _selected(out_rows, scores, boxes_yxyx):
    """Indices of the test's boxes that the layer output holds, in output order (matched on confidence, then on the box)."""
    rows = out_rows[~(out_rows == 0).all(axis=1)]
    sel = []
    b = np.asarray(boxes_yxyx, dtype=np.float32).reshape(-1, 4)
    for r in rows:
        assert r[0] == 1.0
        hits = [i for i in range(len(scores)) if np.float32(scores[i]) == r[1] and i not in sel
                and np.allclose([b[i, 1], b[i, 0], b[i, 3], b[i, 2]], r[2:6], atol=1e-5)]
        assert hits, "output row %s is none of the input boxes" % (r,)
        sel.append(hits[0])
    return sel


@pytest.mark.parametrize("i", range(len(tfk.NMS)))
def test_oracle_tf_nms_reproduces_tensorflows_op_tests(i):
    name, boxes, scores, thr, cap, expected = tfk.nms_case(i)
    got = orc._tf_nms(boxes, scores, thr, cap) if len(scores) else np.zeros(0, np.int64)
    assert np.array_equal(got, expected), "%s: %s != %s" % (name, got, expected)


@pytest.mark.parametrize("i", range(len(tfk.NMS)))
def test_oracle_layer_on_tensorflows_op_tests(i):
    """the layer restatement around `_tf_nms` (threshold, cap, top-k, padding) on the same vectors: the selected SET is the op's
    output and the rows come out by confidence descending."""
    name, boxes, scores, thr, cap, expected = tfk.nms_case(i)
    y = _as_predictions(boxes, scores, n_pad=2)
    out = orc.decode_detections_layer(y, confidence_thresh=-100.0, iou_threshold=thr, top_k=max(cap, 1), nms_max_output_size=max(cap, 1),
                                      normalize_coords=False, exp_mode="det")
    sel = _selected(out[0], scores, boxes)
    assert sel == list(expected), "%s: %s != %s" % (name, sel, list(expected))


@pytest.mark.parametrize("values,k,expected", tfk.TOP_K)
def test_oracle_layer_top_k_tie_order(values, k, expected):
    """tf.nn.top_k: equal confidences keep their order in the padded class-major array -- here index order, the boxes are disjoint."""
    boxes = [[0, 10 * j, 1, 10 * j + 1] for j in range(len(values))]
    y = _as_predictions(boxes, values)
    out = orc.decode_detections_layer(y, confidence_thresh=0.01, iou_threshold=0.5, top_k=k, nms_max_output_size=len(values),
                                      normalize_coords=False, exp_mode="det")
    assert _selected(out[0], values, boxes) == expected


# ---- the HIP path ---------------------------------------------------------------------------------------------------------------------
def _layer(**kw):
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from ssd_keras_amd.keras_layers.keras_layer_DecodeDetections import DecodeDetections
    return torch, DecodeDetections(**kw)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(tfk.NMS)))
def test_hip_layer_reproduces_tensorflows_op_tests(i):
    name, boxes, scores, thr, cap, expected = tfk.nms_case(i)
    y = _as_predictions(boxes, scores, n_pad=2)
    kw = dict(confidence_thresh=-100.0, iou_threshold=thr, top_k=max(cap, 1), nms_max_output_size=max(cap, 1), normalize_coords=False)
    torch, layer = _layer(**kw)
    out = layer(torch.from_numpy(y).cuda()).cpu().numpy()
    sel = _selected(out[0], scores, boxes)
    assert sel == list(expected), "%s: %s != %s" % (name, sel, list(expected))
    assert np.array_equal(out, orc.decode_detections_layer(y, exp_mode="det", **kw)), name


@pytest.mark.gpu
@pytest.mark.parametrize("values,k,expected", tfk.TOP_K)
def test_hip_layer_top_k_tie_order(values, k, expected):
    boxes = [[0, 10 * j, 1, 10 * j + 1] for j in range(len(values))]
    y = _as_predictions(boxes, values)
    torch, layer = _layer(confidence_thresh=0.01, iou_threshold=0.5, top_k=k, nms_max_output_size=len(values), normalize_coords=False)
    out = layer(torch.from_numpy(y).cuda()).cpu().numpy()
    assert _selected(out[0], values, boxes) == expected
