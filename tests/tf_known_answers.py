"""Known-answer vectors PUBLISHED BY TENSORFLOW ITSELF for the two TF ops the in-graph decoder layers rest on
(keras_layers/keras_layer_DecodeDetections.py:195-199 `tf.image.non_max_suppression`, :238-251 `tf.nn.top_k`).

TensorFlow is an un-vendored, un-pinned dependency of the reference (README.md:143-150: "TensorFlow 1.x") and cannot be installed
here, so the restatement `oracle/np_oracle._tf_nms` and the HIP pair-test policy POL_TF32 had only each other as witnesses
(missing #3 / weak #1).  The cases below are the op's own unit tests -- tensorflow/core/kernels/non_max_suppression_op_test.cc,
class NonMaxSuppressionOpTest (TF 1.x; the V2 / V3 classes repeat them with the threshold as an input tensor) -- transcribed as data:
boxes are `[y1, x1, y2, x2]` rows, `expected` is the op's output (selected indices in selection order).

`TOP_K` holds the documented contract of tf.nn.top_k ("If two elements are equal, the lower-index element appears first",
tensorflow/python/ops/nn_ops.py top_k docstring) as small cases.

Test infrastructure only: imported by tests/test_tf_known_answers.py.
"""
import numpy as np

_CLUSTERS = [[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]]
_FLIPPED = [[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, 0.9, 1, -0.1], [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]]
_SCORES = [0.9, 0.75, 0.6, 0.95, 0.5, 0.3]

NMS = [
    # (name in non_max_suppression_op_test.cc, boxes, scores, iou_threshold, max_output_size, expected)
    ("TestSelectFromThreeClusters", _CLUSTERS, _SCORES, 0.5, 3, [3, 0, 5]),
    ("TestSelectFromThreeClustersFlippedCoordinates", _FLIPPED, _SCORES, 0.5, 3, [3, 0, 5]),
    ("TestSelectAtMostTwoBoxesFromThreeClusters", _CLUSTERS, _SCORES, 0.5, 2, [3, 0]),
    ("TestSelectWithNegativeScores", _CLUSTERS, [s - 10.0 for s in _SCORES], 0.5, 6, [3, 0, 5]),
    ("TestSelectAtMostThirtyBoxesFromThreeClusters", _CLUSTERS, _SCORES, 0.5, 30, [3, 0, 5]),
    ("TestSelectSingleBox", [[0, 0, 1, 1]], [0.9], 0.5, 3, [0]),
    ("TestSelectFromTenIdenticalBoxes", [[0, 0, 1, 1]] * 10, [0.9] * 10, 0.5, 3, [0]),
    ("TestEmptyInput", [], [], 0.5, 30, []),
]


def nms_case(i):
    name, boxes, scores, thr, cap, expected = NMS[i]
    return (name, np.asarray(boxes, dtype=np.float32).reshape(-1, 4), np.asarray(scores, dtype=np.float32), thr, cap,
            np.asarray(expected, dtype=np.int64))


TOP_K = [
    # (values, k, expected indices): sorted descending, equal values in index order
    ([1.0, 3.0, 3.0, 2.0, 3.0], 3, [1, 2, 4]),
    ([0.5, 0.5, 0.5, 0.5], 2, [0, 1]),
    ([0.1, 0.9, 0.9, 0.1, 0.5], 4, [1, 2, 4, 0]),
]
