"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under ``ssd_keras_amd/`` may import this module.

A CPU (NumPy) restatement of the per-anchor hot path of pierluigiferrari/ssd_keras:
box math, anchor generation, GT->target encoding (IoU + bipartite/multi matching),
the SSD loss, and prediction decoding (decode + threshold + greedy NMS + top-k).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it -- as the checker / the timed CPU port, never as the product.

Every function cites the reference lines it restates (paths relative to the
reference repo root).  The restatement keeps the reference's *arithmetic*:
dtype flow, operation order, comparison operators, tie breaking and its quirks
(see SURVEY.md Appendix A).  It does not keep its *code*: loops, temporaries and
names are our own.

Pinning: ``tests/golden/make_golden.py`` imports the real reference (NumPy half)
in the build container and stores its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this module against them bit for bit.
The TensorFlow half (SSDLoss, the DecodeDetections layers, L2Normalization) has
no runnable reference here: **parity unpinned** for those, restated from source.

``exp_mode``: the reference decodes widths/heights with ``np.exp`` on float32, which
is not correctly rounded (measured: <=2.33 ulp, 39 % of inputs off by >=1 ulp on
AVX-512 hosts) and therefore not reproducible on another machine.  ``exp_mode='numpy'``
keeps that call (used to pin this file against the reference); ``exp_mode='det'``
uses `det_expf`, a fixed sequence of IEEE-754 double operations that the HIP kernels
and ``oracle/ssd_oracle.c`` repeat verbatim, so GPU-vs-oracle comparisons are bit exact.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# deterministic float32 exp (shared algorithm: csrc/ssdhip_math.h, oracle/ssd_oracle.c)
# --------------------------------------------------------------------------------------
_LOG2E = float.fromhex("0x1.71547652b82fep+0")
_LN2_HI = float.fromhex("0x1.62e42fee00000p-1")
_LN2_LO = float.fromhex("0x1.a39ef35793c76p-33")
_RND = float.fromhex("0x1.8p52")
# 1/n!, n = 0..13, each the double nearest to the exact value
_EXP_C = [float.fromhex(h) for h in (
    "0x1.0000000000000p+0", "0x1.0000000000000p+0", "0x1.0000000000000p-1",
    "0x1.5555555555555p-3", "0x1.5555555555555p-5", "0x1.1111111111111p-7",
    "0x1.6c16c16c16c17p-10", "0x1.a01a01a01a01ap-13", "0x1.a01a01a01a01ap-16",
    "0x1.71de3a556c734p-19", "0x1.27e4fb7789f5cp-22", "0x1.ae64567f544e4p-26",
    "0x1.1eed8eff8d898p-29", "0x1.6124613a86d09p-33")]


def det_expf(x):
    """float32 -> float32 exp through a fixed chain of double adds/multiplies.

    k = rint(x*log2(e)); r = x - k*ln2 (two-term); p = Horner(sum r^n/n!, n<=13);
    result = float32(p * 2^k).  Inputs are clamped to [-104, 89] (outside that the
    float32 result is 0 / inf anyway); NaN propagates.
    """
    x = np.asarray(x)
    xd = x.astype(np.float64)
    nan = np.isnan(xd)
    xd = np.where(nan, 0.0, xd)
    xd = np.minimum(np.maximum(xd, -104.0), 89.0)
    k = (xd * _LOG2E + _RND) - _RND
    r = (xd - k * _LN2_HI) - k * _LN2_LO
    p = np.full_like(r, _EXP_C[13])
    for n in range(12, -1, -1):
        p = p * r + _EXP_C[n]
    out = (p * np.ldexp(1.0, k.astype(np.int64))).astype(np.float32)
    return np.where(nan, np.float32(np.nan), out).astype(np.float32)


def det_exp64(x):
    """float64 -> float64 exp through the same fixed chain of double operations as `det_expf` (no final narrowing):
    k = rint(x*log2(e)); r = x - k*ln2 (two-term); p = Horner(sum r^n/n!, n<=13); result = ldexp(p, k) (one rounding,
    also in the subnormal range).  Truncation error of the polynomial on |r| <= ln2/2 is 4e-18, so the result is within
    ~1 ulp of exp(x) -- what np.exp(float64) promises too, without being the same on every host.  Inputs are clamped to
    [-746, 710] (beyond that the float64 result is 0 / inf anyway); NaN propagates."""
    xd = np.asarray(x, dtype=np.float64)
    nan = np.isnan(xd)
    xd = np.where(nan, 0.0, xd)
    xd = np.minimum(np.maximum(xd, -746.0), 710.0)
    k = (xd * _LOG2E + _RND) - _RND
    r = (xd - k * _LN2_HI) - k * _LN2_LO
    p = np.full_like(r, _EXP_C[13])
    for n in range(12, -1, -1):
        p = p * r + _EXP_C[n]
    with np.errstate(over="ignore", under="ignore"):
        out = np.ldexp(p, k.astype(np.int64).astype(np.int32))
    return np.where(nan, np.nan, out)


def _exp_like_input(v, exp_mode):
    if v.dtype == np.float32 and exp_mode == "det":
        return det_expf(v)
    if v.dtype == np.float64 and exp_mode == "det":
        return det_exp64(v)
    return np.exp(v)


_BORDER = {"half": 0, "include": 1, "exclude": -1}


# --------------------------------------------------------------------------------------
# bounding_box_utils
# --------------------------------------------------------------------------------------
def convert_coordinates(tensor, start_index, conversion, border_pixels="half"):
    """bounding_box_utils/bounding_box_utils.py:24-87.

    Result is a float64 copy; the right-hand sides are evaluated in the *input's*
    dtype (that is what makes the f32 decode path round where it does).
    """
    d = _BORDER[border_pixels]
    i = start_index
    t = tensor
    out = np.array(tensor, dtype=np.float64, copy=True)
    a, b, c, e = t[..., i], t[..., i + 1], t[..., i + 2], t[..., i + 3]
    if conversion == "minmax2centroids":      # (xmin,xmax,ymin,ymax) -> (cx,cy,w,h)
        out[..., i], out[..., i + 1] = (a + b) / 2.0, (c + e) / 2.0
        out[..., i + 2], out[..., i + 3] = b - a + d, e - c + d
    elif conversion == "centroids2minmax":
        out[..., i], out[..., i + 1] = a - c / 2.0, a + c / 2.0
        out[..., i + 2], out[..., i + 3] = b - e / 2.0, b + e / 2.0
    elif conversion == "corners2centroids":   # (xmin,ymin,xmax,ymax) -> (cx,cy,w,h)
        out[..., i], out[..., i + 1] = (a + c) / 2.0, (b + e) / 2.0
        out[..., i + 2], out[..., i + 3] = c - a + d, e - b + d
    elif conversion == "centroids2corners":
        out[..., i], out[..., i + 1] = a - c / 2.0, b - e / 2.0
        out[..., i + 2], out[..., i + 3] = a + c / 2.0, b + e / 2.0
    elif conversion in ("minmax2corners", "corners2minmax"):
        out[..., i + 1], out[..., i + 2] = c, b
    else:
        raise ValueError("Unexpected conversion value.")
    return out


def _corner_cols(coords):
    # column of (xmin, ymin, xmax, ymax) in a 'corners' / 'minmax' box
    return (0, 1, 2, 3) if coords == "corners" else (0, 2, 1, 3)


def iou(boxes1, boxes2, coords="centroids", mode="outer_product", border_pixels="half"):
    """bounding_box_utils.py:283-383 (+ intersection_area_ :226-280).

    Quirk kept (:345): the intersection is always computed with d = 0; only the two
    box areas see ``border_pixels``.
    """
    boxes1, boxes2 = np.asarray(boxes1), np.asarray(boxes2)
    if boxes1.ndim > 2 or boxes2.ndim > 2:
        raise ValueError("boxes must have rank 1 or 2")
    if boxes1.ndim == 1:
        boxes1 = boxes1[None]
    if boxes2.ndim == 1:
        boxes2 = boxes2[None]
    if not (boxes1.shape[1] == boxes2.shape[1] == 4):
        raise ValueError("All boxes must consist of 4 coordinates")
    if mode not in ("outer_product", "element-wise"):
        raise ValueError("`mode` must be one of 'outer_product' and 'element-wise'")
    if coords == "centroids":
        boxes1 = convert_coordinates(boxes1, 0, "centroids2corners")
        boxes2 = convert_coordinates(boxes2, 0, "centroids2corners")
        coords = "corners"
    elif coords not in ("minmax", "corners"):
        raise ValueError("Unexpected value for `coords`.")
    x0, y0, x1, y1 = _corner_cols(coords)
    d = _BORDER[border_pixels]
    if mode == "outer_product":
        p, q = boxes1[:, None, :], boxes2[None, :, :]
    else:
        p, q = boxes1, boxes2
    iw = np.maximum(0, np.minimum(p[..., x1], q[..., x1]) - np.maximum(p[..., x0], q[..., x0]) + 0)
    ih = np.maximum(0, np.minimum(p[..., y1], q[..., y1]) - np.maximum(p[..., y0], q[..., y0]) + 0)
    inter = iw * ih
    area_p = (p[..., x1] - p[..., x0] + d) * (p[..., y1] - p[..., y0] + d)
    area_q = (q[..., x1] - q[..., x0] + d) * (q[..., y1] - q[..., y0] + d)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area_p + area_q - inter)


def convert_coordinates2(tensor, start_index, conversion):
    """bounding_box_utils.py:89-117: float64 copy times a constant 4x4 matrix with entries 0, +-0.5, +-1.  Each output is
    a two-term sum of exact products (plus exact zeros), i.e. one rounding -- the same value `convert_coordinates` gives
    for a float64 input with border_pixels='half'."""
    if conversion not in ("minmax2centroids", "centroids2minmax"):
        raise ValueError("Unexpected conversion value.")
    return convert_coordinates(np.array(tensor, dtype=np.float64), start_index, conversion, "half")


def intersection_area(boxes1, boxes2, coords="centroids", mode="outer_product", border_pixels="half"):
    """bounding_box_utils.py:119-224 (the public function: side lengths DO see border_pixels)."""
    boxes1, boxes2 = np.asarray(boxes1), np.asarray(boxes2)
    if boxes1.ndim > 2 or boxes2.ndim > 2:
        raise ValueError("boxes must have rank 1 or 2")
    if boxes1.ndim == 1:
        boxes1 = boxes1[None]
    if boxes2.ndim == 1:
        boxes2 = boxes2[None]
    if not (boxes1.shape[1] == boxes2.shape[1] == 4):
        raise ValueError("All boxes must consist of 4 coordinates")
    if mode not in ("outer_product", "element-wise"):
        raise ValueError("`mode` must be one of 'outer_product' and 'element-wise'")
    if coords == "centroids":
        boxes1 = convert_coordinates(boxes1, 0, "centroids2corners")
        boxes2 = convert_coordinates(boxes2, 0, "centroids2corners")
        coords = "corners"
    elif coords not in ("minmax", "corners"):
        raise ValueError("Unexpected value for `coords`.")
    x0, y0, x1, y1 = _corner_cols(coords)
    d = _BORDER[border_pixels]
    if mode == "outer_product":
        p, q = boxes1[:, None, :], boxes2[None, :, :]
    else:
        p, q = boxes1, boxes2
    iw = np.maximum(0, np.minimum(p[..., x1], q[..., x1]) - np.maximum(p[..., x0], q[..., x0]) + d)
    ih = np.maximum(0, np.minimum(p[..., y1], q[..., y1]) - np.maximum(p[..., y0], q[..., y0]) + d)
    return iw * ih


# --------------------------------------------------------------------------------------
# matching_utils
# --------------------------------------------------------------------------------------
def match_bipartite_greedy(weight_matrix):
    """ssd_encoder_decoder/matching_utils.py:22-79.

    g rounds; each round takes the largest remaining entry (ties: lowest GT row, then
    lowest anchor column), records it, zeroes its row and column.  When everything left
    is zero the argmaxes are 0/0, i.e. GT 0 is (re)assigned anchor 0 -- kept.
    """
    w = np.array(weight_matrix, copy=True)
    g = w.shape[0]
    matches = np.zeros(g, dtype=np.int64)
    rows = np.arange(g)
    for _ in range(g):
        best_col = np.argmax(w, axis=1)
        gt = int(np.argmax(w[rows, best_col]))
        col = int(best_col[gt])
        matches[gt] = col
        w[gt, :] = 0
        w[:, col] = 0
    return matches


def match_multi(weight_matrix, threshold):
    """matching_utils.py:81-116: per-anchor argmax over GTs, kept where value >= threshold."""
    best_gt = np.argmax(weight_matrix, axis=0)
    val = weight_matrix[best_gt, np.arange(weight_matrix.shape[1])]
    cols = np.nonzero(val >= threshold)[0]
    return best_gt[cols], cols


# --------------------------------------------------------------------------------------
# anchors (SSDInputEncoder.generate_anchor_boxes_for_layer == AnchorBoxes.call)
# --------------------------------------------------------------------------------------
def anchor_boxes_for_layer(img_height, img_width, feature_map_size, aspect_ratios, this_scale, next_scale,
                           two_boxes_for_ar1=True, this_steps=None, this_offsets=None, clip_boxes=False,
                           coords="centroids", normalize_coords=True):
    """ssd_input_encoder.py:420-548 / keras_layer_AnchorBoxes.py:133-243 -> (fh, fw, n_boxes, 4) float64."""
    size = min(img_height, img_width)
    wh = []
    for ar in aspect_ratios:
        if ar == 1:
            wh.append((this_scale * size, this_scale * size))
            if two_boxes_for_ar1:
                s = np.sqrt(this_scale * next_scale) * size
                wh.append((s, s))
        else:
            wh.append((this_scale * size * np.sqrt(ar), this_scale * size / np.sqrt(ar)))
    wh = np.array(wh)
    fh, fw = int(feature_map_size[0]), int(feature_map_size[1])
    if this_steps is None:
        step_h, step_w = img_height / fh, img_width / fw
    elif isinstance(this_steps, (list, tuple)) and len(this_steps) == 2:
        step_h, step_w = this_steps
    else:
        step_h = step_w = this_steps
    if this_offsets is None:
        off_h = off_w = 0.5
    elif isinstance(this_offsets, (list, tuple)) and len(this_offsets) == 2:
        off_h, off_w = this_offsets
    else:
        off_h = off_w = this_offsets
    cy = np.linspace(off_h * step_h, (off_h + fh - 1) * step_h, fh)
    cx = np.linspace(off_w * step_w, (off_w + fw - 1) * step_w, fw)
    boxes = np.zeros((fh, fw, len(wh), 4))
    boxes[..., 0] = cx[None, :, None]
    boxes[..., 1] = cy[:, None, None]
    boxes[..., 2] = wh[:, 0]
    boxes[..., 3] = wh[:, 1]
    boxes = convert_coordinates(boxes, 0, "centroids2corners")
    if clip_boxes:
        for cols, lim in (((0, 2), img_width), ((1, 3), img_height)):
            v = boxes[..., cols]
            v[v >= lim] = lim - 1
            v[v < 0] = 0
            boxes[..., cols] = v
    if normalize_coords:
        boxes[..., [0, 2]] /= img_width
        boxes[..., [1, 3]] /= img_height
    if coords == "centroids":
        boxes = convert_coordinates(boxes, 0, "corners2centroids", border_pixels="half")
    elif coords == "minmax":
        boxes = convert_coordinates(boxes, 0, "corners2minmax", border_pixels="half")
    return boxes


class EncoderOracle:
    """SSDInputEncoder, ssd_input_encoder.py:25-611 (config handling :142-275)."""

    def __init__(self, img_height, img_width, n_classes, predictor_sizes, min_scale=0.1, max_scale=0.9,
                 scales=None, aspect_ratios_global=(0.5, 1.0, 2.0), aspect_ratios_per_layer=None,
                 two_boxes_for_ar1=True, steps=None, offsets=None, clip_boxes=False,
                 variances=(0.1, 0.1, 0.2, 0.2), matching_type="multi", pos_iou_threshold=0.5,
                 neg_iou_limit=0.3, border_pixels="half", coords="centroids", normalize_coords=True,
                 background_id=0):
        predictor_sizes = np.array(predictor_sizes)
        if predictor_sizes.ndim == 1:
            predictor_sizes = predictor_sizes[None]
        L = predictor_sizes.shape[0]
        self.img_height, self.img_width = img_height, img_width
        self.n_classes = n_classes + 1
        self.predictor_sizes = predictor_sizes
        self.scales = np.linspace(min_scale, max_scale, L + 1) if scales is None else np.array(scales)
        self.aspect_ratios = ([list(aspect_ratios_global)] * L if aspect_ratios_per_layer is None
                              else aspect_ratios_per_layer)
        self.two_boxes_for_ar1 = two_boxes_for_ar1
        self.steps = steps if steps is not None else [None] * L
        self.offsets = offsets if offsets is not None else [None] * L
        self.clip_boxes = clip_boxes
        self.variances = np.array(variances, dtype=np.float64)
        self.matching_type = matching_type
        self.pos_iou_threshold = pos_iou_threshold
        self.neg_iou_limit = neg_iou_limit
        self.border_pixels = border_pixels
        self.coords = coords
        self.normalize_coords = normalize_coords
        self.background_id = background_id
        self.boxes_list = [
            anchor_boxes_for_layer(img_height, img_width, predictor_sizes[i], self.aspect_ratios[i],
                                   self.scales[i], self.scales[i + 1], two_boxes_for_ar1, self.steps[i],
                                   self.offsets[i], clip_boxes, coords, normalize_coords)
            for i in range(L)]

    def anchors(self):
        """(N, 4) float64 in `coords` format, predictor layers concatenated (:583-590)."""
        return np.concatenate([b.reshape(-1, 4) for b in self.boxes_list], axis=0)

    def generate_encoding_template(self, batch_size):
        """:550-611 -> (B, N, C+12) float64 = [zeros C | anchors | anchors | variances]."""
        a = self.anchors()
        n = a.shape[0]
        t = np.zeros((batch_size, n, self.n_classes + 12))
        t[:, :, self.n_classes:self.n_classes + 4] = a
        t[:, :, self.n_classes + 4:self.n_classes + 8] = a
        t[:, :, self.n_classes + 8:] = self.variances
        return t

    def __call__(self, ground_truth_labels, diagnostics=False, return_matches=False):
        """:277-418.  `return_matches` additionally returns an int32 (B, N) map:
        >=0 index of the matched GT of that image, -1 background, -2 neutral."""
        B = len(ground_truth_labels)
        C = self.n_classes
        y = self.generate_encoding_template(B)
        y[:, :, self.background_id] = 1
        N = y.shape[1]
        match_map = np.full((B, N), -1, dtype=np.int32)
        eye = np.eye(C)
        for i in range(B):
            gt = np.asarray(ground_truth_labels[i])
            if gt.size == 0:
                continue
            lab = gt.astype(np.float64)
            if np.any(lab[:, 3] - lab[:, 1] <= 0) or np.any(lab[:, 4] - lab[:, 2] <= 0):
                raise DegenerateBoxError("degenerate ground truth bounding boxes for batch item {}".format(i))
            if self.normalize_coords:
                lab[:, [2, 4]] /= self.img_height
                lab[:, [1, 3]] /= self.img_width
            if self.coords == "centroids":
                lab = convert_coordinates(lab, 1, "corners2centroids", self.border_pixels)
            elif self.coords == "minmax":
                lab = convert_coordinates(lab, 1, "corners2minmax")
            one_hot = np.concatenate([eye[lab[:, 0].astype(np.int64)], lab[:, 1:5]], axis=-1)
            sim = iou(lab[:, 1:5], y[i, :, -12:-8], coords=self.coords, mode="outer_product",
                      border_pixels=self.border_pixels)
            bip = match_bipartite_greedy(sim)
            y[i, bip, :-8] = one_hot                      # duplicates: last GT wins
            for g_idx, col in enumerate(bip):
                match_map[i, col] = g_idx
            sim[:, bip] = 0
            if self.matching_type == "multi":
                gts, cols = match_multi(sim, self.pos_iou_threshold)
                y[i, cols, :-8] = one_hot[gts]
                match_map[i, cols] = gts
                sim[:, cols] = 0
            neutral = np.nonzero(np.amax(sim, axis=0) >= self.neg_iou_limit)[0]
            y[i, neutral, self.background_id] = 0
            # a neutral anchor is one whose class vector is now all zero
            was_bg = match_map[i, neutral] == -1
            match_map[i, neutral[was_bg]] = -2
        y_matched = None
        if self.coords == "centroids":                    # :396-400
            y[:, :, [-12, -11]] -= y[:, :, [-8, -7]]
            y[:, :, [-12, -11]] /= y[:, :, [-6, -5]] * y[:, :, [-4, -3]]
            y[:, :, [-10, -9]] /= y[:, :, [-6, -5]]
            y[:, :, [-10, -9]] = np.log(y[:, :, [-10, -9]]) / y[:, :, [-2, -1]]
        elif self.coords == "corners":                    # :401-405
            y[:, :, -12:-8] -= y[:, :, -8:-4]
            y[:, :, [-12, -10]] /= (y[:, :, -6] - y[:, :, -8])[..., None]
            y[:, :, [-11, -9]] /= (y[:, :, -5] - y[:, :, -7])[..., None]
            y[:, :, -12:-8] /= y[:, :, -4:]
        elif self.coords == "minmax":                     # :406-410
            y[:, :, -12:-8] -= y[:, :, -8:-4]
            y[:, :, [-12, -11]] /= (y[:, :, -7] - y[:, :, -8])[..., None]
            y[:, :, [-10, -9]] /= (y[:, :, -5] - y[:, :, -6])[..., None]
            y[:, :, -12:-8] /= y[:, :, -4:]
        out = [y]
        if diagnostics:                                   # :412-416
            y_matched = np.copy(y)
            y_matched[:, :, -12:-8] = 0
            out.append(y_matched)
        if return_matches:
            out.append(match_map)
        return out[0] if len(out) == 1 else tuple(out)


class DegenerateBoxError(Exception):
    """ssd_input_encoder.py:613."""


# --------------------------------------------------------------------------------------
# decoder
# --------------------------------------------------------------------------------------
def _greedy_nms_rows(rows, score_col, box_col, iou_threshold, border_pixels, coords="corners", max_keep=None):
    """The loop shared by greedy_nms / _greedy_nms / _greedy_nms2 / _greedy_nms_debug
    (ssd_output_decoder.py:27-109, 469-486): repeatedly take the first maximum of the
    remaining scores, drop it from the pool, drop everything whose IoU with it is
    > threshold (keep `<=`).  Order of the pool is preserved, so among equal scores the
    earliest (lowest anchor index) wins.  IoU runs in the dtype of `rows`.
    `max_keep`: stop after that many survivors (`tf.image.non_max_suppression(max_output_size=...)`: the first
    max_keep rows of the uncapped result).
    """
    alive = np.arange(rows.shape[0])
    kept = []
    boxes = rows[:, box_col:box_col + 4]
    scores = rows[:, score_col]
    while alive.size:
        j = alive[int(np.argmax(scores[alive]))]
        kept.append(j)
        alive = alive[alive != j]
        if not alive.size or (max_keep is not None and len(kept) >= max_keep):
            break
        sim = iou(boxes[alive], boxes[j], coords=coords, mode="element-wise", border_pixels=border_pixels)
        NMS_WORK["iou_pairs"] += int(alive.size)
        alive = alive[sim <= iou_threshold]
    NMS_WORK["kept"] += len(kept)
    return rows[kept]


# Work counters of the loop above (bench.py's secondary figure, SURVEY 8d: "report IoUs/s"): how many box pairs the
# reference's formulation evaluates -- every kept box against everything still alive -- and how many boxes it keeps.
NMS_WORK = {"iou_pairs": 0, "kept": 0}


def greedy_nms(y_pred_decoded, iou_threshold=0.45, coords="corners", border_pixels="half"):
    """ssd_output_decoder.py:27-75 (rows `[class_id, score, 4 coords]`)."""
    res = [_greedy_nms_rows(np.copy(item), 1, 2, iou_threshold, border_pixels, coords) for item in y_pred_decoded]
    return [r if r.shape[0] else np.array([]) for r in res]          # `np.array(maxima)` of an empty list


def greedy_nms_single(predictions, iou_threshold=0.45, coords="corners", border_pixels="half"):
    """`_greedy_nms` (:77-92): rows `[score, 4 coords]`."""
    return _greedy_nms_rows(np.copy(predictions), 0, 1, iou_threshold, border_pixels, coords)


def greedy_nms_single2(predictions, iou_threshold=0.45, coords="corners", border_pixels="half"):
    """`_greedy_nms2` (:94-109) / `_greedy_nms_debug` (:469-486): rows `[id, score, 4 coords]`."""
    return _greedy_nms_rows(np.copy(predictions), 1, 2, iou_threshold, border_pixels, coords)


def _decode_boxes(y_pred, n_lead, input_coords, normalize_coords, img_height, img_width, exp_mode, order="numpy",
                  variance_encoded_in_target=False):
    """Steps 1-2 of decode_detections (:172-198) / decode_detections_fast (:295-321).

    Returns the (B, N, 4) corner boxes.  dtype: float64 when the reference routes the
    tensor through convert_coordinates (centroids, minmax); the input dtype for 'corners'.
    `order='keras'` uses the DecodeDetections layer's association (d*var)*a + c
    (keras_layer_DecodeDetections.py:124-133); 'debug' uses (d*a)*var + c (:399-405).
    """
    off = np.copy(y_pred[:, :, -12:-8])
    anc, var = y_pred[:, :, -8:-4], y_pred[:, :, -4:]
    if input_coords == "centroids" and variance_encoded_in_target:
        # decode_detections_debug only (:405-409): the targets were not divided by the variances
        wh = _exp_like_input(off[:, :, 2:4], exp_mode) * anc[:, :, 2:4]
        cxy = off[:, :, 0:2] * anc[:, :, 2:4] + anc[:, :, 0:2]
        box = convert_coordinates(np.concatenate([cxy, wh], axis=-1), 0, "centroids2corners")
    elif input_coords == "centroids":
        wh = _exp_like_input(off[:, :, 2:4] * var[:, :, 2:4], exp_mode)
        wh = wh * anc[:, :, 2:4]
        if order == "numpy":
            cxy = off[:, :, 0:2] * (var[:, :, 0:2] * anc[:, :, 2:4])
        elif order == "keras":
            cxy = (off[:, :, 0:2] * var[:, :, 0:2]) * anc[:, :, 2:4]
        else:
            cxy = (off[:, :, 0:2] * anc[:, :, 2:4]) * var[:, :, 0:2]
        cxy = cxy + anc[:, :, 0:2]
        box = convert_coordinates(np.concatenate([cxy, wh], axis=-1), 0, "centroids2corners")
    elif input_coords == "minmax":
        off *= var
        off[:, :, [0, 1]] *= (anc[:, :, 1] - anc[:, :, 0])[..., None]
        off[:, :, [2, 3]] *= (anc[:, :, 3] - anc[:, :, 2])[..., None]
        off += anc
        box = convert_coordinates(off, 0, "minmax2corners")
    elif input_coords == "corners":
        off *= var
        off[:, :, [0, 2]] *= (anc[:, :, 2] - anc[:, :, 0])[..., None]
        off[:, :, [1, 3]] *= (anc[:, :, 3] - anc[:, :, 1])[..., None]
        off += anc
        box = off
    else:
        raise ValueError("Unexpected value for `input_coords`.")
    if normalize_coords:
        box[:, :, [0, 2]] *= img_width
        box[:, :, [1, 3]] *= img_height
    return box


def decode_detections(y_pred, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, input_coords="centroids",
                      normalize_coords=True, img_height=None, img_width=None, border_pixels="half",
                      exp_mode="numpy", with_anchor_index=False, decode_order="numpy", variance_encoded_in_target=False):
    """ssd_output_decoder.py:111-226 (and decode_detections_debug :342-467 when
    `with_anchor_index`: rows get the anchor index prepended).

    Returns a list of B float64 arrays (k_i, 6) `[class, conf, xmin, ymin, xmax, ymax]`;
    an image with nothing left is `np.array([])` (shape (0,)).
    """
    if normalize_coords and (img_height is None or img_width is None):
        raise ValueError("the decoder needs the image size to convert relative to absolute coordinates")
    y_pred = np.asarray(y_pred)
    box = _decode_boxes(y_pred, 0, input_coords, normalize_coords, img_height, img_width, exp_mode, decode_order,
                        variance_encoded_in_target)
    C = y_pred.shape[2] - 12
    # rows live in one array per image in the reference, so everything is box.dtype
    conf = y_pred[:, :, :C].astype(box.dtype)
    ids = np.arange(y_pred.shape[1], dtype=box.dtype)
    out = []
    for b in range(y_pred.shape[0]):
        per_class = []
        for c in range(1, C):
            sel = conf[b, :, c] > confidence_thresh
            if not sel.any():
                continue
            rows = np.concatenate([ids[sel, None], conf[b, sel, c, None], box[b, sel]], axis=1)
            kept = _greedy_nms_rows(rows, 1, 2, iou_threshold, border_pixels)
            res = np.zeros((kept.shape[0], 7))
            res[:, 0], res[:, 1], res[:, 2:] = kept[:, 0], c, kept[:, 1:]
            per_class.append(res)
        if per_class:
            pred = np.concatenate(per_class, axis=0)
            if top_k != "all" and pred.shape[0] > top_k:
                keep = np.argpartition(pred[:, 2], kth=pred.shape[0] - top_k, axis=0)[pred.shape[0] - top_k:]
                pred = pred[keep]
            out.append(pred if with_anchor_index else pred[:, 1:])
        else:
            out.append(np.array([]))
    return out


def decode_detections_fast(y_pred, confidence_thresh=0.5, iou_threshold=0.45, top_k="all",
                           input_coords="centroids", normalize_coords=True, img_height=None, img_width=None,
                           border_pixels="half", exp_mode="numpy", with_anchor_index=False):
    """ssd_output_decoder.py:228-333: class = first argmax over all C scores, drop class 0,
    keep conf >= threshold, ONE class-agnostic NMS, optional top-k."""
    if normalize_coords and (img_height is None or img_width is None):
        raise ValueError("the decoder needs the image size to convert relative to absolute coordinates")
    y_pred = np.asarray(y_pred)
    box = _decode_boxes(y_pred, 0, input_coords, normalize_coords, img_height, img_width, exp_mode)
    cls = np.argmax(y_pred[:, :, :-12], axis=-1).astype(y_pred.dtype).astype(box.dtype)
    conf = np.amax(y_pred[:, :, :-12], axis=-1).astype(box.dtype)
    ids = np.arange(y_pred.shape[1], dtype=box.dtype)
    out = []
    for b in range(y_pred.shape[0]):
        rows = np.concatenate([ids[:, None], cls[b, :, None], conf[b, :, None], box[b]], axis=1)
        rows = rows[np.nonzero(rows[:, 1])]
        rows = rows[rows[:, 2] >= confidence_thresh]
        if iou_threshold:
            rows = _greedy_nms_rows(rows, 2, 3, iou_threshold, border_pixels)
            if rows.shape[0] == 0:
                rows = np.array([])           # np.array([]) of an empty `maxima` list (:109)
        if rows.ndim == 2 and top_k != "all" and rows.shape[0] > top_k:
            keep = np.argpartition(rows[:, 2], kth=rows.shape[0] - top_k, axis=0)[rows.shape[0] - top_k:]
            rows = rows[keep]
        if rows.ndim == 2 and not with_anchor_index:
            rows = rows[:, 1:]
        out.append(rows)
    return out


def _tf_iou_f32(box, others):
    """`IOU()` of tensorflow/core/kernels/non_max_suppression_op.cc (TF 1.x; the dependency is un-vendored and un-pinned:
    README.md:143-150 only says "TensorFlow 1.x"), float32 throughout: corners normalised with std::min / std::max
    ((a < b) ? b : a -- NOT NaN-propagating), `area_i <= 0 || area_j <= 0 -> 0`, IEEE division.  box (4,), others (n, 4)."""
    f = np.float32
    mx = lambda a, b: np.where(a < b, b, a)          # std::max(a, b)
    mn = lambda a, b: np.where(b < a, b, a)          # std::min(a, b)
    with np.errstate(all="ignore"):
        y0i, x0i, y1i, x1i = mn(box[0], box[2]), mn(box[1], box[3]), mx(box[0], box[2]), mx(box[1], box[3])
        y0j, x0j = mn(others[:, 0], others[:, 2]), mn(others[:, 1], others[:, 3])
        y1j, x1j = mx(others[:, 0], others[:, 2]), mx(others[:, 1], others[:, 3])
        area_i = f((y1i - y0i) * (x1i - x0i))
        area_j = ((y1j - y0j) * (x1j - x0j)).astype(f)
        iy0, ix0, iy1, ix1 = mx(y0i, y0j), mx(x0i, x0j), mn(y1i, y1j), mn(x1i, x1j)
        inter = (mx((iy1 - iy0).astype(f), f(0)) * mx((ix1 - ix0).astype(f), f(0))).astype(f)
        out = (inter / ((area_i + area_j).astype(f) - inter).astype(f)).astype(f)
    return np.where((area_i <= 0) | (area_j <= 0), f(0), out)


def _tf_nms(boxes32, scores32, iou_threshold, max_keep):
    """`tf.image.non_max_suppression(boxes, scores, max_output_size, iou_threshold)`: candidates by score descending
    (equal scores: lower index first), a candidate is dropped when its IoU with an already selected box is `>`
    float32(iou_threshold); stops at `max_keep` selections.  Returns the selected indices in selection order."""
    thr = np.float32(iou_threshold)
    alive = np.argsort(-scores32, kind="stable")
    kept = []
    while alive.size and len(kept) < max_keep:
        j = alive[0]
        kept.append(j)
        alive = alive[1:]
        if not alive.size:
            break
        sim = _tf_iou_f32(boxes32[j], boxes32[alive])
        NMS_WORK["iou_pairs"] += int(alive.size)
        alive = alive[~(sim > thr)]
    NMS_WORK["kept"] += len(kept)
    return np.asarray(kept, dtype=np.int64)


def decode_detections_layer(y_pred, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, nms_max_output_size=400,
                            normalize_coords=True, img_height=None, img_width=None, fast=False, exp_mode="det"):
    """Container semantics of the DecodeDetections / DecodeDetectionsFast Keras layers
    (keras_layer_DecodeDetections.py:109-265, keras_layer_DecodeDetectionsFast.py:111-248),
    PARITY UNPINNED (TensorFlow is not installable here):
      * decode in float32 with the layer's association (d*var)*a + c, corners = c -/+ 0.5*wh, *img size in f32;
      * per class (or once, class-agnostic, `fast`): strict `>` threshold evaluated in float32, then
        `tf.image.non_max_suppression(max_output_size=nms_max_output_size)` restated from the TF kernel's source
        (`_tf_nms`: float32 IoU, `>` float32 threshold, zero for non-positive areas);
      * global top-k by confidence, sorted descending (tf.nn.top_k(sorted=True); ties -> lower position in
        the class-major padded array), zero rows as padding -> (B, top_k, 6) float32.
    """
    y_pred = np.asarray(y_pred, dtype=np.float32)
    with np.errstate(all="ignore"):
        box64 = _decode_boxes(y_pred, 0, "centroids", normalize_coords, img_height, img_width, exp_mode, "keras")
        box = box64.astype(np.float32)                    # f32 product == rounded exact product
    B, N, L = y_pred.shape
    C = L - 12
    thr32 = np.float32(confidence_thresh)
    out = np.zeros((B, top_k, 6), dtype=np.float32)
    for b in range(B):
        cand = []
        if fast:
            cls = np.argmax(y_pred[b, :, :C], axis=-1)
            conf = np.amax(y_pred[b, :, :C], axis=-1)
            groups = [(None, (cls != 0) & (conf > thr32), cls, conf)]
        else:
            groups = [(c, y_pred[b, :, c] > thr32, None, y_pred[b, :, c]) for c in range(1, C)]
        pos = 0
        for c, sel, cls, conf in groups:
            if sel.any():
                idx = np.nonzero(sel)[0]
                keep = idx[_tf_nms(box[b, idx], conf[idx], iou_threshold, nms_max_output_size)]
                cid = (cls[keep] if c is None else np.full(keep.size, c)).astype(np.float32)
                rows = np.concatenate([cid[:, None], conf[keep, None], box[b, keep]], axis=1)
                for r, row in enumerate(rows):
                    cand.append((row, pos + r))
            pos += nms_max_output_size
        cand.sort(key=lambda t: (-t[0][1], t[1]))
        for r, (row, _) in enumerate(cand[:top_k]):
            out[b, r] = row
    return out


# --------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------
def ssd_loss(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0, return_parts=False):
    """SSDLoss.compute_loss, keras_loss_function/keras_ssd_loss.py:98-211 (smooth_L1 :53-75,
    log_loss :77-96), float32 like the TF graph.  PARITY UNPINNED (no TensorFlow here).

    Hard-negative mining is global over the flattened batch (:179-183); among equal
    losses at the k-th place the lowest flat index is kept (tf.nn.top_k).
    Returns (B,) float32; with `return_parts` also a dict (n_pos, n_neg_losses, k, keep mask, ...).
    """
    yt = np.asarray(y_true, dtype=np.float32)
    yp = np.asarray(y_pred, dtype=np.float32)
    B, N, L = yp.shape
    C = L - 12
    cls_loss = -np.sum(yt[:, :, :C] * np.log(np.maximum(yp[:, :, :C], np.float32(1e-15))), axis=-1, dtype=np.float32)
    diff = yt[:, :, C:C + 4] - yp[:, :, C:C + 4]
    absd = np.abs(diff)
    loc_loss = np.sum(np.where(absd < 1.0, np.float32(0.5) * diff ** 2, absd - np.float32(0.5)), axis=-1,
                      dtype=np.float32)
    negatives = yt[:, :, 0]
    positives = np.max(yt[:, :, 1:C], axis=-1)
    n_pos = np.sum(positives, dtype=np.float32)
    pos_cls = np.sum(cls_loss * positives, axis=-1, dtype=np.float32)
    neg_all = cls_loss * negatives
    n_neg_losses = int(np.count_nonzero(neg_all))
    k = min(max(int(neg_pos_ratio) * int(n_pos), int(n_neg_min)), n_neg_losses)
    keep = np.zeros(B * N, dtype=np.float32)
    if n_neg_losses > 0:
        order = np.argsort(-neg_all.reshape(-1), kind="stable")[:k]
        keep[order] = 1
    keep = keep.reshape(B, N)
    neg_cls = np.sum(cls_loss * keep, axis=-1, dtype=np.float32) if n_neg_losses > 0 else np.zeros(B, np.float32)
    loc = np.sum(loc_loss * positives, axis=-1, dtype=np.float32)
    total = (pos_cls + neg_cls + np.float32(alpha) * loc) / np.maximum(np.float32(1.0), n_pos)
    total = (total * np.float32(B)).astype(np.float32)
    if return_parts:
        return total, dict(n_pos=float(n_pos), n_neg_losses=n_neg_losses, k=k, keep=keep, cls_loss=cls_loss,
                           loc_loss=loc_loss, positives=positives, neg_all=neg_all)
    return total


def ssd_loss_grad(y_true, y_pred, grad_out, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
    """d(sum_b grad_out[b] * loss[b]) / d y_pred, analytic, float64 accumulation then float32.
    Class columns: -y_true/y_pred where y_pred >= 1e-15, for positives and kept negatives;
    offset columns: -alpha * positives * smoothL1'(y_true - y_pred); last 8 columns: 0."""
    yt = np.asarray(y_true, dtype=np.float32)
    yp = np.asarray(y_pred, dtype=np.float32)
    _, parts = ssd_loss(yt, yp, neg_pos_ratio, n_neg_min, alpha, return_parts=True)
    B, N, L = yp.shape
    C = L - 12
    scale = (np.asarray(grad_out, np.float64) * B / max(1.0, parts["n_pos"]))[:, None]
    w_cls = (parts["positives"].astype(np.float64) + parts["keep"].astype(np.float64)) * scale
    g = np.zeros((B, N, L), dtype=np.float64)
    live = yp[:, :, :C] >= np.float32(1e-15)
    safe = np.where(live, yp[:, :, :C], 1).astype(np.float64)
    g[:, :, :C] = np.where(live, -yt[:, :, :C].astype(np.float64) / safe, 0.0) * w_cls[..., None]
    d = (yt[:, :, C:C + 4] - yp[:, :, C:C + 4]).astype(np.float64)
    dl = np.where(np.abs(d) < 1.0, d, np.sign(d))
    g[:, :, C:C + 4] = -alpha * dl * (parts["positives"].astype(np.float64) * scale)[..., None]
    return g.astype(np.float32)


# --------------------------------------------------------------------------------------
# small layers
# --------------------------------------------------------------------------------------
def l2_normalization(x_nhwc, gamma):
    """keras_layer_L2Normalization.py:61-63: K.l2_normalize(x, axis=3) * gamma with TF's
    x * rsqrt(max(sum(x^2), 1e-12)).  PARITY UNPINNED."""
    x = np.asarray(x_nhwc, dtype=np.float32)
    ss = np.sum(x * x, axis=-1, keepdims=True, dtype=np.float32)
    return (x / np.sqrt(np.maximum(ss, np.float32(1e-12)))) * np.asarray(gamma, np.float32)


def anchor_boxes_layer(batch_size, img_height, img_width, feature_map_size, this_scale, next_scale,
                       aspect_ratios=(0.5, 1.0, 2.0), two_boxes_for_ar1=True, this_steps=None, this_offsets=None,
                       clip_boxes=False, variances=(0.1, 0.1, 0.2, 0.2), coords="centroids", normalize_coords=False):
    """AnchorBoxes.call, keras_layer_AnchorBoxes.py:133-255 -> (B, fh, fw, n_boxes, 8) float32."""
    a = anchor_boxes_for_layer(img_height, img_width, feature_map_size, aspect_ratios, this_scale, next_scale,
                               two_boxes_for_ar1, this_steps, this_offsets, clip_boxes, coords, normalize_coords)
    v = np.zeros_like(a) + np.asarray(variances, dtype=np.float64)
    t = np.concatenate([a, v], axis=-1)[None].astype(np.float32)
    return np.tile(t, (batch_size, 1, 1, 1, 1))


# --------------------------------------------------------------------------------------
# eval_utils/average_precision_evaluator.py
# --------------------------------------------------------------------------------------
def evaluator_num_gt_per_class(labels, eval_neutral, n_classes, class_id_index=0, ignore_neutral_boxes=True):
    """average_precision_evaluator.py:477-536."""
    num = np.zeros(n_classes + 1, dtype=np.int64)
    for i, boxes in enumerate(labels):
        boxes = np.asarray(boxes)
        for j in range(boxes.shape[0]):
            if ignore_neutral_boxes and eval_neutral is not None and eval_neutral[i][j]:
                continue
            num[int(boxes[j, class_id_index])] += 1
    return num


def evaluator_match_predictions(prediction_results, labels, image_ids, eval_neutral, n_classes,
                                gt_format={"class_id": 0, "xmin": 1, "ymin": 2, "xmax": 3, "ymax": 4},
                                ignore_neutral_boxes=True, matching_iou_threshold=0.5, border_pixels="include"):
    """average_precision_evaluator.py:538-736 as it behaves with verbose=True and a stable sort (`kind='mergesort'`): with
    verbose=False the reference iterates `range(len(predictions.shape))`, i.e. over the first prediction only (:650), and
    'quicksort' leaves the order of equal confidences unspecified.  Returns (tp, fp, cum_tp, cum_fp): lists indexed by class id;
    classes without predictions get empty arrays for all four (the reference appends nothing to the cumulative lists, :616-620)."""
    cols = [gt_format[k] for k in ("xmin", "ymin", "xmax", "ymax")]
    neutral_on = ignore_neutral_boxes and eval_neutral is not None
    gt_by_image = {str(image_ids[i]): i for i in range(len(image_ids))}
    tps, fps, ctps, cfps = [[]], [[]], [[]], [[]]
    for class_id in range(1, n_classes + 1):
        preds = prediction_results[class_id]
        n = len(preds)
        tp, fp = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        if n:
            conf = np.array([p[1] for p in preds], dtype=np.float32)
            order = np.argsort(-conf, kind="mergesort")
            matched = {}
            for s, pi in enumerate(order):
                image_id = str(preds[pi][0])
                box = np.array(preds[pi][2:6], dtype=np.float32)          # the structured array stores 'f4' (:629-634)
                i = gt_by_image[image_id]
                gt = np.asarray(labels[i])
                mask = gt[:, gt_format["class_id"]] == class_id if gt.size else np.zeros((0,), dtype=bool)
                gt = gt[mask] if gt.size else gt
                if gt.size == 0:
                    fp[s] = 1
                    continue
                with np.errstate(divide="ignore", invalid="ignore"):
                    overlaps = iou(gt[:, cols], box, coords="corners", mode="element-wise", border_pixels=border_pixels)
                j = int(np.argmax(overlaps))
                if overlaps[j] < matching_iou_threshold:
                    fp[s] = 1
                    continue
                if neutral_on and np.asarray(eval_neutral[i])[mask][j]:
                    continue
                flags = matched.setdefault(image_id, np.zeros(gt.shape[0], dtype=bool))
                if not flags[j]:
                    tp[s] = 1
                    flags[j] = True
                else:
                    fp[s] = 1
        tps.append(tp)
        fps.append(fp)
        ctps.append(np.cumsum(tp))
        cfps.append(np.cumsum(fp))
    return tps, fps, ctps, cfps


def evaluator_precision_recall(cum_tp, cum_fp, num_gt_per_class):
    """average_precision_evaluator.py:738-781."""
    precisions, recalls = [[]], [[]]
    for c in range(1, len(cum_tp)):
        tp, fp = np.asarray(cum_tp[c]), np.asarray(cum_fp[c])
        with np.errstate(divide="ignore", invalid="ignore"):
            precisions.append(np.where(tp + fp > 0, tp / (tp + fp), 0))
            recalls.append(tp / num_gt_per_class[c])
    return precisions, recalls


def evaluator_average_precisions(precisions, recalls, mode="sample", num_recall_points=11):
    """average_precision_evaluator.py:783-884 ('sample': Pascal VOC pre-2010 k-point sampling; 'integrate': post-2010)."""
    aps = [0.0]
    for c in range(1, len(precisions)):
        prec, rec = np.asarray(precisions[c]), np.asarray(recalls[c])
        ap = 0.0
        if mode == "sample":
            for t in np.linspace(0, 1, num_recall_points, endpoint=True):
                sel = prec[rec >= t]
                ap += 0.0 if sel.size == 0 else np.amax(sel)
            ap /= num_recall_points
        elif mode == "integrate":
            ur, ui, _ = np.unique(rec, return_index=True, return_counts=True)
            maxp, dr = np.zeros_like(ur), np.zeros_like(ur)
            for i in range(len(ur) - 2, -1, -1):
                maxp[i] = np.maximum(np.amax(prec[ui[i]:ui[i + 1]]), maxp[i + 1])
                dr[i] = ur[i + 1] - ur[i]
            ap = np.sum(maxp * dr)
        else:
            raise ValueError("`mode` can be either 'sample' or 'integrate'")
        aps.append(ap)
    return aps


# --------------------------------------------------------------------------------------
# data_generator/object_detection_2d_image_boxes_validation_utils.py
# --------------------------------------------------------------------------------------
def box_filter_mask(labels, image_height, image_width, check_overlap=True, check_min_area=True, check_degenerate=True,
                    overlap_criterion="center_point", overlap_bounds=(0.3, 1.0), min_area=16, border_pixels="half",
                    labels_format={"class_id": 0, "xmin": 1, "ymin": 2, "xmax": 3, "ymax": 4}):
    """BoxFilter.__call__ (:147-232) as a boolean mask over the rows of `labels`."""
    labels = np.asarray(labels)
    xmin, ymin, xmax, ymax = (labels_format[k] for k in ("xmin", "ymin", "xmax", "ymax"))
    ok = np.ones(labels.shape[0], dtype=bool)
    if labels.shape[0] == 0:
        return ok
    if check_degenerate:
        ok &= (labels[:, xmax] > labels[:, xmin]) & (labels[:, ymax] > labels[:, ymin])
    if check_min_area:
        ok &= (labels[:, xmax] - labels[:, xmin]) * (labels[:, ymax] - labels[:, ymin]) >= min_area
    if check_overlap:
        lower, upper = overlap_bounds
        if overlap_criterion == "iou":
            with np.errstate(divide="ignore", invalid="ignore"):
                v = iou(np.array([0, 0, image_width, image_height]), labels[:, [xmin, ymin, xmax, ymax]], coords="corners",
                        mode="element-wise", border_pixels=border_pixels)
            ok &= (v > lower) & (v <= upper)
        elif overlap_criterion == "area":
            d = _BORDER[border_pixels]
            box_areas = (labels[:, xmax] - labels[:, xmin] + d) * (labels[:, ymax] - labels[:, ymin] + d)
            cy = np.clip(labels[:, [ymin, ymax]], 0, image_height - 1)
            cx = np.clip(labels[:, [xmin, xmax]], 0, image_width - 1)
            inter = (cx[:, 1] - cx[:, 0] + d) * (cy[:, 1] - cy[:, 0] + d)
            ok &= ((inter > lower * box_areas) if lower == 0.0 else (inter >= lower * box_areas)) & (inter <= upper * box_areas)
        else:
            cy = (labels[:, ymin] + labels[:, ymax]) / 2
            cx = (labels[:, xmin] + labels[:, xmax]) / 2
            ok &= (cy >= 0.0) & (cy <= image_height - 1) & (cx >= 0.0) & (cx <= image_width - 1)
    return ok
