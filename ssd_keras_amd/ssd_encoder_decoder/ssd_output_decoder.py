"""Drop-in for the reference's `ssd_encoder_decoder/ssd_output_decoder.py`, computed on the GPU.

Same function names, keyword arguments, error behaviour and return containers as the
reference (ssd_output_decoder.py:111-226 `decode_detections`, :228-333
`decode_detections_fast`, :342-467 `decode_detections_debug`, :488-530 helpers); the
arithmetic runs in libssdhip.so (`ssdhip_decode_detections`, include/ssdhip.h).

`y_pred` may be a NumPy array (copied to the current GPU) or a CUDA torch tensor (used in
place, no host round trip until the final, small result copy).
"""
from __future__ import annotations

import numpy as np

from .. import _native as nat


def _as_device(y_pred):
    """NumPy array / torch tensor -> CUDA tensor of float32 (the model's output: float32 decode, then the reference's float64
    flow) or float64 (the reference computes in the input's dtype, ssd_output_decoder.py:172-198: everything float64).
    Other dtypes are widened to float64, which is where NumPy's arithmetic with the Python-float thresholds would take them."""
    import torch
    if isinstance(y_pred, np.ndarray):
        if y_pred.dtype not in (np.float32, np.float64):
            y_pred = y_pred.astype(np.float64)
        return nat.to_device(y_pred)
    if not torch.is_tensor(y_pred):
        raise TypeError("y_pred must be a NumPy array or a torch tensor")
    if y_pred.dtype not in (torch.float32, torch.float64):
        y_pred = y_pred.to(torch.float64)
    return nat.to_device(y_pred)


def _check_common(y, normalize_coords, img_height, img_width, input_coords):
    if normalize_coords and ((img_height is None) or (img_width is None)):
        raise ValueError("If relative box coordinates are supposed to be converted to absolute coordinates, the decoder "
                         "needs the image size in order to decode the predictions, but `img_height == {}` and "
                         "`img_width == {}`".format(img_height, img_width))
    if input_coords not in nat.COORDS:
        raise ValueError("Unexpected value for `input_coords`. Supported input coordinate formats are 'minmax', "
                         "'corners' and 'centroids'.")
    if y.dim() != 3 or y.shape[2] < 14:
        raise ValueError("y_pred must have shape (batch, #boxes, #classes + 12)")


def _to_list(out, count, aidx=None, empty_1d=True):
    out = out.cpu().numpy()
    count = count.cpu().numpy()
    aidx = aidx.cpu().numpy() if aidx is not None else None
    res = []
    for b in range(out.shape[0]):
        k = int(count[b])
        if k == 0 and empty_1d:
            res.append(np.array([]))                  # the reference's container for "nothing left" (:223)
            continue
        rows = out[b, :k].astype(np.float64, copy=True)
        if aidx is not None:
            rows = np.concatenate([aidx[b, :k, None].astype(np.float64), rows], axis=1)
        res.append(rows)
    return res


def _nms_tables(tables, score_col, box_col, iou_threshold, coords, border_pixels):
    """Greedy NMS of several row tables in one launch (`ssdhip_greedy_nms`, one workgroup per table).  Returns, per
    table, the kept rows in selection order as float64 NumPy arrays (what `np.array(maxima)` is in the reference)."""
    import torch
    if coords not in nat.COORDS:
        raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
    tabs = []
    for t in tables:
        t = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        tabs.append(np.ascontiguousarray(t, dtype=np.float64))
    widths = {t.shape[1] for t in tabs if t.ndim == 2 and t.shape[0]}
    if len(widths) > 1:
        raise ValueError("all batch items must have the same number of columns")
    if not widths:
        return [np.array([]) for _ in tabs]                     # `np.array(maxima)` of an empty list
    L = widths.pop()
    if L < box_col + 4 or L <= score_col:
        raise ValueError("rows need at least {} columns".format(max(box_col + 4, score_col + 1)))
    nonempty = [t if (t.ndim == 2 and t.shape[0]) else np.zeros((0, L)) for t in tabs]
    off = np.cumsum([0] + [t.shape[0] for t in nonempty]).astype(np.int32)
    cat = np.concatenate(nonempty, axis=0)
    rows_d = nat.to_device(cat)
    kept, cnt = nat.greedy_nms_rows(rows_d, off, score_col, box_col, iou_threshold, coords, border_pixels)
    kept, cnt = kept.cpu().numpy(), cnt.cpu().numpy()
    res = []
    for i, t in enumerate(nonempty):
        k = int(cnt[i])
        res.append(t[kept[off[i]:off[i] + k]] if k else np.array([]))
    return res


def greedy_nms(y_pred_decoded, iou_threshold=0.45, coords='corners', border_pixels='half'):
    '''Reference: ssd_output_decoder.py:27-75.  `y_pred_decoded`: list of `(k, 6)` arrays `[class_id, score, 4 box
    coordinates]`; per batch item, repeatedly keep the highest-scoring box and drop every box whose IoU with it is
    `> iou_threshold`.  Returns the list of kept rows (score descending).'''
    return _nms_tables(list(y_pred_decoded), 1, 2, iou_threshold, coords, border_pixels)


def _greedy_nms(predictions, iou_threshold=0.45, coords='corners', border_pixels='half'):
    '''Reference :77-92: rows `[score, 4 box coordinates]` of one class of one image.'''
    return _nms_tables([predictions], 0, 1, iou_threshold, coords, border_pixels)[0]


def _greedy_nms2(predictions, iou_threshold=0.45, coords='corners', border_pixels='half'):
    '''Reference :94-109: rows `[class_id, score, 4 box coordinates]` of one image.'''
    return _nms_tables([predictions], 1, 2, iou_threshold, coords, border_pixels)[0]


def _greedy_nms_debug(predictions, iou_threshold=0.45, coords='corners', border_pixels='half'):
    '''Reference :469-486: rows `[box_id, score, 4 box coordinates]`.'''
    return _nms_tables([predictions], 1, 2, iou_threshold, coords, border_pixels)[0]


def decode_detections(y_pred, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, input_coords='centroids',
                      normalize_coords=True, img_height=None, img_width=None, border_pixels='half'):
    '''Reference: ssd_output_decoder.py:111-226.  Per image and per non-background class: strict `>`
    confidence threshold, greedy NMS (float64 IoU, keep `<= iou_threshold`), then the `top_k` most
    confident rows.  Returns a list of `batch_size` float64 arrays `(k, 6)`
    `[class_id, confidence, xmin, ymin, xmax, ymax]`; `np.array([])` for an image with nothing left.'''
    y = _as_device(y_pred)
    _check_common(y, normalize_coords, img_height, img_width, input_coords)
    B, N, L = y.shape
    k = 0 if top_k == 'all' else int(top_k)
    rows = (L - 13) * N if k == 0 else k
    out, count, _ = nat.decode(y, confidence_thresh, iou_threshold, k, 0, False, nat.SEM_NUMPY, input_coords,
                               normalize_coords, img_height, img_width, border_pixels, nat.F64, rows)
    return _to_list(out, count)


def decode_detections_fast(y_pred, confidence_thresh=0.5, iou_threshold=0.45, top_k='all', input_coords='centroids',
                           normalize_coords=True, img_height=None, img_width=None, border_pixels='half'):
    '''Reference: ssd_output_decoder.py:228-333.  Class = first argmax over all scores, background
    dropped, `>=` confidence threshold, one class-agnostic NMS (skipped when `iou_threshold` is falsy).'''
    y = _as_device(y_pred)
    _check_common(y, normalize_coords, img_height, img_width, input_coords)
    B, N, L = y.shape
    k = 0 if top_k == 'all' else int(top_k)
    rows = N if k == 0 else k
    no_nms = not iou_threshold
    out, count, aidx = nat.decode(y, confidence_thresh, float('inf') if no_nms else iou_threshold, k, 0, True,
                                  nat.SEM_NUMPY, input_coords, normalize_coords, img_height, img_width, border_pixels,
                                  nat.F64, rows, want_anchor_idx=no_nms)
    if no_nms:       # the reference leaves the rows in anchor order when it skips NMS
        res = _to_list(out, count, aidx, empty_1d=False)
        return [r[np.argsort(r[:, 0], kind='stable')][:, 1:] if r.shape[0] else np.zeros((0, 6)) for r in res]
    return _to_list(out, count)


def decode_detections_debug(y_pred, confidence_thresh=0.01, iou_threshold=0.45, top_k=200, input_coords='centroids',
                            normalize_coords=True, img_height=None, img_width=None, variance_encoded_in_target=False,
                            border_pixels='half'):
    '''Reference: ssd_output_decoder.py:342-467.  As `decode_detections`, rows
    `[box_id, class_id, confidence, xmin, ymin, xmax, ymax]`.'''
    y = _as_device(y_pred)
    _check_common(y, normalize_coords, img_height, img_width, input_coords)
    if variance_encoded_in_target and input_coords == 'centroids':
        # :405-409: the offsets were not divided by the variances, i.e. decode with variances of one -- x * 1.0 is exact, so the
        # kernel's (d * a) * var + c and exp(d * var) * a are bit for bit the reference's d * a + c and exp(d) * a.  (The
        # reference ignores the flag for 'corners' / 'minmax', :416-425.)
        y = y.clone()                                  # never write into the caller's tensor
        y[:, :, -4:] = 1.0
    out, count, aidx = nat.decode(y, confidence_thresh, iou_threshold, int(top_k), 0, False, nat.SEM_DEBUG, input_coords,
                                  normalize_coords, img_height, img_width, border_pixels, nat.F64, int(top_k),
                                  want_anchor_idx=True)
    return _to_list(out, count, aidx, empty_1d=False)


def get_num_boxes_per_pred_layer(predictor_sizes, aspect_ratios, two_boxes_for_ar1):
    '''Reference: ssd_output_decoder.py:488-501.'''
    return [int(predictor_sizes[i][0]) * int(predictor_sizes[i][1]) * (len(aspect_ratios[i]) + (1 if two_boxes_for_ar1 else 0))
            for i in range(len(predictor_sizes))]


def get_pred_layers(y_pred_decoded, num_boxes_per_pred_layer):
    '''Reference: ssd_output_decoder.py:503-530: predictor layer of every row of a
    `decode_detections_debug` result (host-side bookkeeping).'''
    edges = np.cumsum(num_boxes_per_pred_layer)
    res = []
    for item in y_pred_decoded:
        ids = np.asarray(item)[:, 0] if len(item) else np.zeros((0,))
        if np.any(ids < 0) or np.any(ids >= edges[-1]):
            raise ValueError("Box index is out of bounds of the possible indices as given by the values in "
                             "`num_boxes_per_pred_layer`.")
        res.append([int(v) for v in np.searchsorted(edges, ids, side='right')])
    return res
