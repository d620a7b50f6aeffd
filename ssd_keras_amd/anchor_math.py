"""Host-side anchor ("prior") box generation, float64, done once per model configuration.

Mirrors the arithmetic of SSDInputEncoder.generate_anchor_boxes_for_layer
(reference ssd_encoder_decoder/ssd_input_encoder.py:420-548), which AnchorBoxes.call
(keras_layers/keras_layer_AnchorBoxes.py:133-243) repeats verbatim: box sizes from the
scale / aspect ratio, centre grid by np.linspace, -> corners, optional clip, optional
normalisation, -> the requested coordinate format.  Not a per-step cost: the result is
uploaded once and stays resident in HBM.
"""
from __future__ import annotations

import numpy as np


def _pair(v, default):
    if v is None:
        return default, default
    if isinstance(v, (list, tuple)) and len(v) == 2:
        return v[0], v[1]
    return v, v


def n_boxes_for(aspect_ratios, two_boxes_for_ar1):
    return len(aspect_ratios) + (1 if (1 in aspect_ratios) and two_boxes_for_ar1 else 0)


def layer_anchor_boxes(img_height, img_width, feature_map_size, aspect_ratios, this_scale, next_scale,
                       two_boxes_for_ar1=True, this_steps=None, this_offsets=None, clip_boxes=False,
                       coords='centroids', normalize_coords=True, diagnostics=False):
    """(fh, fw, n_boxes, 4) float64 anchors of one predictor layer in `coords` format."""
    size = min(img_height, img_width)
    wh = []
    for ar in aspect_ratios:
        if ar == 1:
            wh.append((this_scale * size, this_scale * size))
            if two_boxes_for_ar1:
                extra = np.sqrt(this_scale * next_scale) * size
                wh.append((extra, extra))
        else:
            wh.append((this_scale * size * np.sqrt(ar), this_scale * size / np.sqrt(ar)))
    wh = np.array(wh)
    fh, fw = int(feature_map_size[0]), int(feature_map_size[1])
    if this_steps is None:
        step_h, step_w = img_height / fh, img_width / fw
    else:
        step_h, step_w = _pair(this_steps, None)
    off_h, off_w = _pair(this_offsets, 0.5)
    cy = np.linspace(off_h * step_h, (off_h + fh - 1) * step_h, fh)
    cx = np.linspace(off_w * step_w, (off_w + fw - 1) * step_w, fw)
    n = wh.shape[0]
    cxg = np.broadcast_to(cx[None, :, None], (fh, fw, n))
    cyg = np.broadcast_to(cy[:, None, None], (fh, fw, n))
    w = np.broadcast_to(wh[:, 0], (fh, fw, n))
    h = np.broadcast_to(wh[:, 1], (fh, fw, n))
    xmin, ymin, xmax, ymax = cxg - w / 2.0, cyg - h / 2.0, cxg + w / 2.0, cyg + h / 2.0
    if clip_boxes:
        xmin, xmax = [np.where(v < 0, 0.0, np.where(v >= img_width, img_width - 1.0, v)) for v in (xmin, xmax)]
        ymin, ymax = [np.where(v < 0, 0.0, np.where(v >= img_height, img_height - 1.0, v)) for v in (ymin, ymax)]
    if normalize_coords:
        xmin, xmax = xmin / img_width, xmax / img_width
        ymin, ymax = ymin / img_height, ymax / img_height
    if coords == 'centroids':
        out = np.stack([(xmin + xmax) / 2.0, (ymin + ymax) / 2.0, xmax - xmin, ymax - ymin], axis=-1)
    elif coords == 'minmax':
        out = np.stack([xmin, xmax, ymin, ymax], axis=-1)
    elif coords == 'corners':
        out = np.stack([xmin, ymin, xmax, ymax], axis=-1)
    else:
        raise ValueError("Unexpected value for `coords`. Supported values are 'minmax', 'corners' and 'centroids'.")
    out = np.ascontiguousarray(out, dtype=np.float64)
    if diagnostics:
        return out, (cy, cx), wh, (step_h, step_w), (off_h, off_w)
    return out
