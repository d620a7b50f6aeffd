"""Seeded synthetic workloads for the parity tests and `bench.py`.

No dataset or checkpoint is reachable from the build or GPU boxes, so every measured or
tested configuration is generated here from `np.random.RandomState(seed)`.  Host-side
NumPy only; nothing here is on the product's hot path.
"""
from __future__ import annotations

import numpy as np

# Anchor configurations of the reference notebooks ---------------------------------------
SSD300_VOC = dict(                       # ssd300_training.ipynb:84-100
    img_height=300, img_width=300, n_classes=20,
    predictor_sizes=[(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)],
    scales=[0.1, 0.2, 0.37, 0.54, 0.71, 0.88, 1.05],
    aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                             [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]],
    two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 100, 300], offsets=[0.5] * 6, clip_boxes=False,
    variances=[0.1, 0.1, 0.2, 0.2], normalize_coords=True)

SSD512_COCO = dict(                      # ssd512_inference.ipynb:92 (COCO scales), 80 classes
    img_height=512, img_width=512, n_classes=80,
    predictor_sizes=[(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)],
    scales=[0.04, 0.1, 0.26, 0.42, 0.58, 0.74, 0.9, 1.06],
    aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0],
                             [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0], [1.0, 2.0, 0.5],
                             [1.0, 2.0, 0.5]],
    two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 128, 256, 512], offsets=[0.5] * 7, clip_boxes=False,
    variances=[0.1, 0.1, 0.2, 0.2], normalize_coords=True)

SSD7_300 = dict(                         # ssd7_training.ipynb:84-90 at 300x300, 5 classes
    img_height=300, img_width=300, n_classes=5,
    predictor_sizes=[(37, 37), (18, 18), (9, 9), (4, 4)],
    scales=[0.08, 0.16, 0.32, 0.64, 0.96], aspect_ratios_global=[0.5, 1.0, 2.0],
    two_boxes_for_ar1=True, steps=None, offsets=None, clip_boxes=False,
    variances=[1.0, 1.0, 1.0, 1.0], normalize_coords=True)

TINY = dict(                             # small grid for exhaustive option sweeps
    img_height=96, img_width=128, n_classes=5,
    predictor_sizes=[(8, 8), (4, 4), (2, 2), (1, 1)],
    min_scale=0.15, max_scale=0.9, aspect_ratios_global=[0.5, 1.0, 2.0],
    two_boxes_for_ar1=True, steps=None, offsets=None, clip_boxes=False,
    variances=[0.1, 0.1, 0.2, 0.2], normalize_coords=True)


def softmax64(z):
    """Row softmax in float64 using only +,-,*,/ and a Taylor/squaring exp, so the same
    bytes come out on every host (np.exp is SIMD-implementation dependent)."""
    z = z.astype(np.float64)
    z = z - z.max(axis=-1, keepdims=True)
    x = z / 1024.0                         # |x| small: 12-term Taylor is exact to double rounding
    e = np.ones_like(x)
    term = np.ones_like(x)
    for n in range(1, 13):
        term = term * x / n
        e = e + term
    for _ in range(10):                    # e^(1024 x) by repeated squaring
        e = e * e
    tot = e[..., 0].copy()
    for c in range(1, e.shape[-1]):        # fixed left-to-right order (np.sum's order is build dependent)
        tot = tot + e[..., c]
    return e / tot[..., None]


def make_y_pred(anchors_var, batch_size, n_classes_incl_bg, bias=7.0, seed=1234, loc_sigma=0.5,
                dtype=np.float32):
    """Prediction tensor (B, N, C+12): class probabilities = softmax(z), z ~ N(0,1) with
    `bias` added to the background logit (+7: trained-model-like 'sparse', 0: random-weights
    'dense'); offsets ~ N(0, loc_sigma^2); the last 8 columns are `anchors_var` (N, 8)."""
    rng = np.random.RandomState(seed)
    n = anchors_var.shape[0]
    c = n_classes_incl_bg
    z = rng.standard_normal((batch_size, n, c))
    z[:, :, 0] += bias
    y = np.empty((batch_size, n, c + 12), dtype=dtype)
    y[:, :, :c] = softmax64(z)
    y[:, :, c:c + 4] = rng.standard_normal((batch_size, n, 4)) * loc_sigma
    y[:, :, c + 4:] = anchors_var
    return y


def make_ground_truth(batch_size, n_classes, img_height, img_width, max_boxes=8, seed=7, min_boxes=1):
    """List of B float64 arrays (g, 5) `[class, xmin, ymin, xmax, ymax]` in absolute pixels."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(batch_size):
        g = int(rng.randint(min_boxes, max_boxes + 1))
        cls = rng.randint(1, n_classes + 1, size=g)
        x0 = rng.uniform(0, 0.8 * img_width, size=g)
        y0 = rng.uniform(0, 0.8 * img_height, size=g)
        w = rng.uniform(0.03 * img_width, 0.5 * img_width, size=g)
        h = rng.uniform(0.03 * img_height, 0.5 * img_height, size=g)
        x1 = np.minimum(x0 + w, img_width - 1)
        y1 = np.minimum(y0 + h, img_height - 1)
        out.append(np.stack([cls.astype(np.float64), x0, y0, x1, y1], axis=1))
    return out
