"""`AnchorBoxes` -- drop-in for keras_layers/keras_layer_AnchorBoxes.py:27-278 as a torch module.

The reference re-creates and tiles the anchor constant in every forward pass
(`K.tile(K.constant(...))`, :253).  Here the `(h, w, n_boxes, 8)` float32 constant is built once
(float64 host math, `anchor_math.layer_anchor_boxes`), kept resident in HBM as a buffer and
returned as a stride-0 batch expansion: no per-step traffic at all.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..anchor_math import layer_anchor_boxes, n_boxes_for


class AnchorBoxes(nn.Module):
    def __init__(self, img_height, img_width, this_scale, next_scale, aspect_ratios=[0.5, 1.0, 2.0],
                 two_boxes_for_ar1=True, this_steps=None, this_offsets=None, clip_boxes=False,
                 variances=[0.1, 0.1, 0.2, 0.2], coords='centroids', normalize_coords=False, **kwargs):
        super().__init__()
        if (this_scale < 0) or (next_scale < 0) or (this_scale > 1):
            raise ValueError("`this_scale` must be in [0, 1] and `next_scale` must be >0, but `this_scale` == {}, "
                             "`next_scale` == {}".format(this_scale, next_scale))
        if len(variances) != 4:
            raise ValueError("4 variance values must be pased, but {} values were received.".format(len(variances)))
        variances = np.array(variances)
        if np.any(variances <= 0):
            raise ValueError("All variances must be >0, but the variances given are {}".format(variances))
        self.img_height, self.img_width = img_height, img_width
        self.this_scale, self.next_scale = this_scale, next_scale
        self.aspect_ratios = aspect_ratios
        self.two_boxes_for_ar1 = two_boxes_for_ar1
        self.this_steps, self.this_offsets = this_steps, this_offsets
        self.clip_boxes = clip_boxes
        self.variances = variances
        self.coords = coords
        self.normalize_coords = normalize_coords
        self.n_boxes = n_boxes_for(aspect_ratios, two_boxes_for_ar1)
        self.name = kwargs.get('name')
        self._cache = {}

    def anchors_f64(self, feature_map_height, feature_map_width):
        """(h, w, n_boxes, 4) float64 -- identical to the encoder's `boxes_list` entry."""
        return layer_anchor_boxes(self.img_height, self.img_width, (feature_map_height, feature_map_width),
                                  self.aspect_ratios, self.this_scale, self.next_scale, self.two_boxes_for_ar1,
                                  self.this_steps, self.this_offsets, self.clip_boxes, self.coords,
                                  self.normalize_coords)

    def constant(self, feature_map_height, feature_map_width, device):
        key = (feature_map_height, feature_map_width, str(device))
        t = self._cache.get(key)
        if t is None:
            a = self.anchors_f64(feature_map_height, feature_map_width)
            v = np.zeros_like(a) + self.variances
            t = torch.from_numpy(np.concatenate([a, v], axis=-1).astype(np.float32)).to(device)
            self._cache[key] = t
        return t

    def build(self, input_shape):
        """Keras' shape hook (reference :128-131): nothing to create here -- the layer has no weights."""
        self.input_shape_ = tuple(input_shape) if input_shape is not None else None

    def forward(self, x):
        """x: the predictor feature map, (B, C, H, W).  Returns (B, H, W, n_boxes, 8) float32:
        4 anchor coordinates + 4 variances (reference :245-255)."""
        b, _, h, w = x.shape
        return self.constant(h, w, x.device).unsqueeze(0).expand(b, -1, -1, -1, -1)

    call = forward                                   # the Keras layer's method name (reference :133)

    def compute_output_shape(self, input_shape):
        batch_size, _, h, w = input_shape
        return (batch_size, h, w, self.n_boxes, 8)

    def get_config(self):
        return {'img_height': self.img_height, 'img_width': self.img_width, 'this_scale': self.this_scale,
                'next_scale': self.next_scale, 'aspect_ratios': list(self.aspect_ratios),
                'two_boxes_for_ar1': self.two_boxes_for_ar1, 'clip_boxes': self.clip_boxes,
                'variances': list(self.variances), 'coords': self.coords, 'normalize_coords': self.normalize_coords}
