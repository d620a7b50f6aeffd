"""`build_model` (SSD7) -- drop-in for the reference builder models/keras_ssd7.py:30-430, as a torch module.

Seven conv + BatchNorm(eps 1e-3, Keras momentum 0.99) + ELU blocks with 2x2 'valid' max-pools
(:277-309); predictor heads on conv4..conv7 (:323-331).
"""
from __future__ import annotations

import numpy as np
from torch import nn
import torch.nn.functional as F

from ._common import SSDModel, he_normal_, make_priorboxes, pool_out, resolve_anchor_config


class SSD7(SSDModel):
    WIDTHS = (32, 48, 64, 64, 48, 48, 32)

    def __init__(self, image_size, n_classes, mode, l2_regularization, scales, aspect_ratios, n_boxes, steps, offsets,
                 two_boxes_for_ar1, clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev,
                 swap_channels, confidence_thresh, iou_threshold, top_k, nms_max_output_size):
        super().__init__(image_size, n_classes, mode, l2_regularization, subtract_mean, divide_by_stddev, swap_channels,
                         confidence_thresh, iou_threshold, top_k, nms_max_output_size, coords, normalize_coords)
        chans = (self.img_channels,) + self.WIDTHS
        self.convs = nn.ModuleList([nn.Conv2d(chans[i], chans[i + 1], 5 if i == 0 else 3, padding=2 if i == 0 else 1)
                                    for i in range(7)])
        self.bns = nn.ModuleList([nn.BatchNorm2d(c, eps=1e-3, momentum=0.01) for c in self.WIDTHS])
        src = self.WIDTHS[3:]
        self.conf_heads = nn.ModuleList([nn.Conv2d(ch, nb * self.n_classes, 3, padding=1) for ch, nb in zip(src, n_boxes)])
        self.loc_heads = nn.ModuleList([nn.Conv2d(ch, nb * 4, 3, padding=1) for ch, nb in zip(src, n_boxes)])
        self.priorboxes = make_priorboxes(self.img_height, self.img_width, scales, aspect_ratios, two_boxes_for_ar1, steps,
                                          offsets, clip_boxes, variances, coords, normalize_coords,
                                          ['anchors4', 'anchors5', 'anchors6', 'anchors7'])
        he_normal_(self)

    def features(self, x):
        feats = []
        for i in range(7):
            x = F.elu(self.bns[i](self.convs[i](x)))
            if i >= 3:
                feats.append(x)
            if i < 6:
                x = self.max_pool(x, 2, 2)
        return feats

    def predictor_sizes(self):
        out = []
        for n in (self.img_height, self.img_width):
            sizes = []
            for i in range(7):
                if i >= 3:
                    sizes.append(n)
                n = pool_out(n, 2, 2)
            out.append(sizes)
        return np.array(list(zip(*out)))


def build_model(image_size, n_classes, mode='training', l2_regularization=0.0, min_scale=0.1, max_scale=0.9, scales=None,
                aspect_ratios_global=[0.5, 1.0, 2.0], aspect_ratios_per_layer=None, two_boxes_for_ar1=True, steps=None,
                offsets=None, clip_boxes=False, variances=[1.0, 1.0, 1.0, 1.0], coords='centroids', normalize_coords=False,
                subtract_mean=None, divide_by_stddev=None, swap_channels=False, confidence_thresh=0.01,
                iou_threshold=0.45, top_k=200, nms_max_output_size=400, return_predictor_sizes=False):
    '''Build the 7-layer SSD (reference keras_ssd7.py:30-54 for the arguments); see `ssd_300`.'''
    scales, ars, n_boxes, steps, offsets = resolve_anchor_config(4, min_scale, max_scale, scales, aspect_ratios_global,
                                                                 aspect_ratios_per_layer, two_boxes_for_ar1, steps,
                                                                 offsets, variances)
    model = SSD7(image_size, n_classes, mode, l2_regularization, scales, ars, n_boxes, steps, offsets, two_boxes_for_ar1,
                 clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev, swap_channels,
                 confidence_thresh, iou_threshold, top_k, nms_max_output_size)
    if return_predictor_sizes:
        return model, model.predictor_sizes()
    return model
