"""`ssd_512` -- drop-in for the reference builder models/keras_ssd512.py:31-477, as a torch module.

As SSD300 with a seventh predictor layer: conv6_2..conv9_2 are all pad-1 stride-2 3x3 convolutions and
conv10_2 is a 4x4 'valid' convolution on a padded 2x2 map (:303-321); 24564 anchors at 512x512.
"""
from __future__ import annotations

import numpy as np
from torch import nn

from ._common import conv_out, he_normal_, make_priorboxes, resolve_anchor_config
from .keras_ssd300 import _VGGBase


class SSD512(_VGGBase):
    SOURCE_CHANNELS = (512, 1024, 512, 256, 256, 256, 256)
    NAMES = ('conv4_3_norm', 'fc7', 'conv6_2', 'conv7_2', 'conv8_2', 'conv9_2', 'conv10_2')

    def __init__(self, image_size, n_classes, mode, l2_regularization, scales, aspect_ratios, n_boxes, steps, offsets,
                 two_boxes_for_ar1, clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev,
                 swap_channels, confidence_thresh, iou_threshold, top_k, nms_max_output_size):
        super().__init__(image_size, n_classes, mode, l2_regularization, subtract_mean, divide_by_stddev, swap_channels,
                         confidence_thresh, iou_threshold, top_k, nms_max_output_size, coords, normalize_coords)
        self._build_vgg(self.img_channels)
        self.conv6_1, self.conv6_2 = nn.Conv2d(1024, 256, 1), nn.Conv2d(256, 512, 3, stride=2, padding=1)
        self.conv7_1, self.conv7_2 = nn.Conv2d(512, 128, 1), nn.Conv2d(128, 256, 3, stride=2, padding=1)
        self.conv8_1, self.conv8_2 = nn.Conv2d(256, 128, 1), nn.Conv2d(128, 256, 3, stride=2, padding=1)
        self.conv9_1, self.conv9_2 = nn.Conv2d(256, 128, 1), nn.Conv2d(128, 256, 3, stride=2, padding=1)
        self.conv10_1, self.conv10_2 = nn.Conv2d(256, 128, 1), nn.Conv2d(128, 256, 4, padding=1)
        self.conf_heads = nn.ModuleList([nn.Conv2d(ch, nb * self.n_classes, 3, padding=1)
                                         for ch, nb in zip(self.SOURCE_CHANNELS, n_boxes)])
        self.loc_heads = nn.ModuleList([nn.Conv2d(ch, nb * 4, 3, padding=1) for ch, nb in zip(self.SOURCE_CHANNELS, n_boxes)])
        self.priorboxes = make_priorboxes(self.img_height, self.img_width, scales, aspect_ratios, two_boxes_for_ar1, steps,
                                          offsets, clip_boxes, variances, coords, normalize_coords,
                                          [n + '_mbox_priorbox' for n in self.NAMES])
        he_normal_(self)

    def trunk_features(self, x):
        """The two source maps the VGG trunk yields (conv4_3 after L2Normalization, fc7): their predictor heads do not depend on
        the extra layers, which `extra_features` derives from fc7."""
        return self._trunk(x)

    def extra_features(self, fc7):
        ca = self.conv_act
        conv6_2 = ca(self.conv6_2, ca(self.conv6_1, fc7))
        conv7_2 = ca(self.conv7_2, ca(self.conv7_1, conv6_2))
        conv8_2 = ca(self.conv8_2, ca(self.conv8_1, conv7_2))
        conv9_2 = ca(self.conv9_2, ca(self.conv9_1, conv8_2))
        conv10_2 = ca(self.conv10_2, ca(self.conv10_1, conv9_2))
        return [conv6_2, conv7_2, conv8_2, conv9_2, conv10_2]

    def features(self, x):
        early = self.trunk_features(x)
        return early + self.extra_features(early[1])

    def predictor_sizes(self):
        out = []
        for n in (self.img_height, self.img_width):
            c43, f7 = self._vgg_sizes(n)
            sizes = [c43, f7]
            m = f7
            for _ in range(4):
                m = conv_out(m, 3, 2, 1)
                sizes.append(m)
            sizes.append(conv_out(m, 4, 1, 1))
            out.append(sizes)
        return np.array(list(zip(*out)))


def ssd_512(image_size, n_classes, mode='training', l2_regularization=0.0005, min_scale=None, max_scale=None, scales=None,
            aspect_ratios_global=None,
            aspect_ratios_per_layer=[[1.0, 2.0, 0.5], [1.0, 2.0, 0.5, 3.0, 1.0/3.0], [1.0, 2.0, 0.5, 3.0, 1.0/3.0],
                                     [1.0, 2.0, 0.5, 3.0, 1.0/3.0], [1.0, 2.0, 0.5, 3.0, 1.0/3.0], [1.0, 2.0, 0.5],
                                     [1.0, 2.0, 0.5]],
            two_boxes_for_ar1=True, steps=[8, 16, 32, 64, 128, 256, 512], offsets=None, clip_boxes=False,
            variances=[0.1, 0.1, 0.2, 0.2], coords='centroids', normalize_coords=True, subtract_mean=[123, 117, 104],
            divide_by_stddev=None, swap_channels=[2, 1, 0], confidence_thresh=0.01, iou_threshold=0.45, top_k=200,
            nms_max_output_size=400, return_predictor_sizes=False):
    '''Build an SSD512 (reference keras_ssd512.py:31-61 for the arguments); see `ssd_300`.'''
    scales, ars, n_boxes, steps, offsets = resolve_anchor_config(7, min_scale, max_scale, scales, aspect_ratios_global,
                                                                 aspect_ratios_per_layer, two_boxes_for_ar1, steps,
                                                                 offsets, variances)
    model = SSD512(image_size, n_classes, mode, l2_regularization, scales, ars, n_boxes, steps, offsets, two_boxes_for_ar1,
                   clip_boxes, variances, coords, normalize_coords, subtract_mean, divide_by_stddev, swap_channels,
                   confidence_thresh, iou_threshold, top_k, nms_max_output_size)
    if return_predictor_sizes:
        return model, model.predictor_sizes()
    return model
