"""Geometric image ops of the original-SSD chain -- drop-in for data_generator/object_detection_2d_geometric_ops.py:27-262
(`Resize`, `ResizeRandomInterp`, `Flip`, `RandomFlip`).

The label arithmetic, the inverters, the random draws and the return conventions are the reference's; the resampling itself
(`cv2.resize`, :70-72) runs on the GPU with the arithmetic of OpenCV's imgproc/resize.cpp for 8-bit images (round 6;
csrc/ssdhip_image.hip, plans built by `_image_ops.resize_plan`): cv::resize's dispatch over the five interpolation modes the chain draws
from -- 11-bit fixed-point coefficients with the two-stage vertical rounding (linear) or `(sum + 2^21) >> 22` (cubic, Lanczos-4), the
area-mode bilinear variant, ResizeArea / ResizeAreaFast, nearest, copy -- checked against cases worked out by hand from that source
(tests/resize_hand_cases.py; no OpenCV binary exists here to pin against).
`Translate`, `Scale`, `Rotate` and their random forms (cv2.warpAffine; the satellite / constant-input-size chains) are not provided."""
from __future__ import annotations

import numpy as np

from . import _image_ops as iop
from .object_detection_2d_image_boxes_validation_utils import BoxFilter

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4      # cv2's values

_DEFAULT_FORMAT = {'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}


class Resize:
    def __init__(self, height, width, interpolation_mode=INTER_LINEAR, box_filter=None, labels_format=_DEFAULT_FORMAT):
        if not (isinstance(box_filter, BoxFilter) or box_filter is None):
            raise ValueError("`box_filter` must be either `None` or a `BoxFilter` object.")
        self.out_height = height
        self.out_width = width
        self.interpolation_mode = interpolation_mode
        self.box_filter = box_filter
        self.labels_format = labels_format

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        xmin, ymin = self.labels_format['xmin'], self.labels_format['ymin']
        xmax, ymax = self.labels_format['xmax'], self.labels_format['ymax']
        out_h, out_w = self.out_height, self.out_width

        image = iop.resize(image, out_h, out_w, self.interpolation_mode)

        def inverter(labels):
            # (the reference's inverter addresses the columns one to the RIGHT of the label format's -- predictions carry a
            # confidence column after the class id, :76-78)
            labels = np.copy(labels)
            labels[:, [ymin + 1, ymax + 1]] = np.round(labels[:, [ymin + 1, ymax + 1]] * (img_height / out_h), decimals=0)
            labels[:, [xmin + 1, xmax + 1]] = np.round(labels[:, [xmin + 1, xmax + 1]] * (img_width / out_w), decimals=0)
            return labels

        if labels is None:
            return (image, inverter) if return_inverter else image
        labels = np.copy(labels)
        labels[:, [ymin, ymax]] = np.round(labels[:, [ymin, ymax]] * (out_h / img_height), decimals=0)
        labels[:, [xmin, xmax]] = np.round(labels[:, [xmin, xmax]] * (out_w / img_width), decimals=0)
        if self.box_filter is not None:
            self.box_filter.labels_format = self.labels_format
            labels = self.box_filter(labels=labels, image_height=out_h, image_width=out_w)
        return (image, labels, inverter) if return_inverter else (image, labels)


class ResizeRandomInterp:
    def __init__(self, height, width,
                 interpolation_modes=[INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4],
                 box_filter=None, labels_format=_DEFAULT_FORMAT):
        if not (isinstance(interpolation_modes, (list, tuple))):
            raise ValueError("`interpolation_mode` must be a list or tuple.")
        self.height = height
        self.width = width
        self.interpolation_modes = interpolation_modes
        self.box_filter = box_filter
        self.labels_format = labels_format
        self.resize = Resize(height=self.height, width=self.width, box_filter=self.box_filter, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        self.resize.interpolation_mode = np.random.choice(self.interpolation_modes)
        self.resize.labels_format = self.labels_format
        return self.resize(image, labels, return_inverter)


class Flip:
    """A view of the image and mirrored box coordinates (reference :150-200; `return_inverter` is accepted and ignored there too)."""

    def __init__(self, dim='horizontal', labels_format=_DEFAULT_FORMAT):
        if not (dim in {'horizontal', 'vertical'}):
            raise ValueError("`dim` can be one of 'horizontal' and 'vertical'.")
        self.dim = dim
        self.labels_format = labels_format

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        xmin, ymin = self.labels_format['xmin'], self.labels_format['ymin']
        xmax, ymax = self.labels_format['xmax'], self.labels_format['ymax']
        if self.dim == 'horizontal':
            image = image[:, ::-1]
            if labels is None:
                return image
            labels = np.copy(labels)
            labels[:, [xmin, xmax]] = img_width - labels[:, [xmax, xmin]]
            return image, labels
        image = image[::-1]
        if labels is None:
            return image
        labels = np.copy(labels)
        labels[:, [ymin, ymax]] = img_height - labels[:, [ymax, ymin]]
        return image, labels


class RandomFlip:
    def __init__(self, dim='horizontal', prob=0.5, labels_format=_DEFAULT_FORMAT):
        self.dim = dim
        self.prob = prob
        self.labels_format = labels_format
        self.flip = Flip(dim=self.dim, labels_format=self.labels_format)

    def __call__(self, image, labels=None):
        if np.random.uniform(0, 1) < (1.0 - self.prob):
            return image if labels is None else (image, labels)
        self.flip.labels_format = self.labels_format
        return self.flip(image, labels)
