"""Drop-in for the reference's data_generator/object_detection_2d_image_boxes_validation_utils.py:
`BoundGenerator` :28-77, `BoxFilter` :79-232, `ImageValidator` :234-322 -- which boxes, and which images / patches, meet
overlap / size requirements with respect to an image size.

The per-box tests run on the GPU (`ssdhip_box_filter`, csrc/ssdhip_boxes.hip), one thread per box, in float64 (exact for the integer
label arrays the dataset parsers produce and for float64 labels; float32 labels are widened first, whereas NumPy would evaluate them
in float32).  The reference validates one image per call; `BoxFilter.filter_batch` / `ImageValidator.validate_batch` take a whole
batch (e.g. all candidate patches of a random-crop step) in one launch, which is the shape a GPU input pipeline needs.
"""
from __future__ import annotations

import numpy as np

from .. import _native as nat


class BoundGenerator:
    '''Generates pairs of lower / upper bounds from a sample space (reference :28-77; host-side sampling).'''

    def __init__(self, sample_space=((0.1, None), (0.3, None), (0.5, None), (0.7, None), (0.9, None), (None, None)), weights=None):
        if (weights is not None) and len(weights) != len(sample_space):
            raise ValueError("`weights` must either be `None` for uniform distribution or have the same length as `sample_space`.")
        self.sample_space = []
        for bound_pair in sample_space:
            if len(bound_pair) != 2:
                raise ValueError("All elements of the sample space must be 2-tuples.")
            bound_pair = list(bound_pair)
            if bound_pair[0] is None:
                bound_pair[0] = 0.0
            if bound_pair[1] is None:
                bound_pair[1] = 1.0
            if bound_pair[0] > bound_pair[1]:
                raise ValueError("For all sample space elements, the lower bound cannot be greater than the upper bound.")
            self.sample_space.append(bound_pair)
        self.sample_space_size = len(self.sample_space)
        self.weights = [1.0 / self.sample_space_size] * self.sample_space_size if weights is None else weights

    def __call__(self):
        i = np.random.choice(self.sample_space_size, p=self.weights)
        return self.sample_space[i]


class BoxFilter:
    '''Returns all bounding boxes that are valid with respect to the defined criteria (reference :79-232).'''

    def __init__(self, check_overlap=True, check_min_area=True, check_degenerate=True, overlap_criterion='center_point',
                 overlap_bounds=(0.3, 1.0), min_area=16, labels_format={'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4},
                 border_pixels='half'):
        if not isinstance(overlap_bounds, (list, tuple, BoundGenerator)):
            raise ValueError("`overlap_bounds` must be either a 2-tuple of scalars or a `BoundGenerator` object.")
        if isinstance(overlap_bounds, (list, tuple)) and (overlap_bounds[0] > overlap_bounds[1]):
            raise ValueError("The lower bound must not be greater than the upper bound.")
        if overlap_criterion not in {'iou', 'area', 'center_point'}:
            raise ValueError("`overlap_criterion` must be one of 'iou', 'area', or 'center_point'.")
        self.overlap_criterion = overlap_criterion
        self.overlap_bounds = overlap_bounds
        self.min_area = min_area
        self.check_overlap = check_overlap
        self.check_min_area = check_min_area
        self.check_degenerate = check_degenerate
        self.labels_format = labels_format
        self.border_pixels = border_pixels

    def _bounds(self):
        return self.overlap_bounds() if isinstance(self.overlap_bounds, BoundGenerator) else self.overlap_bounds

    def masks_batch(self, labels_list, image_heights, image_widths):
        '''Boolean keep mask per label array; image i is validated against (image_heights[i], image_widths[i]).  One set of bounds is
        drawn for the whole batch when `overlap_bounds` is a `BoundGenerator`.'''
        lf = self.labels_format
        cols = [lf['xmin'], lf['ymin'], lf['xmax'], lf['ymax']]
        arrs = [np.asarray(lab) for lab in labels_list]
        counts = [a.shape[0] if a.ndim == 2 else 0 for a in arrs]
        if sum(counts) == 0:
            return [np.zeros((0,), dtype=bool) for _ in arrs]
        boxes = np.concatenate([a[:, cols].astype(np.float64) for a, c in zip(arrs, counts) if c], axis=0)
        box_image = np.repeat(np.arange(len(arrs), dtype=np.int32), counts)
        hw = np.stack([np.asarray(image_heights, dtype=np.float64) if image_heights is not None else np.zeros(len(arrs)),
                       np.asarray(image_widths, dtype=np.float64) if image_widths is not None else np.zeros(len(arrs))], axis=1)
        lower, upper = self._bounds() if self.check_overlap else (0.0, 1.0)
        keep = nat.box_filter(boxes, box_image, hw, self.check_overlap, self.check_min_area, self.check_degenerate,
                              self.overlap_criterion, lower, upper, self.min_area, self.border_pixels).cpu().numpy().astype(bool)
        off = np.cumsum([0] + counts)
        return [keep[off[i]:off[i + 1]] for i in range(len(arrs))]

    def masks_patches(self, labels, patch_ymin, patch_xmin, patch_heights, patch_widths):
        '''Keep masks (T, g) of ONE image's boxes against T candidate patches in one launch: patch t sees the boxes translated into
        its own coordinate system (minus (patch_xmin[t], patch_ymin[t]), evaluated in the dtype of `labels` like the reference's in-place
        `labels[:, [xmin, xmax]] -= patch_xmin`) and tests them against (patch_heights[t], patch_widths[t]).  This is the shape of a
        random-crop step: every trial of a sampling round validated at once.'''
        lf = self.labels_format
        cols = [lf['xmin'], lf['ymin'], lf['xmax'], lf['ymax']]
        labels = np.asarray(labels)
        T = len(patch_ymin)
        g = labels.shape[0] if labels.ndim == 2 else 0
        if g == 0 or T == 0:
            return np.zeros((T, g), dtype=bool)
        px, py = np.asarray(patch_xmin), np.asarray(patch_ymin)
        shift = np.stack([px, py, px, py], axis=1).astype(labels.dtype)
        boxes = (labels[:, cols][None, :, :] - shift[:, None, :]).astype(np.float64).reshape(T * g, 4)
        box_image = np.repeat(np.arange(T, dtype=np.int32), g)
        hw = np.stack([np.asarray(patch_heights, dtype=np.float64), np.asarray(patch_widths, dtype=np.float64)], axis=1)
        lower, upper = self._bounds() if self.check_overlap else (0.0, 1.0)
        keep = nat.box_filter(boxes, box_image, hw, self.check_overlap, self.check_min_area, self.check_degenerate,
                              self.overlap_criterion, lower, upper, self.min_area, self.border_pixels).cpu().numpy().astype(bool)
        return keep.reshape(T, g)

    def filter_batch(self, labels_list, image_heights, image_widths):
        '''`[labels[mask] for each image]` in one launch.'''
        masks = self.masks_batch(labels_list, image_heights, image_widths)
        return [np.copy(np.asarray(lab))[m] if np.asarray(lab).ndim == 2 else np.copy(np.asarray(lab)) for lab, m in zip(labels_list, masks)]

    def __call__(self, labels, image_height=None, image_width=None):
        '''Reference :147-232: the rows of `labels` that pass all checks (a copy).'''
        if self.check_overlap and (image_height is None or image_width is None):
            raise ValueError("`image_height` and `image_width` are required when `check_overlap` is set")
        return self.filter_batch([labels], [image_height if image_height is not None else 0], [image_width if image_width is not None else 0])[0]


class ImageValidator:
    '''Returns `True` if a given minimum number of bounding boxes meets given overlap requirements with an image of a given
    height and width (reference :234-322).'''

    def __init__(self, overlap_criterion='center_point', bounds=(0.3, 1.0), n_boxes_min=1,
                 labels_format={'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}, border_pixels='half'):
        if not ((isinstance(n_boxes_min, int) and n_boxes_min > 0) or n_boxes_min == 'all'):
            raise ValueError("`n_boxes_min` must be a positive integer or 'all'.")
        self.overlap_criterion = overlap_criterion
        self.bounds = bounds
        self.n_boxes_min = n_boxes_min
        self.labels_format = labels_format
        self.border_pixels = border_pixels
        self.box_filter = BoxFilter(check_overlap=True, check_min_area=False, check_degenerate=False,
                                    overlap_criterion=self.overlap_criterion, overlap_bounds=self.bounds,
                                    labels_format=self.labels_format, border_pixels=self.border_pixels)

    def validate_batch(self, labels_list, image_heights, image_widths):
        '''One boolean per image / patch.'''
        self.box_filter.overlap_bounds = self.bounds
        self.box_filter.labels_format = self.labels_format
        masks = self.box_filter.masks_batch(labels_list, image_heights, image_widths)
        if self.n_boxes_min == 'all':
            return [bool(m.sum() == len(m)) for m in masks]
        return [bool(m.sum() >= self.n_boxes_min) for m in masks]

    def validate_patches(self, labels, patch_ymin, patch_xmin, patch_heights, patch_widths):
        '''One boolean per candidate patch of ONE image (see `BoxFilter.masks_patches`), one launch for all of them.'''
        self.box_filter.overlap_bounds = self.bounds
        self.box_filter.labels_format = self.labels_format
        masks = self.box_filter.masks_patches(labels, patch_ymin, patch_xmin, patch_heights, patch_widths)
        n_valid = masks.sum(axis=1)
        if self.n_boxes_min == 'all':
            return n_valid == masks.shape[1]
        return n_valid >= self.n_boxes_min

    def __call__(self, labels, image_height, image_width):
        '''Reference :286-322.'''
        return self.validate_batch([labels], [image_height], [image_width])[0]
