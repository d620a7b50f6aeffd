"""Drop-in for the reference's data_generator/object_detection_2d_patch_sampling_ops.py: patch
sampling for data augmentation -- `PatchCoordinateGenerator` :24-178, `CropPad` :180-339, `Crop` :341-382, `Pad` :384-421,
`RandomPatch` :423-581, `RandomPatchInf` :583-742, `RandomMaxCropFixedAR` :744-821, `RandomPadFixedAR` :823-881.

What the reference spends its time on here is the search for a valid patch: up to `n_trials_max` (50 in SSDRandomCrop) candidate
patches per round, each validated by an `ImageValidator` call (an IoU / overlap test of every ground truth box against the patch) in
a Python loop.  Here one round is ONE launch: all candidate patches of the round are generated first -- consuming NumPy's global
random stream in exactly the reference's order --, `ImageValidator.validate_patches` tests every (patch, box) pair on the GPU
(`ssdhip_box_filter`), the first valid candidate wins, and the random stream is put back to where the reference would have
stopped drawing (the round is replayed from its saved state up to the winning trial), so a seeded run yields the same patches,
labels and subsequent random numbers as the reference.  Images stay host arrays as in the reference's generator: a crop is a slice.

Reference quirk not replicated: `CropPad.__call__` does `labels = np.copy(labels)` before testing `labels is None`, so calling it
(or any op built on it) without labels raises IndexError; here the image-only call returns the image (and the inverter).
"""
from __future__ import annotations

import numpy as np

from . import _image_ops as iop

from .object_detection_2d_image_boxes_validation_utils import BoundGenerator, BoxFilter, ImageValidator

_DEFAULT_FORMAT = {'class_id': 0, 'xmin': 1, 'ymin': 2, 'xmax': 3, 'ymax': 4}


class PatchCoordinateGenerator:
    '''Generates random patch coordinates that meet specified requirements (reference :24-178; host-side sampling, the draws
    are made in the reference's order: height, width / aspect ratio, then ymin, then xmin).'''

    def __init__(self, img_height=None, img_width=None, must_match='h_w', min_scale=0.3, max_scale=1.0, scale_uniformly=False,
                 min_aspect_ratio=0.5, max_aspect_ratio=2.0, patch_ymin=None, patch_xmin=None, patch_height=None, patch_width=None,
                 patch_aspect_ratio=None):
        if must_match not in {'h_w', 'h_ar', 'w_ar'}:
            raise ValueError("`must_match` must be either of 'h_w', 'h_ar' and 'w_ar'.")
        if min_scale >= max_scale:
            raise ValueError("It must be `min_scale < max_scale`.")
        if min_aspect_ratio >= max_aspect_ratio:
            raise ValueError("It must be `min_aspect_ratio < max_aspect_ratio`.")
        if scale_uniformly and not ((patch_height is None) and (patch_width is None)):
            raise ValueError("If `scale_uniformly == True`, `patch_height` and `patch_width` must both be `None`.")
        self.img_height = img_height
        self.img_width = img_width
        self.must_match = must_match
        self.min_scale = min_scale
        self.max_scale = max_scale
        self.scale_uniformly = scale_uniformly
        self.min_aspect_ratio = min_aspect_ratio
        self.max_aspect_ratio = max_aspect_ratio
        self.patch_ymin = patch_ymin
        self.patch_xmin = patch_xmin
        self.patch_height = patch_height
        self.patch_width = patch_width
        self.patch_aspect_ratio = patch_aspect_ratio

    def _scaled(self, fixed, extent):
        return int(np.random.uniform(self.min_scale, self.max_scale) * extent) if fixed is None else fixed

    def _aspect(self):
        if self.patch_aspect_ratio is None:
            return np.random.uniform(self.min_aspect_ratio, self.max_aspect_ratio)
        return self.patch_aspect_ratio

    @staticmethod
    def _position(fixed, extent, size):
        if fixed is not None:
            return fixed
        room = extent - size                       # >= 0: positions of a crop inside the image; < 0: of the image on the canvas
        return np.random.randint(0, room + 1) if room >= 0 else np.random.randint(room, 1)

    def __call__(self):
        '''Returns `(ymin, xmin, height, width)` of the generated patch.'''
        if self.must_match == 'h_w':
            if self.scale_uniformly:
                factor = np.random.uniform(self.min_scale, self.max_scale)
                height, width = int(factor * self.img_height), int(factor * self.img_width)
            else:
                height = self._scaled(self.patch_height, self.img_height)
                width = self._scaled(self.patch_width, self.img_width)
        elif self.must_match == 'h_ar':
            height = self._scaled(self.patch_height, self.img_height)
            width = int(height * self._aspect())
        else:
            width = self._scaled(self.patch_width, self.img_width)
            height = int(width / self._aspect())
        ymin = self._position(self.patch_ymin, self.img_height, height)
        xmin = self._position(self.patch_xmin, self.img_width, width)
        return (ymin, xmin, height, width)


def _identity_inverter(labels):
    return labels


class CropPad:
    '''Crops and/or pads an image deterministically: the output is the `patch_height` x `patch_width` window whose top left corner
    sits at (`patch_ymin`, `patch_xmin`) of the image's coordinate system; what lies outside the image is `background`
    (reference :180-339).'''

    def __init__(self, patch_ymin, patch_xmin, patch_height, patch_width, clip_boxes=True, box_filter=None, background=(0, 0, 0),
                 labels_format=_DEFAULT_FORMAT):
        if not (isinstance(box_filter, BoxFilter) or box_filter is None):
            raise ValueError("`box_filter` must be either `None` or a `BoxFilter` object.")
        self.patch_height = patch_height
        self.patch_width = patch_width
        self.patch_ymin = patch_ymin
        self.patch_xmin = patch_xmin
        self.clip_boxes = clip_boxes
        self.box_filter = box_filter
        self.background = background
        self.labels_format = labels_format

    def _window(self, image):
        '''The patch as a fresh uint8 array: background canvas + the part of the image the window covers.'''
        img_height, img_width = image.shape[:2]
        top, left, height, width = self.patch_ymin, self.patch_xmin, self.patch_height, self.patch_width
        if isinstance(image, iop.GeoImage):              # a lazy image (SSDDataAugmentation.augment_batch): record, do not copy
            return image.window(top, left, height, width, self.background)
        if image.ndim == 3:
            canvas = np.zeros((height, width, 3), dtype=np.uint8)
            canvas[:, :] = self.background
        else:
            canvas = np.zeros((height, width), dtype=np.uint8)
            canvas[:, :] = self.background[0]
        y0, y1 = max(top, 0), min(top + height, img_height)            # the image rows / columns inside the window
        x0, x1 = max(left, 0), min(left + width, img_width)
        if y1 > y0 and x1 > x0:
            canvas[y0 - top:y1 - top, x0 - left:x1 - left] = image[y0:y1, x0:x1]
        return canvas

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        if (self.patch_ymin > img_height) or (self.patch_xmin > img_width):
            raise ValueError("The given patch doesn't overlap with the input image.")
        lf = self.labels_format
        xmin, ymin, xmax, ymax = lf['xmin'], lf['ymin'], lf['xmax'], lf['ymax']
        top, left = self.patch_ymin, self.patch_xmin
        patch = self._window(image)

        def inverter(predictions):                  # predictions carry one more leading column than labels (reference :310-315)
            predictions = np.copy(predictions)
            predictions[:, [ymin + 1, ymax + 1]] += top
            predictions[:, [xmin + 1, xmax + 1]] += left
            return predictions

        if labels is None:
            return (patch, inverter) if return_inverter else patch
        labels = np.copy(labels)
        labels[:, [ymin, ymax]] -= top
        labels[:, [xmin, xmax]] -= left
        if self.box_filter is not None:
            self.box_filter.labels_format = self.labels_format
            labels = self.box_filter(labels=labels, image_height=self.patch_height, image_width=self.patch_width)
        if self.clip_boxes:
            labels[:, [ymin, ymax]] = np.clip(labels[:, [ymin, ymax]], a_min=0, a_max=self.patch_height - 1)
            labels[:, [xmin, xmax]] = np.clip(labels[:, [xmin, xmax]], a_min=0, a_max=self.patch_width - 1)
        return (patch, labels, inverter) if return_inverter else (patch, labels)


class Crop:
    '''Crops off the specified numbers of pixels from the borders of images (reference :341-382).'''

    def __init__(self, crop_top, crop_bottom, crop_left, crop_right, clip_boxes=True, box_filter=None, labels_format=_DEFAULT_FORMAT):
        self.crop_top = crop_top
        self.crop_bottom = crop_bottom
        self.crop_left = crop_left
        self.crop_right = crop_right
        self.clip_boxes = clip_boxes
        self.box_filter = box_filter
        self.labels_format = labels_format
        self.crop = CropPad(patch_ymin=self.crop_top, patch_xmin=self.crop_left, patch_height=None, patch_width=None,
                            clip_boxes=self.clip_boxes, box_filter=self.box_filter, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        self.crop.patch_height = img_height - self.crop_top - self.crop_bottom
        self.crop.patch_width = img_width - self.crop_left - self.crop_right
        self.crop.labels_format = self.labels_format
        return self.crop(image, labels, return_inverter)


class Pad:
    '''Pads images by the specified numbers of pixels on each side (reference :384-421).'''

    def __init__(self, pad_top, pad_bottom, pad_left, pad_right, background=(0, 0, 0), labels_format=_DEFAULT_FORMAT):
        self.pad_top = pad_top
        self.pad_bottom = pad_bottom
        self.pad_left = pad_left
        self.pad_right = pad_right
        self.background = background
        self.labels_format = labels_format
        self.pad = CropPad(patch_ymin=-self.pad_top, patch_xmin=-self.pad_left, patch_height=None, patch_width=None, clip_boxes=False,
                           box_filter=None, background=self.background, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        self.pad.patch_height = img_height + self.pad_top + self.pad_bottom
        self.pad.patch_width = img_width + self.pad_left + self.pad_right
        self.pad.labels_format = self.labels_format
        return self.pad(image, labels, return_inverter)


def _unaltered(image, labels, return_inverter, inverter):
    if labels is None:
        return (image, inverter) if return_inverter else image
    return (image, labels, inverter) if return_inverter else (image, labels)


class _PatchSearch:
    '''The search both random-patch ops share: one round = up to `n_trials_max` candidate patches, the first valid one is cut.'''

    def _prepare(self, image):
        img_height, img_width = image.shape[:2]
        self.patch_coord_generator.img_height = img_height
        self.patch_coord_generator.img_width = img_width
        if self.image_validator is not None:
            self.image_validator.labels_format = self.labels_format
        self.sample_patch.labels_format = self.labels_format

    def _cut(self, coords, image, labels, return_inverter):
        sp = self.sample_patch
        sp.patch_ymin, sp.patch_xmin, sp.patch_height, sp.patch_width = coords
        return sp(image, labels, return_inverter)

    def _aspect_ok(self, coords):
        return True

    def _round(self, image, labels, return_inverter):
        '''Returns the cut patch of the first acceptable trial of this round, or None when all `n_trials_max` trials fail.'''
        n_trials = max(1, self.n_trials_max)
        gen = self.patch_coord_generator
        validator = self.image_validator
        if labels is None or validator is None:
            for _ in range(n_trials):                                    # any patch (of an acceptable shape) will do
                coords = gen()
                if self._aspect_ok(coords):
                    return self._cut(coords, image, labels, return_inverter)
            return None
        if isinstance(validator.bounds, BoundGenerator):
            # the validator itself draws random bounds per call: the stream position depends on every validation, so the trials
            # are validated one by one like the reference does
            for _ in range(n_trials):
                coords = gen()
                if self._aspect_ok(coords) and validator.validate_patches(labels, [coords[0]], [coords[1]], [coords[2]], [coords[3]])[0]:
                    return self._cut(coords, image, labels, return_inverter)
            return None
        # ---- every trial of the round in one launch -------------------------------------------------------------------
        start = np.random.get_state()
        trials = [gen() for _ in range(n_trials)]
        shaped = [t for t, c in enumerate(trials) if self._aspect_ok(c)]
        winner = None
        if shaped:
            ys, xs, hs, ws = (np.array([trials[t][k] for t in shaped]) for k in range(4))
            valid = validator.validate_patches(labels, ys, xs, hs, ws)
            hits = np.flatnonzero(valid)
            if hits.size:
                winner = shaped[int(hits[0])]
        if winner is None:
            return None                                                  # the stream has consumed all trials, as in the reference
        np.random.set_state(start)                                       # replay up to the winner: the stream stops where the
        for _ in range(winner + 1):                                      # reference's loop would have returned
            coords = gen()
        return self._cut(coords, image, labels, return_inverter)


class RandomPatch(_PatchSearch):
    '''Randomly samples a patch from an image; may fail to produce one, in which case it returns `None`s (`can_fail`) or the
    unaltered input (reference :423-581).'''

    def __init__(self, patch_coord_generator, box_filter=None, image_validator=None, n_trials_max=3, clip_boxes=True, prob=1.0,
                 background=(0, 0, 0), can_fail=False, labels_format=_DEFAULT_FORMAT):
        if not isinstance(patch_coord_generator, PatchCoordinateGenerator):
            raise ValueError("`patch_coord_generator` must be an instance of `PatchCoordinateGenerator`.")
        if not (isinstance(image_validator, ImageValidator) or image_validator is None):
            raise ValueError("`image_validator` must be either `None` or an `ImageValidator` object.")
        self.patch_coord_generator = patch_coord_generator
        self.box_filter = box_filter
        self.image_validator = image_validator
        self.n_trials_max = n_trials_max
        self.clip_boxes = clip_boxes
        self.prob = prob
        self.background = background
        self.can_fail = can_fail
        self.labels_format = labels_format
        self.sample_patch = CropPad(patch_ymin=None, patch_xmin=None, patch_height=None, patch_width=None, clip_boxes=self.clip_boxes,
                                    box_filter=self.box_filter, background=self.background, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        p = np.random.uniform(0, 1)
        if p < (1.0 - self.prob):
            return _unaltered(image, labels, return_inverter, _identity_inverter)
        self._prepare(image)
        out = self._round(image, labels, return_inverter)
        if out is not None:
            return out
        if self.can_fail:
            n_out = (1 if labels is None else 2) + (1 if return_inverter else 0)
            return None if n_out == 1 else (None,) * n_out
        return _unaltered(image, labels, return_inverter, None)


class RandomPatchInf(_PatchSearch):
    '''Randomly samples a patch from an image, round after round until a valid patch is found or -- with probability
    `1 - prob` per round -- the input is returned unaltered; a `bound_generator` draws new validator bounds every round
    (reference :583-742).'''

    def __init__(self, patch_coord_generator, box_filter=None, image_validator=None, bound_generator=None, n_trials_max=50,
                 clip_boxes=True, prob=0.857, background=(0, 0, 0), labels_format=_DEFAULT_FORMAT):
        if not isinstance(patch_coord_generator, PatchCoordinateGenerator):
            raise ValueError("`patch_coord_generator` must be an instance of `PatchCoordinateGenerator`.")
        if not (isinstance(image_validator, ImageValidator) or image_validator is None):
            raise ValueError("`image_validator` must be either `None` or an `ImageValidator` object.")
        if not (isinstance(bound_generator, BoundGenerator) or bound_generator is None):
            raise ValueError("`bound_generator` must be either `None` or a `BoundGenerator` object.")
        self.patch_coord_generator = patch_coord_generator
        self.box_filter = box_filter
        self.image_validator = image_validator
        self.bound_generator = bound_generator
        self.n_trials_max = n_trials_max
        self.clip_boxes = clip_boxes
        self.prob = prob
        self.background = background
        self.labels_format = labels_format
        self.sample_patch = CropPad(patch_ymin=None, patch_xmin=None, patch_height=None, patch_width=None, clip_boxes=self.clip_boxes,
                                    box_filter=self.box_filter, background=self.background, labels_format=self.labels_format)

    def _aspect_ok(self, coords):
        gen = self.patch_coord_generator
        return gen.min_aspect_ratio <= coords[3] / coords[2] <= gen.max_aspect_ratio

    def __call__(self, image, labels=None, return_inverter=False):
        self._prepare(image)
        while True:
            p = np.random.uniform(0, 1)
            if p < (1.0 - self.prob):
                return _unaltered(image, labels, return_inverter, _identity_inverter)
            if not ((self.image_validator is None) or (self.bound_generator is None)):
                self.image_validator.bounds = self.bound_generator()
            out = self._round(image, labels, return_inverter)
            if out is not None:
                return out


class RandomMaxCropFixedAR:
    '''Crops the largest possible patch of a given fixed aspect ratio from an image (reference :744-821).'''

    def __init__(self, patch_aspect_ratio, box_filter=None, image_validator=None, n_trials_max=3, clip_boxes=True,
                 labels_format=_DEFAULT_FORMAT):
        self.patch_aspect_ratio = patch_aspect_ratio
        self.box_filter = box_filter
        self.image_validator = image_validator
        self.n_trials_max = n_trials_max
        self.clip_boxes = clip_boxes
        self.labels_format = labels_format
        self.random_patch = RandomPatch(patch_coord_generator=PatchCoordinateGenerator(), box_filter=self.box_filter,
                                        image_validator=self.image_validator, n_trials_max=self.n_trials_max,
                                        clip_boxes=self.clip_boxes, prob=1.0, can_fail=False, labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        if img_width / img_height < self.patch_aspect_ratio:
            patch_width = img_width
            patch_height = int(round(patch_width / self.patch_aspect_ratio))
        else:
            patch_height = img_height
            patch_width = int(round(patch_height * self.patch_aspect_ratio))
        self.random_patch.patch_coord_generator = PatchCoordinateGenerator(img_height=img_height, img_width=img_width, must_match='h_w',
                                                                           patch_height=patch_height, patch_width=patch_width)
        self.random_patch.labels_format = self.labels_format
        return self.random_patch(image, labels, return_inverter)


class RandomPadFixedAR:
    '''Adds the minimal padding that turns an image into a patch of the given fixed aspect ratio containing the entire image
    (reference :823-881).'''

    def __init__(self, patch_aspect_ratio, background=(0, 0, 0), labels_format=_DEFAULT_FORMAT):
        self.patch_aspect_ratio = patch_aspect_ratio
        self.background = background
        self.labels_format = labels_format
        self.random_patch = RandomPatch(patch_coord_generator=PatchCoordinateGenerator(), box_filter=None, image_validator=None,
                                        n_trials_max=1, clip_boxes=False, background=self.background, prob=1.0,
                                        labels_format=self.labels_format)

    def __call__(self, image, labels=None, return_inverter=False):
        img_height, img_width = image.shape[:2]
        if img_width < img_height:
            patch_height = img_height
            patch_width = int(round(patch_height * self.patch_aspect_ratio))
        else:
            patch_width = img_width
            patch_height = int(round(patch_width / self.patch_aspect_ratio))
        self.random_patch.patch_coord_generator = PatchCoordinateGenerator(img_height=img_height, img_width=img_width, must_match='h_w',
                                                                           patch_height=patch_height, patch_width=patch_width)
        self.random_patch.labels_format = self.labels_format
        return self.random_patch(image, labels, return_inverter)
