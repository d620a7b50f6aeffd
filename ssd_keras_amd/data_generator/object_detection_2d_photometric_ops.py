"""Photometric image ops -- drop-in for data_generator/object_detection_2d_photometric_ops.py:23-480.

Same classes, arguments, random draws (one `np.random.uniform(0, 1)` per random op and call, a second draw only when the op fires:
`p >= 1 - prob`) and return conventions (`image` or `(image, labels)`); the pixel arithmetic runs on the GPU
(csrc/ssdhip_image.hip through `_image_ops`), one NumPy image at a time as the reference's callers hand them over, or a whole CUDA
batch through `SSDPhotometricDistortions.distort_batch` (data_augmentation_chain_original_ssd.py).  What the reference gets from
OpenCV -- cv2.cvtColor, cv2.LUT, cv2.equalizeHist -- is restated from OpenCV's published algorithms (8-bit HSV with H in [0, 180),
float32 HSV with H in [0, 360)); the NumPy arithmetic around it follows the reference's expressions for the array's dtype.

Two deliberate differences: `Gamma.__call__` works (the reference's reads an undefined global `table`, :359, and raises NameError),
and `ConvertColor(current='HSV', to='GRAY')` raises ValueError (the reference asks cv2 for a constant that does not exist, :54)."""
from __future__ import annotations

import numpy as np

from . import _image_ops as iop


def _ret(image, labels):
    return image if labels is None else (image, labels)


def _fires(prob):
    """One uniform draw per call of a random op; it acts when the draw reaches 1 - prob (so prob = 0 never acts: the draw is < 1)."""
    return np.random.uniform(0, 1) >= (1.0 - prob)


def _check_range(lower, upper):
    if lower >= upper:
        raise ValueError("`upper` must be greater than `lower`.")


def _assign(image, result):
    """The reference's `image[:, :, c] = ...` ops modify the caller's array: so do these."""
    if isinstance(image, np.ndarray) and image.flags.writeable and image.dtype == result.dtype and image.shape == result.shape:
        image[...] = result
        return image
    return result


class ConvertColor:
    """RGB <-> HSV / grey (reference :23-59, a wrapper around cv2.cvtColor)."""

    def __init__(self, current='RGB', to='HSV', keep_3ch=True):
        if not ((current in {'RGB', 'HSV'}) and (to in {'RGB', 'HSV', 'GRAY'})):
            raise NotImplementedError
        self.current = current
        self.to = to
        self.keep_3ch = keep_3ch

    def __call__(self, image, labels=None):
        if self.current == 'RGB' and self.to == 'HSV':
            image = iop.run(image, [("rgb2hsv", 0)])
        elif self.current == 'RGB' and self.to == 'GRAY':
            image = iop.run(image, [("rgb2gray", 0)])
            if not self.keep_3ch:
                image = image[..., 0]
        elif self.current == 'HSV' and self.to == 'RGB':
            image = iop.run(image, [("hsv2rgb", 0)])
        elif self.current == 'HSV' and self.to == 'GRAY':
            raise ValueError("OpenCV has no HSV -> grey conversion (the reference fails here too)")
        return _ret(image, labels)


class ConvertDataType:
    """uint8 <-> float32 (reference :62-85): `np.round(image).astype(np.uint8)` / `image.astype(np.float32)`."""

    def __init__(self, to='uint8'):
        if not (to == 'uint8' or to == 'float32'):
            raise ValueError("`to` can be either of 'uint8' or 'float32'.")
        self.to = to

    def __call__(self, image, labels=None):
        image = iop.run(image, [("to_u8" if self.to == 'uint8' else "to_f32", 0)])
        return _ret(image, labels)


class ConvertTo3Channels:
    """1- and 4-channel images -> 3 channels (reference :88-107); no arithmetic."""

    def __init__(self):
        pass

    def __call__(self, image, labels=None):
        if image.ndim == 2:
            image = np.stack([image] * 3, axis=-1)
        elif image.ndim == 3:
            if image.shape[2] == 1:
                image = np.concatenate([image] * 3, axis=-1)
            elif image.shape[2] == 4:
                image = image[:, :, :3]
        return _ret(image, labels)


class Hue:
    """`image[:, :, 0] = (image[:, :, 0] + delta) % 180.0` on HSV images (reference :110-132)."""

    def __init__(self, delta):
        if not (-180 <= delta <= 180):
            raise ValueError("`delta` must be in the closed interval `[-180, 180]`.")
        self.delta = delta

    def __call__(self, image, labels=None):
        image = _assign(image, iop.run(image, [("hue", self.delta)]))
        return _ret(image, labels)


class RandomHue:
    def __init__(self, max_delta=18, prob=0.5):
        if not (0 <= max_delta <= 180):
            raise ValueError("`max_delta` must be in the closed interval `[0, 180]`.")
        self.max_delta = max_delta
        self.prob = prob
        self.change_hue = Hue(delta=0)

    def draw(self):
        """The op's random draws in the reference's order -> the program steps of this call."""
        if not _fires(self.prob):
            return []
        self.change_hue.delta = np.random.uniform(-self.max_delta, self.max_delta)
        return [("hue", self.change_hue.delta)]

    def __call__(self, image, labels=None):
        steps = self.draw()
        if steps:
            return self.change_hue(image, labels)
        return _ret(image, labels)


class Saturation:
    """`image[:, :, 1] = np.clip(image[:, :, 1] * factor, 0, 255)` on HSV images (reference :166-188)."""

    def __init__(self, factor):
        if factor <= 0.0:
            raise ValueError("It must be `factor > 0`.")
        self.factor = factor

    def __call__(self, image, labels=None):
        image = _assign(image, iop.run(image, [("saturation", self.factor)]))
        return _ret(image, labels)


class RandomSaturation:
    def __init__(self, lower=0.3, upper=2.0, prob=0.5):
        _check_range(lower, upper)
        self.lower = lower
        self.upper = upper
        self.prob = prob
        self.change_saturation = Saturation(factor=1.0)

    def draw(self):
        if not _fires(self.prob):
            return []
        self.change_saturation.factor = np.random.uniform(self.lower, self.upper)
        return [("saturation", self.change_saturation.factor)]

    def __call__(self, image, labels=None):
        if self.draw():
            return self.change_saturation(image, labels)
        return _ret(image, labels)


class Brightness:
    """`np.clip(image + delta, 0, 255)` on RGB images (reference :225-245)."""

    def __init__(self, delta):
        self.delta = delta

    def __call__(self, image, labels=None):
        image = iop.run(image, [("brightness", self.delta)])
        return _ret(image, labels)


class RandomBrightness:
    def __init__(self, lower=-84, upper=84, prob=0.5):
        _check_range(lower, upper)
        self.lower = float(lower)
        self.upper = float(upper)
        self.prob = prob
        self.change_brightness = Brightness(delta=0)

    def draw(self):
        if not _fires(self.prob):
            return []
        self.change_brightness.delta = np.random.uniform(self.lower, self.upper)
        return [("brightness", self.change_brightness.delta)]

    def __call__(self, image, labels=None):
        if self.draw():
            return self.change_brightness(image, labels)
        return _ret(image, labels)


class Contrast:
    """`np.clip(127.5 + factor * (image - 127.5), 0, 255)` on RGB images (reference :281-303)."""

    def __init__(self, factor):
        if factor <= 0.0:
            raise ValueError("It must be `factor > 0`.")
        self.factor = factor

    def __call__(self, image, labels=None):
        image = iop.run(image, [("contrast", self.factor)])
        return _ret(image, labels)


class RandomContrast:
    def __init__(self, lower=0.5, upper=1.5, prob=0.5):
        _check_range(lower, upper)
        self.lower = lower
        self.upper = upper
        self.prob = prob
        self.change_contrast = Contrast(factor=1.0)

    def draw(self):
        if not _fires(self.prob):
            return []
        self.change_contrast.factor = np.random.uniform(self.lower, self.upper)
        return [("contrast", self.change_contrast.factor)]

    def __call__(self, image, labels=None):
        if self.draw():
            return self.change_contrast(image, labels)
        return _ret(image, labels)


class Gamma:
    """Gamma correction of uint8 RGB images through the reference's 256-entry table (:340-362; see the module docstring)."""

    def __init__(self, gamma):
        if gamma <= 0.0:
            raise ValueError("It must be `gamma > 0`.")
        self.gamma = gamma
        self.gamma_inv = 1.0 / gamma
        self.table = np.array([((i / 255.0) ** self.gamma_inv) * 255 for i in np.arange(0, 256)]).astype("uint8")

    def __call__(self, image, labels=None):
        image = iop.lut(image, self.table, 0xff)
        return _ret(image, labels)


class RandomGamma:
    def __init__(self, lower=0.25, upper=2.0, prob=0.5):
        _check_range(lower, upper)
        self.lower = lower
        self.upper = upper
        self.prob = prob

    def __call__(self, image, labels=None):
        if not _fires(self.prob):
            return _ret(image, labels)
        return Gamma(gamma=np.random.uniform(self.lower, self.upper))(image, labels)


class HistogramEqualization:
    """`image[:, :, 2] = cv2.equalizeHist(image[:, :, 2])` on uint8 HSV images (reference :397-410)."""

    def __init__(self):
        pass

    def __call__(self, image, labels=None):
        image = _assign(image, iop.equalize_channel(image, 2))
        return _ret(image, labels)


class RandomHistogramEqualization:
    def __init__(self, prob=0.5):
        self.prob = prob
        self.equalize = HistogramEqualization()

    def __call__(self, image, labels=None):
        return self.equalize(image, labels) if _fires(self.prob) else _ret(image, labels)


class ChannelSwap:
    """`image[:, :, order]` (reference :438-454); no arithmetic."""

    def __init__(self, order):
        self.order = order

    def __call__(self, image, labels=None):
        image = image[:, :, self.order]
        return _ret(image, labels)


class RandomChannelSwap:
    def __init__(self, prob=0.5):
        self.prob = prob
        # every permutation of the three channels but the identity, in the reference's order (:470-472)
        self.permutations = ((0, 2, 1),
                             (1, 0, 2), (1, 2, 0),
                             (2, 0, 1), (2, 1, 0))
        self.swap_channels = ChannelSwap(order=(0, 1, 2))

    def draw(self):
        if not _fires(self.prob):
            return []
        self.swap_channels.order = self.permutations[np.random.randint(5)]
        return [("swap", iop.swap_code(self.swap_channels.order))]

    def __call__(self, image, labels=None):
        if self.draw():
            return self.swap_channels(image, labels)
        return _ret(image, labels)
