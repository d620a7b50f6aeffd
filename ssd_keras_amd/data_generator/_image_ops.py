"""Host side of the image half of the augmentation: per-image programs for `ssdhip_image_program`, tap tables for
`ssdhip_image_resize_u8`, histogram equalisation and look-up tables -- the pixel work itself runs in csrc/ssdhip_image.hip.
Images come in as the reference's ops take them (one NumPy array (H, W, 3), uint8 or float32) or as a CUDA tensor (B, H, W, 3);
NumPy in -> NumPy out (one upload, one download), tensor in -> tensor out (nothing crosses PCIe)."""
from __future__ import annotations

import numpy as np

from .. import _native as nat

U8, F32, F64 = "uint8", "float32", "float64"
_NP = {U8: np.uint8, F32: np.float32, F64: np.float64}


def end_dtype(dtype, steps):
    """The dtype a program ends in, following NumPy's rules for the reference's expressions (see csrc/ssdhip_image.hip)."""
    tag = str(np.dtype(dtype))
    if tag not in _NP:
        raise TypeError("images are uint8, float32 or float64 arrays, not %s" % tag)
    for name, _ in steps:
        if name == "to_f32":
            tag = F32
        elif name == "to_u8":
            tag = U8
        elif name in ("brightness", "contrast"):
            tag = F32 if tag == F32 else F64
        elif name in ("rgb2hsv", "hsv2rgb", "rgb2gray") and tag == F64:
            raise TypeError("colour conversions take uint8 or float32 images (cv2.cvtColor does not take float64)")
    return tag


def encode(steps):
    """[(name, argument), ...] -> (ops int32[16], args float64[16])."""
    if len(steps) > nat.IMG_PROG - 1:
        raise ValueError("a program holds at most %d steps" % (nat.IMG_PROG - 1))
    ops = np.zeros(nat.IMG_PROG, dtype=np.int32)
    args = np.zeros(nat.IMG_PROG, dtype=np.float64)
    for k, (name, arg) in enumerate(steps):
        ops[k] = nat.IMG_OPS[name]
        args[k] = float(arg)
    return ops, args


def swap_code(order):
    o = tuple(int(v) for v in order)
    if len(o) != 3 or any(v not in (0, 1, 2) for v in o):
        raise ValueError("a channel order is three indices out of (0, 1, 2)")
    return o[0] + 4 * o[1] + 16 * o[2]


def _torch_dtype(tag):
    import torch
    return {U8: torch.uint8, F32: torch.float32, F64: torch.float64}[tag]


def run(image, steps):
    """One program on one image (NumPy (H, W, 3)) or the same program on every image of a CUDA batch (B, H, W, 3)."""
    import torch
    if isinstance(image, np.ndarray):
        if image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("expected an (H, W, 3) image")
        out_tag = end_dtype(image.dtype, steps)
        ops, args = encode(steps)
        dev = nat.to_device(np.ascontiguousarray(image)[None])
        out = nat.image_program(dev, ops[None], args[None], _torch_dtype(out_tag))
        return out[0].cpu().numpy()
    out_tag = end_dtype(str(image.dtype).replace("torch.", ""), steps)
    ops, args = encode(steps)
    b = image.shape[0]
    return nat.image_program(image.contiguous(), np.repeat(ops[None], b, 0), np.repeat(args[None], b, 0), _torch_dtype(out_tag))


def run_batch(images, programs):
    """Per-image programs on a CUDA uint8 / float32 batch (B, H, W, 3): `programs[i]` is image i's step list; all must end in one dtype."""
    tags = {end_dtype(str(images.dtype).replace("torch.", ""), p) for p in programs}
    if len(tags) != 1:
        raise ValueError("the programs of a batch must end in the same dtype")
    enc = [encode(p) for p in programs]
    return nat.image_program(images.contiguous(), np.stack([e[0] for e in enc]), np.stack([e[1] for e in enc]), _torch_dtype(tags.pop()))


# ---- lazy geometry: the geometric ops of a chain recorded instead of executed ----------------------------------------------------------
class GeoImage:
    """An image that EXISTS ONLY AS INDEX MAPS into an image resident on the device: what the geometric ops of the augmentation chain
    (expansion / crop = a window with a background, flips = reversed slices, the final resize) do to it is recorded as two index arrays
    -- `ys[i]` / `xs[j]` = the row / column of the original image that row i / column j of the current virtual image shows, -1 = a
    position filled with `background` -- and executed once, for a whole batch, by `ssdhip_image_resize_gather_u8`.  Quacks like the
    (H, W, 3) uint8 array the ops expect: `shape`, `ndim`, `dtype`, slicing with plain / reversed slices."""

    ndim = 3
    dtype = np.dtype(np.uint8)

    def __init__(self, ys, xs, background=None, resized=None):
        self.ys, self.xs, self.background, self.resized = ys, xs, background, resized

    @classmethod
    def of(cls, height, width):
        return cls(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64))

    @property
    def shape(self):
        if self.resized is not None:
            return (self.resized[0], self.resized[1], 3)
        return (len(self.ys), len(self.xs), 3)

    def __getitem__(self, key):
        if self.resized is not None:
            raise TypeError("a resized lazy image is final")
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) > 2 or not all(isinstance(k, slice) for k in key):
            raise TypeError("lazy images take row / column slices only")
        ys = self.ys[key[0]]
        xs = self.xs[key[1]] if len(key) > 1 else self.xs
        return GeoImage(ys, xs, self.background)

    def window(self, top, left, height, width, background):
        """CropPad's window (data_generator/object_detection_2d_patch_sampling_ops.py: the patch, background where it leaves the image)."""
        h, w = len(self.ys), len(self.xs)
        rows = np.arange(top, top + height)
        cols = np.arange(left, left + width)
        ys = np.where((rows >= 0) & (rows < h), self.ys[np.clip(rows, 0, h - 1)], -1)
        xs = np.where((cols >= 0) & (cols < w), self.xs[np.clip(cols, 0, w - 1)], -1)
        bg = tuple(int(v) for v in background)
        padded = bool(top < 0 or left < 0 or top + height > h or left + width > w)     # THIS window leaves the (virtual) image
        had = bool((self.ys < 0).any() or (self.xs < 0).any())
        if padded and had and self.background is not None and tuple(self.background) != bg:
            raise NotImplementedError("two paddings with different background colours cannot be composed lazily")
        return GeoImage(ys, xs, bg if (padded or self.background is None) else self.background)

    def resize(self, out_h, out_w, interp):
        return GeoImage(self.ys, self.xs, self.background, resized=(int(out_h), int(out_w), int(interp)))

    def taps(self, n_taps):
        """(ix, wx, iy, wy): the resize's taps composed with the index maps, padded to n_taps per output position (weight 0)."""
        out_h, out_w, interp = self.resized
        al = _area_linear(len(self.ys), len(self.xs), out_h, out_w, interp)
        res = []
        for idx_map, n_dst in ((self.xs, out_w), (self.ys, out_h)):
            i, w = axis_taps(len(idx_map), n_dst, interp, al)
            t = i.shape[1]
            if t > n_taps:
                raise ValueError("%d taps needed, %d provided" % (t, n_taps))
            src = idx_map[i].astype(np.int32)
            ii = np.zeros((n_dst, n_taps), dtype=np.int32)
            ww = np.zeros((n_dst, n_taps), dtype=np.float64)
            ii[:, :t], ww[:, :t] = src, w
            res += [ii, ww]
        return res

    def raw_taps(self):
        """(ix, wx, iy, wy) of `taps`, unpadded: (out_w, tx) / (out_h, ty) with this image's own tap counts."""
        out_h, out_w, interp = self.resized
        al = _area_linear(len(self.ys), len(self.xs), out_h, out_w, interp)
        res = []
        for idx_map, n_dst in ((self.xs, out_w), (self.ys, out_h)):
            i, w = axis_taps(len(idx_map), n_dst, interp, al)
            res += [idx_map[i].astype(np.int32), w]
        return res

    def n_taps(self):
        out_h, out_w, interp = self.resized
        al = _area_linear(len(self.ys), len(self.xs), out_h, out_w, interp)
        return max(axis_taps(len(self.xs), out_w, interp, al)[0].shape[1], axis_taps(len(self.ys), out_h, interp, al)[0].shape[1])


def gather_batch(images, lazies):
    """Execute the recorded geometry of a batch: images (B, H, W, 3) CUDA uint8, lazies[i] a resized GeoImage of image i -> the
    (B, out_h, out_w, 3) uint8 batch, ONE launch."""
    sizes = {l.resized[:2] for l in lazies}
    if len(sizes) != 1:
        raise ValueError("every image of a batch must end in the same size")
    out_h, out_w = sizes.pop()
    # every image's taps ONCE (round 4 built them twice: first to learn the tap count, then padded), written straight into the batch tables
    raw = [l.raw_taps() for l in lazies]
    n = max(max(r[0].shape[1], r[2].shape[1]) for r in raw)
    n = 1 if n <= 1 else (2 if n <= 2 else (4 if n <= 4 else (8 if n <= 8 else 16 if n <= 16 else n)))
    b = len(lazies)
    ix, wx = np.zeros((b, out_w, n), dtype=np.int32), np.zeros((b, out_w, n), dtype=np.float64)
    iy, wy = np.zeros((b, out_h, n), dtype=np.int32), np.zeros((b, out_h, n), dtype=np.float64)
    for k, (sx, vx, sy, vy) in enumerate(raw):
        ix[k, :, :sx.shape[1]], wx[k, :, :sx.shape[1]] = sx, vx
        iy[k, :, :sy.shape[1]], wy[k, :, :sy.shape[1]] = sy, vy
    bg = np.array([(l.background if l.background is not None else (0, 0, 0)) for l in lazies], dtype=np.uint8)
    return nat.image_resize_gather_u8(images.contiguous(), out_h, out_w, ix, wx, iy, wy, bg)


# ---- cv2.resize as separable taps ------------------------------------------------------------------------------------------------
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4


def _kernel_cubic(t):
    a = -0.75
    t = np.abs(t)
    near = ((a + 2) * t - (a + 3)) * t * t + 1
    far = ((a * t - 5 * a) * t + 8 * a) * t - 4 * a
    return np.where(t <= 1, near, np.where(t < 2, far, 0.0))


def _kernel_lanczos4(t):
    t = np.asarray(t, dtype=np.float64)
    inside = (np.abs(t) >= 1e-12) & (np.abs(t) < 4)
    safe = np.where(inside, t, 1.0)
    val = 4 * np.sin(np.pi * safe) * np.sin(np.pi * safe / 4) / (np.pi * np.pi * safe * safe)
    return np.where(inside, val, np.where(np.abs(t) < 1e-12, 1.0, 0.0))


def axis_taps(n_src, n_dst, interp, area_linear=False):
    """Source indices (n_dst, T) int32 and float64 weights (n_dst, T) of one axis for an OpenCV interpolation mode: pixel centres
    (src = (dst + 0.5) * scale - 0.5), replicated border.  INTER_AREA is the box filter only when BOTH axes shrink (the caller decides:
    `area_linear=False`); otherwise cv2.resize emulates it on both axes with its `area_mode` bilinear variant (imgproc/resize.cpp):
    sx = floor(dx scale), fx = (float)((dx + 1) - (sx + 1) / scale), 0 if fx <= 0 else its fractional part -- an integer enlargement
    then replicates pixels (the first version decided per axis and used the centre-based linear kernel)."""
    scale = n_src / n_dst
    i = np.arange(n_dst, dtype=np.float64)
    if interp == INTER_AREA and area_linear:
        sx = np.floor(i * scale)
        fx = ((i + 1) - (sx + 1) * (1.0 / scale)).astype(np.float32)
        fx = np.where(fx <= 0, np.float32(0), fx - np.floor(fx)).astype(np.float64)
        last = sx >= n_src - 1
        sx = np.where(last, n_src - 1, sx).astype(np.int64)
        fx = np.where(last, 0.0, fx)
        return np.clip(np.stack([sx, sx + 1], axis=1), 0, n_src - 1).astype(np.int32), np.stack([1.0 - fx, fx], axis=1)
    if interp == INTER_NEAREST:
        return np.minimum(np.floor(i * scale), n_src - 1).astype(np.int32)[:, None], np.ones((n_dst, 1))
    if interp == INTER_AREA and scale >= 1:
        lo, hi = i * scale, (i + 1) * scale
        first = np.floor(lo).astype(np.int64)
        cells = first[:, None] + np.arange(int(np.ceil(scale)) + 1)[None, :]
        w = np.clip(np.minimum(cells + 1.0, hi[:, None]) - np.maximum(cells.astype(np.float64), lo[:, None]), 0.0, None)
        return np.clip(cells, 0, n_src - 1).astype(np.int32), w / w.sum(axis=1, keepdims=True)
    center = (i + 0.5) * scale - 0.5
    base = np.floor(center)
    frac = center - base
    if interp in (INTER_LINEAR, INTER_AREA):
        offs, w = np.array([0, 1]), np.stack([1.0 - frac, frac], axis=1)
    elif interp == INTER_CUBIC:
        offs = np.array([-1, 0, 1, 2])
        w = _kernel_cubic(frac[:, None] - offs[None, :])
    elif interp == INTER_LANCZOS4:
        offs = np.arange(-3, 5)
        w = _kernel_lanczos4(frac[:, None] - offs[None, :])
        w = w / w.sum(axis=1, keepdims=True)
    else:
        raise ValueError("interpolation mode %r is not one of cv2's INTER_NEAREST .. INTER_LANCZOS4 (0 .. 4)" % (interp,))
    return np.clip(base[:, None].astype(np.int64) + offs[None, :], 0, n_src - 1).astype(np.int32), w


def _area_linear(h, w, out_h, out_w, interp):
    """cv2.resize: INTER_AREA is the true area (box) filter only if scale_x >= 1 and scale_y >= 1."""
    return int(interp) == INTER_AREA and not (w >= out_w and h >= out_h)


def resize(image, out_h, out_w, interp):
    """cv2.resize(image, dsize=(out_w, out_h), interpolation=interp) for uint8 images: NumPy (H, W[, C]) or CUDA (B, H, W, C)."""
    if isinstance(image, GeoImage):
        return image.resize(out_h, out_w, interp)
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise TypeError("resize takes uint8 images")
        src = image if image.ndim == 3 else image[:, :, None]
        al = _area_linear(src.shape[0], src.shape[1], out_h, out_w, interp)
        ix, wx = axis_taps(src.shape[1], out_w, int(interp), al)
        iy, wy = axis_taps(src.shape[0], out_h, int(interp), al)
        out = nat.image_resize_u8(nat.to_device(np.ascontiguousarray(src)[None]), out_h, out_w, ix, wx, iy, wy)[0].cpu().numpy()
        return out if image.ndim == 3 else out[:, :, 0]
    al = _area_linear(int(image.shape[1]), int(image.shape[2]), out_h, out_w, interp)
    ix, wx = axis_taps(int(image.shape[2]), out_w, int(interp), al)
    iy, wy = axis_taps(int(image.shape[1]), out_h, int(interp), al)
    return nat.image_resize_u8(image.contiguous(), out_h, out_w, ix, wx, iy, wy)


# ---- cv2.LUT / cv2.equalizeHist ----------------------------------------------------------------------------------------------------
def lut(image, table, channel_mask):
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise TypeError("look-up tables apply to uint8 images")
        return nat.image_lut_u8(nat.to_device(np.ascontiguousarray(image)), np.asarray(table, dtype=np.uint8), channel_mask).cpu().numpy()
    return nat.image_lut_u8(image.contiguous(), np.asarray(table, dtype=np.uint8), channel_mask)


def equalize_table(hist):
    """cv2.equalizeHist's table from a 256-bin histogram: bins up to the first occupied one map to 0, the rest to the cumulative
    count scaled to 255 (float32 scale, rounded to nearest even); a constant plane is left alone."""
    hist = np.asarray(hist, dtype=np.int64)
    total = int(hist.sum())
    occupied = np.nonzero(hist)[0]
    if occupied.size == 0 or hist[occupied[0]] == total:
        return np.arange(256, dtype=np.uint8)
    first = int(occupied[0])
    scale = np.float32(255.0) / np.float32(total - hist[first])
    below = np.cumsum(hist) - hist[first]
    below[:first + 1] = 0
    return np.clip(np.rint((below.astype(np.float32) * scale).astype(np.float32)), 0, 255).astype(np.uint8)


def equalize_channel(image, channel):
    """image with cv2.equalizeHist applied to one channel (uint8; NumPy (H, W, C) or one CUDA image (H, W, C))."""
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise TypeError("histogram equalisation takes uint8 images")
        dev = nat.to_device(np.ascontiguousarray(image))
        table = equalize_table(nat.image_hist_u8(dev, channel).cpu().numpy())
        return nat.image_lut_u8(dev, table, 1 << channel).cpu().numpy()
    table = equalize_table(nat.image_hist_u8(image.contiguous(), channel).cpu().numpy())
    return nat.image_lut_u8(image.contiguous(), table, 1 << channel)
