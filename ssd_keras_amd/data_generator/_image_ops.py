"""Host side of the image half of the augmentation: per-image programs for `ssdhip_image_program`, tap tables for
`ssdhip_image_resize_cv_u8` (cv2.resize's own 8-bit arithmetic), histogram equalisation and look-up tables -- the pixel work itself runs in csrc/ssdhip_image.hip.
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
    position filled with `background` -- and executed once, for a whole batch, by `ssdhip_image_resize_gather_cv_u8`.  Quacks like the
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

    def plan(self):
        """(kind, area, ix, wx, iy, wy) of the final resize (`resize_plan`), its tap indices composed with the index maps: a tap
        addresses a column / row of the ORIGINAL image, or -1 = a position the expansion filled with the background colour."""
        out_h, out_w, interp = self.resized
        kind, ix, wx, iy, wy, area = resize_plan(len(self.ys), len(self.xs), out_h, out_w, interp)
        return kind, area, self.xs[ix].astype(np.int32), wx, self.ys[iy].astype(np.int32), wy


def gather_batch(images, lazies):
    """Execute the recorded geometry of a batch: images (B, H, W, 3) CUDA uint8, lazies[i] a resized GeoImage of image i -> the
    (B, out_h, out_w, 3) uint8 batch, ONE launch (every image with its own cv2.resize arithmetic: `kinds`)."""
    sizes = {l.resized[:2] for l in lazies}
    if len(sizes) != 1:
        raise ValueError("every image of a batch must end in the same size")
    out_h, out_w = sizes.pop()
    plans = [l.plan() for l in lazies]
    n = max(max(pl[2].shape[1], pl[4].shape[1]) for pl in plans)
    n = 1 if n <= 1 else (2 if n <= 2 else (4 if n <= 4 else (8 if n <= 8 else 16 if n <= 16 else n)))
    b = len(lazies)
    ix, wx = np.zeros((b, out_w, n), dtype=np.int32), np.zeros((b, out_w, n), dtype=np.float64)
    iy, wy = np.zeros((b, out_h, n), dtype=np.int32), np.zeros((b, out_h, n), dtype=np.float64)
    kinds = np.zeros((b, 4), dtype=np.int32)                      # kind, area, taps per column, taps per row
    for k, (kind, area, sx, vx, sy, vy) in enumerate(plans):
        ix[k, :, :sx.shape[1]], wx[k, :, :sx.shape[1]] = sx, vx
        iy[k, :, :sy.shape[1]], wy[k, :, :sy.shape[1]] = sy, vy
        kinds[k] = (kind, area, sx.shape[1], sy.shape[1])
    bg = np.array([(l.background if l.background is not None else (0, 0, 0)) for l in lazies], dtype=np.uint8)
    return nat.image_resize_gather_cv_u8(images.contiguous(), out_h, out_w, kinds, ix, wx, iy, wy, bg)


# ---- cv2.resize for 8-bit images: the tables of OpenCV's imgproc/resize.cpp (3.4 / 4.x), the arithmetic runs in csrc/ssdhip_image.hip ----
# cv::resize on CV_8U (round 6; the first version resampled with float64 weights and one rounding, off by one grey level here and there):
#   LINEAR / CUBIC / LANCZOS4, and AREA when an axis grows ("area_mode" bilinear): per output column / row the float32 kernel values become
#     11-bit fixed-point `short`s (saturate_cast<short>(c * 2048), nearest-even); rows are int32 sums of uchar x short; the vertical pass is
#     uchar((((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2) for the linear kernel and saturate((sum + 2^21) >> 22) for the
#     4- and 8-tap ones.  Columns reset (sx, fx) at the borders for the linear kernel; rows are clamped with their coefficients kept;
#   AREA with both scales >= 1: integer scales -> block sums, saturate(cvRound(sum * (1.f / area))), the 2 x 2 block (sum + 2) >> 2 (also
#     what INTER_LINEAR does at exactly 2 x 2); otherwise computeResizeAreaTab's float32 alpha / beta accumulated in float32, cvRound;
#   NEAREST: min(floor(dst * (1 / (n_dst / n_src))), n_src - 1); equal sizes: a copy.
# A plan = (kind, ix, wx, iy, wy, area): tap indices and the table values as float64 (shorts / float32 alphas / ones are all exact there).
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
KIND_NEAREST, KIND_LINEAR, KIND_KERNEL, KIND_AREA, KIND_AREA_FAST, KIND_AREA_FAST2, KIND_COPY = 0, 1, 2, 3, 4, 5, 6
COEF_SCALE = 2048                                           # 1 << INTER_RESIZE_COEF_BITS
_f32 = np.float32
_PI = 3.1415926535897932384626433832795                     # CV_PI


def _scales(n_src, n_dst):
    inv = float(n_dst) / float(n_src)                       # inv_scale = (double)dsize / ssize;  scale = 1. / inv_scale
    return inv, 1.0 / inv


def _to_short(coef):
    """saturate_cast<short>(float32 coefficient * 2048): nearest-even, clamped."""
    v = np.asarray(coef, dtype=_f32) * _f32(COEF_SCALE)
    return np.clip(np.rint(v.astype(np.float64)), -32768, 32767)


def _kernel_cubic(x):
    """interpolateCubic, A = -0.75, every operation in float32: (n,) -> (n, 4)."""
    x = np.asarray(x, dtype=_f32)
    a = _f32(-0.75)
    u = x + _f32(1)
    c0 = ((a * u - _f32(5) * a) * u + _f32(8) * a) * u - _f32(4) * a
    c1 = ((a + _f32(2)) * x - (a + _f32(3))) * x * x + _f32(1)
    v = _f32(1) - x
    c2 = ((a + _f32(2)) * v - (a + _f32(3))) * v * v + _f32(1)
    return np.stack([c0, c1, c2, _f32(1) - c0 - c1 - c2], axis=1).astype(_f32)


import math as _math
_INV_FACT = [1.0 / _math.factorial(_k) for _k in range(24)]      # exact integers, ONE rounding each (the literals of csrc/ssdhip_augment.hip)


def sincos_near_minus_pi(y):
    """(sin y, cos y) for y in [-pi, -3 pi / 4] by a FIXED chain of float64 operations (t = y + pi; Taylor polynomials of degree 23 / 22 in
    Horner form, signs folded into the subtractions): the GPU runs the same chain (csrc/ssdhip_augment.hip), so host- and device-built
    Lanczos tables agree to the bit.  Within 1 ulp of the C library's."""
    t = np.asarray(y, dtype=np.float64) + _PI
    t2 = t * t
    s = np.full_like(t2, _INV_FACT[23])
    for k in range(21, 0, -2):
        s = _INV_FACT[k] - s * t2
    c = np.full_like(t2, _INV_FACT[22])
    for k in range(20, -1, -2):
        c = _INV_FACT[k] - c * t2
    return -(s * t), -c


def _kernel_lanczos4(x):
    """interpolateLanczos4: (n,) float32 -> (n, 8) float32."""
    x = np.asarray(x, dtype=_f32).reshape(-1)
    r = 0.70710678118654752440084436210485
    cs = ((1, 0), (-r, -r), (0, 1), (r, -r), (-1, 0), (r, r), (0, -1), (-r, r))
    out = np.zeros((x.shape[0], 8), dtype=_f32)
    centre = x < np.finfo(_f32).eps                           # x < FLT_EPSILON
    out[centre, 3] = 1
    rest = ~centre
    if rest.any():
        x3 = x[rest] + _f32(3)
        s0, c0 = sincos_near_minus_pi(-(x3.astype(np.float64)) * _PI * 0.25)
        co = np.empty((x3.shape[0], 8), dtype=_f32)
        for i in range(8):
            y = -((x3 - _f32(i)).astype(np.float64)) * _PI * 0.25
            co[:, i] = ((cs[i][0] * s0 + cs[i][1] * c0) / (y * y)).astype(_f32)
        total = np.zeros(x3.shape[0], dtype=_f32)
        for i in range(8):
            total = total + co[:, i]
        out[rest] = co * (_f32(1) / total)[:, None]
    return out


def _coords(n_src, n_dst, area_mode):
    """(sx, fx) per destination coordinate before the border rules: fx = (float)((d + 0.5) scale - 0.5), or the area-mode variant."""
    inv, scale = _scales(n_src, n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    if area_mode:
        sx = np.floor(d * scale).astype(np.int64)
        fx = ((d + 1) - (sx + 1) * inv).astype(_f32)
        return sx, np.where(fx <= 0, _f32(0), fx - np.floor(fx)).astype(_f32)
    fx = ((d + 0.5) * scale - 0.5).astype(_f32)
    sx = np.floor(fx).astype(np.int64)
    return sx, (fx - sx.astype(_f32)).astype(_f32)


def fixed_axis(n_src, n_dst, interp, area_mode, horizontal):
    """Tap indices (n_dst, k) int32 and `short` coefficients (n_dst, k) float64 of one axis of the fixed-point paths."""
    sx, fx = _coords(n_src, n_dst, area_mode)
    if interp in (INTER_LINEAR, INTER_AREA):
        if horizontal:
            lo, hi = sx < 0, sx >= n_src - 1
            fx = np.where(lo | hi, _f32(0), fx).astype(_f32)
            sx = np.where(lo, 0, np.where(hi, n_src - 1, sx))
        offs, coef = np.array([0, 1]), np.stack([_f32(1) - fx, fx], axis=1)
    elif interp == INTER_CUBIC:
        offs, coef = np.array([-1, 0, 1, 2]), _kernel_cubic(fx)
    elif interp == INTER_LANCZOS4:
        offs, coef = np.arange(-3, 5), _kernel_lanczos4(fx)
    else:
        raise ValueError("interpolation mode %r is not one of cv2's INTER_NEAREST .. INTER_LANCZOS4 (0 .. 4)" % (interp,))
    return np.clip(sx[:, None] + offs[None, :], 0, n_src - 1).astype(np.int32), _to_short(coef)


def area_axis(n_src, n_dst):
    """computeResizeAreaTab as padded per-destination taps: indices (n_dst, T) int32, float32 alphas (n_dst, T) as float64, zeros behind a
    row's own entries."""
    _, scale = _scales(n_src, n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f1 = d * scale
    f2 = f1 + scale
    cell = np.minimum(scale, n_src - f1)
    s2 = np.minimum(np.floor(f2).astype(np.int64), n_src - 1)
    s1 = np.minimum(np.ceil(f1).astype(np.int64), s2)
    head = (s1 - f1) > 1e-3
    tail = (f2 - s2) > 1e-3
    count = head.astype(np.int64) + (s2 - s1) + tail.astype(np.int64)
    T = max(1, int(count.max()))
    idx = np.zeros((n_dst, T), dtype=np.int64)
    alpha = np.zeros((n_dst, T), dtype=_f32)
    pos = np.arange(T)[None, :]
    first = (s1 - head.astype(np.int64))[:, None]                     # the row's first source index
    idx[:] = first + pos
    alpha[:] = (1.0 / cell)[:, None].astype(_f32)
    head_val = ((s1 - f1) / cell).astype(_f32)
    tail_val = (np.minimum(np.minimum(f2 - s2, 1.0), cell) / cell).astype(_f32)
    alpha[head, 0] = head_val[head]
    rows = np.nonzero(tail)[0]
    alpha[rows, count[rows] - 1] = tail_val[rows]
    beyond = pos >= count[:, None]
    alpha[beyond] = 0
    last = np.take_along_axis(idx, np.maximum(count - 1, 0)[:, None], axis=1)
    idx = np.where(beyond, last, idx)
    return np.clip(idx, 0, n_src - 1).astype(np.int32), alpha.astype(np.float64)


def resize_plan(src_h, src_w, dst_h, dst_w, interp):
    """cv::resize's dispatch for these sizes -> (kind, ix, wx, iy, wy, area)."""
    interp = int(interp)
    if interp not in (INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4):
        raise ValueError("interpolation mode %r is not one of cv2's INTER_NEAREST .. INTER_LANCZOS4 (0 .. 4)" % (interp,))
    ones = lambda i: np.ones(i.shape, dtype=np.float64)
    if (src_h, src_w) == (dst_h, dst_w):
        ix, iy = np.arange(dst_w, dtype=np.int32)[:, None], np.arange(dst_h, dtype=np.int32)[:, None]
        return KIND_COPY, ix, ones(ix), iy, ones(iy), 1
    if interp == INTER_NEAREST:
        tabs = []
        for n_src, n_dst in ((src_w, dst_w), (src_h, dst_h)):
            i = np.minimum(np.floor(np.arange(n_dst, dtype=np.float64) * _scales(n_src, n_dst)[1]), n_src - 1).astype(np.int32)[:, None]
            tabs += [i, ones(i)]
        return (KIND_NEAREST,) + tuple(tabs) + (1,)
    scale_x, scale_y = _scales(src_w, dst_w)[1], _scales(src_h, dst_h)[1]
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    eps = np.finfo(np.float64).eps
    fast = abs(scale_x - isx) < eps and abs(scale_y - isy) < eps
    if interp == INTER_LINEAR and fast and isx == 2 and isy == 2:
        interp = INTER_AREA
    if interp == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        if fast:
            ix = (np.arange(dst_w)[:, None] * isx + np.arange(isx)[None, :]).astype(np.int32)
            iy = (np.arange(dst_h)[:, None] * isy + np.arange(isy)[None, :]).astype(np.int32)
            return (KIND_AREA_FAST2 if (isx, isy) == (2, 2) else KIND_AREA_FAST), ix, ones(ix), iy, ones(iy), isx * isy
        ix, ax = area_axis(src_w, dst_w)
        iy, ay = area_axis(src_h, dst_h)
        return KIND_AREA, ix, ax, iy, ay, 1
    area_mode = interp == INTER_AREA
    ix, cx = fixed_axis(src_w, dst_w, interp, area_mode, True)
    iy, cy = fixed_axis(src_h, dst_h, interp, area_mode, False)
    return (KIND_LINEAR if interp in (INTER_LINEAR, INTER_AREA) else KIND_KERNEL), ix, cx, iy, cy, 1


def resize(image, out_h, out_w, interp):
    """cv2.resize(image, dsize=(out_w, out_h), interpolation=interp) for uint8 images: NumPy (H, W[, C]) or CUDA (B, H, W, C)."""
    if isinstance(image, GeoImage):
        return image.resize(out_h, out_w, interp)
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise TypeError("resize takes uint8 images")
        src = image if image.ndim == 3 else image[:, :, None]
        kind, ix, wx, iy, wy, area = resize_plan(src.shape[0], src.shape[1], out_h, out_w, interp)
        out = nat.image_resize_cv_u8(nat.to_device(np.ascontiguousarray(src)[None]), out_h, out_w, kind, area, ix, wx, iy, wy)[0].cpu().numpy()
        return out if image.ndim == 3 else out[:, :, 0]
    kind, ix, wx, iy, wy, area = resize_plan(int(image.shape[1]), int(image.shape[2]), out_h, out_w, interp)
    return nat.image_resize_cv_u8(image.contiguous(), out_h, out_w, kind, area, ix, wx, iy, wy)


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
