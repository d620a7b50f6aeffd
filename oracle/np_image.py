"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): NumPy restatement of the OpenCV primitives the reference's image augmentation
calls -- data_generator/object_detection_2d_photometric_ops.py:44-54 (cv2.cvtColor), :359 (cv2.LUT), :407 (cv2.equalizeHist),
object_detection_2d_geometric_ops.py:70-72 (cv2.resize).

PARITY UNPINNED for these primitives: OpenCV (opencv-python, the reference's un-pinned dependency, README.md:143-150) is not installed
here and the reference holds no image fixtures.  The 8-bit colour conversions, the LUT and the histogram equalisation follow the
algorithms of OpenCV's imgproc sources as published (fixed-point tables of color_hsv / color_yuv, equalizeHist's scale-and-round);
`resize` is the documented sampling geometry (pixel centres, src = (dst + 0.5) * scale - 0.5, replicated border) with float64
weights and ONE rounding at the end -- OpenCV's own 8-bit paths use 11-bit fixed-point weights, so a real cv2 result can differ from
this one by one grey level.  What IS pinned with these primitives standing in for cv2: everything the reference itself does around
them (tests/golden/make_golden.py gen_image_ops: the real reference classes run with a cv2 stub built on this module -- dtype
conversions, NumPy arithmetic and clipping, the order of random draws, label arithmetic)."""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
COLOR_RGB2HSV, COLOR_HSV2RGB, COLOR_RGB2GRAY = 41, 55, 7
HSV_SHIFT = 12


def _round_half_even_div(num, den):
    """saturate_cast<int>(num / den) for non-negative integers: nearest, ties to even (cvRound)."""
    q, r = divmod(num, den)
    twice = 2 * r
    if twice > den or (twice == den and (q & 1)):
        q += 1
    return q


SDIV = np.array([0] + [_round_half_even_div(255 << HSV_SHIFT, i) for i in range(1, 256)], dtype=np.int64)
HDIV180 = np.array([0] + [_round_half_even_div(180 << HSV_SHIFT, 6 * i) for i in range(1, 256)], dtype=np.int64)


def rgb2hsv_u8(img):
    """8-bit RGB -> HSV, H in [0, 180) (OpenCV RGB2HSV_b: integer arithmetic with the two division tables)."""
    r, g, b = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    s = (diff * SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * HDIV180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT          # arithmetic shift: floor for negative values
    h = np.where(h < 0, h + 180, h)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # (b, g, r) picks of tab[]


def _hsv2rgb_float(h, s, v, hscale):
    """OpenCV HSV2RGB_f on float32 planes (h already in the input's unit, hscale = 6 / hrange)."""
    f32 = np.float32
    h = (h * f32(hscale)).astype(f32)
    h = np.where(h < 0, h + f32(6), h).astype(f32)            # (one wrap is enough for the ranges the callers produce)
    h = np.where(h >= 6, h - f32(6), h).astype(f32)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f32(0), h).astype(f32)
    one = f32(1)
    tab = np.stack([v, (v * (one - s)).astype(f32), (v * (one - (s * h).astype(f32)).astype(f32)).astype(f32),
                    (v * (one - (s * (one - h).astype(f32)).astype(f32)).astype(f32)).astype(f32)], axis=-1)
    pick = _SECTOR[sector]                                    # (..., 3) indices into tab for (b, g, r)
    b = np.take_along_axis(tab, pick[..., 0:1], axis=-1)[..., 0]
    g = np.take_along_axis(tab, pick[..., 1:2], axis=-1)[..., 0]
    r = np.take_along_axis(tab, pick[..., 2:3], axis=-1)[..., 0]
    grey = s == 0
    return np.where(grey, v, r).astype(f32), np.where(grey, v, g).astype(f32), np.where(grey, v, b).astype(f32)


def hsv2rgb_u8(img):
    """8-bit HSV (H in [0, 180)) -> RGB (OpenCV HSV2RGB_b: through float32, one rounding at the end)."""
    f32 = np.float32
    h = img[..., 0].astype(f32)
    s = (img[..., 1].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    v = (img[..., 2].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    r, g, b = _hsv2rgb_float(h, s, v, 6.0 / 180.0)
    out = np.stack([r, g, b], axis=-1)
    return np.clip(np.rint((out * f32(255.0)).astype(f32)), 0, 255).astype(np.uint8)


def rgb2hsv_f32(img):
    """float32 RGB -> HSV, H in [0, 360) (OpenCV RGB2HSV_f)."""
    f32 = np.float32
    eps = np.finfo(f32).eps
    r, g, b = (img[..., k].astype(f32) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = (v - vmin).astype(f32)
    s = (diff / (np.abs(v) + eps).astype(f32)).astype(f32)
    d = (f32(60.0) / (diff + eps).astype(f32)).astype(f32)
    h = np.where(v == r, ((g - b).astype(f32) * d).astype(f32),
                 np.where(v == g, (((b - r).astype(f32) * d).astype(f32) + f32(120.0)).astype(f32),
                          (((r - g).astype(f32) * d).astype(f32) + f32(240.0)).astype(f32))).astype(f32)
    h = np.where(h < 0, h + f32(360.0), h).astype(f32)
    return np.stack([h, s, v], axis=-1).astype(f32)


def hsv2rgb_f32(img):
    r, g, b = _hsv2rgb_float(img[..., 0].astype(np.float32), img[..., 1].astype(np.float32), img[..., 2].astype(np.float32), 6.0 / 360.0)
    return np.stack([r, g, b], axis=-1).astype(np.float32)


def rgb2gray(img):
    """RGB -> one grey plane: 8-bit fixed point (4899, 9617, 1868) >> 14, float32 0.299 / 0.587 / 0.114 (OpenCV RGB2Gray)."""
    if img.dtype == np.uint8:
        r, g, b = (img[..., k].astype(np.int64) for k in range(3))
        return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)
    f32 = np.float32
    r, g, b = (img[..., k].astype(f32) for k in range(3))
    return ((r * f32(0.299)).astype(f32) + (g * f32(0.587)).astype(f32) + (b * f32(0.114)).astype(f32)).astype(f32)


def cvt_color(img, code):
    if img.dtype not in (np.uint8, np.float32):
        raise TypeError("cvtColor: 8-bit or float32 images")
    if code == COLOR_RGB2HSV:
        return rgb2hsv_u8(img) if img.dtype == np.uint8 else rgb2hsv_f32(img)
    if code == COLOR_HSV2RGB:
        return hsv2rgb_u8(img) if img.dtype == np.uint8 else hsv2rgb_f32(img)
    if code == COLOR_RGB2GRAY:
        return rgb2gray(img)
    raise ValueError("colour conversion code %r is not one the reference can reach" % (code,))


def lut(img, table):
    if img.dtype != np.uint8:
        raise TypeError("LUT: 8-bit images")
    return np.asarray(table)[img]


def equalize_hist_table(hist):
    """The 256-entry table of cv2.equalizeHist from the plane's histogram."""
    hist = np.asarray(hist, dtype=np.int64)
    total = int(hist.sum())
    nz = np.nonzero(hist)[0]
    i0 = int(nz[0]) if nz.size else 0
    if nz.size == 0 or hist[i0] == total:
        return np.full(256, i0, dtype=np.uint8), True                      # a constant plane stays what it is
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    csum = np.cumsum(hist) - hist[i0]
    csum[:i0 + 1] = 0
    tab = np.clip(np.rint((csum.astype(np.float32) * scale).astype(np.float32)), 0, 255).astype(np.uint8)
    return tab, False


def equalize_hist(plane):
    if plane.dtype != np.uint8 or plane.ndim != 2:
        raise TypeError("equalizeHist: one 8-bit plane")
    tab, _ = equalize_hist_table(np.bincount(plane.reshape(-1), minlength=256))
    return tab[plane]


# ---- resize: separable resampling with per-output-coordinate taps --------------------------------------------------------------
def _cubic_w(t, a=-0.75):
    t = np.abs(t)
    return np.where(t <= 1, ((a + 2) * t - (a + 3)) * t * t + 1, np.where(t < 2, ((a * t - 5 * a) * t + 8 * a) * t - 4 * a, 0.0))


def _lanczos_w(t, a=4):
    t = np.asarray(t, dtype=np.float64)
    out = np.where(np.abs(t) < 1e-12, 1.0, 0.0)
    nzm = (np.abs(t) >= 1e-12) & (np.abs(t) < a)
    tt = np.where(nzm, t, 1.0)
    val = a * np.sin(np.pi * tt) * np.sin(np.pi * tt / a) / (np.pi * np.pi * tt * tt)
    return np.where(nzm, val, out)


def resize_taps(n_src, n_dst, interp, area_linear=False):
    """(index [n_dst, T] int32, weight [n_dst, T] float64) of one axis: dst[i] = sum_t weight[i, t] * src[index[i, t]].
    `area_linear`: INTER_AREA when NOT both axes shrink -- OpenCV's resize() then emulates it "using some variant of bilinear" on BOTH
    axes (imgproc/resize.cpp, 3.x / 4.x: `area_mode`): sx = floor(dx scale), fx = (float)((dx + 1) - (sx + 1) / scale), fx <= 0 -> 0 else
    its fractional part; taps (sx, sx + 1) with weights (1 - fx, fx), clamped at the last pixel."""
    scale = n_src / n_dst
    i = np.arange(n_dst, dtype=np.float64)
    if interp == INTER_AREA and area_linear:
        sx = np.floor(i * scale)
        fx = ((i + 1) - (sx + 1) * (1.0 / scale)).astype(np.float32)
        fx = np.where(fx <= 0, np.float32(0), fx - np.floor(fx)).astype(np.float64)
        last = sx >= n_src - 1
        sx = np.where(last, n_src - 1, sx).astype(np.int64)
        fx = np.where(last, 0.0, fx)
        idx = np.clip(np.stack([sx, sx + 1], axis=1), 0, n_src - 1).astype(np.int32)
        return idx, np.stack([1.0 - fx, fx], axis=1)
    if interp == INTER_NEAREST:
        idx = np.minimum(np.floor(i * scale), n_src - 1).astype(np.int32)[:, None]
        return idx, np.ones((n_dst, 1))
    if interp == INTER_AREA and scale >= 1:
        # box filter: the overlap of [i * scale, (i + 1) * scale) with each source cell, normalised
        lo, hi = i * scale, (i + 1) * scale
        first = np.floor(lo).astype(np.int64)
        T = int(np.ceil(scale)) + 1
        idx = first[:, None] + np.arange(T)[None, :]
        w = np.clip(np.minimum(idx + 1.0, hi[:, None]) - np.maximum(idx.astype(np.float64), lo[:, None]), 0.0, None)
        w = w / w.sum(axis=1, keepdims=True)
        return np.clip(idx, 0, n_src - 1).astype(np.int32), w
    center = (i + 0.5) * scale - 0.5
    base = np.floor(center)
    frac = center - base
    if interp in (INTER_LINEAR, INTER_AREA):                                # (AREA when enlarging: the linear kernel)
        offs = np.array([0, 1])
        w = np.stack([1.0 - frac, frac], axis=1)
    elif interp == INTER_CUBIC:
        offs = np.array([-1, 0, 1, 2])
        w = _cubic_w(frac[:, None] - offs[None, :])
    elif interp == INTER_LANCZOS4:
        offs = np.arange(-3, 5)
        w = _lanczos_w(frac[:, None] - offs[None, :])
        w = w / w.sum(axis=1, keepdims=True)
    else:
        raise ValueError("interpolation mode %r" % (interp,))
    idx = np.clip(base[:, None].astype(np.int64) + offs[None, :], 0, n_src - 1).astype(np.int32)
    return idx, w


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, dsize=(width, height), interpolation) for 8-bit images [H, W] / [H, W, C] (see the module docstring)."""
    if img.dtype != np.uint8:
        raise TypeError("resize: 8-bit images")
    wo, ho = int(dsize[0]), int(dsize[1])
    src = img if img.ndim == 3 else img[:, :, None]
    al = int(interpolation) == INTER_AREA and not (src.shape[1] >= wo and src.shape[0] >= ho)     # cv2: the box filter only if BOTH axes shrink
    ix, wx = resize_taps(src.shape[1], wo, int(interpolation), al)
    iy, wy = resize_taps(src.shape[0], ho, int(interpolation), al)
    srcd = src.astype(np.float64)
    acc = np.zeros((ho, wo, src.shape[2]))
    for j in range(iy.shape[1]):                                            # rows outer, columns inner: the kernel's order
        rows = srcd[iy[:, j]]                                               # [ho, W, C]
        racc = np.zeros((ho, wo, src.shape[2]))
        for t in range(ix.shape[1]):
            racc = racc + wx[None, :, t, None] * rows[:, ix[:, t]]
        acc = acc + wy[:, j, None, None] * racc
    out = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


# ---- the pointwise programs of csrc/ssdhip_image.hip (ssdhip_image_program), restated in NumPy ------------------------------------
OPS = {0: "end", 1: "to_f32", 2: "to_u8", 3: "brightness", 4: "contrast", 5: "saturation", 6: "hue", 7: "rgb2hsv", 8: "hsv2rgb",
       9: "rgb2gray", 10: "swap"}


def run_program(image, ops, args):
    """One image (H, W, 3) through one program: every step is the reference's own NumPy expression for that op
    (object_detection_2d_photometric_ops.py:81-83, :129, :185, :242, :300, :451) or the cvtColor restatement above."""
    img = np.array(image, copy=True)
    for o, a in zip(ops, args):
        name = OPS[int(o)]
        a = float(a)
        if name == "end":
            break
        if name == "to_f32":
            img = img.astype(np.float32)
        elif name == "to_u8":
            img = np.clip(np.round(img, decimals=0), 0, 255).astype(np.uint8)
        elif name == "brightness":
            img = np.clip(img + a, 0, 255)
        elif name == "contrast":
            img = np.clip(127.5 + a * (img - 127.5), 0, 255)
        elif name == "saturation":
            img[:, :, 1] = np.clip(img[:, :, 1] * a, 0, 255)
        elif name == "hue":
            img[:, :, 0] = (img[:, :, 0] + a) % 180.0
        elif name == "rgb2hsv":
            img = cvt_color(img, COLOR_RGB2HSV)
        elif name == "hsv2rgb":
            img = cvt_color(img, COLOR_HSV2RGB)
        elif name == "rgb2gray":
            img = np.stack([rgb2gray(img)] * 3, axis=-1)
        elif name == "swap":
            code = int(a)
            img = img[:, :, [code & 3, (code >> 2) & 3, (code >> 4) & 3]]
    return img
