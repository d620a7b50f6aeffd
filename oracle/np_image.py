"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): NumPy restatement of the OpenCV primitives the reference's image augmentation
calls -- data_generator/object_detection_2d_photometric_ops.py:44-54 (cv2.cvtColor), :359 (cv2.LUT), :407 (cv2.equalizeHist),
object_detection_2d_geometric_ops.py:70-72 (cv2.resize).

PARITY UNPINNED for these primitives: OpenCV (opencv-python, the reference's un-pinned dependency, README.md:143-150) is not installed
here and the reference holds no image fixtures.  The 8-bit colour conversions, the LUT and the histogram equalisation follow the
algorithms of OpenCV's imgproc sources as published (fixed-point tables of color_hsv / color_yuv, equalizeHist's scale-and-round);
`resize` (round 6) follows imgproc/resize.cpp's 8-bit arithmetic itself: 11-bit fixed-point coefficients, int32 rows, the two-stage
vertical rounding of the linear path and the `(... + (1 << 21)) >> 22` of the cubic / Lanczos ones, ResizeAreaFast / ResizeArea for
shrinking INTER_AREA (the block comment in front of the resize functions lists every rule and the two that stay platform-defined in
OpenCV itself).  What IS pinned with these primitives standing in for cv2: everything the reference itself does around
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


# ---- cv2.resize for 8-bit images, with the ARITHMETIC of OpenCV's imgproc/resize.cpp (3.4 / 4.x, the C++ reference paths) -----------
# Round 6.  cv::resize on CV_8U is not "weights x pixels, one rounding":
#   * INTER_LINEAR / INTER_CUBIC / INTER_LANCZOS4 (and INTER_AREA when an axis grows: `area_mode`) use 11-bit FIXED-POINT coefficients
#     (INTER_RESIZE_COEF_BITS = 11): per output column / row the float32 kernel values are rounded to `short`
#     (saturate_cast<short>(coeff * 2048), cvRound = nearest-even), HResize* accumulates uchar x short products in int32 rows, and
#     VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> computes
#         dst = uchar(( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 ) >> 2)
#     (the two-stage form its SIMD twin _mm_mulhi_epi16 dictates), VResizeCubic / VResizeLanczos4
#         dst = saturate_cast<uchar>((S0 b0 + S1 b1 + ... + (1 << 21)) >> 22);
#   * horizontally the linear kernel resets (sx, fx) at the borders (sx < 0 -> (0, 0); sx >= width - 1 -> (width - 1, 0)), vertically
#     the ROWS are clamped and the coefficients kept; cubic / Lanczos clamp their tap indices on both axes;
#   * INTER_AREA with both scales >= 1: integer scales -> ResizeAreaFast (sum of the block, saturate_cast<uchar>(sum * (1.f / area)),
#     the 2 x 2 block as (sum + 2) >> 2); otherwise ResizeArea with computeResizeAreaTab's float32 alpha / beta tables accumulated in
#     float32 in table order and saturate_cast<uchar> (cvRound) at the end;
#   * INTER_LINEAR at exactly 2 x 2 shrinking IS the fast area path; equal sizes are a copy; INTER_NEAREST takes
#     min(floor(dst * (1 / (dst_size / src_size))), src - 1).
# What remains of the reservation in the module docstring: OpenCV's SIMD builds run the cubic vertical pass in float32 (identical except
# on exact ties of the last rounding), and interpolateLanczos4 calls the platform's sin / cos -- here `det_sincos`, a fixed Horner chain
# shared with the product (within 1 ulp of libm's; the float32 cast and the 11-bit rounding absorb that except on measure-zero ties).
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS
KIND_NEAREST, KIND_LINEAR, KIND_KERNEL, KIND_AREA, KIND_AREA_FAST, KIND_AREA_FAST2, KIND_COPY = 0, 1, 2, 3, 4, 5, 6
_F32 = np.float32
_CV_PI = 3.1415926535897932384626433832795


def _sat_short(v):
    """saturate_cast<short>(float): cvRound (nearest, ties to even), clamped."""
    return np.clip(np.rint(np.asarray(v, dtype=np.float64)), -32768, 32767).astype(np.int64)


def _inv_scales(n_src, n_dst):
    inv = float(n_dst) / float(n_src)                     # inv_scale_x = (double)dsize.width / ssize.width
    return inv, 1.0 / inv                                 # scale_x = 1. / inv_scale_x


def cv_interpolate_cubic(x):
    """interpolateCubic (imgproc/resize.cpp), float32 operation by operation; x float32 array -> (n, 4) float32."""
    x = np.asarray(x, dtype=_F32)
    A = _F32(-0.75)
    x1 = x + _F32(1)
    c0 = ((A * x1 - _F32(5) * A) * x1 + _F32(8) * A) * x1 - _F32(4) * A
    c1 = ((A + _F32(2)) * x - (A + _F32(3))) * x * x + _F32(1)
    y = _F32(1) - x
    c2 = ((A + _F32(2)) * y - (A + _F32(3))) * y * y + _F32(1)
    c3 = _F32(1) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(_F32)


def det_sincos(y):
    """sin and cos of y in [-pi, -3 pi / 4] (interpolateLanczos4's y0) as a fixed chain of float64 +, -, *: t = y + pi in [0, pi / 4],
    sin y = -sin t, cos y = -cos t, Taylor polynomials of degree 23 / 22 by Horner.  Same bits on every host and on the GPU."""
    y = np.asarray(y, dtype=np.float64)
    t = y + _CV_PI
    t2 = t * t
    s = np.full_like(t2, 1.0 / 25852016738884976640000.0)            # 1 / 23!
    for k in (21, 19, 17, 15, 13, 11, 9, 7, 5, 3, 1):
        s = s * t2
        s = _SIN_C[k] - s
    s = s * t
    c = np.full_like(t2, 1.0 / 1124000727777607680000.0)             # 1 / 22!
    for k in (20, 18, 16, 14, 12, 10, 8, 6, 4, 2, 0):
        c = c * t2
        c = _COS_C[k] - c
    return -s, -c


def _fact(n):
    r = 1
    for i in range(2, n + 1):
        r *= i
    return r


# alternating Taylor coefficients written so that every Horner step is `c_k - acc * t^2`: sin t = t (1/1! - t2 (1/3! - t2 (1/5! - ...)))
_SIN_C = {k: 1.0 / _fact(k) for k in range(1, 23, 2)}
_COS_C = {k: 1.0 / _fact(k) for k in range(0, 22, 2)}


def cv_interpolate_lanczos4(x):
    """interpolateLanczos4 (imgproc/resize.cpp); x float32 array -> (n, 8) float32."""
    x = np.asarray(x, dtype=_F32).reshape(-1)
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]], dtype=np.float64)
    out = np.zeros((x.shape[0], 8), dtype=_F32)
    tiny = x < np.finfo(_F32).eps                           # x < FLT_EPSILON: the centre tap alone
    out[tiny, 3] = 1
    xs = x[~tiny]
    if xs.size:
        x3 = xs + _F32(3)                                   # float32: (x + 3)
        y0 = -(x3.astype(np.float64)) * _CV_PI * 0.25
        s0, c0 = det_sincos(y0)
        co = np.empty((xs.shape[0], 8), dtype=_F32)
        for i in range(8):
            y = -((x3 - _F32(i)).astype(np.float64)) * _CV_PI * 0.25        # -(x + 3 - i) * CV_PI * 0.25, the difference in float32
            co[:, i] = ((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y)).astype(_F32)
        total = np.zeros(xs.shape[0], dtype=_F32)
        for i in range(8):
            total = total + co[:, i]                        # float sum, in tap order
        total = _F32(1) / total
        out[~tiny] = co * total[:, None]
    return out


def cv_coords(n_src, n_dst, area_mode):
    """(sx int64, fx float32) of every destination coordinate, before any border handling (resize.cpp: the dx / dy loops)."""
    inv, scale = _inv_scales(n_src, n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    if not area_mode:
        fx = ((d + 0.5) * scale - 0.5).astype(_F32)
        sx = np.floor(fx).astype(np.int64)
        fx = (fx - sx.astype(_F32)).astype(_F32)
    else:
        sx = np.floor(d * scale).astype(np.int64)
        fx = ((d + 1) - (sx + 1) * inv).astype(_F32)
        fx = np.where(fx <= 0, _F32(0), fx - np.floor(fx)).astype(_F32)
    return sx, fx


def cv_fixed_axis(n_src, n_dst, interp, area_mode, horizontal):
    """Tap indices (n_dst, k) and the `short` coefficients (n_dst, k) of one axis for the fixed-point paths."""
    sx, fx = cv_coords(n_src, n_dst, area_mode)
    if interp in (INTER_LINEAR, INTER_AREA):
        if horizontal:                                      # only the x loop resets the pair at the borders
            lo, hi = sx < 0, sx >= n_src - 1
            fx = np.where(lo | hi, _F32(0), fx).astype(_F32)
            sx = np.where(lo, 0, np.where(hi, n_src - 1, sx))
        coef = np.stack([_F32(1) - fx, fx], axis=1)
        offs = np.array([0, 1])
    elif interp == INTER_CUBIC:
        coef, offs = cv_interpolate_cubic(fx), np.array([-1, 0, 1, 2])
    elif interp == INTER_LANCZOS4:
        coef, offs = cv_interpolate_lanczos4(fx), np.arange(-3, 5)
    else:
        raise ValueError("interpolation mode %r" % (interp,))
    idx = np.clip(sx[:, None] + offs[None, :], 0, n_src - 1).astype(np.int32)
    return idx, _sat_short(coef.astype(_F32) * _F32(INTER_RESIZE_COEF_SCALE))


def cv_area_tab(n_src, n_dst):
    """computeResizeAreaTab as padded per-destination taps: (index (n_dst, T) int32, alpha (n_dst, T) float32, zero beyond a row's
    own entries -- adding `pixel * 0.f` leaves a float32 sum unchanged)."""
    _, scale = _inv_scales(n_src, n_dst)
    rows = []
    for dx in range(n_dst):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, n_src - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, n_src - 1)
        sx1 = min(sx1, sx2)
        ent = []
        if sx1 - fsx1 > 1e-3:
            ent.append((sx1 - 1, _F32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            ent.append((sx, _F32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            ent.append((sx2, _F32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        rows.append(ent)
    T = max(1, max(len(r) for r in rows))
    idx = np.zeros((n_dst, T), dtype=np.int32)
    alpha = np.zeros((n_dst, T), dtype=_F32)
    for d, ent in enumerate(rows):
        for k, (si, a) in enumerate(ent):
            idx[d, k], alpha[d, k] = si, a
        if ent:
            idx[d, len(ent):] = ent[-1][0]
    return idx, alpha


def cv_resize_plan(src_h, src_w, dst_h, dst_w, interp):
    """What cv::resize does for these sizes: (kind, ix, wx, iy, wy, area).  ix / iy (n_dst, T) source indices, wx / wy the table
    values as float64 (shorts for the fixed-point kinds, float32 alphas for the area filter, ones otherwise)."""
    interp = int(interp)
    ones = lambda i: np.ones(i.shape, dtype=np.float64)
    if (src_h, src_w) == (dst_h, dst_w):
        iy, ix = np.arange(dst_h, dtype=np.int32)[:, None], np.arange(dst_w, dtype=np.int32)[:, None]
        return KIND_COPY, ix, ones(ix), iy, ones(iy), 1
    if interp == INTER_NEAREST:
        res = []
        for n_src, n_dst in ((src_w, dst_w), (src_h, dst_h)):
            _, scale = _inv_scales(n_src, n_dst)
            i = np.minimum(np.floor(np.arange(n_dst, dtype=np.float64) * scale), n_src - 1).astype(np.int32)[:, None]
            res += [i, ones(i)]
        return (KIND_NEAREST,) + tuple(res) + (1,)
    _, scale_x = _inv_scales(src_w, dst_w)
    _, scale_y = _inv_scales(src_h, dst_h)
    iscale_x, iscale_y = int(np.rint(scale_x)), int(np.rint(scale_y))             # saturate_cast<int>(double) = cvRound
    eps = np.finfo(np.float64).eps
    fast = abs(scale_x - iscale_x) < eps and abs(scale_y - iscale_y) < eps
    if interp == INTER_LINEAR and fast and iscale_x == 2 and iscale_y == 2:
        interp = INTER_AREA
    if interp == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        if fast:
            ix = (np.arange(dst_w)[:, None] * iscale_x + np.arange(iscale_x)[None, :]).astype(np.int32)
            iy = (np.arange(dst_h)[:, None] * iscale_y + np.arange(iscale_y)[None, :]).astype(np.int32)
            kind = KIND_AREA_FAST2 if (iscale_x, iscale_y) == (2, 2) else KIND_AREA_FAST
            return kind, ix, ones(ix), iy, ones(iy), iscale_x * iscale_y
        ix, ax = cv_area_tab(src_w, dst_w)
        iy, ay = cv_area_tab(src_h, dst_h)
        return KIND_AREA, ix, ax.astype(np.float64), iy, ay.astype(np.float64), 1
    area_mode = interp == INTER_AREA
    ix, cx = cv_fixed_axis(src_w, dst_w, interp, area_mode, True)
    iy, cy = cv_fixed_axis(src_h, dst_h, interp, area_mode, False)
    kind = KIND_LINEAR if interp in (INTER_LINEAR, INTER_AREA) else KIND_KERNEL
    return kind, ix, cx.astype(np.float64), iy, cy.astype(np.float64), 1


def cv_apply_plan(src, plan, background=None):
    """The pixel arithmetic of a plan on src (H, W, C) uint8 -> (dst_h, dst_w, C) uint8.  Index -1 (the augmentation's canvas) reads
    `background` (C,)."""
    kind, ix, wx, iy, wy, area = plan
    C = src.shape[2]

    def rows_of(j):                                         # the source rows of vertical tap j, gathered at horizontal tap k: (Ho, Wo, C)
        def at(k):
            r, c = iy[:, j], ix[:, k]
            v = src[np.clip(r, 0, None)][:, np.clip(c, 0, None)].astype(np.int64)
            if background is not None:
                hole = (r[:, None] < 0) | (c[None, :] < 0)
                v = np.where(hole[:, :, None], np.asarray(background, dtype=np.int64)[None, None, :], v)
            return v
        return at

    if kind in (KIND_NEAREST, KIND_COPY):
        return rows_of(0)(0).astype(np.uint8)
    if kind == KIND_LINEAR:
        a, b = wx.astype(np.int64), wy.astype(np.int64)
        S = []
        for j in range(2):
            at = rows_of(j)
            S.append(at(0) * a[None, :, 0, None] + at(1) * a[None, :, 1, None])          # HResizeLinear: int32 row
        v = ((b[:, 0, None, None] * (S[0] >> 4)) >> 16) + ((b[:, 1, None, None] * (S[1] >> 4)) >> 16)
        return (((v + 2) >> 2) & 0xff).astype(np.uint8)                                   # uchar(...): a plain cast
    if kind == KIND_KERNEL:
        a, b = wx.astype(np.int64), wy.astype(np.int64)
        acc = np.zeros((iy.shape[0], ix.shape[0], C), dtype=np.int64)
        for j in range(iy.shape[1]):
            at = rows_of(j)
            row = np.zeros_like(acc)
            for k in range(ix.shape[1]):
                row = row + at(k) * a[None, :, k, None]
            acc = acc + row * b[:, j, None, None]
        acc = ((acc + (1 << 31)) & 0xffffffff) - (1 << 31)                                # int arithmetic wraps at 32 bits
        return np.clip((acc + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    if kind == KIND_AREA:
        a, b = wx.astype(_F32), wy.astype(_F32)
        total = None
        for j in range(iy.shape[1]):
            at = rows_of(j)
            buf = np.zeros((iy.shape[0], ix.shape[0], C), dtype=_F32)
            for k in range(ix.shape[1]):
                buf = buf + at(k).astype(_F32) * a[None, :, k, None]                       # buf[dx] += S[sx] * alpha
            term = b[:, j, None, None] * buf
            total = term if total is None else total + term                               # sum[dx] = beta * buf  /  += beta * buf
        return np.clip(np.rint(total.astype(np.float64)), 0, 255).astype(np.uint8)        # saturate_cast<uchar>(float): cvRound
    total = np.zeros((iy.shape[0], ix.shape[0], C), dtype=np.int64)
    for j in range(iy.shape[1]):
        at = rows_of(j)
        for k in range(ix.shape[1]):
            total = total + at(k)
    if kind == KIND_AREA_FAST2:
        return ((total + 2) >> 2).astype(np.uint8)
    scale = _F32(1.0) / _F32(area)                                                         # scale = 1.f / (scale_x * scale_y)
    return np.clip(np.rint((total.astype(_F32) * scale).astype(np.float64)), 0, 255).astype(np.uint8)


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, dsize=(width, height), interpolation) for 8-bit images [H, W] / [H, W, C] (see the block comment above)."""
    if img.dtype != np.uint8:
        raise TypeError("resize: 8-bit images")
    wo, ho = int(dsize[0]), int(dsize[1])
    src = img if img.ndim == 3 else img[:, :, None]
    out = cv_apply_plan(src, cv_resize_plan(src.shape[0], src.shape[1], ho, wo, interpolation))
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
