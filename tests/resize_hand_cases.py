"""cv2.resize on 8-bit images: cases worked out BY HAND from the arithmetic OpenCV publishes in modules/imgproc/src/resize.cpp (3.4 / 4.x,
the C++ reference paths) -- VERDICT r5 item 7.  Shared by the CPU tests (oracle/np_image.resize; the product's tables evaluated by the
oracle's pixel rule) and the GPU tests (the product's kernels).  Every case: (name, source image, (width, height), interpolation,
expected image), the derivation in the comment above it.  OpenCV itself is not installed anywhere this project runs: these are what its
source says, not what a binary returned."""
import numpy as np

NEAREST, LINEAR, CUBIC, AREA, LANCZOS4 = 0, 1, 2, 3, 4
u8 = lambda a: np.array(a, dtype=np.uint8)

CASES = []

# --- INTER_LINEAR, a 1 x 2 row [0, 255] -> 1 x 4.  scale_x = 0.5: fx = (dx + 0.5) 0.5 - 0.5 = -0.25, 0.25, 0.75, 1.25.
#   dx 0: sx = -1 < 0 -> (sx, fx) = (0, 0): coefficients (2048, 0), row value 0 * 2048 = 0
#   dx 1: sx 0, fx 0.25: (1536, 512): 0 * 1536 + 255 * 512 = 130560
#   dx 2: sx 0, fx 0.75: (512, 1536): 255 * 1536 = 391680
#   dx 3: sx 1 >= width - 1 -> (1, 0): 255 * 2048 = 522240
#   vertical: one row in, one row out, fy = 0 -> b = (2048, 0), both taps the clamped row 0:
#   dst = (((2048 (S >> 4)) >> 16) + 0 + 2) >> 2: 0 -> 0; 130560 >> 4 = 8160, x 2048 >> 16 = 255, (255 + 2) >> 2 = 64;
#   391680 >> 4 = 24480 -> 765 -> 191; 522240 >> 4 = 32640 -> 1020 -> 255.
CASES.append(("linear_2_to_4", u8([[0, 255]]), (4, 1), LINEAR, u8([[0, 64, 191, 255]])))

# --- the same kernel vertically, a 2 x 1 column [0, 255]^T -> 4 x 1: the ROW loop does not reset (sy, fy): fy = -0.25 -> sy = -1, fy = 0.75,
#   rows clip(-1) = 0 and clip(0) = 0 with b = (512, 1536); fy = 1.25 -> sy = 1, fy = 0.25, rows 1 and clip(2) = 1 with b = (1536, 512).
#   Horizontally width 1 -> 1: fx = 0, sx = 0 >= width - 1: (2048, 0): S = p * 2048, S >> 4 = 128 p.
#   dy 0: p = 0 -> 0.   dy 1: sy 0, fy 0.25: ((1536 * 0) >> 16) + ((512 * 32640) >> 16 = 255) = 255 -> (255 + 2) >> 2 = 64
#   dy 2: sy 0, fy 0.75: (1536 * 32640) >> 16 = 765 -> 191.   dy 3: both rows 255: ((1536 * 32640) >> 16) + ((512 * 32640) >> 16) = 765 + 255 -> 255
CASES.append(("linear_2_to_4_vertical", u8([[0], [255]]), (1, 4), LINEAR, u8([[0], [64], [191], [255]])))

# --- the two-stage rounding shows: a 1 x 2 row [1, 2] -> 1 x 3.  scale = 2 / 3 (inv 1.5): fx = -1/6 -> reset (0, 0); dx 1: fx = 0.5:
#   (1024, 1024): 1 * 1024 + 2 * 1024 = 3072; dx 2: fx = 7/6 -> sx 1 -> reset (1, 0): 2 * 2048 = 4096.
#   vertical b = (2048, 0): 2048 >> 4 = 128 -> (2048 * 128) >> 16 = 4 -> (4 + 2) >> 2 = 1;  3072 >> 4 = 192 -> 6 -> (6 + 2) >> 2 = 2
#   (the exact value 1.5 goes UP: + 2 >> 2 rounds half up);  4096 >> 4 = 256 -> 8 -> 10 >> 2 = 2.
CASES.append(("linear_half_goes_up", u8([[1, 2]]), (3, 1), LINEAR, u8([[1, 2, 2]])))

# --- INTER_LINEAR at exactly 2 x 2 shrinking IS ResizeAreaFast: (a + b + c + d + 2) >> 2.  [[1, 2], [3, 4]] -> (10 + 2) >> 2 = 3 (the bilinear
#   value 2.5 would round to 2 under cvRound); a 2 x 4 image -> 1 x 2: blocks (1 + 2 + 5 + 6 + 2) >> 2 = 4, (3 + 4 + 7 + 9 + 2) >> 2 = 6.
CASES.append(("linear_2x2_is_area_fast", u8([[1, 2], [3, 4]]), (1, 1), LINEAR, u8([[3]])))
CASES.append(("area_2x2_blocks", u8([[1, 2, 3, 4], [5, 6, 7, 9]]), (2, 1), AREA, u8([[4, 6]])))

# --- ResizeAreaFast at other integer scales: saturate_cast<uchar>(sum * (1.f / area)), cvRound = nearest-even.
#   4 x 1 -> 1 x 1 (scales 4 and 1, area 4): [1, 2, 3, 4]: 10 * 0.25 = 2.5 -> 2 (half to even; the 2 x 2 rule would give 3)
#   [1, 2, 3, 5]: 11 * 0.25 = 2.75 -> 3.   3 x 3 -> 1 x 1 (area 9, 1.f / 9 = 0.11111111f): nine 7s: 63 * 0.11111111f = 7.0000000 -> 7;
#   [[0, 0, 0], [0, 0, 0], [0, 0, 5]]: 5 * 0.11111111f = 0.5555556 -> 1;  sum 4: 0.44444445 -> 0.
CASES.append(("area_fast_4x1_tie_to_even", u8([[1, 2, 3, 4]]), (1, 1), AREA, u8([[2]])))
CASES.append(("area_fast_4x1", u8([[1, 2, 3, 5]]), (1, 1), AREA, u8([[3]])))
CASES.append(("area_fast_3x3_sevens", u8([[7, 7, 7]] * 3), (1, 1), AREA, u8([[7]])))
CASES.append(("area_fast_3x3_five", u8([[0, 0, 0], [0, 0, 0], [0, 0, 5]]), (1, 1), AREA, u8([[1]])))
CASES.append(("area_fast_3x3_four", u8([[0, 0, 0], [0, 4, 0], [0, 0, 0]]), (1, 1), AREA, u8([[0]])))

# --- ResizeArea (non-integer shrink), 3 x 3 -> 2 x 2, scale 1.5.  computeResizeAreaTab: dx 0: cells [0, 1.5): pixel 0 with 1 / 1.5 and pixel 1
#   with 0.5 / 1.5; dx 1: [1.5, 3): pixel 1 with 0.5 / 1.5, pixel 2 with 1 / 1.5 (float32: 0.6666667, 0.33333334).
#   Image rows [0, 30, 60], [0, 30, 60], [90, 90, 90]:
#   horizontal: row 0 / 1: dx 0: 0 * 0.6666667 + 30 * 0.33333334 = 10.0, dx 1: 30 * 0.33333334 + 60 * 0.6666667 = 10.0 + 40.000004 = 50.000004
#               row 2: 90 * 0.6666667 + 90 * 0.33333334 = 60.000004 + 30.0 = 90.0 (float32) both columns
#   vertical dy 0: 0.6666667 * 10 + 0.33333334 * 10 = 6.666667 + 3.3333335 = 10.000001 -> 10;  column 1: 33.333336 + 16.666668 = 50.000004 -> 50
#            dy 1: 0.33333334 * 10 + 0.6666667 * 90 = 3.3333335 + 60.000004 = 63.333336 -> 63;  column 1: 16.666668 + 60.000004 = 76.66667 -> 77
CASES.append(("area_3_to_2", u8([[0, 30, 60], [0, 30, 60], [90, 90, 90]]), (2, 2), AREA, u8([[10, 50], [63, 77]])))

# --- INTER_AREA when an axis grows is cv2's "area_mode" bilinear on BOTH axes: 1 x 2 -> 1 x 4: sx = floor(dx 0.5) = 0, 0, 1, 1 and
#   fx = (dx + 1) - (sx + 1) 2 = -1, 0, -1, 0 -> 0: coefficients (2048, 0): pixel replication.
CASES.append(("area_enlarging_replicates", u8([[10, 200]]), (4, 1), AREA, u8([[10, 10, 200, 200]])))

# --- INTER_CUBIC, a ramp 1 x 8 [0, 10, ..., 70] -> 1 x 4.  fx = 2 dx + 0.5: sx = 2 dx, fx = 0.5: interpolateCubic(0.5) =
#   (-0.09375, 0.59375, 0.59375, -0.09375) -> shorts (-192, 1216, 1216, -192); taps sx - 1 .. sx + 2, clamped.
#   dx 0: pixels (0, 0, 10, 20): 12160 - 3840 = 8320;  dx 1: (10, 20, 30, 40): -1920 + 24320 + 36480 - 7680 = 51200
#   dx 2: (30, 40, 50, 60): -5760 + 48640 + 60800 - 11520 = 92160;  dx 3: (50, 60, 70, 70): -9600 + 72960 + 85120 - 13440 = 135040
#   vertical fy = 0: (0, 2048, 0, 0) on the clamped row: (S 2048 + 2^21) >> 22 = (S + 1024) >> 11: 9344 >> 11 = 4; 52224 >> 11 = 25;
#   93184 >> 11 = 45; 136064 >> 11 = 66.
CASES.append(("cubic_8_to_4_ramp", u8([[0, 10, 20, 30, 40, 50, 60, 70]]), (4, 1), CUBIC, u8([[4, 25, 45, 66]])))

# --- INTER_CUBIC overshoot saturates: an edge [0, 0, 0, 0, 255, 255, 255, 255] -> 1 x 4: dx 1: (0, 0, 0, 255): 255 * -192 = -48960:
#   (-48960 + 1024) >> 11 = -24 -> 0;  dx 2: (0, 255, 255, 255): 255 * 2240 = 571200: 572224 >> 11 = 279 -> 255.
CASES.append(("cubic_overshoot_saturates", u8([[0, 0, 0, 0, 255, 255, 255, 255]]), (4, 1), CUBIC, u8([[0, 0, 255, 255]])))

# --- INTER_NEAREST 1 x 6 -> 1 x 4: floor(dx * (1 / (4 / 6))) = floor(0, 1.5, 3.0.., 4.5) = 0, 1, 3, 4.
CASES.append(("nearest_6_to_4", u8([[10, 11, 12, 13, 14, 15]]), (4, 1), NEAREST, u8([[10, 11, 13, 14]])))

# --- equal sizes are a copy in every mode (cv::resize returns src.copyTo(dst) first thing)
_same = u8([[3, 250, 7], [0, 128, 255]])
for _m in range(5):
    CASES.append(("copy_mode_%d" % _m, _same, (3, 2), _m, _same))

# the fixed-point tables themselves: (n_src, n_dst, interpolation, horizontal) -> (indices, shorts)
TABLES = [
    # linear 2 -> 4, columns: resets at both ends; rows: no reset, clamped indices with the fractions kept (0.75 | 0.25 | 0.75 | 0.25)
    (2, 4, LINEAR, True, [[0, 1], [0, 1], [0, 1], [1, 1]], [[2048, 0], [1536, 512], [512, 1536], [2048, 0]]),
    (2, 4, LINEAR, False, [[0, 0], [0, 1], [0, 1], [1, 1]], [[512, 1536], [1536, 512], [512, 1536], [1536, 512]]),
    # cubic 8 -> 4: fx = 0.5 everywhere;  5 -> 4 (scale 1.25): fx = 0.125, 0.375, 0.625, 0.875
    (8, 4, CUBIC, True, [[0, 0, 1, 2], [1, 2, 3, 4], [3, 4, 5, 6], [5, 6, 7, 7]], [[-192, 1216, 1216, -192]] * 4),
    # cubic at fx = 0.25 (4 -> 8, dx = 1: 1.5 * 0.5 - 0.5 = 0.25): (-0.10546875, 0.87890625, 0.26171875, -0.03515625) * 2048
    (4, 8, CUBIC, True, None, {1: [-216, 1800, 536, -72]}),
]
