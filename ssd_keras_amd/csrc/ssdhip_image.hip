// ssdhip_image.hip -- the image half of the reference's training-time augmentation on gfx950 (MI355X): SURVEY 8f row 4.
//
// The reference distorts one NumPy image at a time on the host through OpenCV (data_generator/object_detection_2d_photometric_ops.py
// :23-480, object_detection_2d_geometric_ops.py:27-148, the chain data_augmentation_chain_original_ssd.py:146-280); a batch-32 step of
// the detector takes 2.3 ms here, so the input pipeline has to live where the images are.  Three kernels:
//   * pixel_program_kernel   every POINTWISE operation of the photometric ops as a per-image program of up to 16 (op, argument) steps
//                            run by one thread per pixel: dtype conversions (the reference's uint8 <-> float32 round trips with
//                            their roundings), brightness / contrast / saturation / hue in NumPy's arithmetic for the array's
//                            dtype (float32 ops for float32 images, float64 for uint8 ones, truncating in-place stores), 8-bit and
//                            float32 RGB <-> HSV, grey, channel permutations.  One launch distorts a whole batch, each image with
//                            its own program (the random draws stay on the host, in the reference's order);
//   * resize_taps_kernel     cv2.resize as separable resampling: the host builds, per output column / row, the source indices and
//                            float64 weights of its taps (nearest 1, linear 2, cubic 4, Lanczos 8, area ceil(scale) + 1) -- one
//                            kernel serves every interpolation mode; float64 accumulation in a fixed order, one rounding;
//   * hist_u8_kernel / lut_u8_kernel   cv2.equalizeHist (histogram on the device, the 256-entry table on the host) and cv2.LUT.
// Arithmetic is written operation by operation (-ffp-contract=off): bit-identical to the NumPy restatement in oracle/np_image.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

enum { IMG_U8 = 0, IMG_F32 = 1, IMG_F64 = 2 };
enum { OP_END = 0, OP_TO_F32 = 1, OP_TO_U8 = 2, OP_BRIGHTNESS = 3, OP_CONTRAST = 4, OP_SATURATION = 5, OP_HUE = 6, OP_RGB2HSV = 7,
       OP_HSV2RGB = 8, OP_RGB2GRAY = 9, OP_SWAP = 10 };
constexpr int IMG_PROG = 16;

__device__ __forceinline__ double img_clip255(double v) { return v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); }
__device__ __forceinline__ float img_clip255f(float v) { return v < 0.f ? 0.f : (v > 255.f ? 255.f : v); }
// np.remainder (the sign of the divisor), float32 / float64
__device__ __forceinline__ float img_modf(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f) { if ((b < 0.f) != (m < 0.f)) m += b; } else m = copysignf(0.f, b);
    return m;
}
__device__ __forceinline__ double img_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; } else m = copysign(0.0, b);
    return m;
}
// value stored into a uint8 array by `a[..] = float_expression`: C truncation (the callers keep it inside [0, 255])
__device__ __forceinline__ double img_trunc_u8(double v) { return (double)(unsigned char)(int)v; }

__device__ __forceinline__ int img_div_table(int numerator_shifted, int i, int six) {      // saturate_cast<int>(num / (six * i)): nearest, ties to even
    if (i == 0) return 0;
    const int den = six * i;
    int q = numerator_shifted / den;
    const int r = numerator_shifted - q * den;
    if (2 * r > den || (2 * r == den && (q & 1))) ++q;
    return q;
}

// sdiv / hdiv: OpenCV's two 256-entry division tables (saturate_cast<int>((255 << 12) / v), ((180 << 12) / (6 diff))) in LDS, or null:
// the entries computed on the spot (two integer divisions per pixel -- what the first version did for every pixel)
__device__ __forceinline__ void img_rgb2hsv_u8(double (&c)[3], const int* sdiv = nullptr, const int* hdiv = nullptr) {
    const int r = (int)c[0], g = (int)c[1], b = (int)c[2];
    const int v = max(max(r, g), b), vmin = min(min(r, g), b), diff = v - vmin;
    const int s = (diff * (sdiv ? sdiv[v] : img_div_table(255 << 12, v, 1)) + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * (hdiv ? hdiv[diff] : img_div_table(180 << 12, diff, 6)) + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    c[0] = (double)min(max(h, 0), 255); c[1] = (double)s; c[2] = (double)v;
}

__device__ __forceinline__ void img_hsv2rgb_float(float h, float s, float v, float hscale, float (&rgb)[3]) {
    if (s == 0.f) { rgb[0] = rgb[1] = rgb[2] = v; return; }
    h = h * hscale;
    if (h < 0.f) h = h + 6.f;
    if (h >= 6.f) h = h - 6.f;
    int sector = (int)floorf(h);
    h = h - (float)sector;
    if (sector < 0 || sector >= 6) { sector = 0; h = 0.f; }
    float tab[4];
    tab[0] = v;
    tab[1] = v * (1.f - s);
    tab[2] = v * (1.f - s * h);
    tab[3] = v * (1.f - s * (1.f - h));
    const int pick[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};      // (b, g, r)
    rgb[2] = tab[pick[sector][0]]; rgb[1] = tab[pick[sector][1]]; rgb[0] = tab[pick[sector][2]];
}

__device__ __forceinline__ void img_rgb2hsv_f32(double (&c)[3]) {
    const float r = (float)c[0], g = (float)c[1], b = (float)c[2];
    const float eps = 1.1920928955078125e-07f;
    const float v = fmaxf(fmaxf(r, g), b), vmin = fminf(fminf(r, g), b);
    const float diff = v - vmin;
    const float s = diff / (fabsf(v) + eps);
    const float d = 60.f / (diff + eps);
    float h = v == r ? (g - b) * d : (v == g ? (b - r) * d + 120.f : (r - g) * d + 240.f);
    if (h < 0.f) h = h + 360.f;
    c[0] = (double)h; c[1] = (double)s; c[2] = (double)v;
}

// One program step on N pixels (the same step for all of them: a program is per image).  `tag` = the dtype the reference's array has at
// this point of the chain; arithmetic per dtype as NumPy / OpenCV do it (see the file header).
template <int N>
__device__ __forceinline__ void img_apply(double (&px)[N][3], int& tag, const int o, const double a, const int* sdiv, const int* hdiv) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double (&c)[3] = px[n];
        if (o == OP_TO_F32) {
            if (tag == IMG_F64) { c[0] = (double)(float)c[0]; c[1] = (double)(float)c[1]; c[2] = (double)(float)c[2]; }
        } else if (o == OP_TO_U8) {                                  // np.round(image, 0).astype(uint8)
#pragma unroll
            for (int q = 0; q < 3; ++q) c[q] = img_clip255(tag == IMG_F32 ? (double)rintf((float)c[q]) : rint(c[q]));
        } else if (o == OP_BRIGHTNESS) {                             // np.clip(image + delta, 0, 255)
            if (tag == IMG_F32) {
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = (double)img_clip255f((float)c[q] + (float)a);
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = img_clip255(c[q] + a);
            }
        } else if (o == OP_CONTRAST) {                               // np.clip(127.5 + factor * (image - 127.5), 0, 255)
            if (tag == IMG_F32) {
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = (double)img_clip255f(127.5f + (float)a * ((float)c[q] - 127.5f));
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = img_clip255(127.5 + a * (c[q] - 127.5));
            }
        } else if (o == OP_SATURATION) {                             // image[:, :, 1] = np.clip(image[:, :, 1] * factor, 0, 255)
            if (tag == IMG_F32) c[1] = (double)img_clip255f((float)c[1] * (float)a);
            else if (tag == IMG_U8) c[1] = img_trunc_u8(img_clip255(c[1] * a));
            else c[1] = img_clip255(c[1] * a);
        } else if (o == OP_HUE) {                                    // image[:, :, 0] = (image[:, :, 0] + delta) % 180.0
            if (tag == IMG_F32) c[0] = (double)img_modf((float)c[0] + (float)a, 180.f);
            else if (tag == IMG_U8) c[0] = img_trunc_u8(img_mod(c[0] + a, 180.0));
            else c[0] = img_mod(c[0] + a, 180.0);
        } else if (o == OP_RGB2HSV) {
            if (tag == IMG_U8) img_rgb2hsv_u8(c, sdiv, hdiv);
            else img_rgb2hsv_f32(c);
        } else if (o == OP_HSV2RGB) {
            float rgb[3];
            if (tag == IMG_U8) {
                img_hsv2rgb_float((float)c[0], (float)c[1] * (float)(1.0 / 255.0), (float)c[2] * (float)(1.0 / 255.0), (float)(6.0 / 180.0), rgb);
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = (double)img_clip255f(rintf(rgb[q] * 255.f));
            } else {
                img_hsv2rgb_float((float)c[0], (float)c[1], (float)c[2], (float)(6.0 / 360.0), rgb);
#pragma unroll
                for (int q = 0; q < 3; ++q) c[q] = (double)rgb[q];
            }
        } else if (o == OP_RGB2GRAY) {
            double gr;
            if (tag == IMG_U8) {
                gr = (double)(((int)c[0] * 4899 + (int)c[1] * 9617 + (int)c[2] * 1868 + (1 << 13)) >> 14);
            } else {
                const float t = (float)c[0] * 0.299f + (float)c[1] * 0.587f;
                gr = (double)(t + (float)c[2] * 0.114f);
            }
            c[0] = c[1] = c[2] = gr;
        } else if (o == OP_SWAP) {                                   // image[:, :, order], order packed as o0 + 4 o1 + 16 o2
            const int code = (int)a;
            const double t0 = c[code & 3], t1 = c[(code >> 2) & 3], t2 = c[(code >> 4) & 3];
            c[0] = t0; c[1] = t1; c[2] = t2;
        }
    }
    // the dtype after the step (the same for every pixel)
    if (o == OP_TO_F32) tag = IMG_F32;
    else if (o == OP_TO_U8) tag = IMG_U8;
    else if ((o == OP_BRIGHTNESS || o == OP_CONTRAST) && tag != IMG_F32) tag = IMG_F64;
}

// x: [n_images][pixels][3] uint8, float32 or float64; y: uint8 / float32 / float64 as out_tag says; ops / args: [n_images][IMG_PROG].
__global__ __launch_bounds__(256) void pixel_program_kernel(const void* __restrict__ x, int in_tag, void* __restrict__ y, int out_tag,
                                                            long long pixels, const int* __restrict__ ops, const double* __restrict__ args) {
    const int img = blockIdx.y;
    const int* op = ops + (size_t)img * IMG_PROG;
    const double* arg = args + (size_t)img * IMG_PROG;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
        const size_t e = ((size_t)img * pixels + p) * 3;
        double c[3];
        int tag = in_tag;
        if (in_tag == IMG_U8) {
            const unsigned char* s = static_cast<const unsigned char*>(x) + e;
            c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
        } else if (in_tag == IMG_F32) {
            const float* s = static_cast<const float*>(x) + e;
            c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
        } else {
            const double* s = static_cast<const double*>(x) + e;
            c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
        }
        double px[1][3] = {{c[0], c[1], c[2]}};
        for (int k = 0; k < IMG_PROG; ++k) {
            const int o = op[k];
            if (o == OP_END) break;
            img_apply<1>(px, tag, o, arg[k], nullptr, nullptr);
        }
        c[0] = px[0][0]; c[1] = px[0][1]; c[2] = px[0][2];
        if (out_tag == IMG_U8) {
            unsigned char* d = static_cast<unsigned char*>(y) + e;
            d[0] = (unsigned char)c[0]; d[1] = (unsigned char)c[1]; d[2] = (unsigned char)c[2];
        } else if (out_tag == IMG_F32) {
            float* d = static_cast<float*>(y) + e;
            d[0] = (float)c[0]; d[1] = (float)c[1]; d[2] = (float)c[2];
        } else {
            double* d = static_cast<double*>(y) + e;
            d[0] = c[0]; d[1] = c[1]; d[2] = c[2];
        }
    }
}

// The uint8 -> uint8 form (the photometric distortions of a whole batch): FOUR pixels = 12 bytes = three dwords per thread and trip, so
// a wave moves 768 contiguous bytes per load / store instruction instead of 64 single bytes at a 3-byte stride; every program step is
// decoded once for the four pixels; OpenCV's two 8-bit HSV division tables sit in LDS (built once per workgroup) instead of two integer
// divisions per pixel.  Same arithmetic, same results as pixel_program_kernel.  pixels % 4 == 0 (the host checks; else the generic kernel).
__global__ __launch_bounds__(256) void pixel_program_u8x4_kernel(const unsigned int* __restrict__ x, unsigned int* __restrict__ y, long long pixels,
                                                                 const int* __restrict__ ops, const double* __restrict__ args) {
    __shared__ int sdiv[256], hdiv[256];
    sdiv[threadIdx.x] = img_div_table(255 << 12, (int)threadIdx.x, 1);
    hdiv[threadIdx.x] = img_div_table(180 << 12, (int)threadIdx.x, 6);
    __syncthreads();
    const int img = blockIdx.y;
    const int* op = ops + (size_t)img * IMG_PROG;
    const double* arg = args + (size_t)img * IMG_PROG;
    const long long groups = pixels / 4;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
        const size_t e = ((size_t)img * groups + g) * 3;                 // dword index of the group's 12 bytes
        const unsigned int w0 = x[e], w1 = x[e + 1], w2 = x[e + 2];
        double px[4][3];
        px[0][0] = w0 & 255u; px[0][1] = (w0 >> 8) & 255u; px[0][2] = (w0 >> 16) & 255u;
        px[1][0] = w0 >> 24; px[1][1] = w1 & 255u; px[1][2] = (w1 >> 8) & 255u;
        px[2][0] = (w1 >> 16) & 255u; px[2][1] = w1 >> 24; px[2][2] = w2 & 255u;
        px[3][0] = (w2 >> 8) & 255u; px[3][1] = (w2 >> 16) & 255u; px[3][2] = w2 >> 24;
        int tag = IMG_U8;
        for (int k = 0; k < IMG_PROG; ++k) {
            const int o = op[k];
            if (o == OP_END) break;
            img_apply<4>(px, tag, o, arg[k], sdiv, hdiv);
        }
        unsigned int b[12];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 3; ++q) b[n * 3 + q] = (unsigned int)(unsigned char)px[n][q];
        y[e] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        y[e + 1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        y[e + 2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    }
}

// x [B,H,W,C] uint8 -> y [B,Ho,Wo,C] uint8.  ix [Wo][nx] / wx, iy [Ho][ny] / wy: the taps of every output column / row.
// out = rint(sum_j wy[j] * (sum_t wx[t] * x[iy[j]][ix[t]])), rows outer, columns inner, float64, clipped to [0, 255].
__global__ __launch_bounds__(256) void resize_taps_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W,
                                                          int Ho, int Wo, int C, const int* __restrict__ ix, const double* __restrict__ wx,
                                                          int nx, const int* __restrict__ iy, const double* __restrict__ wy, int ny) {
    const int b = blockIdx.y;
    const long long total = (long long)Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % C);
        const long long t = i / C;
        const int xo = (int)(t % Wo), yo = (int)(t / Wo);
        const unsigned char* src = x + (size_t)b * H * W * C;
        double acc = 0.0;
        for (int j = 0; j < ny; ++j) {
            const unsigned char* row = src + (size_t)iy[yo * ny + j] * W * C;
            double racc = 0.0;
            for (int k = 0; k < nx; ++k) racc = racc + wx[xo * nx + k] * (double)row[(size_t)ix[xo * nx + k] * C + ch];
            acc = acc + wy[yo * ny + j] * racc;
        }
        const double r = rint(acc);
        y[(size_t)b * total + i] = (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
    }
}

// The same resampling with the work laid out for the memory system: one workgroup = 256 consecutive values of ONE output row (grid
// y = row, z = image), so the row's vertical taps are wave-uniform (scalar loads), a thread keeps ITS column's horizontal taps in
// registers for all the rows it combines (nx <= 8: every mode but strong 'area' shrinks), there is no 64-bit division per value, and the
// 256 gathers of a tap fall into one short stretch of one source row.  Same operations in the same order: identical results.
__global__ __launch_bounds__(256) void resize_rows_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W,
                                                          int Ho, int Wo, int C, const int* __restrict__ ix, const double* __restrict__ wx,
                                                          int nx, const int* __restrict__ iy, const double* __restrict__ wy, int ny) {
    const int v = (int)blockIdx.x * 256 + (int)threadIdx.x, yo = (int)blockIdx.y, b = (int)blockIdx.z;
    if (v >= Wo * C) return;
    const int xo = v / C, ch = v - xo * C;
    int ixr[8];
    double wxr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ixr[k] = k < nx ? ix[xo * nx + k] * C + ch : 0;
        wxr[k] = k < nx ? wx[xo * nx + k] : 0.0;
    }
    const unsigned char* src = x + (size_t)b * H * W * C;
    double acc = 0.0;
    for (int j = 0; j < ny; ++j) {
        const unsigned char* row = src + (size_t)iy[yo * ny + j] * W * C;
        double racc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nx) racc = racc + wxr[k] * (double)row[ixr[k]];
        acc = acc + wy[yo * ny + j] * racc;
    }
    const double r = rint(acc);
    y[((size_t)b * Ho + yo) * Wo * C + v] = (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}

// The geometric half of the augmentation chain for a whole BATCH in one launch: every image has its OWN tap tables -- the taps of
// cv2.resize on the image's crop, composed on the host with the crop / expansion / flip index maps, so a tap addresses a column (row) of
// the ORIGINAL image, or -1 = a position the expansion filled with the background colour.  Nothing but the final 300 x 300 batch is ever
// materialised (the reference builds the expanded canvas, the crop and the flipped view per image on the host).  Layout and arithmetic as
// resize_rows_kernel: out = rint(sum_j wy[j] (sum_k wx[k] pixel)), float64, rows outer; unused taps carry weight 0 and a valid index.
__global__ __launch_bounds__(256) void resize_gather_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W,
                                                            int Ho, int Wo, int C, const int* __restrict__ ix, const double* __restrict__ wx,
                                                            int nx, const int* __restrict__ iy, const double* __restrict__ wy, int ny,
                                                            const unsigned char* __restrict__ background) {
    const int v = (int)blockIdx.x * 256 + (int)threadIdx.x, yo = (int)blockIdx.y, b = (int)blockIdx.z;
    if (v >= Wo * C) return;
    const int xo = v / C, ch = v - xo * C;
    const double bg = (double)background[b * C + ch];
    const int* ixb = ix + ((size_t)b * Wo + xo) * nx;
    const double* wxb = wx + ((size_t)b * Wo + xo) * nx;
    const int* iyb = iy + ((size_t)b * Ho + yo) * ny;
    const double* wyb = wy + ((size_t)b * Ho + yo) * ny;
    const unsigned char* src = x + (size_t)b * H * W * C;
    double acc = 0.0;
    for (int j = 0; j < ny; ++j) {
        const int sy = iyb[j];
        const unsigned char* row = src + (size_t)(sy < 0 ? 0 : sy) * W * C;
        double racc = 0.0;
        for (int k = 0; k < nx; ++k) {
            const int sx = ixb[k];
            const double pv = (sy < 0 || sx < 0) ? bg : (double)row[(size_t)sx * C + ch];
            racc = racc + wxb[k] * pv;
        }
        acc = acc + wyb[j] * racc;
    }
    const double r = rint(acc);
    y[((size_t)b * Ho + yo) * Wo * C + v] = (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// cv2.resize on 8-bit images with OpenCV's OWN arithmetic (round 6; imgproc/resize.cpp, the C++ reference paths -- the kernels above
// resample with float64 weights and one rounding, one grey level off here and there).  The host (data_generator/_image_ops.py
// resize_plan) or the device (csrc/ssdhip_augment.hip) builds the plan: `kind`, tap indices, and the tables as float64 values that hold
// 11-bit fixed-point shorts / float32 area weights / ones exactly.  Per output value, rows j outer, columns k inner:
//   COPY / NEAREST   the single tap
//   LINEAR           S_j = p[j][0] a0 + p[j][1] a1 (int32);  uchar((((b0 (S_0 >> 4)) >> 16) + ((b1 (S_1 >> 4)) >> 16) + 2) >> 2)
//                    (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>: the two-stage form of its mulhi SIMD twin)
//   KERNEL           cubic / Lanczos-4: S_j = sum_k p a_k, saturate((sum_j S_j b_j + (1 << 21)) >> 22), int32 (wrapping) arithmetic
//   AREA             ResizeArea: buf_j = sum_k float(p) alpha_k, total = sum_j beta_j buf_j, all float32 in table order, cvRound, saturate
//   AREA_FAST / 2    ResizeAreaFast: the block's integer sum; saturate(cvRound(float(sum) * (1.f / area))); 2 x 2: (sum + 2) >> 2
// Unused taps of a padded row carry weight 0 and a valid index (a float32 sum is unchanged by + p * 0.f).
// ---------------------------------------------------------------------------------------------------------------------------------------
enum { CV_NEAREST = 0, CV_LINEAR = 1, CV_KERNEL = 2, CV_AREA = 3, CV_AREA_FAST = 4, CV_AREA_FAST2 = 5, CV_COPY = 6 };

template <typename Px>
__device__ __forceinline__ unsigned char cv_resample(const int kind, const int area, const int nx, const int ny, const double* __restrict__ wx,
                                                     const double* __restrict__ wy, Px px) {
    if (kind == CV_NEAREST || kind == CV_COPY) return (unsigned char)px(0, 0);
    if (kind == CV_LINEAR) {
        const int a0 = (int)wx[0], a1 = (int)wx[1], b0 = (int)wy[0], b1 = (int)wy[1];
        const int s0 = px(0, 0) * a0 + px(0, 1) * a1, s1 = px(1, 0) * a0 + px(1, 1) * a1;
        return (unsigned char)((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2);
    }
    if (kind == CV_KERNEL) {
        unsigned int acc = 0u;                                        // unsigned: the wrap-around of the reference's int arithmetic, defined
        for (int j = 0; j < ny; ++j) {
            unsigned int row = 0u;
            for (int k = 0; k < nx; ++k) row += (unsigned int)(px(j, k) * (int)wx[k]);
            acc += row * (unsigned int)(int)wy[j];
        }
        const int v = ((int)(acc + (1u << 21))) >> 22;
        return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    if (kind == CV_AREA) {
        float total = 0.f;
        for (int j = 0; j < ny; ++j) {
            float buf = 0.f;
            for (int k = 0; k < nx; ++k) buf = buf + (float)px(j, k) * (float)wx[k];
            const float term = (float)wy[j] * buf;
            total = j == 0 ? term : total + term;
        }
        const float r = rintf(total);
        return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
    }
    int sum = 0;
    for (int j = 0; j < ny; ++j)
        for (int k = 0; k < nx; ++k) sum += px(j, k);
    if (kind == CV_AREA_FAST2) return (unsigned char)((sum + 2) >> 2);
    const float r = rintf((float)sum * (1.f / (float)area));
    return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

// one plan for the whole batch: grid (blocks over Wo C, Ho, B); ix / wx [Wo][nx], iy / wy [Ho][ny]
__global__ __launch_bounds__(256) void resize_cv_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W,
                                                        int Ho, int Wo, int C, int kind, int area, const int* __restrict__ ix,
                                                        const double* __restrict__ wx, int nx, const int* __restrict__ iy,
                                                        const double* __restrict__ wy, int ny) {
    const int v = (int)blockIdx.x * 256 + (int)threadIdx.x, yo = (int)blockIdx.y, b = (int)blockIdx.z;
    if (v >= Wo * C) return;
    const int xo = v / C, ch = v - xo * C;
    const unsigned char* src = x + (size_t)b * H * W * C;
    const int* ixr = ix + (size_t)xo * nx;
    const int* iyr = iy + (size_t)yo * ny;
    auto px = [&](int j, int k) -> int { return (int)src[((size_t)iyr[j] * W + ixr[k]) * C + ch]; };
    y[((size_t)b * Ho + yo) * Wo * C + v] = cv_resample(kind, area, nx, ny, wx + (size_t)xo * nx, wy + (size_t)yo * ny, px);
}

// every image its own plan (the augmentation chain's gather launch): plan [B][4] = kind, area, taps per column, taps per row (<= nx, ny =
// the tables' strides); ix / wx [B][Wo][nx], iy / wy [B][Ho][ny]; index -1 = the expansion canvas -> background [B][C]
__global__ __launch_bounds__(256) void resize_gather_cv_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H,
                                                               int W, int Ho, int Wo, int C, const int* __restrict__ plan,
                                                               const int* __restrict__ ix, const double* __restrict__ wx, int nx,
                                                               const int* __restrict__ iy, const double* __restrict__ wy, int ny,
                                                               const unsigned char* __restrict__ background) {
    const int v = (int)blockIdx.x * 256 + (int)threadIdx.x, yo = (int)blockIdx.y, b = (int)blockIdx.z;
    if (v >= Wo * C) return;
    const int xo = v / C, ch = v - xo * C;
    const int kind = plan[b * 4], area = plan[b * 4 + 1], tx = plan[b * 4 + 2], ty = plan[b * 4 + 3];
    const int bg = (int)background[b * C + ch];
    const int* ixb = ix + ((size_t)b * Wo + xo) * nx;
    const int* iyb = iy + ((size_t)b * Ho + yo) * ny;
    const unsigned char* src = x + (size_t)b * H * W * C;
    auto px = [&](int j, int k) -> int {
        const int sy = iyb[j], sx = ixb[k];
        return (sy < 0 || sx < 0) ? bg : (int)src[((size_t)sy * W + sx) * C + ch];
    };
    y[((size_t)b * Ho + yo) * Wo * C + v] = cv_resample(kind, area, tx < nx ? tx : nx, ty < ny ? ty : ny, wx + ((size_t)b * Wo + xo) * nx,
                                                        wy + ((size_t)b * Ho + yo) * ny, px);
}

__global__ __launch_bounds__(256) void hist_u8_kernel(const unsigned char* __restrict__ x, long long n_pixels, int C, int channel,
                                                      unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_pixels; p += (long long)gridDim.x * 256)
        atomicAdd(&h[x[(size_t)p * C + channel]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// y = x with table[] applied to the channels whose bit is set in channel_mask
__global__ __launch_bounds__(256) void lut_u8_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, long long n_values,
                                                     int C, int channel_mask, const unsigned char* __restrict__ table) {
    __shared__ unsigned char t[256];
    t[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_values; i += (long long)gridDim.x * 256) {
        const unsigned char v = x[i];
        y[i] = ((channel_mask >> (int)(i % C)) & 1) ? t[v] : v;
    }
}

}  // namespace ssdhip

using namespace ssdhip;

static unsigned img_blocks(long long work, unsigned cap) {
    long long b = (work + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

extern "C" int ssdhip_image_program(const void* x, int in_dtype, void* y, int out_dtype, int n_images, long long pixels_per_image,
                                    const int* ops_dev, const double* args_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !ops_dev || !args_dev || n_images <= 0 || n_images > 65535 || pixels_per_image <= 0) return SSDHIP_E_BADARG;
    if (in_dtype < IMG_U8 || in_dtype > IMG_F64 || out_dtype < IMG_U8 || out_dtype > IMG_F64) return SSDHIP_E_BADARG;
    if (in_dtype == IMG_U8 && out_dtype == IMG_U8 && pixels_per_image % 4 == 0 && !(((uintptr_t)x | (uintptr_t)y) & 3)) {
        hipLaunchKernelGGL(pixel_program_u8x4_kernel, dim3(img_blocks(pixels_per_image / 4, 2048), n_images), dim3(256), 0, stream,
                           static_cast<const unsigned int*>(x), static_cast<unsigned int*>(y), pixels_per_image, ops_dev, args_dev);
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(pixel_program_kernel, dim3(img_blocks(pixels_per_image, 4096), n_images), dim3(256), 0, stream, x, in_dtype, y, out_dtype,
                       pixels_per_image, ops_dev, args_dev);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_resize_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* ix_dev,
                                      const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !ix_dev || !wx_dev || !iy_dev || !wy_dev) return SSDHIP_E_BADARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || C > 4 || nx <= 0 || ny <= 0 || nx > 64 || ny > 64) return SSDHIP_E_BADARG;
    if (nx <= 8 && Ho <= 65535 && (long long)Wo * C < 0x7fffff00LL) {
        hipLaunchKernelGGL(resize_rows_kernel, dim3((unsigned)((Wo * C + 255) / 256), (unsigned)Ho, (unsigned)B), dim3(256), 0, stream,
                           static_cast<const unsigned char*>(x), static_cast<unsigned char*>(y), H, W, Ho, Wo, C, ix_dev, wx_dev, nx, iy_dev, wy_dev, ny);
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(resize_taps_kernel, dim3(img_blocks((long long)Ho * Wo * C, 4096), B), dim3(256), 0, stream,
                       static_cast<const unsigned char*>(x), static_cast<unsigned char*>(y), H, W, Ho, Wo, C, ix_dev, wx_dev, nx, iy_dev, wy_dev, ny);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_resize_gather_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* ix_dev,
                                             const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny,
                                             const void* background_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !ix_dev || !wx_dev || !iy_dev || !wy_dev || !background_dev) return SSDHIP_E_BADARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || Ho <= 0 || Ho > 65535 || Wo <= 0 || C <= 0 || C > 4 || nx <= 0 || ny <= 0 || nx > 64 || ny > 64)
        return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(resize_gather_kernel, dim3((unsigned)((Wo * C + 255) / 256), (unsigned)Ho, (unsigned)B), dim3(256), 0, stream,
                       static_cast<const unsigned char*>(x), static_cast<unsigned char*>(y), H, W, Ho, Wo, C, ix_dev, wx_dev, nx, iy_dev, wy_dev, ny,
                       static_cast<const unsigned char*>(background_dev));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_resize_cv_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, int kind, int area,
                                         const int* ix_dev, const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev, int ny,
                                         void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !ix_dev || !wx_dev || !iy_dev || !wy_dev) return SSDHIP_E_BADARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || Ho <= 0 || Ho > 65535 || Wo <= 0 || C <= 0 || C > 4 || nx <= 0 || ny <= 0 || nx > 64 ||
        ny > 64 || kind < 0 || kind > CV_COPY || area < 1)
        return SSDHIP_E_BADARG;
    if ((kind == CV_LINEAR && (nx != 2 || ny != 2)) || ((kind == CV_NEAREST || kind == CV_COPY) && (nx != 1 || ny != 1))) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(resize_cv_kernel, dim3((unsigned)((Wo * C + 255) / 256), (unsigned)Ho, (unsigned)B), dim3(256), 0, stream,
                       static_cast<const unsigned char*>(x), static_cast<unsigned char*>(y), H, W, Ho, Wo, C, kind, area, ix_dev, wx_dev, nx,
                       iy_dev, wy_dev, ny);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_resize_gather_cv_u8(const void* x, void* y, int B, int H, int W, int Ho, int Wo, int C, const int* plan_dev,
                                                const int* ix_dev, const double* wx_dev, int nx, const int* iy_dev, const double* wy_dev,
                                                int ny, const void* background_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !plan_dev || !ix_dev || !wx_dev || !iy_dev || !wy_dev || !background_dev) return SSDHIP_E_BADARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || Ho <= 0 || Ho > 65535 || Wo <= 0 || C <= 0 || C > 4 || nx < 2 || ny < 2 || nx > 64 || ny > 64)
        return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(resize_gather_cv_kernel, dim3((unsigned)((Wo * C + 255) / 256), (unsigned)Ho, (unsigned)B), dim3(256), 0, stream,
                       static_cast<const unsigned char*>(x), static_cast<unsigned char*>(y), H, W, Ho, Wo, C, plan_dev, ix_dev, wx_dev, nx,
                       iy_dev, wy_dev, ny, static_cast<const unsigned char*>(background_dev));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_hist_u8(const void* x, long long n_pixels, int C, int channel, unsigned int* hist_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !hist_dev || n_pixels <= 0 || C <= 0 || channel < 0 || channel >= C) return SSDHIP_E_BADARG;
    if (zero_async(hist_dev, 256 * sizeof(unsigned int), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(hist_u8_kernel, dim3(img_blocks(n_pixels, 1024)), dim3(256), 0, stream, static_cast<const unsigned char*>(x), n_pixels, C,
                       channel, hist_dev);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_image_lut_u8(const void* x, void* y, long long n_values, int C, int channel_mask, const void* table_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || !table_dev || n_values <= 0 || C <= 0 || C > 8) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(lut_u8_kernel, dim3(img_blocks(n_values, 4096)), dim3(256), 0, stream, static_cast<const unsigned char*>(x),
                       static_cast<unsigned char*>(y), n_values, C, channel_mask, static_cast<const unsigned char*>(table_dev));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
