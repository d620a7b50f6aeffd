// ssdhip_train.hip -- memory-bound glue of the TRAINING step's backward pass on gfx950 (MI355X), bf16 NHWC.
//
// The reference trains through Keras/TensorFlow (ssd300_training.ipynb: model.fit_generator with SSDLoss): the backward of
// Conv2D(activation='relu') and MaxPooling2D (models/keras_ssd300.py:274-313) are TF graph ops.  PyTorch-ROCm runs them as
// threshold_backward + a bf16 reduction per layer (bias gradient) and max_pool2d_backward; here
//   * relu_bwd_bias_kernel   ONE pass over (dL/dy, y): writes dL/dy masked by y > 0 (the tensor both convolution gradients consume)
//                            and per-workgroup float32 partial sums of it per channel (the bias gradient, summed in a fixed order:
//                            no atomics, bit-reproducible);
//   * maxpool_bwd_kernel     dL/dx of max-pooling by GATHER: each input pixel recomputes the arg-max of the windows that cover it
//                            (first maximum in row-major window order, NaN wins: max_pool2d's rule) and sums their gradients --
//                            deterministic, no atomics, one write per element.
//   * maxpool2_relu_bwd_bias_kernel   the two above in ONE pass for the Conv2D(relu) -> MaxPooling2D(2, 2) pairs (pool1 .. pool3:
//                            the largest maps of the step): the pre-pool activation is read once, the pooled gradient once, the masked
//                            full-resolution gradient written once -- the unmasked one (368 MB after conv1_2 at batch 32) is never
//                            written or re-read.  Bit-identical to the two kernels run one after the other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

__device__ __forceinline__ float tb2f(u32 h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ u32 tf2b(float f) {             // round to nearest even, NaN stays NaN (as c10::BFloat16)
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// gy, y, out: [n_pixels][C] bf16 as uint4 (8 channels); partial: [gridDim.x][C] float32.  cvec = C / 8 divides 256.
// MASK = false: no activation behind the layer (the packed predictor heads): channel sums of gy only, nothing written to `out`.
template <bool MASK>
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const uint4* __restrict__ gy, const uint4* __restrict__ y,
                                                            uint4* __restrict__ out, float* __restrict__ partial, u32 n_pixels,
                                                            u32 cvec) {
    __shared__ float red[256 * 8];
    const u32 tid = threadIdx.x;
    const u32 cg = tid % cvec, pl = tid / cvec, ppb = 256u / cvec;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (u32 px = blockIdx.x * ppb + pl; px < n_pixels; px += gridDim.x * ppb) {
        const size_t i = (size_t)px * cvec + cg;
        const uint4 g = gy[i];
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (MASK) v = y[i];
        const u32 gw[4] = {g.x, g.y, g.z, g.w}, vw[4] = {v.x, v.y, v.z, v.w};
        u32 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // threshold_backward(grad, y, 0): zero where y <= 0 (a NaN activation lets the gradient through)
            const u32 lo = (MASK && tb2f(vw[q] & 0xffffu) <= 0.f) ? 0u : (gw[q] & 0xffffu);
            const u32 hi = (MASK && tb2f(vw[q] >> 16) <= 0.f) ? 0u : (gw[q] >> 16);
            o[q] = lo | (hi << 16);
            acc[2 * q] += tb2f(lo);
            acc[2 * q + 1] += tb2f(hi);
        }
        if (MASK) out[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[tid * 8 + q] = acc[q];
    __syncthreads();
    if (pl == 0) {                                              // fixed order over the pixel lanes: reproducible sums
        for (u32 j = 1; j < ppb; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += red[(j * cvec + cg) * 8 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) partial[(size_t)blockIdx.x * cvec * 8 + cg * 8 + q] = acc[q];
    }
}

// x: [B,H,W,C] input of the pooling, gy: [B,Ho,Wo,C] gradient of its output, gx: [B,H,W,C].  One thread per (input pixel, 8 channels).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gy,
                                                          uint4* __restrict__ gx, int B, int H, int W, u32 cvec, int k, int s, int p,
                                                          int Ho, int Wo) {
    const u32 total = (u32)B * H * W * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 cg = i % cvec;
        u32 t = i / cvec;
        const int w = t % W; t /= W;
        const int h = t % H;
        const int b = t / H;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // windows (oh, ow) with oh*s - p <= h < oh*s - p + k
        const int oh0 = max((h + p - k + s) / s, 0), oh1 = min((h + p) / s, Ho - 1);
        const int ow0 = max((w + p - k + s) / s, 0), ow1 = min((w + p) / s, Wo - 1);
        for (int oh = oh0; oh <= oh1; ++oh)
            for (int ow = ow0; ow <= ow1; ++ow) {
                const int h0 = max(oh * s - p, 0), h1 = min(oh * s - p + k, H);
                const int w0 = max(ow * s - p, 0), w1 = min(ow * s - p + k, W);
                if (h < h0 || h >= h1 || w < w0 || w >= w1) continue;
                float best[8];
                int arg[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { best[q] = -__builtin_inff(); arg[q] = -1; }
                for (int hi = h0; hi < h1; ++hi)
                    for (int wi = w0; wi < w1; ++wi) {
                        const uint4 v = x[((size_t)(b * H + hi) * W + wi) * cvec + cg];
                        const u32 vw[4] = {v.x, v.y, v.z, v.w};
                        const int pos = hi * W + wi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a = tb2f(vw[q] & 0xffffu), c = tb2f(vw[q] >> 16);
                            if (a > best[2 * q] || a != a) { best[2 * q] = a; arg[2 * q] = pos; }          // max_pool2d: first maximum, NaN wins
                            if (c > best[2 * q + 1] || c != c) { best[2 * q + 1] = c; arg[2 * q + 1] = pos; }
                        }
                    }
                const uint4 g = gy[((size_t)(b * Ho + oh) * Wo + ow) * cvec + cg];
                const u32 gw[4] = {g.x, g.y, g.z, g.w};
                const int me = h * W + w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (arg[2 * q] == me) acc[2 * q] += tb2f(gw[q] & 0xffffu);
                    if (arg[2 * q + 1] == me) acc[2 * q + 1] += tb2f(gw[q] >> 16);
                }
            }
        gx[i] = make_uint4(tf2b(acc[0]) | (tf2b(acc[1]) << 16), tf2b(acc[2]) | (tf2b(acc[3]) << 16), tf2b(acc[4]) | (tf2b(acc[5]) << 16),
                           tf2b(acc[6]) | (tf2b(acc[7]) << 16));
    }
}

// maxpool_bwd_kernel for kernel 3, stride 1, padding 1 (pool5: Ho = H, Wo = W).  A pixel lies in up to nine windows and the gather kernel
// re-read each of them: 81 loads per thread.  Here the pixel's 5 x 5 neighbourhood is loaded ONCE (25 loads in flight together) and the nine
// arg-max searches run on registers, in the gather kernel's order (windows by (oh, ow), a window's pixels by (row, column), first
// maximum, a NaN wins), so the float32 sums and their one rounding are the same.
__global__ __launch_bounds__(256) void maxpool3s1_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gy, uint4* __restrict__ gx,
                                                             int B, int H, int W, u32 cvec) {
    const u32 total = (u32)B * H * W * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 cg = i % cvec;
        u32 t = i / cvec;
        const int w = t % W; t /= W;
        const int h = t % H;
        const int b = t / H;
        bool okr[5], okc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { okr[j] = (unsigned)(h + j - 2) < (unsigned)H; okc[j] = (unsigned)(w + j - 2) < (unsigned)W; }
        uint4 nb[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c)
                nb[r][c] = (okr[r] && okc[c]) ? x[((size_t)(b * H + h + r - 2) * W + (w + c - 2)) * cvec + cg] : make_uint4(0u, 0u, 0u, 0u);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                if (!(okr[dh + 2] && okc[dw + 2])) continue;                 // the window centred on (h + dh, w + dw)
                const uint4 g = gy[((size_t)(b * H + h + dh) * W + (w + dw)) * cvec + cg];
                const u32 gw[4] = {g.x, g.y, g.z, g.w};
                const int me = (1 - dh) * 3 + (1 - dw);                      // this pixel's place in the window's scan
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float best0 = -__builtin_inff(), best1 = -__builtin_inff();
                    int arg0 = -1, arg1 = -1;
#pragma unroll
                    for (int eh = -1; eh <= 1; ++eh)
#pragma unroll
                        for (int ew = -1; ew <= 1; ++ew) {
                            const int r = dh + eh + 2, c = dw + ew + 2;
                            const uint4 v = nb[r][c];
                            const u32 wd = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
                            const float a = tb2f(wd & 0xffffu), cc = tb2f(wd >> 16);
                            const bool ok = okr[r] && okc[c];
                            if (ok && (a > best0 || a != a)) { best0 = a; arg0 = (eh + 1) * 3 + (ew + 1); }
                            if (ok && (cc > best1 || cc != cc)) { best1 = cc; arg1 = (eh + 1) * 3 + (ew + 1); }
                        }
                    if (arg0 == me) acc[2 * q] += tb2f(gw[q] & 0xffffu);
                    if (arg1 == me) acc[2 * q + 1] += tb2f(gw[q] >> 16);
                }
            }
        gx[i] = make_uint4(tf2b(acc[0]) | (tf2b(acc[1]) << 16), tf2b(acc[2]) | (tf2b(acc[3]) << 16), tf2b(acc[4]) | (tf2b(acc[5]) << 16),
                           tf2b(acc[6]) | (tf2b(acc[7]) << 16));
    }
}

// The same pooling (3 x 3, stride 1, padding 1) on a SMALL map, one workgroup per (image, 32 channels): every window's arg-max is found ONCE
// (phase A: the thread of the window's centre, nine loads; a 4-bit place per channel into LDS), then every pixel collects the gradients of the
// windows that point at it (phase B: nine codes from LDS, nine gradient loads) -- the neighbourhood kernel above repeats each search nine
// times (85 us on pool5 at batch 32).  Same windows in the same order, same sums.
__global__ __launch_bounds__(256) void maxpool3s1_bwd_img_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gy, uint4* __restrict__ gx,
                                                                 int H, int W, u32 cvec) {
    extern __shared__ u32 code[];                            // [H W][4]
    const int b = blockIdx.y, items = H * W * 4;
    const size_t img = (size_t)b * H * W;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int px = it >> 2, h = px / W, w = px - h * W;
        const u32 cg = blockIdx.x * 4u + (u32)(it & 3);
        u32 packed = 0;
        uint4 v[9];
        bool ok[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const int hh = h + e / 3 - 1, ww = w + e % 3 - 1;
            ok[e] = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            v[e] = ok[e] ? x[(img + (size_t)hh * W + ww) * cvec + cg] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float best0 = -__builtin_inff(), best1 = -__builtin_inff();
            u32 arg0 = 15u, arg1 = 15u;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const u32 wd = q == 0 ? v[e].x : q == 1 ? v[e].y : q == 2 ? v[e].z : v[e].w;
                const float a = tb2f(wd & 0xffffu), c = tb2f(wd >> 16);
                if (ok[e] && (a > best0 || a != a)) { best0 = a; arg0 = (u32)e; }          // max_pool2d: first maximum, NaN wins
                if (ok[e] && (c > best1 || c != c)) { best1 = c; arg1 = (u32)e; }
            }
            packed |= (arg0 << (8 * q)) | (arg1 << (8 * q + 4));
        }
        code[it] = packed;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {
        const int px = it >> 2, h = px / W, w = px - h * W, j = it & 3;
        const u32 cg = blockIdx.x * 4u + (u32)j;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int hh = h + dh, ww = w + dw;
                if ((unsigned)hh >= (unsigned)H || (unsigned)ww >= (unsigned)W) continue;
                const u32 c = code[(hh * W + ww) * 4 + j];
                const uint4 g = gy[(img + (size_t)hh * W + ww) * cvec + cg];
                const u32 gw[4] = {g.x, g.y, g.z, g.w};
                const u32 me = (u32)((1 - dh) * 3 + (1 - dw));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (((c >> (8 * q)) & 15u) == me) acc[2 * q] += tb2f(gw[q] & 0xffffu);
                    if (((c >> (8 * q + 4)) & 15u) == me) acc[2 * q + 1] += tb2f(gw[q] >> 16);
                }
            }
        gx[(img + (size_t)px) * cvec + cg] = make_uint4(tf2b(acc[0]) | (tf2b(acc[1]) << 16), tf2b(acc[2]) | (tf2b(acc[3]) << 16),
                                                         tf2b(acc[4]) | (tf2b(acc[5]) << 16), tf2b(acc[6]) | (tf2b(acc[7]) << 16));
    }
}

// maxpool_bwd_kernel for kernel 2, stride 2, no padding, windows clipped to the map (Ho = ceil(H/2), Wo = ceil(W/2): pool1 .. pool4): the windows
// do not overlap, so one thread per WINDOW and 8 channels finds the arg-max once and writes the window's four gradients (0 + g rounded at
// the arg-max, +0 elsewhere: the gather kernel's values) -- a quarter of its loads and searches.
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ gy, uint4* __restrict__ gx,
                                                           int H, int W, int Ho, int Wo, u32 n_windows, u32 cvec) {
    const size_t rowp = (size_t)W * cvec;
    const u32 total = n_windows * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 cg = i % cvec, wn = i / cvec;
        const int ow = (int)(wn % (u32)Wo);
        const u32 t = wn / (u32)Wo;
        const int oh = (int)(t % (u32)Ho), b = (int)(t / (u32)Ho);
        const int h0 = 2 * oh, w0 = 2 * ow;
        const bool ok[4] = {true, w0 + 1 < W, h0 + 1 < H, (w0 + 1 < W) && (h0 + 1 < H)};
        const size_t p00 = ((size_t)(b * H + h0) * W + w0) * cvec + cg;
        const size_t pos[4] = {p00, p00 + cvec, p00 + rowp, p00 + rowp + cvec};
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ok[k] ? x[pos[k]] : make_uint4(0u, 0u, 0u, 0u);
        const uint4 g = gy[i];
        const u32 gw[4] = {g.x, g.y, g.z, g.w};
        u32 o[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float best0 = -__builtin_inff(), best1 = -__builtin_inff();
            int arg0 = -1, arg1 = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 w = q == 0 ? v[k].x : q == 1 ? v[k].y : q == 2 ? v[k].z : v[k].w;
                const float a = tb2f(w & 0xffffu), c = tb2f(w >> 16);
                if (ok[k] && (a > best0 || a != a)) { best0 = a; arg0 = k; }          // max_pool2d: first maximum, NaN wins
                if (ok[k] && (c > best1 || c != c)) { best1 = c; arg1 = k; }
            }
            const u32 glo = tf2b(0.f + tb2f(gw[q] & 0xffffu)), ghi = tf2b(0.f + tb2f(gw[q] >> 16));
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][q] = (arg0 == k ? glo : 0u) | ((arg1 == k ? ghi : 0u) << 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ok[k]) gx[pos[k]] = make_uint4(o[k][0], o[k][1], o[k][2], o[k][3]);
    }
}

// y: [B,H,W,C] post-ReLU activation (the pooling's input), gp: [B,Ho,Wo,C] gradient of the pooled map (2x2 windows, stride 2, clipped to
// the map: Ho = ceil(H/2), Wo = ceil(W/2)), out: [B,H,W,C] = dL/dy masked by y > 0, partial: [gridDim.x][C] float32 channel sums of out.
// One thread per WINDOW and 8 channels (round 5; one thread per pixel re-read its window's four activations and searched the maximum
// four times over: 2.2 TB/s of HBM traffic at four times the load instructions): the four activations and the pooled gradient are five
// 16-byte loads in flight together, the arg-max is found once, the four gradients leave as four stores.  The window scan order, the
// first-maximum and NaN rules are max_pool2d's (as maxpool_bwd_kernel); the channel sums add a thread's pixels in that order.
__global__ __launch_bounds__(256) void maxpool2_relu_bwd_bias_kernel(const uint4* __restrict__ y, const uint4* __restrict__ gp,
                                                                     uint4* __restrict__ out, float* __restrict__ partial, int H, int W,
                                                                     int Ho, int Wo, u32 n_windows, u32 cvec) {
    __shared__ float red[256 * 8];
    const u32 tid = threadIdx.x;
    const u32 cg = tid % cvec, wl = tid / cvec, wpb = 256u / cvec;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t rowp = (size_t)W * cvec;
    for (u32 wn = blockIdx.x * wpb + wl; wn < n_windows; wn += gridDim.x * wpb) {
        const int ow = (int)(wn % (u32)Wo);
        const u32 t = wn / (u32)Wo;
        const int oh = (int)(t % (u32)Ho), b = (int)(t / (u32)Ho);
        const int h0 = 2 * oh, w0 = 2 * ow;
        const bool ok[4] = {true, w0 + 1 < W, h0 + 1 < H, (w0 + 1 < W) && (h0 + 1 < H)};
        const size_t p00 = ((size_t)(b * H + h0) * W + w0) * cvec + cg;
        const size_t pos[4] = {p00, p00 + cvec, p00 + rowp, p00 + rowp + cvec};
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ok[k] ? y[pos[k]] : make_uint4(0u, 0u, 0u, 0u);
        const uint4 g = gp[(size_t)wn * cvec + cg];
        const u32 gw[4] = {g.x, g.y, g.z, g.w};
        u32 o[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float best0 = -__builtin_inff(), best1 = -__builtin_inff();
            int arg0 = -1, arg1 = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 w = k == 0 ? (q == 0 ? v[0].x : q == 1 ? v[0].y : q == 2 ? v[0].z : v[0].w)
                            : k == 1 ? (q == 0 ? v[1].x : q == 1 ? v[1].y : q == 2 ? v[1].z : v[1].w)
                            : k == 2 ? (q == 0 ? v[2].x : q == 1 ? v[2].y : q == 2 ? v[2].z : v[2].w)
                                     : (q == 0 ? v[3].x : q == 1 ? v[3].y : q == 2 ? v[3].z : v[3].w);
                const float a = tb2f(w & 0xffffu), c = tb2f(w >> 16);
                if (ok[k] && (a > best0 || a != a)) { best0 = a; arg0 = k; }          // max_pool2d: first maximum, NaN wins
                if (ok[k] && (c > best1 || c != c)) { best1 = c; arg1 = k; }
            }
            // maxpool_bwd_kernel's value (0 + g, rounded: -0 becomes +0) where the pixel is the window's arg-max, then the ReLU mask
            const u32 glo = tf2b(0.f + tb2f(gw[q] & 0xffffu)), ghi = tf2b(0.f + tb2f(gw[q] >> 16));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 w = k == 0 ? (q == 0 ? v[0].x : q == 1 ? v[0].y : q == 2 ? v[0].z : v[0].w)
                            : k == 1 ? (q == 0 ? v[1].x : q == 1 ? v[1].y : q == 2 ? v[1].z : v[1].w)
                            : k == 2 ? (q == 0 ? v[2].x : q == 1 ? v[2].y : q == 2 ? v[2].z : v[2].w)
                                     : (q == 0 ? v[3].x : q == 1 ? v[3].y : q == 2 ? v[3].z : v[3].w);
                const u32 lo = (arg0 != k || tb2f(w & 0xffffu) <= 0.f) ? 0u : glo;
                const u32 hi = (arg1 != k || tb2f(w >> 16) <= 0.f) ? 0u : ghi;
                o[k][q] = lo | (hi << 16);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ok[k]) {
                out[pos[k]] = make_uint4(o[k][0], o[k][1], o[k][2], o[k][3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += tb2f(o[k][q] & 0xffffu); acc[2 * q + 1] += tb2f(o[k][q] >> 16); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) red[tid * 8 + q] = acc[q];
    __syncthreads();
    if (wl == 0) {                                              // fixed order over the window lanes: reproducible sums
        for (u32 j = 1; j < wpb; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += red[(j * cvec + cg) * 8 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) partial[(size_t)blockIdx.x * cvec * 8 + cg * 8 + q] = acc[q];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The FIRST layer's backward in one pass (round 5): conv1_1 (3 -> 64 channels, 3 x 3 'same', ReLU; models/keras_ssd300.py:274) needs no
// data gradient, so its masked output gradient is only ever summed -- into the bias gradient and into dW[co][kh][kw][ci] = sum over
// pixels of g[p][co] x[p + tap][ci].  The three-launch form wrote the masked gradient (368 MB at 300 x 300 / batch 32) for MIOpen's weight
// gradient to read back: 1.1 GB + 0.39 GB of traffic, ~390 us.  Here a tile of 64 pixels of one image row is masked in registers, laid
// into LDS TRANSPOSED ([channel][pixel], so that an MFMA lane's eight consecutive K values -- pixels -- are one 16-byte read) next to
// the tile's im2col patch [k = (kh 3 + kw) 3 + ci][pixel], and four waves accumulate D[64 channels][32 k] with
// v_mfma_f32_32x32x16_bf16 over the pixels (K).  Reads gy + y + x once, writes per-workgroup partial sums (added in order by the caller).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));
typedef float tr_f32x16 __attribute__((ext_vector_type(16)));
constexpr int C11B_PITCH = 144;                              // bytes per LDS row: 64 pixels + padding, 16-byte aligned

__global__ __launch_bounds__(256) void conv1_1_bwd_kernel(const uint4* __restrict__ gy, const uint4* __restrict__ y,
                                                          const unsigned short* __restrict__ x, float* __restrict__ wpart,
                                                          float* __restrict__ bpart, int H, int W, int tiles_per_row, int n_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char gT[64 * C11B_PITCH];       // masked gradient [channel][pixel] bf16
    __shared__ __attribute__((aligned(16))) unsigned char pT[32 * C11B_PITCH];       // im2col patch   [k][pixel] bf16, rows 27 .. 31 zero
    __shared__ unsigned short xs[3][66 * 3 + 2];                                     // the tile's three input rows, one pixel of halo each side
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7, pp = tid >> 3;                   // 8 channels x one PAIR of pixels per thread
    float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tr_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int i = tid; i < 32 * C11B_PITCH / 4; i += 256) reinterpret_cast<u32*>(pT)[i] = 0u;
    const int cohalf = wave & 1, pxhalf = wave >> 1;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tr = tile % tiles_per_row, rowid = tile / tiles_per_row;           // rowid = b H + h
        const int h = rowid % H, w0 = tr * 64;
        const int npx = min(64, W - w0);
        // ---- loads: the pair's gradient and activation vectors, the three input rows ----
        uint4 g[2], a[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int px = 2 * pp + e;
            const bool ok = px < npx;
            const size_t at = ((size_t)rowid * W + w0 + px) * 8 + cg;
            g[e] = ok ? gy[at] : make_uint4(0u, 0u, 0u, 0u);
            a[e] = ok ? y[at] : make_uint4(0u, 0u, 0u, 0u);
        }
        for (int i = tid; i < 3 * 198; i += 256) {
            const int r = i / 198, j = i - r * 198;
            const int col = w0 - 1 + j / 3, row = h + r - 1;
            unsigned short v = 0;
            if ((unsigned)row < (unsigned)H && (unsigned)col < (unsigned)W)
                v = x[((size_t)(rowid + r - 1) * W + col) * 3 + (j % 3)];            // rowid + r - 1 stays inside the image: row is
            xs[r][j] = v;
        }
        // ---- ReLU mask (threshold_backward: zero where y <= 0, a NaN activation lets the gradient through), bias sums, transposed store ----
        u32 m[2][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const u32 gw[4] = {g[e].x, g[e].y, g[e].z, g[e].w}, vw[4] = {a[e].x, a[e].y, a[e].z, a[e].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 lo = (tb2f(vw[q] & 0xffffu) <= 0.f) ? 0u : (gw[q] & 0xffffu);
                const u32 hi = (tb2f(vw[q] >> 16) <= 0.f) ? 0u : (gw[q] >> 16);
                m[e][q] = lo | (hi << 16);
                bacc[2 * q] += tb2f(lo);
                bacc[2 * q + 1] += tb2f(hi);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                         // channels cg 8 + 2 q and + 1: the pair's two pixels side by side in one word
            *reinterpret_cast<u32*>(gT + (cg * 8 + 2 * q) * C11B_PITCH + pp * 4) = (m[0][q] & 0xffffu) | (m[1][q] << 16);
            *reinterpret_cast<u32*>(gT + (cg * 8 + 2 * q + 1) * C11B_PITCH + pp * 4) = (m[0][q] >> 16) | (m[1][q] & 0xffff0000u);
        }
        __syncthreads();
        // ---- im2col, transposed: pT[k][px] = xs[kh][(px + kw) 3 + ci] ----
        for (int i = tid; i < 27 * 32; i += 256) {
            const int k = i >> 5, q = i & 31;
            const int kh = k / 9, rem = k - kh * 9;           // rem = kw 3 + ci
            const u32 v0 = xs[kh][(2 * q) * 3 + rem], v1 = xs[kh][(2 * q + 1) * 3 + rem];
            *reinterpret_cast<u32*>(pT + k * C11B_PITCH + q * 4) = v0 | (v1 << 16);
        }
        __syncthreads();
        // ---- D[32 channels of this wave][32 k] += G[32][16 pixels] P[16 pixels][32]: two K-steps per wave and tile ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px0 = (pxhalf * 2 + ks) * 16 + (lane >> 5) * 8;
            const tr_bf16x8 fa = *reinterpret_cast<const tr_bf16x8*>(gT + (cohalf * 32 + (lane & 31)) * C11B_PITCH + px0 * 2);
            const tr_bf16x8 fb = *reinterpret_cast<const tr_bf16x8*>(pT + (lane & 31) * C11B_PITCH + px0 * 2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        __syncthreads();                                      // the next tile overwrites gT / pT / xs
    }
    // ---- the two pixel halves of a channel half are added through LDS; D row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5), column = lane & 31 ----
    float* dsum = red;                                        // [2 channel halves][16][64 lanes]
    if (pxhalf == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dsum[(cohalf * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (pxhalf == 0) {
        const int k = lane & 31;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = acc[i] + dsum[(cohalf * 16 + i) * 64 + lane];
            const int co = cohalf * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            if (k < 27) wpart[((size_t)blockIdx.x * 64 + co) * 27 + k] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) red[tid * 8 + q] = bacc[q];
    __syncthreads();
    if (pp == 0) {                                            // fixed order over the pixel-pair lanes: reproducible sums
        for (int j = 1; j < 32; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) bacc[q] += red[(j * 8 + cg) * 8 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) bpart[(size_t)blockIdx.x * 64 + cg * 8 + q] = bacc[q];
    }
}

// z[b, off + s i, off + s j, :] = gy[b, i, j, :], zeros everywhere else (one thread per 16 bytes of z): the data gradient of a strided or
// 'valid' 3 x 3 convolution is the 3 x 3 'same' convolution of THIS map with the transposed, tap-flipped filters (see the entry point).
__global__ __launch_bounds__(256) void embed_strided_kernel(const uint4* __restrict__ gy, uint4* __restrict__ z, u32 n, int H, int W, int Ho, int Wo,
                                                            u32 cvec, int stride, int off) {
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const u32 c = i % cvec, px = i / cvec;
        const u32 w = px % (u32)W, r = px / (u32)W, h = r % (u32)H, b = r / (u32)H;
        const int hh = (int)h - off, ww = (int)w - off;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (hh >= 0 && ww >= 0 && hh % stride == 0 && ww % stride == 0) {
            const int ho = hh / stride, wo = ww / stride;
            if (ho < Ho && wo < Wo) v = gy[(((size_t)b * Ho + ho) * Wo + wo) * cvec + c];
        }
        z[i] = v;
    }
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" int ssdhip_relu_bwd_bias_blocks(long long n_pixels, int C) {
    if (n_pixels <= 0 || C <= 0 || C % 8) return 0;
    const int cvec = C / 8;
    if (cvec > 256 || (256 % cvec)) return 0;
    const long long ppb = 256 / cvec;
    long long blocks = (n_pixels + ppb * 8 - 1) / (ppb * 8);           // >= 8 pixels per pixel lane
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" int ssdhip_relu_bwd_bias_nhwc_bf16(const void* gy, const void* y, void* out, float* partial, long long n_pixels, int C,
                                              int n_blocks, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gy || !y || !out || !partial || n_pixels <= 0 || n_pixels > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (n_blocks <= 0 || n_blocks != ssdhip_relu_bwd_bias_blocks(n_pixels, C)) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(relu_bwd_bias_kernel<true>, dim3(n_blocks), dim3(256), 0, stream, static_cast<const uint4*>(gy), static_cast<const uint4*>(y),
                       static_cast<uint4*>(out), partial, (u32)n_pixels, (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Channel sums of a bf16 map alone (the bias gradient of a layer WITHOUT activation: the packed predictor heads): partial [n_blocks][C]
// float32 per-workgroup sums, n_blocks = ssdhip_relu_bwd_bias_blocks(n_pixels, C); the caller -- or the reduction launch of the layer's
// weight gradient, ssdhip_conv3x3_wgrad_bias_nhwc_bf16 -- adds the rows in order.
extern "C" int ssdhip_channel_sums_nhwc_bf16(const void* gy, float* partial, long long n_pixels, int C, int n_blocks, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gy || !partial || n_pixels <= 0 || n_pixels > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (n_blocks <= 0 || n_blocks != ssdhip_relu_bwd_bias_blocks(n_pixels, C)) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(relu_bwd_bias_kernel<false>, dim3(n_blocks), dim3(256), 0, stream, static_cast<const uint4*>(gy), static_cast<const uint4*>(gy),
                       static_cast<uint4*>(nullptr), partial, (u32)n_pixels, (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Backward of Conv2D(relu) -> MaxPooling2D(2, 2, clipped windows) up to the convolution, in one pass: y [B,H,W,C] the post-ReLU
// activation, gp [B,ceil(H/2),ceil(W/2),C] the pooled map's gradient; out = the full-resolution gradient masked by y > 0, partial
// [n_blocks][C] float32 per-workgroup channel sums of it (n_blocks = ssdhip_relu_bwd_bias_blocks(B H W, C); the caller adds them).
extern "C" int ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16(const void* y, const void* gp, void* out, float* partial, int B, int H, int W, int C,
                                                       int n_blocks, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!y || !gp || !out || !partial || B <= 0 || H <= 0 || W <= 0) return SSDHIP_E_BADARG;
    const long long n = (long long)B * H * W;
    if (n > 0x7fffffffLL || n_blocks <= 0 || n_blocks != ssdhip_relu_bwd_bias_blocks(n, C)) return SSDHIP_E_BADARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    hipLaunchKernelGGL(maxpool2_relu_bwd_bias_kernel, dim3(n_blocks), dim3(256), 0, stream, static_cast<const uint4*>(y),
                       static_cast<const uint4*>(gp), static_cast<uint4*>(out), partial, H, W, Ho, Wo, (u32)((long long)B * Ho * Wo), (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

namespace ssdhip {

// ---------------------------------------------------------------------------------------------------------------
// L2Normalization (keras_layers/keras_layer_L2Normalization.py:61-63) for the float32 model and the training step: forward with the
// pixel's inverse norm saved, backward in one pass.  One wave per pixel, float32 math; T = float (4 channels per 16 bytes) or bf16
// (8 channels).  With inv = rsqrt(max(s, 1e-12)), s = sum_c x_c^2:
//     y_c  = x_c inv gamma_c
//     dx_c = inv gamma_c dy_c - x_c inv^3 sum_k(dy_k gamma_k x_k)        (the second term only where s >= 1e-12: a clamped norm is a constant)
//     dgamma_c = sum over pixels of dy_c x_c inv                          (per-wave partial sums, added in a fixed order by the caller)
// ---------------------------------------------------------------------------------------------------------------
constexpr int L2_MAXIT = 8;                                  // channel vectors per lane: C <= 64 * 8 * (4 | 8)

template <bool BF16>
__device__ __forceinline__ void l2_load(const void* base, size_t vec, float (&v)[8]) {
    if constexpr (BF16) {
        const uint4 r = reinterpret_cast<const uint4*>(base)[vec];
        const u32 w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = tb2f(w[q] & 0xffffu); v[2 * q + 1] = tb2f(w[q] >> 16); }
    } else {
        const float4 r = reinterpret_cast<const float4*>(base)[vec];
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
        v[4] = v[5] = v[6] = v[7] = 0.f;
    }
}
template <bool BF16>
__device__ __forceinline__ void l2_store(void* base, size_t vec, const float (&v)[8]) {
    if constexpr (BF16)
        reinterpret_cast<uint4*>(base)[vec] = make_uint4(tf2b(v[0]) | (tf2b(v[1]) << 16), tf2b(v[2]) | (tf2b(v[3]) << 16),
                                                         tf2b(v[4]) | (tf2b(v[5]) << 16), tf2b(v[6]) | (tf2b(v[7]) << 16));
    else
        reinterpret_cast<float4*>(base)[vec] = make_float4(v[0], v[1], v[2], v[3]);
}

template <bool BF16>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma, void* __restrict__ y,
                                                         float* __restrict__ inv_out, u32 n_pixels, u32 cvec) {
    constexpr int CPV = BF16 ? 8 : 4;
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (u32 px = wave; px < n_pixels; px += nwaves) {
        float ss = 0.f;
        for (u32 j = lane; j < cvec; j += 64u) {
            float v[8];
            l2_load<BF16>(x, (size_t)px * cvec + j, v);
#pragma unroll
            for (int e = 0; e < CPV; ++e) ss += v[e] * v[e];
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        if (lane == 0 && inv_out) inv_out[px] = inv;
        for (u32 j = lane; j < cvec; j += 64u) {
            float v[8], o[8];
            l2_load<BF16>(x, (size_t)px * cvec + j, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = e < CPV ? (v[e] * inv) * gamma[j * CPV + e] : 0.f;
            l2_store<BF16>(y, (size_t)px * cvec + j, o);
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ inv_in, void* __restrict__ dx,
                                                         float* __restrict__ dgamma_partial, u32 n_pixels, u32 cvec) {
    constexpr int CPV = BF16 ? 8 : 4;
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    float dg[L2_MAXIT][8];
#pragma unroll
    for (int it = 0; it < L2_MAXIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[it][e] = 0.f;
    for (u32 px = wave; px < n_pixels; px += nwaves) {
        const float inv = inv_in[px];
        float dot = 0.f;                                     // sum_k dy_k gamma_k x_k
#pragma unroll
        for (int it = 0; it < L2_MAXIT; ++it) {
            const u32 j = lane + 64u * it;
            if (j < cvec) {
                float xv[8], gv[8];
                l2_load<BF16>(x, (size_t)px * cvec + j, xv);
                l2_load<BF16>(dy, (size_t)px * cvec + j, gv);
#pragma unroll
                for (int e = 0; e < CPV; ++e) {
                    dot += (gv[e] * gamma[j * CPV + e]) * xv[e];
                    dg[it][e] += (gv[e] * xv[e]) * inv;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        // inv = rsqrt(max(s, 1e-12)): the norm is a constant (1e6) where it was clamped
        const float k3 = inv < 0.999e6f ? (inv * inv) * inv * dot : 0.f;
#pragma unroll
        for (int it = 0; it < L2_MAXIT; ++it) {
            const u32 j = lane + 64u * it;
            if (j < cvec) {
                float xv[8], gv[8], o[8];
                l2_load<BF16>(x, (size_t)px * cvec + j, xv);
                l2_load<BF16>(dy, (size_t)px * cvec + j, gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = e < CPV ? (inv * gamma[j * CPV + e]) * gv[e] - xv[e] * k3 : 0.f;
                l2_store<BF16>(dx, (size_t)px * cvec + j, o);
            }
        }
    }
    // this wave's share of dgamma: row `wave` of dgamma_partial [n_waves][C]
#pragma unroll
    for (int it = 0; it < L2_MAXIT; ++it) {
        const u32 j = lane + 64u * it;
        if (j < cvec)
#pragma unroll
            for (int e = 0; e < CPV; ++e) dgamma_partial[(size_t)wave * (cvec * CPV) + j * CPV + e] = dg[it][e];
    }
}

}  // namespace ssdhip

// Waves (= rows of dgamma_partial) the backward launches for n_pixels pixels; 0: shape not supported.
extern "C" int ssdhip_l2_normalize_bwd_waves(long long n_pixels, int C, int is_bf16) {
    const int cpv = is_bf16 ? 8 : 4;
    if (n_pixels <= 0 || n_pixels > 0x7fffffffLL || C <= 0 || (C % cpv) || C / cpv > 64 * ssdhip::L2_MAXIT) return 0;
    long long blocks = (n_pixels + 4 * 16 - 1) / (4 * 16);            // >= 16 pixels per wave
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return (int)blocks * 4;
}

extern "C" int ssdhip_l2_normalize_fwd(const void* x, const float* gamma, void* y, float* inv_norm, long long n_pixels, int C, int is_bf16,
                                       void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int cpv = is_bf16 ? 8 : 4;
    if (!x || !gamma || !y || n_pixels <= 0 || n_pixels > 0x7fffffffLL || C <= 0 || (C % cpv)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    long long blocks = (n_pixels + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (is_bf16)
        hipLaunchKernelGGL(l2norm_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, x, gamma, y, inv_norm, (u32)n_pixels, (u32)(C / cpv));
    else
        hipLaunchKernelGGL(l2norm_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, gamma, y, inv_norm, (u32)n_pixels, (u32)(C / cpv));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_l2_normalize_bwd(const void* x, const void* dy, const float* gamma, const float* inv_norm, void* dx,
                                       float* dgamma_partial, int n_waves, long long n_pixels, int C, int is_bf16, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !dy || !gamma || !inv_norm || !dx || !dgamma_partial) return SSDHIP_E_BADARG;
    if (n_waves <= 0 || n_waves != ssdhip_l2_normalize_bwd_waves(n_pixels, C, is_bf16)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return SSDHIP_E_BADARG;
    const int cpv = is_bf16 ? 8 : 4;
    if (is_bf16)
        hipLaunchKernelGGL(l2norm_bwd_kernel<true>, dim3(n_waves / 4), dim3(256), 0, stream, x, dy, gamma, inv_norm, dx, dgamma_partial,
                           (u32)n_pixels, (u32)(C / cpv));
    else
        hipLaunchKernelGGL(l2norm_bwd_kernel<false>, dim3(n_waves / 4), dim3(256), 0, stream, x, dy, gamma, inv_norm, dx, dgamma_partial,
                           (u32)n_pixels, (u32)(C / cpv));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_maxpool_bwd_nhwc_bf16(const void* x, const void* gy, void* gx, int B, int H, int W, int C, int kernel, int stride,
                                            int pad, int Ho, int Wo, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !gy || !gx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || kernel < 1 || stride < 1 || pad < 0 || pad >= kernel || Ho <= 0 || Wo <= 0)
        return SSDHIP_E_BADARG;
    const long long total = (long long)B * H * W * (C / 8);
    if (total > 0x7fffffffLL) return SSDHIP_E_BADARG;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (kernel == 2 && stride == 2 && pad == 0 && Ho == (H + 1) / 2 && Wo == (W + 1) / 2) {
        const long long nwin = (long long)B * Ho * Wo;
        long long wb = (nwin * (C / 8) + 255) / 256;
        if (wb > 256 * 32) wb = 256 * 32;
        hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)wb), dim3(256), 0, stream, static_cast<const uint4*>(x),
                           static_cast<const uint4*>(gy), static_cast<uint4*>(gx), H, W, Ho, Wo, (u32)nwin, (u32)(C / 8));
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    if (kernel == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W && (C % 32) == 0 && (size_t)H * W * 16 <= 48 * 1024 && B <= 65535) {
        hipLaunchKernelGGL(maxpool3s1_bwd_img_kernel, dim3((unsigned)(C / 32), (unsigned)B), dim3(256), (size_t)H * W * 16, stream,
                           static_cast<const uint4*>(x), static_cast<const uint4*>(gy), static_cast<uint4*>(gx), H, W, (u32)(C / 8));
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    if (kernel == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W) {
        hipLaunchKernelGGL(maxpool3s1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const uint4*>(x),
                           static_cast<const uint4*>(gy), static_cast<uint4*>(gx), B, H, W, (u32)(C / 8));
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const uint4*>(x),
                       static_cast<const uint4*>(gy), static_cast<uint4*>(gx), B, H, W, (u32)(C / 8), kernel, stride, pad, Ho, Wo);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// conv1_1's backward (3 -> 64 channels, 3 x 3 'same', ReLU, no data gradient) in one pass: gy, y [B,H,W,64] bf16 (the gradient of the
// post-ReLU output and that output), x [B,H,W,3] bf16; wpart [n_blocks][64][27] float32 with k = (kh 3 + kw) 3 + ci, bpart
// [n_blocks][64] float32: per-workgroup partial sums the caller adds in order.  n_blocks = ssdhip_conv1_1_bwd_blocks(B, H, W).
extern "C" int ssdhip_conv1_1_bwd_blocks(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long long tiles = (long long)B * H * ((W + 63) / 64);
    return (int)(tiles < 1024 ? tiles : 1024);
}

extern "C" int ssdhip_conv1_1_bwd_nhwc_bf16(const void* gy, const void* y, const void* x, float* wpart, float* bpart, int B, int H, int W,
                                            int n_blocks, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gy || !y || !x || !wpart || !bpart || B <= 0 || H <= 0 || W <= 0) return SSDHIP_E_BADARG;
    if ((((uintptr_t)gy | (uintptr_t)y) & 15) || ((uintptr_t)x & 1)) return SSDHIP_E_BADARG;
    const long long tiles = (long long)B * H * ((W + 63) / 64);
    if (tiles > 0x7fffffffLL || (long long)B * H * W > 0x3fffffffLL || n_blocks != ssdhip_conv1_1_bwd_blocks(B, H, W)) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(conv1_1_bwd_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, static_cast<const uint4*>(gy),
                       static_cast<const uint4*>(y), static_cast<const unsigned short*>(x), wpart, bpart, H, W, (W + 63) / 64, (int)tiles);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// The data gradient of a 3 x 3 convolution with stride s and padding pad (dilation 1) through a 3 x 3 'same' convolution (round 6:
// conv6_2 / conv7_2 -- stride 2 behind ZeroPadding2D -- and the 'valid' conv8_2 / conv9_2, models/keras_ssd300.py:299-313):
// dX[r] = sum_k dY[(r + pad - k) / s] w[k] = sum_k' Z[r + k' - 1] w[2 - k'] with Z[s i + 1 - pad] = dY[i] and zeros elsewhere.  This
// entry builds Z: gy [B, Ho, Wo, C] bf16 -> z [B, H, W, C] bf16 (the convolution's INPUT size), offset = 1 - pad in {0, 1}; the caller
// then runs its forward kernel on z with the transposed, tap-flipped filters.  C % 8 == 0.
extern "C" int ssdhip_embed_strided_nhwc_bf16(const void* gy, void* z, int B, int Ho, int Wo, int C, int H, int W, int stride, int offset,
                                              void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!gy || !z || B <= 0 || Ho <= 0 || Wo <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || stride < 1 || offset < 0) return SSDHIP_E_BADARG;
    if (offset + (long long)stride * (Ho - 1) >= H + 1 || offset + (long long)stride * (Wo - 1) >= W + 1) return SSDHIP_E_BADARG;   // (a last tap row may fall off: pad > 0)
    if ((((uintptr_t)gy | (uintptr_t)z) & 15)) return SSDHIP_E_BADARG;
    const long long n = (long long)B * H * W * (C / 8);
    if (n > 0x7fffffffLL) return SSDHIP_E_BADARG;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_strided_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const uint4*>(gy), static_cast<uint4*>(z), (u32)n,
                       H, W, Ho, Wo, (u32)(C / 8), stride, offset);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
