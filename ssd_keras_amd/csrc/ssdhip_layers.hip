// ssdhip_layers.hip -- the memory-bound glue of the SSD graph between the convolutions, on gfx950 (MI355X).
//
// The reference expresses these as separate Keras layers / TF ops (file:line relative to the reference root):
//   * conv bias + ReLU (+ MaxPooling2D)   models/keras_ssd300.py:274-313 (Conv2D(activation='relu') + MaxPooling2D)
//   * L2Normalization                     keras_layers/keras_layer_L2Normalization.py:61-63
//   * input mean/scale/channel swap       models/keras_ssd300.py:247-272 (Lambda layers)
//   * Reshape + Concatenate + softmax + AnchorBoxes tiling + final Concatenate -> (B, N, C+12)
//                                         models/keras_ssd300.py:363-419, keras_layers/keras_layer_AnchorBoxes.py:245-255
// PyTorch-ROCm runs each of them as 2-7 elementwise kernels (bias add, clamp, pool, copies, cat); here each is ONE pass:
// HBM bytes = read the conv output once + write the result once.  All tensors are NHWC (torch channels_last), bf16
// activations, 16-byte (8 x bf16) vector accesses; arithmetic in float32 with one rounding to bf16, which makes
// bias+ReLU(+pool) bit-identical to the PyTorch sequence conv -> add(bf16) -> clamp_min(0) -> max_pool2d.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_heads.h"

namespace ssdhip {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(u32 h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ u32 f2bf(float f) {            // round to nearest even, NaN stays NaN (as c10::BFloat16)
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two bf16 lanes of one dword: out = act(x + b)
__device__ __forceinline__ u32 bias_act2(u32 x, u32 b, bool relu) {
    u32 lo = f2bf(bf2f(x & 0xffffu) + bf2f(b & 0xffffu));
    u32 hi = f2bf(bf2f(x >> 16) + bf2f(b >> 16));
    if (relu) {                                            // clamp_min(0) on the rounded value; NaN passes through
        if ((lo & 0x8000u) && (lo & 0x7fffu) <= 0x7f80u) lo = 0;
        if ((hi & 0x8000u) && (hi & 0x7fffu) <= 0x7f80u) hi = 0;
    }
    return lo | (hi << 16);
}
__device__ __forceinline__ uint4 bias_act8(uint4 x, uint4 b, bool relu) {
    return make_uint4(bias_act2(x.x, b.x, relu), bias_act2(x.y, b.y, relu), bias_act2(x.z, b.z, relu), bias_act2(x.w, b.w, relu));
}
// max of two bf16 as torch's max_pool2d takes it (NaN propagates)
__device__ __forceinline__ u32 bfmax1(u32 a, u32 b) {
    const float fa = bf2f(a), fb = bf2f(b);
    return (fa > fb || fa != fa) ? a : b;
}
__device__ __forceinline__ u32 bfmax2(u32 a, u32 b) { return bfmax1(a & 0xffffu, b & 0xffffu) | (bfmax1(a >> 16, b >> 16) << 16); }
__device__ __forceinline__ uint4 bfmax8(uint4 a, uint4 b) { return make_uint4(bfmax2(a.x, b.x), bfmax2(a.y, b.y), bfmax2(a.z, b.z), bfmax2(a.w, b.w)); }

// ---------------------------------------------------------------------------------------------------------------
// y = act(x + bias[c]); x, y: [n_pixels, C] bf16 (may alias), C % 8 == 0.  4 independent 16-byte loads per thread.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_kernel(const uint4* __restrict__ x, const uint4* __restrict__ bias,
                                                       uint4* __restrict__ y, u32 nvec, u32 cvec, int relu) {
    const u32 stride = gridDim.x * 256u;
    for (u32 i0 = blockIdx.x * 256u + threadIdx.x; i0 < nvec; i0 += 4u * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u32 i = i0 + u * stride; if (i < nvec) v[u] = x[i]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = i0 + u * stride;
            if (i < nvec) y[i] = bias_act8(v[u], bias ? bias[i % cvec] : make_uint4(0, 0, 0, 0), relu != 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// y[b,ho,wo,c] = max over the k x k window (stride s, padding p, clipped to the map) of act(x[b,hi,wi,c] + bias[c]).
// One thread per (output pixel, 8 channels); consecutive threads walk the channel vectors of a pixel, then wo.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_maxpool_kernel(const uint4* __restrict__ x, const uint4* __restrict__ bias,
                                                               uint4* __restrict__ y, int B, int H, int W, u32 cvec, int k, int s,
                                                               int p, int Ho, int Wo, int relu) {
    const u32 total = (u32)B * Ho * Wo * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 cg = i % cvec;
        u32 t = i / cvec;
        const int wo = t % Wo; t /= Wo;
        const int ho = t % Ho;
        const int b = t / Ho;
        const int h0 = max(ho * s - p, 0), h1 = min(ho * s - p + k, H);
        const int w0 = max(wo * s - p, 0), w1 = min(wo * s - p + k, W);
        const uint4 bv = bias ? bias[cg] : make_uint4(0, 0, 0, 0);
        uint4 best = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);      // -inf
        if (k == 3 && s == 1) {
            // pool5 (MaxPooling2D(3, 1, 'same'), models/keras_ssd300.py:296): the nine loads of the window go out together (clamped
            // coordinates, a window element outside the map is replaced by -inf afterwards) instead of one dependent load per trip
            // of a doubly nested loop with runtime bounds (35 -> ~12 us at batch 32)
            uint4 v[9];
            bool in[9];
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const int hi = ho - p + dh, wi = wo - p + dw;
                    in[dh * 3 + dw] = hi >= 0 && hi < H && wi >= 0 && wi < W;
                    const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
                    v[dh * 3 + dw] = x[((size_t)(b * H + hc) * W + wc) * cvec + cg];
                }
#pragma unroll
            for (int q = 0; q < 9; ++q)
                if (in[q]) best = bfmax8(bias_act8(v[q], bv, relu != 0), best);
            y[i] = best;
            continue;
        }
        for (int hi = h0; hi < h1; ++hi)
            for (int wi = w0; wi < w1; ++wi)
                best = bfmax8(bias_act8(x[((size_t)(b * H + hi) * W + wi) * cvec + cg], bv, relu != 0), best);
        y[i] = best;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pool5 (MaxPooling2D(3, 1, 'same'), models/keras_ssd300.py:296, keras_ssd512.py twin) when a 64-channel slab of the whole map fits
// in LDS: one workgroup per (image, 64 channels) reads its slab ONCE (H W x 128 bytes) and takes the nine-element maxima from LDS.
// The per-output kernel above reads every input nine times through L1 / L2: 106 MB of cache traffic for an 11.8 MB map, 27 us
// in the batch-32 step (r04n timeline) -- bound by what a CU can pull from L2.
// ---------------------------------------------------------------------------------------------------------------
// The maxima are taken on an ORDER-PRESERVING 16-bit integer image of the values (one v_pk_max_i16 per pair and window element, the
// float compare + select of bfmax8 is ~15 instructions per pair): a bf16 pattern v maps to v ^ ((v >> 15) & 0x7fff) as a signed
// 16-bit integer -- negative values in reversed magnitude order below the positive ones -- after NaNs have lost their sign, which
// puts every NaN above +inf: a NaN in the window wins, as in max_pool2d.  (-0 maps below +0: a window of zeros of both signs gives
// +0 where the compare-based maximum keeps the first one; equal as numbers.)  The map is applied once per element on the way into
// LDS and inverted once per output.
constexpr int POOL3_THREADS = 1024;
typedef short l_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short l_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 l_ordered2(u32 v) {
    const u32 mag = v & 0x7fff7fffu;
    // 1 in a half that holds a NaN (magnitude above 0x7f80), by a saturating 16-bit subtraction
    const l_u16x2 over = __builtin_elementwise_sub_sat(__builtin_bit_cast(l_u16x2, mag), __builtin_bit_cast(l_u16x2, 0x7f807f80u));
    const l_u16x2 one = __builtin_elementwise_min(over, __builtin_bit_cast(l_u16x2, 0x00010001u));
    v &= ~(__builtin_bit_cast(u32, one) << 15);            // NaNs become positive
    const u32 neg = __builtin_bit_cast(u32, __builtin_bit_cast(l_s16x2, v) >> 15);   // 0xffff in a negative half
    return v ^ (neg & 0x7fff7fffu);
}
__device__ __forceinline__ u32 l_unordered2(u32 t) {
    const u32 neg = __builtin_bit_cast(u32, __builtin_bit_cast(l_s16x2, t) >> 15);
    return t ^ (neg & 0x7fff7fffu);
}
__device__ __forceinline__ u32 l_pkmax_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(l_s16x2, a), __builtin_bit_cast(l_s16x2, b)));
}
__global__ __launch_bounds__(POOL3_THREADS) void pool3x3s1_slab_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int H, int W,
                                                                       u32 cvec, u32 slabs) {
    extern __shared__ uint4 pool3_tile[];                  // [H W][8 chunks of 8 channels], order-preserving integers
    const u32 b = blockIdx.x / slabs, slab = blockIdx.x - b * slabs;
    const u32 n = (u32)(H * W) * 8u;
    const size_t base = (size_t)b * (size_t)(H * W) * cvec + slab * 8u;
    for (u32 i = threadIdx.x; i < n; i += POOL3_THREADS) {
        const uint4 v = x[base + (size_t)(i >> 3) * cvec + (i & 7u)];
        pool3_tile[i] = make_uint4(l_ordered2(v.x), l_ordered2(v.y), l_ordered2(v.z), l_ordered2(v.w));
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < n; i += POOL3_THREADS) {
        const int px = (int)(i >> 3), c = (int)(i & 7u);
        const int h = px / W, w = px - h * W;
        uint4 best = make_uint4(0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u);      // below every value
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int hi = h + dh, wi = w + dw;
                const bool in = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W);
                const uint4 v = pool3_tile[(in ? hi * W + wi : px) * 8 + c];               // outside the map: the centre again (no effect)
                best = make_uint4(l_pkmax_i16(v.x, best.x), l_pkmax_i16(v.y, best.y), l_pkmax_i16(v.z, best.z), l_pkmax_i16(v.w, best.w));
            }
        y[base + (size_t)px * cvec + (u32)c] = make_uint4(l_unordered2(best.x), l_unordered2(best.y), l_unordered2(best.z), l_unordered2(best.w));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// L2Normalization: y = x * rsqrt(max(sum_c x^2, 1e-12)) * gamma[c]; one wave per pixel, float32 math.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const uint4* __restrict__ x, const float* __restrict__ gamma,
                                                     uint4* __restrict__ y, u32 n_pixels, u32 cvec) {
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256u) >> 6;
    for (u32 px = wave; px < n_pixels; px += nwaves) {
        const uint4* row = x + (size_t)px * cvec;
        float ss = 0.f;
        for (u32 j = lane; j < cvec; j += 64u) {
            const uint4 v = row[j];
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float a = bf2f(w[q] & 0xffffu), c = bf2f(w[q] >> 16); ss += a * a; ss += c * c; }
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        for (u32 j = lane; j < cvec; j += 64u) {
            const uint4 v = row[j];
            const u32 w[4] = {v.x, v.y, v.z, v.w};
            u32 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g0 = gamma[j * 8 + q * 2], g1 = gamma[j * 8 + q * 2 + 1];
                o[q] = f2bf((bf2f(w[q] & 0xffffu) * inv) * g0) | (f2bf((bf2f(w[q] >> 16) * inv) * g1) << 16);
            }
            y[(size_t)px * cvec + j] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pool4 + conv4_3_norm in ONE pass (round 6): MaxPooling2D(2, 2, 'same') and L2Normalization both read the conv4_3 map
// (models/keras_ssd300.py:287 and :316); as two kernels the 47 MB map crossed HBM twice (15.9 + 21 us at batch 32).  One wave per
// 2 x 2 window, lane = 8 of the 512 channels: the window's four pixels are read once, their maximum goes to the pooled map, each
// pixel's normalised row (the arithmetic of l2norm_kernel, operation for operation) to the normalised map.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool2_l2norm_kernel(const uint4* __restrict__ x, const float* __restrict__ gamma, uint4* __restrict__ y_pool,
                                                           uint4* __restrict__ y_norm, int B, int H, int W, int Ho, int Wo) {
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const u32 nwaves = (gridDim.x * 256u) >> 6;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gamma[lane * 8 + e];
    const u32 total = (u32)B * Ho * Wo;
    for (u32 o = wave; o < total; o += nwaves) {
        const int wo = o % Wo;
        const int ho = (o / Wo) % Ho;
        const int b = o / (Wo * Ho);
        uint4 v[4];
        bool in[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 2 * ho + (q >> 1), w = 2 * wo + (q & 1);
            in[q] = h < H && w < W;
            v[q] = in[q] ? x[((size_t)(b * H + h) * W + w) * 64 + lane] : make_uint4(0, 0, 0, 0);
        }
        uint4 best = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);      // -inf
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (in[q]) best = bfmax8(v[q], best);
        y_pool[(size_t)o * 64 + lane] = best;
        float ss[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 w4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float a = bf2f(w4[t] & 0xffffu), c = bf2f(w4[t] >> 16); acc += a * a; acc += c * c; }
            ss[q] = acc;
        }
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ss[q] += __shfl_xor(ss[q], off);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!in[q]) continue;
            const float inv = rsqrtf(fmaxf(ss[q], 1e-12f));
            const u32 w4[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
            u32 r[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                r[t] = f2bf((bf2f(w4[t] & 0xffffu) * inv) * g[2 * t]) | (f2bf((bf2f(w4[t] >> 16) * inv) * g[2 * t + 1]) << 16);
            const int h = 2 * ho + (q >> 1), w = 2 * wo + (q & 1);
            y_norm[((size_t)(b * H + h) * W + w) * 64 + lane] = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Input pipeline: out[b,h,w,c'] = bf16((img[b,h,w,swap[c']] - mean[swap[c']]) * scale[swap[c']]), img float32 NHWC.
// ---------------------------------------------------------------------------------------------------------------
struct PreParams { float mean[4]; float scale[4]; int swap[4]; int has_scale; };

__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, u32 n_pixels,
                                                         int Cin, PreParams pp) {
    for (u32 px = blockIdx.x * 256u + threadIdx.x; px < n_pixels; px += gridDim.x * 256u) {
        float v[4];
        for (int c = 0; c < Cin; ++c) v[c] = img[(size_t)px * Cin + c];
        for (int c = 0; c < Cin; ++c) {
            const int sc = pp.swap[c];
            float t = v[sc] - pp.mean[sc];
            if (pp.has_scale) t = t / pp.scale[sc];
            out[(size_t)px * Cin + c] = (bf16_t)f2bf(t);
        }
    }
}

// Three channels, four pixels per thread (round 5): three 16-byte loads and three 8-byte stores instead of twelve 4-byte loads and twelve
// 2-byte stores with strides of 12 and 6 bytes across the lanes.  Same arithmetic per element.
__global__ __launch_bounds__(256) void preprocess3_kernel(const float4* __restrict__ img, uint2* __restrict__ out, u32 n_quads, PreParams pp) {
    for (u32 q = blockIdx.x * 256u + threadIdx.x; q < n_quads; q += gridDim.x * 256u) {
        const float4 a = img[(size_t)q * 3], b = img[(size_t)q * 3 + 1], c = img[(size_t)q * 3 + 2];
        const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        u32 o[12];
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int sc = pp.swap[ch];
                float t = (sc == 0 ? v[px * 3] : sc == 1 ? v[px * 3 + 1] : v[px * 3 + 2]) - pp.mean[sc];
                if (pp.has_scale) t = t / pp.scale[sc];
                o[px * 3 + ch] = f2bf(t);
            }
        out[(size_t)q * 3] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        out[(size_t)q * 3 + 1] = make_uint2(o[4] | (o[5] << 16), o[6] | (o[7] << 16));
        out[(size_t)q * 3 + 2] = make_uint2(o[8] | (o[9] << 16), o[10] | (o[11] << 16));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Prediction assembly.  For predictor layer l with n_l anchors (= h*w*n_boxes, anchor a <-> (y, x, box) row-major):
//   conf_l [B, n_l, C] bf16 logits (the NHWC conv output read as Keras' Reshape((-1, C)) reads it), loc_l [B, n_l, 4]
//   y_pred[b, off_l + a, :] = [softmax(conf + bias) (C) | loc + bias (4) | anchor (4) | variances (4)]  float32
// grid (tiles over all layers, B); a tile = TA anchors of one layer; rows are built in LDS and stored coalesced.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_kernel(HeadParams hp, const float* __restrict__ anchors_var, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, b = blockIdx.y, L = hp.C + 12;
    int l, a0, na;
    head_tile_of(hp, (int)blockIdx.x, l, a0, na);
    float* rows = reinterpret_cast<float*>(smem_raw);                                   // [TA][L]
    hbf16_t* cl = reinterpret_cast<hbf16_t*>(smem_raw + (size_t)hp.TA * L * sizeof(float));
    head_build_rows(hp, anchors_var, l, b, a0, na, rows, cl, tid, 256);
    float* dst = y + ((size_t)b * hp.N + hp.anchor_off[l] + a0) * L;
    for (int i = tid; i < na * L; i += 256) dst[i] = rows[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the prediction assembly for the TRAINING step (round 5): d loss / d y_pred [B, N, C+12] float32 -> the gradient of every
// source map's PACKED head output [B, h, w, Cp] bf16 ([conf n_boxes C | loc n_boxes 4 | zero padding]: what _PackedHeadFn returns and
// its backward reads).  Per anchor: softmax backward on the class columns (dlogit_c = p_c (g_c - sum_k g_k p_k), p from the saved
// y_pred), the four offset columns pass through, the anchor / variance columns have no producer; one rounding to bf16, as the
// framework's cast of the float32 gradient.  Replaces the backward of Reshape + Concatenate + softmax + Concatenate
// (models/keras_ssd300.py:363-419): ~15 framework launches (slices into zero-filled maps, concatenation splits, softmax backward).
// grid (tiles, B); a tile = whole pixels of one layer (at most 128 anchors), both row tiles staged in LDS, the packed rows built in
// LDS and stored as one contiguous run.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HG_TILE = 128;
struct HeadGradParams {
    bf16_t* out[MAX_PRED_LAYERS];
    int n_anchors[MAX_PRED_LAYERS], n_boxes[MAX_PRED_LAYERS], stride[MAX_PRED_LAYERS], tile_start[MAX_PRED_LAYERS + 1],
        anchor_off[MAX_PRED_LAYERS], tile_anchors[MAX_PRED_LAYERS];
    int n_layers, N, C;
};

__global__ __launch_bounds__(HG_TILE) void head_grad_kernel(HeadGradParams hp, const float* __restrict__ y_pred, const float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, b = blockIdx.y, C = hp.C, L = C + 12;
    int l = 0;
    while (l + 1 < hp.n_layers && (int)blockIdx.x >= hp.tile_start[l + 1]) ++l;
    const int nb = hp.n_boxes[l], ta = hp.tile_anchors[l];
    const int a0 = ((int)blockIdx.x - hp.tile_start[l]) * ta, na = min(ta, hp.n_anchors[l] - a0);
    const int pix0 = a0 / nb, npx = na / nb;                 // tiles hold whole pixels
    const int rowb = hp.stride[l] * 2;
    float* lds = reinterpret_cast<float*>(smem_raw);
    const size_t half = ((size_t)HG_TILE * L + 4 + 3) / 4 * 4;
    unsigned char* stage = smem_raw + 2 * half * sizeof(float);
    const size_t off = ((size_t)b * hp.N + hp.anchor_off[l] + a0) * (size_t)L;
    const float* yp = tile_copy_f32(lds, y_pred + off, na * L, tid, HG_TILE);
    const float* gp = tile_copy_f32(lds + half, grad + off, na * L, tid, HG_TILE);
    for (int i = tid; i < npx * rowb / 4; i += HG_TILE) reinterpret_cast<u32*>(stage)[i] = 0u;
    __syncthreads();
    if (tid < na) {
        const int pix = tid / nb, box = tid - pix * nb;
        const float* pr = yp + (size_t)tid * L;
        const float* gr = gp + (size_t)tid * L;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += gr[c] * pr[c];
        bf16_t* row = reinterpret_cast<bf16_t*>(stage + (size_t)pix * rowb);
        for (int c = 0; c < C; ++c) row[box * C + c] = (bf16_t)f2bf(pr[c] * (gr[c] - dot));
        for (int k = 0; k < 4; ++k) row[nb * C + box * 4 + k] = (bf16_t)f2bf(gr[C + k]);
    }
    __syncthreads();
    const size_t npix_l = (size_t)(hp.n_anchors[l] / nb);
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(hp.out[l]) + ((size_t)b * npix_l + pix0) * rowb);
    const uint4* src = reinterpret_cast<const uint4*>(stage);
    for (int i = tid; i < npx * rowb / 16; i += HG_TILE) dst[i] = src[i];
}

static inline int grid_for(size_t work_items, int per_block) {
    size_t g = (work_items + per_block - 1) / per_block;
    const size_t cap = 256 * 16;                 // 16 workgroups per CU: enough loads in flight, few tail blocks
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" int ssdhip_bias_act_nhwc_bf16(const void* x, const void* bias, void* y, long long n_pixels, int C, int relu,
                                         void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || n_pixels <= 0 || C <= 0 || (C & 7)) return SSDHIP_E_BADARG;
    const long long nvec = n_pixels * (C / 8);
    if (nvec > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid_for((size_t)nvec, 1024)), dim3(256), 0, stream, static_cast<const uint4*>(x),
                       static_cast<const uint4*>(bias), static_cast<uint4*>(y), (u32)nvec, (u32)(C / 8), relu);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_bias_act_maxpool_nhwc_bf16(const void* x, const void* bias, void* y, int B, int H, int W, int C,
                                                 int kernel, int stride, int pad, int Ho, int Wo, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || kernel <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0)
        return SSDHIP_E_BADARG;
    if (pad >= kernel || (Ho - 1) * stride - pad >= H || (Wo - 1) * stride - pad >= W) return SSDHIP_E_BADARG;   // empty window
    const long long total = (long long)B * Ho * Wo * (C / 8);
    if (total > 0x7fffffffLL || (long long)B * H * W * (C / 8) > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) return SSDHIP_E_BADARG;
    if (kernel == 3 && stride == 1 && pad == 1 && Ho == H && Wo == W && !bias && !relu && C % 64 == 0 && (long long)H * W * 128 <= 160 * 1024 && x != y) {
        const size_t lds = (size_t)H * W * 128;
        // the attribute is per DEVICE: one process may drive several GPUs (state per device id: 0 unknown, 1 set, 2 refused)
        static signed char big_lds_state[64] = {0};
        int devid = 0;
        bool big_lds = false;
        if (lds > 64 * 1024 && hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < 64) {
            if (big_lds_state[devid] == 0) {
                big_lds_state[devid] = hipFuncSetAttribute(reinterpret_cast<const void*>(pool3x3s1_slab_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : 2;
                if (big_lds_state[devid] == 2) (void)hipGetLastError();   // a refusal must not surface as the next launch's error
            }
            big_lds = big_lds_state[devid] == 1;
        }
        if (lds <= 64 * 1024 || big_lds) {
            hipLaunchKernelGGL(pool3x3s1_slab_kernel, dim3((unsigned)(B * (C / 64))), dim3(POOL3_THREADS), lds, stream, static_cast<const uint4*>(x),
                               static_cast<uint4*>(y), H, W, (u32)(C / 8), (u32)(C / 64));
            if (hipGetLastError() == hipSuccess) return SSDHIP_OK;
            // a refused launch falls through to the generic pooling kernel below
        }
    }
    hipLaunchKernelGGL(bias_act_maxpool_kernel, dim3(grid_for((size_t)total, 256)), dim3(256), 0, stream,
                       static_cast<const uint4*>(x), static_cast<const uint4*>(bias), static_cast<uint4*>(y), B, H, W, (u32)(C / 8),
                       kernel, stride, pad, Ho, Wo, relu);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// MaxPooling2D(pool_size=(2, 2), strides=(2, 2), padding='same') + L2Normalization of the SAME 512-channel map in one pass (pool4 and
// conv4_3_norm, models/keras_ssd300.py:287, 316): x [B, H, W, 512] bf16 -> y_pool [B, ceil(H / 2), ceil(W / 2), 512], y_norm [B, H, W, 512];
// bit-identical to ssdhip_bias_act_maxpool_nhwc_bf16 (no bias, no activation) and ssdhip_l2_normalize_nhwc_bf16.
extern "C" int ssdhip_pool2_l2_normalize_nhwc_bf16(const void* x, const float* gamma, void* y_pool, void* y_norm, int B, int H, int W, int C,
                                                   void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !gamma || !y_pool || !y_norm || B <= 0 || H <= 0 || W <= 0 || C != 512) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y_pool | (uintptr_t)y_norm) & 15) return SSDHIP_E_BADARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if ((long long)B * H * W * 64 > 0x7fffffffLL) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(pool2_l2norm_kernel, dim3(grid_for((size_t)B * Ho * Wo, 4)), dim3(256), 0, stream, static_cast<const uint4*>(x), gamma,
                       static_cast<uint4*>(y_pool), static_cast<uint4*>(y_norm), B, H, W, Ho, Wo);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_l2_normalize_nhwc_bf16(const void* x, const float* gamma, void* y, long long n_pixels, int C, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !gamma || !y || n_pixels <= 0 || C <= 0 || (C & 7) || n_pixels > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(l2norm_kernel, dim3(grid_for((size_t)n_pixels, 4)), dim3(256), 0, stream, static_cast<const uint4*>(x), gamma,
                       static_cast<uint4*>(y), (u32)n_pixels, (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

namespace ssdhip {

// ---------------------------------------------------------------------------------------------------------------
// Reference-precision path (models/precise.py), the layers that are not 64-channel GEMMs.
//   x3_split_kernel   float32 [n, C] -> float16 [n, 2 C] = [hi | lo], hi = fl16(v), lo = fl16(v - hi): ONE pass instead of the six
//                     elementwise / concatenation kernels of the PyTorch formulation.
//   x3_merge_kernel   the inverse: float32 v = hi + lo (exact).
//   conv1_1_x3_kernel conv1_1 (models/keras_ssd300.py:274: 3 -> 64 channels, 3x3 'same', ReLU) in float32 FMA arithmetic
//                     (fmaf: ONE rounding per term, taps outer, channels inner) with the split output written directly.
//                     K = 27 is no GEMM: 5 GFLOP of vector work against 0.7 GB of output; MIOpen's float32 path for a 3-channel
//                     NHWC input is its naive kernel (5.2 ms at batch 32, r03w).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 l_split2(float a, float b, u32& lo_out) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    lo_out = (u32)__builtin_bit_cast(unsigned short, la) | ((u32)__builtin_bit_cast(unsigned short, lb) << 16);
    return (u32)__builtin_bit_cast(unsigned short, ha) | ((u32)__builtin_bit_cast(unsigned short, hb) << 16);
}

// one thread per 8 channels of a pixel; cvec = C / 8
__global__ __launch_bounds__(256) void x3_split_kernel(const float4* __restrict__ x, uint4* __restrict__ y, u32 n_pixels, u32 cvec) {
    const u32 total = n_pixels * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 px = i / cvec, cg = i - px * cvec;
        const float4 a = x[(size_t)i * 2], b = x[(size_t)i * 2 + 1];
        u32 l0, l1, l2, l3;
        const u32 h0 = l_split2(a.x, a.y, l0), h1 = l_split2(a.z, a.w, l1), h2 = l_split2(b.x, b.y, l2), h3 = l_split2(b.z, b.w, l3);
        y[(size_t)px * (2 * cvec) + cg] = make_uint4(h0, h1, h2, h3);
        y[(size_t)px * (2 * cvec) + cvec + cg] = make_uint4(l0, l1, l2, l3);
    }
}

__device__ __forceinline__ float l_h2f(u32 h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }

__global__ __launch_bounds__(256) void x3_merge_kernel(const uint4* __restrict__ x, float4* __restrict__ y, u32 n_pixels, u32 cvec) {
    const u32 total = n_pixels * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 px = i / cvec, cg = i - px * cvec;
        const uint4 h = x[(size_t)px * (2 * cvec) + cg], l = x[(size_t)px * (2 * cvec) + cvec + cg];
        const u32 hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
        float o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[2 * q] = l_h2f(hw[q] & 0xffffu) + l_h2f(lw[q] & 0xffffu);
            o[2 * q + 1] = l_h2f(hw[q] >> 16) + l_h2f(lw[q] >> 16);
        }
        y[(size_t)i * 2] = make_float4(o[0], o[1], o[2], o[3]);
        y[(size_t)i * 2 + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// x [B, H, W, 3] float32, w [64, 3, 3, 3] float32 (co, kh, kw, ci), bias [64] or null; y [B, H, W, 128] float16 = [hi | lo] of
// relu(conv + bias).  256 threads = 32 pixel quads x 8 channel groups of 8: a thread computes FOUR consecutive pixels of a tile of 128
// consecutive (flattened) pixels; every filter value read from LDS feeds four FMAs.  The tile's inputs -- three flattened runs of 130
// pixels, one per filter row -- are staged in LDS by coalesced loads: in the first two versions every thread fetched its 108 input
// values from global memory itself (eight lanes per address), and the kernel ran at 1.07 TB/s of output (668 / 690 us at batch 32,
// r03x / r03ze) instead of the ~3 TB/s a write-only kernel reaches.  fmaf: one rounding per term, taps outer, channels inner.
constexpr int C11_PX = 4;
constexpr int C11_TILE = 32 * C11_PX;                    // pixels per tile
constexpr int C11_RUN = (C11_TILE + 2) * 3;              // floats of one staged run: pixels p0 - 1 .. p0 + TILE of one filter row
// PRE (round 6): the graph's input Lambdas (identity / mean subtraction / stddev division / channel swap, models/keras_ssd300.py:254-264)
// applied while the tile's input runs are staged -- out[c] = (x[swap[c]] - mean[swap[c]]) / div[swap[c]], the framework's float32
// operations in its order, zeros stay zeros outside the image -- instead of three framework passes over the batch (52 us of the step).
struct C11Pre {
    float mean[3], div[3];
    int swap[3];
    int on, has_div;
};
template <bool PRE>
__global__ __launch_bounds__(256, 4) void conv1_1_x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         uint4* __restrict__ y, int H, int W, u32 n_pixels, int relu, C11Pre pre) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 64];
    __shared__ float bl[64];
    __shared__ float sx[3][C11_RUN + 2];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) { const int co = i / 27, k = i - co * 27; wl[k * 64 + co] = w[i]; }
    if (threadIdx.x < 64) bl[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const u32 n_tiles = (n_pixels + C11_TILE - 1) / C11_TILE;
    const long long n_floats = (long long)n_pixels * 3;
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u32 p0 = tile * C11_TILE;
        __syncthreads();                                 // the previous tile's readers are done (first trip: the filters are in place)
        for (int i = threadIdx.x; i < 3 * C11_RUN; i += 256) {
            const int r = i / C11_RUN, k = i - r * C11_RUN;
            const long long g = ((long long)p0 - 1 + (long long)(r - 1) * W) * 3 + k;
            if constexpr (PRE) {
                const int c = k % 3, sc = c == 0 ? pre.swap[0] : (c == 1 ? pre.swap[1] : pre.swap[2]);
                float v = 0.f;
                if (g >= 0 && g < n_floats) {
                    v = x[g - c + sc] - (sc == 0 ? pre.mean[0] : (sc == 1 ? pre.mean[1] : pre.mean[2]));
                    if (pre.has_div) v = v / (sc == 0 ? pre.div[0] : (sc == 1 ? pre.div[1] : pre.div[2]));
                }
                sx[r][k] = v;
            } else {
                sx[r][k] = (g >= 0 && g < n_floats) ? x[g] : 0.f;
            }
        }
        __syncthreads();
        const u32 px0 = p0 + pl * C11_PX;
        // the filter reads below are tile-invariant: left visible, LLVM hoists all 216 values out of the tile loop (283 VGPRs, one
        // wave per SIMD -- the first two versions -- or 868 bytes of scratch under a register cap); an opaque zero keeps them in LDS
        int wz = 0;
        asm volatile("" : "+v"(wz));
        const float* wt = wl + cg * 8 + wz;
        int wq[C11_PX], hq[C11_PX];
#pragma unroll
        for (int u = 0; u < C11_PX; ++u) {
            const u32 px = min(px0 + u, n_pixels - 1);
            wq[u] = (int)(px % (u32)W);
            hq[u] = (int)((px / (u32)W) % (u32)H);
        }
        // packed float32 FMAs (v_pk_fma_f32: two channels per instruction): the kernel's VALU work is 864 FMAs per thread and tile
        typedef float c11_f2 __attribute__((ext_vector_type(2)));
        c11_f2 acc[C11_PX][4];
#pragma unroll
        for (int u = 0; u < C11_PX; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][e] = (c11_f2){0.f, 0.f};
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {                  // not unrolled: one filter row's 72 values in flight at a time
            // the quad's six input pixels of this filter row (local pixels pl * 4 - 1 .. pl * 4 + 4 -> run offsets pl * 12 .. + 17)
            float in6[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) in6[k] = sx[kh][pl * (C11_PX * 3) + k];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v[C11_PX][3];
#pragma unroll
                for (int u = 0; u < C11_PX; ++u) {
                    const int hh = hq[u] + kh - 1, ww = wq[u] + kw - 1;
                    const bool in = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) v[u][ci] = in ? in6[(u + kw) * 3 + ci] : 0.f;
                }
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float4 w0 = *reinterpret_cast<const float4*>(wt + ((kh * 3 + kw) * 3 + ci) * 64);
                    const float4 w1 = *reinterpret_cast<const float4*>(wt + ((kh * 3 + kw) * 3 + ci) * 64 + 4);
                    const c11_f2 wr[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
#pragma unroll
                    for (int u = 0; u < C11_PX; ++u) {
                        const c11_f2 vv = {v[u][ci], v[u][ci]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[u][e] = __builtin_elementwise_fma(vv, wr[e], acc[u][e]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < C11_PX; ++u) {
            if (px0 + u >= n_pixels) break;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = acc[u][e >> 1][e & 1] + bl[cg * 8 + e];
                o[e] = relu ? (t > 0.f ? t : (t != t ? t : 0.f)) : t;
            }
            u32 l0, l1, l2, l3;
            const u32 h0 = l_split2(o[0], o[1], l0), h1 = l_split2(o[2], o[3], l1), h2 = l_split2(o[4], o[5], l2), h3 = l_split2(o[6], o[7], l3);
            y[(size_t)(px0 + u) * 16 + cg] = make_uint4(h0, h1, h2, h3);
            y[(size_t)(px0 + u) * 16 + 8 + cg] = make_uint4(l0, l1, l2, l3);
        }
    }
}

// L2Normalization (keras_layers/keras_layer_L2Normalization.py:62-70) of a PAIR map (round 6): x, y [n_pixels][2 C] float16 = [hi | lo];
// the true activation is (hi + lo) * scale.  One wave per pixel, the arithmetic of l2norm_fwd_kernel on the float32 sums hi + lo (what
// x3_merge would hand it): ss = sum v^2 in lane-strided order + butterfly, inv = rsqrt(max(scale^2 ss, 1e-12)), out = ((v scale) inv)
// gamma, re-split.  Replaces merge -> float32 normalisation -> split (three passes over the conv4_3 map on the reference-precision
// step's critical path: 32 + 46 + 34 us).
__global__ __launch_bounds__(256) void x3_l2norm_kernel(const uint4* __restrict__ x, const float* __restrict__ gamma, uint4* __restrict__ y,
                                                        u32 n_pixels, u32 cvec, float scale) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (u32 px = wave; px < n_pixels; px += nwaves) {
        float ss = 0.f;
        for (u32 j = lane; j < cvec; j += 64u) {
            const uint4 h = x[(size_t)px * (2 * cvec) + j], l = x[(size_t)px * (2 * cvec) + cvec + j];
            const u32 hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = (l_h2f(hw[q] & 0xffffu) + l_h2f(lw[q] & 0xffffu)) * scale, b = (l_h2f(hw[q] >> 16) + l_h2f(lw[q] >> 16)) * scale;
                ss += a * a;
                ss += b * b;
            }
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float inv = rsqrtf(fmaxf(ss, 1e-12f));
        for (u32 j = lane; j < cvec; j += 64u) {
            const uint4 h = x[(size_t)px * (2 * cvec) + j], l = x[(size_t)px * (2 * cvec) + cvec + j];
            const u32 hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
            float o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = (l_h2f(hw[q] & 0xffffu) + l_h2f(lw[q] & 0xffffu)) * scale, b = (l_h2f(hw[q] >> 16) + l_h2f(lw[q] >> 16)) * scale;
                o[2 * q] = (a * inv) * gamma[j * 8 + 2 * q];
                o[2 * q + 1] = (b * inv) * gamma[j * 8 + 2 * q + 1];
            }
            u32 l0, l1, l2, l3;
            const u32 h0 = l_split2(o[0], o[1], l0), h1 = l_split2(o[2], o[3], l1), h2 = l_split2(o[4], o[5], l2), h3 = l_split2(o[6], o[7], l3);
            y[(size_t)px * (2 * cvec) + j] = make_uint4(h0, h1, h2, h3);
            y[(size_t)px * (2 * cvec) + cvec + j] = make_uint4(l0, l1, l2, l3);
        }
    }
}

}  // namespace ssdhip

extern "C" int ssdhip_x3_split_nhwc(const float* x, void* y, long long n_pixels, int C, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || n_pixels <= 0 || C <= 0 || (C & 7) || n_pixels * (C / 8) > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(ssdhip::x3_split_kernel, dim3(grid_for((size_t)n_pixels * (C / 8), 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x), static_cast<uint4*>(y), (u32)n_pixels, (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

namespace ssdhip {
// MaxPooling2D on (hi, lo) float16 pair maps (round 6): the pair of the window's largest VALUE, without leaving the pair representation
// (before: merge to float32, the framework's pooling, split again -- three passes and 4-byte values through HBM).  hi + lo is exact in
// float32 (22 significant bits), so the comparison is the float32 pooling's; NaNs win as in the framework's kernel.  One thread per
// (output pixel, 8 channels); x [B, H, W, 2 C] = [hi | lo], windows k x k, stride s, padding p, clipped to the map.
__global__ __launch_bounds__(256) void x3_maxpool_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int B, int H, int W, u32 cvec,
                                                         int k, int s, int p, int Ho, int Wo) {
    const u32 total = (u32)B * Ho * Wo * cvec;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const u32 cg = i % cvec;
        u32 t = i / cvec;
        const int wo = t % Wo; t /= Wo;
        const int ho = t % Ho;
        const int b = t / Ho;
        const int h0 = max(ho * s - p, 0), h1 = min(ho * s - p + k, H);
        const int w0 = max(wo * s - p, 0), w1 = min(wo * s - p + k, W);
        float best[8];
        u32 bh[8], bl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -__builtin_inff(); bh[e] = 0xfc00u; bl[e] = 0u; }   // (-inf, 0): an empty window's pair
        for (int hi = h0; hi < h1; ++hi)
            for (int wi = w0; wi < w1; ++wi) {
                const size_t row = ((size_t)(b * H + hi) * W + wi) * (2 * cvec);
                const uint4 vh = x[row + cg], vl = x[row + cvec + cg];
                const u32 wh[4] = {vh.x, vh.y, vh.z, vh.w}, wl[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const u32 h16 = half ? (wh[q] >> 16) : (wh[q] & 0xffffu), l16 = half ? (wl[q] >> 16) : (wl[q] & 0xffffu);
                        const float v = l_h2f(h16) + l_h2f(l16);
                        const int e = 2 * q + half;
                        if (v > best[e] || v != v) { best[e] = v; bh[e] = h16; bl[e] = l16; }
                    }
            }
        const size_t orow = ((size_t)(b * Ho + ho) * Wo + wo) * (2 * cvec);
        y[orow + cg] = make_uint4(bh[0] | (bh[1] << 16), bh[2] | (bh[3] << 16), bh[4] | (bh[5] << 16), bh[6] | (bh[7] << 16));
        y[orow + cvec + cg] = make_uint4(bl[0] | (bl[1] << 16), bl[2] | (bl[3] << 16), bl[4] | (bl[5] << 16), bl[6] | (bl[7] << 16));
    }
}
}  // namespace ssdhip

// MaxPooling2D(k, strides = s, zero... 'same' / ceil-mode windows clipped to the map) of a pair map: x [B, H, W, 2 C] float16 -> y
// [B, Ho, Wo, 2 C]; Ho, Wo given by the caller (models/keras_ssd300.py:287 pool4 = (2, 2) 'same', :296 pool5 = (3, 3) / 1 'same').
extern "C" int ssdhip_x3_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int kernel, int stride, int pad, int Ho, int Wo,
                                      void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || kernel < 1 || stride < 1 || pad < 0 || pad >= kernel || Ho <= 0 || Wo <= 0)
        return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    if ((long long)B * Ho * Wo * (C / 8) > 0x7fffffffLL || (long long)B * H * W * (C / 4) > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if ((Ho - 1) * stride - pad >= H || (Wo - 1) * stride - pad >= W) return SSDHIP_E_BADARG;          // every window meets the map
    hipLaunchKernelGGL(ssdhip::x3_maxpool_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 8), 256)), dim3(256), 0, stream,
                       static_cast<const uint4*>(x), static_cast<uint4*>(y), B, H, W, (u32)(C / 8), kernel, stride, pad, Ho, Wo);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_x3_merge_nhwc(const void* x, float* y, long long n_pixels, int C, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !y || n_pixels <= 0 || C <= 0 || (C & 7) || n_pixels * (C / 8) > 0x7fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(ssdhip::x3_merge_kernel, dim3(grid_for((size_t)n_pixels * (C / 8), 256)), dim3(256), 0, stream,
                       static_cast<const uint4*>(x), reinterpret_cast<float4*>(y), (u32)n_pixels, (u32)(C / 8));
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_conv1_1_x3_nhwc(const float* x, const float* weight, const float* bias, void* y, int B, int H, int W, int relu,
                                      void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W > 0x3fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    const long long n = (long long)B * H * W;
    long long blocks = (n + ssdhip::C11_TILE - 1) / ssdhip::C11_TILE;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ssdhip::conv1_1_x3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x, weight, bias, static_cast<uint4*>(y), H, W,
                       (u32)n, relu ? 1 : 0, ssdhip::C11Pre{});
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// ... with the graph's input pipeline in front (models/keras_ssd300.py:254-264): images [B, H, W, 3] float32 as they come from the
// generator; mean_h / divide_h / swap_h: HOST arrays of three entries or NULL (no subtraction / no division / identity order).  conv1_1
// sees (images[swap[c]] - mean[swap[c]]) / divide[swap[c]] -- bit for bit the framework's float32 expression.
extern "C" int ssdhip_conv1_1_x3_pre_nhwc(const float* images, const float* weight, const float* bias, void* y, int B, int H, int W, int relu,
                                          const float* mean_h, const float* divide_h, const int* swap_h, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!images || !weight || !y || B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W > 0x3fffffffLL) return SSDHIP_E_BADARG;
    if (((uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    ssdhip::C11Pre pre;
    for (int c = 0; c < 3; ++c) {
        pre.mean[c] = mean_h ? mean_h[c] : 0.f;
        pre.div[c] = divide_h ? divide_h[c] : 1.f;
        pre.swap[c] = swap_h ? swap_h[c] : c;
        if (pre.swap[c] < 0 || pre.swap[c] > 2) return SSDHIP_E_BADARG;
    }
    pre.on = 1;
    pre.has_div = divide_h ? 1 : 0;
    const long long n = (long long)B * H * W;
    long long blocks = (n + ssdhip::C11_TILE - 1) / ssdhip::C11_TILE;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ssdhip::conv1_1_x3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, images, weight, bias, static_cast<uint4*>(y), H, W,
                       (u32)n, relu ? 1 : 0, pre);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_preprocess_nhwc_f32_to_bf16(const float* images, void* out, long long n_pixels, int channels,
                                                  const float* mean_h, const float* divide_h, const int* swap_h, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!images || !out || n_pixels <= 0 || n_pixels > 0x7fffffffLL || channels < 1 || channels > 4) return SSDHIP_E_BADARG;
    PreParams pp;
    for (int c = 0; c < 4; ++c) {
        pp.mean[c] = (mean_h && c < channels) ? mean_h[c] : 0.f;
        pp.scale[c] = (divide_h && c < channels) ? divide_h[c] : 1.f;
        pp.swap[c] = (swap_h && c < channels) ? swap_h[c] : c;
        if (pp.swap[c] < 0 || pp.swap[c] >= 4) return SSDHIP_E_BADARG;
    }
    pp.has_scale = divide_h ? 1 : 0;
    if (channels == 3 && !(n_pixels & 3) && !((uintptr_t)images & 15) && !((uintptr_t)out & 7) && pp.swap[0] < 3 && pp.swap[1] < 3 && pp.swap[2] < 3) {
        hipLaunchKernelGGL(preprocess3_kernel, dim3(grid_for((size_t)(n_pixels / 4), 256)), dim3(256), 0, stream,
                           reinterpret_cast<const float4*>(images), static_cast<uint2*>(out), (u32)(n_pixels / 4), pp);
        return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((size_t)n_pixels, 256)), dim3(256), 0, stream, images,
                       static_cast<bf16_t*>(out), (u32)n_pixels, channels, pp);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_assemble_predictions_strided_bf16(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                                        const void* const* conf_bias_h, const void* const* loc_bias_h,
                                                        const int* n_anchors_h, const int* n_boxes_h,
                                                        const int* conf_stride_h, const int* loc_stride_h,
                                                        const float* anchors_var, int B, int N, int C, float* y_pred,
                                                        void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!anchors_var || !y_pred || B <= 0) return SSDHIP_E_BADARG;
    HeadParams hp;
    int tiles = 0;
    const int rc = head_fill_params(hp, n_layers, conf_h, loc_h, conf_bias_h, loc_bias_h, n_anchors_h, n_boxes_h, conf_stride_h,
                                    loc_stride_h, N, C, 60 * 1024, &tiles);
    if (rc != SSDHIP_OK) return rc;
    hipLaunchKernelGGL(head_kernel, dim3(tiles, B), dim3(256), head_tile_lds(hp.TA, C), stream, hp, anchors_var, y_pred);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// The same from FLOAT32 head outputs whose bias the convolution has already added (the reference-precision path, models/precise.py:
// the packed conf + loc maps of ssdhip_conv2d_x3_nhwc_f16 with out_f32): strides count float32 elements.
extern "C" int ssdhip_assemble_predictions_strided_f32(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                                       const int* n_anchors_h, const int* n_boxes_h, const int* conf_stride_h,
                                                       const int* loc_stride_h, const float* anchors_var, int B, int N, int C,
                                                       float* y_pred, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!anchors_var || !y_pred || B <= 0) return SSDHIP_E_BADARG;
    HeadParams hp;
    int tiles = 0;
    const int rc = head_fill_params(hp, n_layers, conf_h, loc_h, nullptr, nullptr, n_anchors_h, n_boxes_h, conf_stride_h, loc_stride_h,
                                    N, C, 60 * 1024, &tiles, 1);
    if (rc != SSDHIP_OK) return rc;
    hipLaunchKernelGGL(head_kernel, dim3(tiles, B), dim3(256), head_tile_lds(hp.TA, C), stream, hp, anchors_var, y_pred);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_assemble_predictions_bf16(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                                const void* const* conf_bias_h, const void* const* loc_bias_h,
                                                const int* n_anchors_h, const int* n_boxes_h, const float* anchors_var,
                                                int B, int N, int C, float* y_pred, void* stream) {
    return ssdhip_assemble_predictions_strided_bf16(n_layers, conf_h, loc_h, conf_bias_h, loc_bias_h, n_anchors_h, n_boxes_h,
                                                    nullptr, nullptr, anchors_var, B, N, C, y_pred, stream);
}

// LDS bytes head_grad_kernel needs for these source maps: two HG_TILE-anchor row tiles (predictions, their gradient) + the widest
// packed stage.  0 = arguments the backward would refuse.  The launch below takes up to HG_MAX_LDS (a CU has 160 KB; beyond 64 KB the
// kernel is opted in per device) -- the host side gates the one-launch training assembly on THIS function (a fixed class
// count in the gate and the LDS test here disagreed for 36-40 classes on 4-box maps).
constexpr size_t HG_MAX_LDS = 160 * 1024 - 64;
extern "C" size_t ssdhip_assemble_backward_lds_bytes(int n_layers, const int* n_boxes_h, const int* stride_h, int C) {
    if (n_layers <= 0 || n_layers > MAX_PRED_LAYERS || !n_boxes_h || !stride_h || C < 2) return 0;
    size_t max_stage = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int nb = n_boxes_h[l];
        if (nb <= 0 || nb > HG_TILE || (stride_h[l] & 7) || stride_h[l] < nb * (C + 4)) return 0;
        const size_t st = (size_t)(HG_TILE / nb) * stride_h[l] * 2;
        max_stage = st > max_stage ? st : max_stage;
    }
    return 2 * (((size_t)HG_TILE * (C + 12) + 4 + 3) / 4 * 4) * sizeof(float) + max_stage + 16;
}

// Backward of ssdhip_assemble_predictions_strided_bf16 for packed heads (the training step): grad_pred, y_pred [B, N, C+12] float32 (the
// gradient of the loss with respect to the assembled predictions, and those predictions); grad_heads_h[l]: the gradient of source map
// l's packed head output [B, n_anchors[l] / n_boxes[l], stride[l]] bf16 with channels [conf n_boxes C | loc n_boxes 4 | padding] --
// written whole, padding channels zero.  stride[l] (elements per pixel) must be a multiple of 8 and hold n_boxes[l] (C + 4) values.
extern "C" int ssdhip_assemble_predictions_backward_bf16(int n_layers, void* const* grad_heads_h, const int* n_anchors_h, const int* n_boxes_h,
                                                         const int* stride_h, const float* y_pred, const float* grad_pred, int B, int N, int C,
                                                         void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_layers <= 0 || n_layers > MAX_PRED_LAYERS || !grad_heads_h || !n_anchors_h || !n_boxes_h || !stride_h || !y_pred || !grad_pred ||
        B <= 0 || B > 65535 || N <= 0 || C < 2)
        return SSDHIP_E_BADARG;
    HeadGradParams hp;
    hp.n_layers = n_layers; hp.N = N; hp.C = C;
    int off = 0, tiles = 0;
    size_t max_stage = 0;
    for (int l = 0; l < MAX_PRED_LAYERS; ++l) {
        const bool on = l < n_layers;
        hp.out[l] = on ? static_cast<bf16_t*>(grad_heads_h[l]) : nullptr;
        hp.n_anchors[l] = on ? n_anchors_h[l] : 0;
        hp.n_boxes[l] = on ? n_boxes_h[l] : 1;
        hp.stride[l] = on ? stride_h[l] : 0;
        hp.tile_start[l] = tiles;
        hp.anchor_off[l] = off;
        hp.tile_anchors[l] = 1;
        if (on) {
            const int nb = hp.n_boxes[l];
            if (!hp.out[l] || ((uintptr_t)hp.out[l] & 15) || hp.n_anchors[l] <= 0 || nb <= 0 || nb > HG_TILE || hp.n_anchors[l] % nb ||
                (hp.stride[l] & 7) || hp.stride[l] < nb * (C + 4))
                return SSDHIP_E_BADARG;
            hp.tile_anchors[l] = HG_TILE / nb * nb;
            off += hp.n_anchors[l];
            tiles += (hp.n_anchors[l] + hp.tile_anchors[l] - 1) / hp.tile_anchors[l];
            const size_t st = (size_t)(hp.tile_anchors[l] / nb) * hp.stride[l] * 2;
            max_stage = st > max_stage ? st : max_stage;
        }
    }
    hp.tile_start[MAX_PRED_LAYERS] = tiles;
    if (off != N) return SSDHIP_E_BADARG;
    const size_t lds = 2 * (((size_t)HG_TILE * (C + 12) + 4 + 3) / 4 * 4) * sizeof(float) + max_stage + 16;
    if (lds > HG_MAX_LDS) return SSDHIP_E_BADARG;
    if (lds > 48 * 1024) {                                   // opted in once per device for the whole range (no per-call attribute traffic)
        static int opted[64] = {0};
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return SSDHIP_E_LAUNCH;
        if (opted[devid] == 0)
            opted[devid] = hipFuncSetAttribute(reinterpret_cast<const void*>(head_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)HG_MAX_LDS) == hipSuccess ? 1 : -1;
        if (opted[devid] < 0) return SSDHIP_E_LAUNCH;
    }
    hipLaunchKernelGGL(head_grad_kernel, dim3(tiles, B), dim3(HG_TILE), lds, stream, hp, y_pred, grad_pred);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// L2Normalization of a pair map: x, y [n_pixels][2 C] float16 = [hi | lo] (x3_split's layout), gamma [C] float32; the true input is
// (hi + lo) * scale (the layer's power-of-two divisor), the output is stored with divisor 1.  C % 8 == 0.  The float32 result of
// ssdhip_l2_normalize_fwd on the merged map, re-split.
extern "C" int ssdhip_x3_l2_normalize_nhwc(const void* x, const float* gamma, void* y, long long n_pixels, int C, float scale, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !gamma || !y || n_pixels <= 0 || n_pixels > 0x7fffffffLL || C <= 0 || (C & 7) || !(scale > 0.f)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return SSDHIP_E_BADARG;
    long long blocks = (n_pixels + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(x3_l2norm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const uint4*>(x), gamma, static_cast<uint4*>(y),
                       (u32)n_pixels, (u32)(C / 8), scale);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
