// Prediction assembly shared by head_kernel (csrc/ssdhip_layers.hip: rows -> y_pred in HBM) and scan_heads_kernel
// (csrc/ssdhip_decode.hip: rows stay in LDS and are decoded at once).  Reference: Reshape + Concatenate + softmax + AnchorBoxes
// + Concatenate, models/keras_ssd300.py:363-419 and keras_layers/keras_layer_AnchorBoxes.py:245-255.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_tile.h"

namespace ssdhip {

typedef unsigned short hbf16_t;

__device__ __forceinline__ float h_bf2f(u32 h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ u32 h_f2bf(float f) {          // round to nearest even, NaN stays NaN (as c10::BFloat16)
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int MAX_PRED_LAYERS = 8;
struct HeadParams {
    const hbf16_t* conf[MAX_PRED_LAYERS];
    const hbf16_t* loc[MAX_PRED_LAYERS];
    const hbf16_t* conf_bias[MAX_PRED_LAYERS];    // [n_boxes*C] or null
    const hbf16_t* loc_bias[MAX_PRED_LAYERS];     // [n_boxes*4] or null
    int n_anchors[MAX_PRED_LAYERS];
    int n_boxes[MAX_PRED_LAYERS];
    int conf_stride[MAX_PRED_LAYERS];             // elements between consecutive pixels of the conf / loc source
    int loc_stride[MAX_PRED_LAYERS];              // (n_boxes*C and n_boxes*4 when the heads are separate, dense tensors)
    int tile_start[MAX_PRED_LAYERS + 1];          // first tile of layer l
    int anchor_off[MAX_PRED_LAYERS];
    int n_layers, N, C, TA;
    int row_dma;                                  // packed bf16 heads: the tile's pixel rows by LDS-DMA (0 with SSDHIP_HEADS_DMA=0: the element-wise gather, A/B)
    int src_f32;                                  // the head outputs are float32 (the reference-precision path, models/precise.py): conf / loc
                                                  // point at float32 values, strides count float32 elements, biases must be null
};

// LDS bytes of one tile: [TA][C+12] float rows + the logits' staging area -- [TA][C] + [TA][4] bf16, or (packed heads, round 5) the
// tile's whole pixel rows as LDS-DMA leaves them: up to TA / n_boxes + 2 rows of 16-byte chunks, the last load's idle lanes, the biases
__host__ __device__ inline size_t head_stage_bytes(int TA, int C) {
    return (size_t)TA * (C + 4) * sizeof(hbf16_t) + (size_t)16 * TA + (size_t)64 * (C + 4) + 1024;
}
__host__ __device__ inline size_t head_tile_lds(int TA, int C) { return (size_t)TA * (C + 12) * sizeof(float) + head_stage_bytes(TA, C); }

// Which layer / anchors the tile `tile_id` of the launch covers.
__device__ __forceinline__ void head_tile_of(const HeadParams& hp, int tile_id, int& l, int& a0, int& na) {
    l = 0;
    while (l + 1 < hp.n_layers && tile_id >= hp.tile_start[l + 1]) ++l;
    a0 = (tile_id - hp.tile_start[l]) * hp.TA;
    na = min(hp.TA, hp.n_anchors[l] - a0);
}

// Builds rows[a][0..C+11] = [softmax(conf + bias) | loc + bias | anchor | variances] for anchors a0..a0+na-1 of layer l, image b.
// `rows`: [TA][C+12] float in LDS, `cl`: [TA][C] + [TA][4] bf16 staging in LDS.  Ends with a __syncthreads().
__device__ __forceinline__ void head_build_rows(const HeadParams& hp, const float* __restrict__ anchors_var, int l, int b, int a0, int na,
                                                float* rows, hbf16_t* cl, int tid, int nthreads) {
    const int TA = hp.TA, C = hp.C, L = C + 12;
    hbf16_t* ll = cl + (size_t)TA * C;
    const int nb = hp.n_boxes[l];
    if (hp.src_f32) {
        // float32 head outputs (bias already added by the convolution): the values go straight into the float32 rows -- the same index
        // walk as the bf16 packed form below, one element per thread and trip
        const size_t px0 = (size_t)b * (hp.n_anchors[l] / nb);
        {
            const int step_a = nthreads / C, step_c = nthreads - step_a * C;
            int c = tid % C, a = tid / C;
            const int ga0 = a0 + a;
            int pix = ga0 / nb, box = ga0 - pix * nb;
            const float* src = reinterpret_cast<const float*>(hp.conf[l]) + px0 * hp.conf_stride[l];
            const int cs = hp.conf_stride[l];
            const int q0 = step_a / nb, r0 = step_a - q0 * nb, q1 = (step_a + 1) / nb, r1 = step_a + 1 - q1 * nb;
            for (int i = tid; i < na * C; i += nthreads) {
                rows[(size_t)a * L + c] = src[(size_t)pix * cs + box * C + c];
                c += step_c;
                const bool carry = c >= C;
                c -= carry ? C : 0;
                a += step_a + (carry ? 1 : 0);
                box += carry ? r1 : r0;
                pix += carry ? q1 : q0;
                if (box >= nb) { box -= nb; ++pix; }
            }
        }
        {
            const int step_a = nthreads >> 2;
            const int k = tid & 3;
            int a = tid >> 2;
            const int ga0 = a0 + a;
            int pix = ga0 / nb, box = ga0 - pix * nb;
            const float* src = reinterpret_cast<const float*>(hp.loc[l]) + px0 * hp.loc_stride[l];
            const int ls = hp.loc_stride[l];
            const int q0 = step_a / nb, r0 = step_a - q0 * nb;
            for (int i = tid; i < na * 4; i += nthreads) {
                rows[(size_t)a * L + C + k] = src[(size_t)pix * ls + box * 4 + k];
                a += step_a;
                box += r0;
                pix += q0;
                if (box >= nb) { box -= nb; ++pix; }
            }
        }
        __syncthreads();
        for (int a = tid; a < na; a += nthreads) {
            float* r = rows + (size_t)a * L;
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, r[c]);
            float sum = 0.f;
            for (int c = 0; c < C; ++c) { const float e = expf(r[c] - mx); r[c] = e; sum += e; }
            for (int c = 0; c < C; ++c) r[c] = r[c] / sum;
            const float* av = anchors_var + (size_t)(hp.anchor_off[l] + a0 + a) * 8;
            for (int k = 0; k < 8; ++k) r[C + 4 + k] = av[k];
        }
        __syncthreads();
        return;
    }
    // ---- packed bf16 heads, rows by LDS-DMA (round 5).  The element-wise gather below keeps ONE two-byte load in flight per thread
    //      (25 dependent trips per tile) and the softmax loop read its bias from global memory per class: together the larger part of
    //      scan_heads_kernel's 52 us.  A packed head's pixel row is [conf n_boxes C | loc n_boxes 4 | padding] bf16: the tile's pixel
    //      rows come in as 16-byte chunks (tile_dma16: every load of the workgroup in flight at once, no registers), the biases are
    //      staged in LDS once, the anchor's 8 template floats are requested before the wait. ----
    {
        const int stride = hp.conf_stride[l];
        const int U = nb * (C + 4) * 2, cpr = (U + 15) >> 4, pitch = cpr * 16;
        const int pix0 = a0 / nb, npx = (a0 + na - 1) / nb - pix0 + 1, nchunks = npx * cpr;
        const size_t dma_area = (size_t)((nchunks + 63) >> 6) * 1024;
        const size_t npix_l = (size_t)(hp.n_anchors[l] / nb);
        const size_t src_bytes = (size_t)gridDim.y * npix_l * (size_t)stride * 2;
        const bool dma = hp.row_dma && hp.loc[l] == hp.conf[l] + (size_t)nb * C && hp.loc_stride[l] == stride && !(stride & 7) && stride * 2 >= pitch &&
                         !((uintptr_t)hp.conf[l] & 15) && dma_area + (size_t)U <= head_stage_bytes(TA, C) && src_bytes < 0x7fffff00ull &&
                         !(nthreads & 63);
        if (dma) {
            unsigned char* st = reinterpret_cast<unsigned char*>(cl);
            const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)st;
            const tile_i32x4 rs = tile_rsrc(hp.conf[l], (u32)src_bytes);
            const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const u32 row0 = (u32)(((size_t)b * npix_l + pix0) * (size_t)stride * 2);
            for (int q0 = wave * 64; q0 < nchunks; q0 += nthreads) {
                const int q = q0 + lane, r = q / cpr, j = q - r * cpr;
                tile_dma16(q < nchunks ? row0 + (u32)r * (u32)(stride * 2) + (u32)j * 16u : TILE_OOB, rs, lds0 + (u32)q0 * 16u);
            }
            hbf16_t* bs = reinterpret_cast<hbf16_t*>(st + dma_area);                     // [nb C] conf bias | [nb 4] loc bias
            const bool has_cb = hp.conf_bias[l] != nullptr, has_lb = hp.loc_bias[l] != nullptr;
            if (has_cb) for (int i = tid; i < nb * C; i += nthreads) bs[i] = hp.conf_bias[l][i];
            if (has_lb) for (int i = tid; i < nb * 4; i += nthreads) bs[nb * C + i] = hp.loc_bias[l][i];
            const bool av16 = !((uintptr_t)anchors_var & 15);
            float4 av0 = make_float4(0.f, 0.f, 0.f, 0.f), av1 = av0;
            if (av16 && tid < na) {
                const float4* avp = reinterpret_cast<const float4*>(anchors_var + (size_t)(hp.anchor_off[l] + a0 + tid) * 8);
                av0 = avp[0]; av1 = avp[1];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int a = tid; a < na; a += nthreads) {
                const int ga = a0 + a, pix = ga / nb, box = ga - pix * nb;
                const hbf16_t* prow = reinterpret_cast<const hbf16_t*>(st + (size_t)(pix - pix0) * pitch);
                const hbf16_t* src = prow + box * C;
                const hbf16_t* cb = bs + box * C;
                float* r = rows + (size_t)a * L;
                float mx = -INFINITY;
                int c = 0;
                for (; c + 4 <= C; c += 4) {                            // four classes' LDS reads in flight together
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float x = h_bf2f(src[c + u]);
                        // the PyTorch path rounds conv + bias to bf16 before the float32 softmax: keep that rounding
                        v[u] = has_cb ? h_bf2f(h_f2bf(x + h_bf2f(cb[c + u]))) : x;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { r[c + u] = v[u]; mx = fmaxf(mx, v[u]); }
                }
                for (; c < C; ++c) {
                    const float x = h_bf2f(src[c]);
                    const float v = has_cb ? h_bf2f(h_f2bf(x + h_bf2f(cb[c]))) : x;
                    r[c] = v;
                    mx = fmaxf(mx, v);
                }
                float sum = 0.f;
                for (c = 0; c + 4 <= C; c += 4) {
                    float e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = r[c + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { e[u] = expf(e[u] - mx); sum += e[u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[c + u] = e[u];
                }
                for (; c < C; ++c) { const float e = expf(r[c] - mx); r[c] = e; sum += e; }
                for (c = 0; c + 4 <= C; c += 4) {
                    float e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = r[c + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) r[c + u] = e[u] / sum;
                }
                for (; c < C; ++c) r[c] = r[c] / sum;
                const hbf16_t* lsrc = prow + nb * C + box * 4;
                const hbf16_t* lb = bs + nb * C + box * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[C + k] = has_lb ? h_bf2f(h_f2bf(h_bf2f(lsrc[k]) + h_bf2f(lb[k]))) : h_bf2f(lsrc[k]);
                if (av16 && a == tid) {
                    r[C + 4] = av0.x; r[C + 5] = av0.y; r[C + 6] = av0.z; r[C + 7] = av0.w;
                    r[C + 8] = av1.x; r[C + 9] = av1.y; r[C + 10] = av1.z; r[C + 11] = av1.w;
                } else {
                    const float* av = anchors_var + (size_t)(hp.anchor_off[l] + a0 + a) * 8;
                    for (int k = 0; k < 8; ++k) r[C + 4 + k] = av[k];
                }
            }
            __syncthreads();
            return;
        }
    }
    if (hp.conf_stride[l] == nb * C && hp.loc_stride[l] == nb * 4) {                    // dense heads: contiguous spans
        const hbf16_t* csrc = hp.conf[l] + ((size_t)b * hp.n_anchors[l] + a0) * C;
        const hbf16_t* lsrc = hp.loc[l] + ((size_t)b * hp.n_anchors[l] + a0) * 4;
        for (int i = tid; i < na * C; i += nthreads) cl[i] = csrc[i];
        for (int i = tid; i < na * 4; i += nthreads) ll[i] = lsrc[i];
    } else {                                                                            // heads packed into one wider conv output
        // element i = (anchor a0 + i / C, class i % C) = (pixel, box, class): the three indices advance by constant steps with
        // carries -- four integer divisions per ELEMENT (~150 instructions, 21 trips per thread) made this loop the bulk of the
        // kernel (58 us in the step, r02z)
        const size_t px0 = (size_t)b * (hp.n_anchors[l] / nb);
        {
            const int step_a = nthreads / C, step_c = nthreads - step_a * C;
            int c = tid % C;
            const int ga0 = a0 + tid / C;
            int pix = ga0 / nb, box = ga0 - pix * nb;
            const hbf16_t* src = hp.conf[l] + px0 * hp.conf_stride[l];
            const int cs = hp.conf_stride[l];
            const int q0 = step_a / nb, r0 = step_a - q0 * nb, q1 = (step_a + 1) / nb, r1 = step_a + 1 - q1 * nb;   // anchors -> (pixels, boxes)
            for (int i = tid; i < na * C; i += nthreads) {
                cl[i] = src[(size_t)pix * cs + box * C + c];
                c += step_c;
                const bool carry = c >= C;
                c -= carry ? C : 0;
                box += carry ? r1 : r0;
                pix += carry ? q1 : q0;
                if (box >= nb) { box -= nb; ++pix; }
            }
        }
        {
            const int step_a = nthreads >> 2;                                           // nthreads is a multiple of 64
            const int k = tid & 3;
            const int ga0 = a0 + (tid >> 2);
            int pix = ga0 / nb, box = ga0 - pix * nb;
            const hbf16_t* src = hp.loc[l] + px0 * hp.loc_stride[l];
            const int ls = hp.loc_stride[l];
            const int q0 = step_a / nb, r0 = step_a - q0 * nb;
            for (int i = tid; i < na * 4; i += nthreads) {
                ll[i] = src[(size_t)pix * ls + box * 4 + k];
                box += r0;
                pix += q0;
                if (box >= nb) { box -= nb; ++pix; }
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < na; a += nthreads) {
        const int box = (a0 + a) % nb;                        // (one division per row)
        float* r = rows + (size_t)a * L;
        const hbf16_t* cb = hp.conf_bias[l] ? hp.conf_bias[l] + box * C : nullptr;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) {
            // the PyTorch path rounds conv + bias to bf16 before the float32 softmax: keep that rounding
            const float v = cb ? h_bf2f(h_f2bf(h_bf2f(cl[a * C + c]) + h_bf2f(cb[c]))) : h_bf2f(cl[a * C + c]);
            r[c] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
        for (int c = 0; c < C; ++c) { const float e = expf(r[c] - mx); r[c] = e; sum += e; }
        for (int c = 0; c < C; ++c) r[c] = r[c] / sum;
        const hbf16_t* lb = hp.loc_bias[l] ? hp.loc_bias[l] + box * 4 : nullptr;
        for (int k = 0; k < 4; ++k) r[C + k] = lb ? h_bf2f(h_f2bf(h_bf2f(ll[a * 4 + k]) + h_bf2f(lb[k]))) : h_bf2f(ll[a * 4 + k]);
        const float* av = anchors_var + (size_t)(hp.anchor_off[l] + a0 + a) * 8;
        for (int k = 0; k < 8; ++k) r[C + 4 + k] = av[k];
    }
    __syncthreads();
}

// Host side: validate the per-layer arrays of the C ABI and fill HeadParams.  Returns SSDHIP_OK or SSDHIP_E_BADARG; *tiles_out =
// number of tiles (grid.x), hp.TA = anchors per tile such that head_tile_lds(TA, C) + extra_lds_per_tile fits max_lds bytes.
static inline int head_fill_params(HeadParams& hp, int n_layers, const void* const* conf_h, const void* const* loc_h,
                                   const void* const* conf_bias_h, const void* const* loc_bias_h, const int* n_anchors_h,
                                   const int* n_boxes_h, const int* conf_stride_h, const int* loc_stride_h, int N, int C,
                                   size_t max_lds, int* tiles_out, int src_f32 = 0) {
    if (n_layers <= 0 || n_layers > MAX_PRED_LAYERS || !conf_h || !loc_h || !n_anchors_h || !n_boxes_h || N <= 0 || C < 2 || C > 1024)
        return SSDHIP_E_BADARG;
    int TA = 256;
    while (TA > 32 && head_tile_lds(TA, C) > max_lds) TA >>= 1;
    hp.n_layers = n_layers; hp.N = N; hp.C = C; hp.TA = TA; hp.src_f32 = src_f32 ? 1 : 0;
    { const char* e = getenv("SSDHIP_HEADS_DMA"); hp.row_dma = !(e && e[0] == '0'); }
    if (src_f32 && (conf_bias_h || loc_bias_h)) {
        for (int l = 0; l < n_layers; ++l)
            if ((conf_bias_h && conf_bias_h[l]) || (loc_bias_h && loc_bias_h[l])) return SSDHIP_E_BADARG;   // float32 heads carry their bias already
    }
    int off = 0, tiles = 0;
    for (int l = 0; l < MAX_PRED_LAYERS; ++l) {
        const bool on = l < n_layers;
        hp.conf[l] = on ? static_cast<const hbf16_t*>(conf_h[l]) : nullptr;
        hp.loc[l] = on ? static_cast<const hbf16_t*>(loc_h[l]) : nullptr;
        hp.conf_bias[l] = (on && conf_bias_h) ? static_cast<const hbf16_t*>(conf_bias_h[l]) : nullptr;
        hp.loc_bias[l] = (on && loc_bias_h) ? static_cast<const hbf16_t*>(loc_bias_h[l]) : nullptr;
        hp.n_anchors[l] = on ? n_anchors_h[l] : 0;
        hp.n_boxes[l] = on ? n_boxes_h[l] : 1;
        hp.conf_stride[l] = on ? (conf_stride_h ? conf_stride_h[l] : n_boxes_h[l] * C) : 0;
        hp.loc_stride[l] = on ? (loc_stride_h ? loc_stride_h[l] : n_boxes_h[l] * 4) : 0;
        hp.tile_start[l] = tiles;
        hp.anchor_off[l] = off;
        if (on) {
            if (!hp.conf[l] || !hp.loc[l] || hp.n_anchors[l] <= 0 || hp.n_boxes[l] <= 0 || hp.n_anchors[l] % hp.n_boxes[l]) return SSDHIP_E_BADARG;
            if (hp.conf_stride[l] < hp.n_boxes[l] * C || hp.loc_stride[l] < hp.n_boxes[l] * 4) return SSDHIP_E_BADARG;
            off += hp.n_anchors[l];
            tiles += (hp.n_anchors[l] + TA - 1) / TA;
        }
    }
    hp.tile_start[MAX_PRED_LAYERS] = tiles;
    if (off != N) return SSDHIP_E_BADARG;
    *tiles_out = tiles;
    return SSDHIP_OK;
}

}  // namespace ssdhip
