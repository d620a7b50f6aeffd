// ssdhip_encode.hip -- ground truth -> SSD training targets on gfx950 (MI355X).
//
// Replaces SSDInputEncoder.__call__ (reference ssd_encoder_decoder/ssd_input_encoder.py:277-418) together
// with iou() (bounding_box_utils/bounding_box_utils.py:283-383), match_bipartite_greedy / match_multi
// (ssd_encoder_decoder/matching_utils.py:22-116) and the template tiling (generate_encoding_template :550-611).
// Everything is float64 in the reference's operation order (compiled with -ffp-contract=off), so the match
// assignment is bit exact; only log() may differ from NumPy's in the last place.
//
// Three kernels on the caller's stream; the (g, N) similarity matrix of the reference is never materialised:
//   R rowmax_kernel    grid (anchor tiles, B): each ground truth row's best (IoU, first column) inside the tile.
//   M match_kernel     grid (B): match_bipartite_greedy for one image -- the g sequential rounds on the row maxima; a row is
//                      re-scanned (its IoUs recomputed by 1024 threads, taken columns masked) only when its column was taken.
//   F finalize_kernel  grid (anchor tiles, B): per anchor its g IoUs again (a few hundred float64 operations), the 'multi' match
//                      (first argmax over GT, >= threshold), the neutral test, one-hot / box / offset encoding; rows are staged
//                      in LDS and written as 16-byte stores (float32) and / or float64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

constexpr int ENC_THREADS = 256;
constexpr int ENC_MAX_GT = 1024;

struct EncodeParams {
    int B, N, C, L, tiles, max_gt;
    int matching_multi, coords, normalize, border, background_id;
    double img_h, img_w, pos_thr, neg_limit;
};

struct GtBox {            // one ground truth box, prepared as the reference prepares it (:339-350)
    double lab[4];        // in `coords` format, normalised if requested: what gets written to matched anchors
    PxBox<double> cr;     // the 'corners' view iou() works on (after its own centroids->corners conversion) + area
    int cls;
};

__device__ __forceinline__ double border_d(int border) {
    return border == SSDHIP_BORDER_INCLUDE ? 1.0 : (border == SSDHIP_BORDER_EXCLUDE ? -1.0 : 0.0);
}

// box in `coords` format -> the corner view + area used by iou() (bounding_box_utils.py:334-337, 364-378)
__device__ __forceinline__ PxBox<double> corner_view(const double v[4], int coords, double d) {
    PxBox<double> r;
    if (coords == SSDHIP_CENTROIDS) {            // centroids2corners with border 'half'
        r.x0 = v[0] - v[2] / 2.0;
        r.y0 = v[1] - v[3] / 2.0;
        r.x1 = v[0] + v[2] / 2.0;
        r.y1 = v[1] + v[3] / 2.0;
    } else if (coords == SSDHIP_MINMAX) {
        r.x0 = v[0]; r.x1 = v[1]; r.y0 = v[2]; r.y1 = v[3];
    } else {
        r.x0 = v[0]; r.y0 = v[1]; r.x1 = v[2]; r.y1 = v[3];
    }
    r.area = box_area<double>(r.x0, r.y0, r.x1, r.y1, d);
    return r;
}

__device__ __forceinline__ void load_gt(GtBox& out, const double* __restrict__ row, const EncodeParams& p) {
    double xmin = row[1], ymin = row[2], xmax = row[3], ymax = row[4];
    if (p.normalize) {                           // :339-341
        ymin /= p.img_h; ymax /= p.img_h;
        xmin /= p.img_w; xmax /= p.img_w;
    }
    const double d = border_d(p.border);
    if (p.coords == SSDHIP_CENTROIDS) {          // corners2centroids with the encoder's border_pixels (:345)
        out.lab[0] = (xmin + xmax) / 2.0;
        out.lab[1] = (ymin + ymax) / 2.0;
        out.lab[2] = xmax - xmin + d;
        out.lab[3] = ymax - ymin + d;
    } else if (p.coords == SSDHIP_MINMAX) {
        out.lab[0] = xmin; out.lab[1] = xmax; out.lab[2] = ymin; out.lab[3] = ymax;
    } else {
        out.lab[0] = xmin; out.lab[1] = ymin; out.lab[2] = xmax; out.lab[3] = ymax;
    }
    out.cr = corner_view(out.lab, p.coords, d);
    out.cls = (int)row[0];
}

// (value, column) ordering of np.argmax along a row: larger value first, then the lower column
__device__ __forceinline__ bool better(double v, int c, double bv, int bc) { return v > bv || (v == bv && c < bc); }

__device__ __forceinline__ void wave_argmax(double& v, int& c) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oc = __shfl_xor(c, off);
        if (better(ov, oc, v, c)) { v = ov; c = oc; }
    }
}

__device__ __forceinline__ PxBox<double> load_anchor_view(const double* __restrict__ anchors, int n, const EncodeParams& p) {
    const double a[4] = {anchors[(size_t)n * 4], anchors[(size_t)n * 4 + 1], anchors[(size_t)n * 4 + 2], anchors[(size_t)n * 4 + 3]};
    return corner_view(a, p.coords, border_d(p.border));
}

// ======================================================================================
// R: per-tile row maxima
// ======================================================================================
// grid (anchor tiles, B): IoU of the tile's 256 anchors with the image's ground truth boxes, reduced at once to each row's best
// (value, first column) inside the tile -- what the bipartite matching starts from.  The similarities themselves are not stored.
__global__ __launch_bounds__(ENC_THREADS) void rowmax_kernel(EncodeParams p, const double* __restrict__ anchors,
                                                             const double* __restrict__ gt, const int* __restrict__ gt_off,
                                                             double* __restrict__ part_val, int* __restrict__ part_col) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    GtBox* gts = reinterpret_cast<GtBox*>(smem_raw);
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = gt_off[b], g = min(gt_off[b + 1] - g0, p.max_gt);   // the LDS tables hold max_gt boxes: never index beyond them
    if (g <= 0) return;
    for (int r = tid; r < g; r += ENC_THREADS) load_gt(gts[r], gt + (size_t)(g0 + r) * 5, p);
    __syncthreads();
    const int n = tile * ENC_THREADS + tid;
    const bool active = n < p.N;
    PxBox<double> an = {};
    if (active) an = load_anchor_view(anchors, n, p);
    // per WAVE partial maxima (part_* [row][tile][wave]): no workgroup combine, no barrier per row -- the matching kernel reduces
    // the tiles x 4 partials of a row with one wave
    const size_t pw = (size_t)p.tiles * (ENC_THREADS / 64);
    for (int r = 0; r < g; ++r) {
        double v = -1.0;                           // IoU >= 0, so inactive lanes never win
        int c = 0x7fffffff;
        if (active) { v = iou_px<double>(gts[r].cr, an); c = n; }
        wave_argmax(v, c);
        if (lane == 0) {
            part_val[(size_t)(g0 + r) * pw + tile * (ENC_THREADS / 64) + wave] = v;
            part_col[(size_t)(g0 + r) * pw + tile * (ENC_THREADS / 64) + wave] = c;
        }
    }
}

// ======================================================================================
// M: bipartite matching of one image, similarities recomputed on the fly
// ======================================================================================
// One workgroup of MATCH_THREADS per image.  match_bipartite_greedy (matching_utils.py:22-79) needs, per ground truth row, the
// first arg-max over ALL anchors, then g sequential rounds; neither needs the (g, N) similarity matrix in memory: the row maxima
// come from rowmax_kernel's per-tile results, and a row is re-scanned -- its IoUs recomputed, the taken columns masked by an
// LDS bitmap -- only when the column it pointed to was just taken.  Output: matches[g0 + r] = anchor of row r.
// Quirks kept: an all-zero row yields column 0, and once everything left is zero GT 0 is re-assigned anchor 0.
constexpr int MATCH_THREADS = 1024;

// The g sequential rounds of match_bipartite_greedy on the row maxima (match_kernel's second half); all
// MATCH_THREADS threads of the workgroup call it.  wv / wc: NW-entry LDS scratch, pick_col: one LDS int.
__device__ __forceinline__ void match_rounds(const EncodeParams& p, const double* __restrict__ anchors, const int g, const GtBox* gts,
                                             double* rowval, int* rowcol, int* match, int* rowgone, u32* colgone, double* wv, int* wc,
                                             int* pick_col_p) {
    constexpr int NW = MATCH_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int round = 0; round < g; ++round) {
        // the largest remaining entry: first argmax over rows of the per-row first argmax (:63-68)
        if (wave == 0) {
            double bv = -1.0;
            int br = 0x7fffffff;
            for (int r = lane; r < g; r += 64) {
                const double v = rowgone[r] ? 0.0 : rowval[r];
                if (better(v, r, bv, br)) { bv = v; br = r; }
            }
            wave_argmax(bv, br);
            if (lane == 0) {
                const int col = rowgone[br] ? 0 : rowcol[br];
                *pick_col_p = col;
                match[br] = col;
                rowgone[br] = 1;
                colgone[col >> 5] |= 1u << (col & 31);
            }
        }
        __syncthreads();
        const int col = *pick_col_p;
        // rows that pointed at the column just taken need a new maximum
        for (int r = 0; r < g; ++r) {
            if (rowgone[r] || rowcol[r] != col || !(rowval[r] > 0.0)) continue;      // uniform across the block
            double bv = -1.0;
            int bc = 0x7fffffff;
            for (int n = tid; n < p.N; n += MATCH_THREADS) {
                double v = 0.0;
                if (!((colgone[n >> 5] >> (n & 31)) & 1u)) v = iou_px<double>(gts[r].cr, load_anchor_view(anchors, n, p));
                if (v > bv) { bv = v; bc = n; }
            }
            wave_argmax(bv, bc);
            if (lane == 0) { wv[wave] = bv; wc[wave] = bc; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NW; ++w) if (better(wv[w], wc[w], bv, bc)) { bv = wv[w]; bc = wc[w]; }
                if (!(bv > 0.0)) bc = 0;
                rowval[r] = bv; rowcol[r] = bc;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(EncodeParams p, const double* __restrict__ anchors,
                                                              const double* __restrict__ gt, const int* __restrict__ gt_off,
                                                              const double* __restrict__ part_val, const int* __restrict__ part_col,
                                                              int* __restrict__ matches) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NW = MATCH_THREADS / 64;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = gt_off[b], g = min(gt_off[b + 1] - g0, p.max_gt);   // the LDS tables hold max_gt boxes: never index beyond them
    if (g <= 0) return;
    GtBox* gts = reinterpret_cast<GtBox*>(smem_raw);                               // [max_gt]
    double* rowval = reinterpret_cast<double*>(gts + p.max_gt);                    // [max_gt]
    int* rowcol = reinterpret_cast<int*>(rowval + p.max_gt);                       // [max_gt]
    int* match = rowcol + p.max_gt;                                                // [max_gt]
    int* rowgone = match + p.max_gt;                                               // [max_gt]
    u32* colgone = reinterpret_cast<u32*>(rowgone + p.max_gt);                     // [(N+31)/32] bitmap
    __shared__ double wv[1][NW];
    __shared__ int wc[1][NW];
    __shared__ int pick_col;

    for (int r = tid; r < g; r += MATCH_THREADS) { load_gt(gts[r], gt + (size_t)(g0 + r) * 5, p); match[r] = 0; rowgone[r] = 0; }
    for (int i = tid; i < (p.N + 31) / 32; i += MATCH_THREADS) colgone[i] = 0;
    __syncthreads();

    // ---- first arg-max of every row over all anchors: the best of the per-tile maxima (rowmax_kernel) ----
    {
        const int pw = p.tiles * (ENC_THREADS / 64);                               // partials per row (rowmax_kernel: [tile][wave])
        for (int r = wave; r < g; r += NW) {                                        // one wave per row: the loads of a row go out together
            double bv = -1.0;
            int bc = 0x7fffffff;
            for (int t = lane; t < pw; t += 64) {
                const double v = part_val[(size_t)(g0 + r) * pw + t];
                const int c = part_col[(size_t)(g0 + r) * pw + t];
                if (better(v, c, bv, bc)) { bv = v; bc = c; }
            }
            wave_argmax(bv, bc);
            if (lane == 0) {
                if (!(bv > 0.0)) bc = 0;                                            // np.argmax of an all-zero row
                rowval[r] = bv; rowcol[r] = bc;
            }
        }
    }
    __syncthreads();

    match_rounds(p, anchors, g, gts, rowval, rowcol, match, rowgone, colgone, &wv[0][0], &wc[0][0], &pick_col);
    for (int r = tid; r < g; r += MATCH_THREADS) matches[g0 + r] = match[r];
}

// ======================================================================================
// F: targets
// ======================================================================================
// grid (anchor tiles, B).  Per anchor: is it a bipartite match (y_encoded[i, bipartite_matches, :-8] = labels_one_hot, :363: with
// duplicates the highest GT wins), else its similarities with the image's ground truth boxes -- recomputed, g IoUs per anchor --
// give the 'multi' match (first argmax over GT, >= threshold, :105-109) and the neutral test (:388-390); then one-hot / box /
// offset encoding.  Rows are staged in LDS (float32, or float64 when the float64 copy is requested) and leave as 16-byte stores.
template <typename TileT>
__global__ __launch_bounds__(ENC_THREADS) void finalize_kernel(EncodeParams p, const double* __restrict__ anchors,
                                                               const double* __restrict__ variances,
                                                               const double* __restrict__ gt, const int* __restrict__ gt_off,
                                                               const int* __restrict__ matches,
                                                               float* __restrict__ y32, double* __restrict__ y64,
                                                               int* __restrict__ match_gt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int TA = blockDim.x;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int a0 = blockIdx.x * TA;
    const int na = min(TA, p.N - a0);
    const int L = p.L, C = p.C;
    const int g0 = gt_off[b], g = min(gt_off[b + 1] - g0, p.max_gt);   // the LDS tables hold max_gt boxes: never index beyond them
    const size_t base = ((size_t)b * p.N + a0) * (size_t)L;
    // the LDS image keeps the 16-byte phase of the float32 destination, so that whole uint4 chunks can be copied out
    const int phase = (sizeof(TileT) == 4 && y32) ? (int)((((uintptr_t)(y32 + base)) & 15u) >> 2) : 0;
    TileT* tile = reinterpret_cast<TileT*>(smem_raw) + phase;                 // [TA][L] staged rows
    GtBox* gts = reinterpret_cast<GtBox*>(smem_raw + (((size_t)TA * L + 4) * sizeof(TileT) + 15) / 16 * 16);   // [max_gt]
    int* smatch = reinterpret_cast<int*>(gts + p.max_gt);                     // [max_gt]
    for (int r = tid; r < g; r += TA) { load_gt(gts[r], gt + (size_t)(g0 + r) * 5, p); smatch[r] = matches[g0 + r]; }
    __syncthreads();

    if (tid < na) {
        const int n = a0 + tid;
        double a[4], var[4];
        for (int k = 0; k < 4; ++k) { a[k] = anchors[(size_t)n * 4 + k]; var[k] = variances[k]; }
        int gt_idx = -1;
        bool neutral = false;
        if (g > 0) {
            int bip = -1;                                                    // the highest GT row whose bipartite match is this anchor
            for (int r = 0; r < g; ++r) if (smatch[r] == n) bip = r;
            const bool colzero = bip >= 0;                                   // similarities[:, bipartite_matches] = 0 (:366)
            double best = 0.0;
            int best_r = 0;
            if (!colzero) {
                const PxBox<double> an = corner_view(a, p.coords, border_d(p.border));
                best = iou_px<double>(gts[0].cr, an);
                for (int r = 1; r < g; ++r) {                                // np.argmax over GT: first maximum (:105)
                    const double v = iou_px<double>(gts[r].cr, an);
                    if (v > best) { best = v; best_r = r; }
                }
            }
            const bool multi = p.matching_multi && (best >= p.pos_thr);      // :109
            gt_idx = multi ? best_r : bip;
            const double bg_sim = (colzero || multi) ? 0.0 : best;           // matched columns are zero by now (:381)
            neutral = bg_sim >= p.neg_limit;                                 // :388-390
        }
        TileT* row = tile + (size_t)tid * L;
        for (int c = 0; c < C; ++c) row[c] = (TileT)0;
        double box[4] = {a[0], a[1], a[2], a[3]};                            // template: anchor in place of the GT
        if (gt_idx >= 0) {
            const GtBox& gb = gts[gt_idx];
            if (gb.cls >= 0 && gb.cls < C) row[gb.cls] = (TileT)1;
            for (int k = 0; k < 4; ++k) box[k] = gb.lab[k];
        } else {
            row[p.background_id] = (TileT)1;
        }
        if (neutral) row[p.background_id] = (TileT)0;
        if (match_gt) {
            int code = gt_idx;
            if (gt_idx < 0) code = neutral ? -2 : -1;
            match_gt[(size_t)b * p.N + n] = code;
        }
        double t[4];
        if (p.coords == SSDHIP_CENTROIDS) {                                  // :396-400
            t[0] = (box[0] - a[0]) / (a[2] * var[0]);
            t[1] = (box[1] - a[1]) / (a[3] * var[1]);
            t[2] = log(box[2] / a[2]) / var[2];
            t[3] = log(box[3] / a[3]) / var[3];
        } else if (p.coords == SSDHIP_CORNERS) {                             // :401-405
            const double w = a[2] - a[0], h = a[3] - a[1];
            t[0] = ((box[0] - a[0]) / w) / var[0];
            t[1] = ((box[1] - a[1]) / h) / var[1];
            t[2] = ((box[2] - a[2]) / w) / var[2];
            t[3] = ((box[3] - a[3]) / h) / var[3];
        } else {                                                             // minmax :406-410
            const double w = a[1] - a[0], h = a[3] - a[2];
            t[0] = ((box[0] - a[0]) / w) / var[0];
            t[1] = ((box[1] - a[1]) / w) / var[1];
            t[2] = ((box[2] - a[2]) / h) / var[2];
            t[3] = ((box[3] - a[3]) / h) / var[3];
        }
        for (int k = 0; k < 4; ++k) { row[C + k] = (TileT)t[k]; row[C + 4 + k] = (TileT)a[k]; row[C + 8 + k] = (TileT)var[k]; }
    }
    __syncthreads();
    const int total = na * L;
    if (sizeof(TileT) == 8) {
        const double* dt = reinterpret_cast<const double*>(tile);
        if (y64) for (int i = tid; i < total; i += TA) y64[base + i] = dt[i];
        if (y32) for (int i = tid; i < total; i += TA) y32[base + i] = (float)dt[i];
    } else if (y32) {
        const float* ft = reinterpret_cast<const float*>(tile);
        float* dst = y32 + base;
        const int head = min(total, (4 - phase) & 3);
        if (tid < head) dst[tid] = ft[tid];
        const int nvec = (total - head) >> 2;
        const float4* vsrc = reinterpret_cast<const float4*>(ft + head);
        float4* vdst = reinterpret_cast<float4*>(dst + head);
        for (int i = tid; i < nvec; i += TA) vdst[i] = vsrc[i];
        const int done = head + (nvec << 2);
        if (tid < total - done) dst[done + tid] = ft[done + tid];
    }
}

struct EncodeWs {
    size_t part_val, part_col, matches, total;
};

static inline size_t enc_align(size_t v) { return (v + 255) / 256 * 256; }

static EncodeWs encode_ws_layout(int B, int N, int G_total) {
    const size_t G = G_total > 0 ? (size_t)G_total : 1;
    const size_t tiles = (size_t)(N + ENC_THREADS - 1) / ENC_THREADS;
    EncodeWs w;
    size_t o = 0;
    w.part_val = o; o = enc_align(o + G * tiles * (ENC_THREADS / 64) * sizeof(double));      // per-wave partial row maxima
    w.part_col = o; o = enc_align(o + G * tiles * (ENC_THREADS / 64) * sizeof(int));
    w.matches = o;  o = enc_align(o + G * sizeof(int));
    w.total = o;
    return w;
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" size_t ssdhip_encode_workspace_bytes(int B, int N, int C, int G_total) {
    if (B <= 0 || N <= 0 || C < 1 || G_total < 0) return 0;
    return encode_ws_layout(B, N, G_total).total;
}

extern "C" int ssdhip_encode(const double* anchors, const double* variances, const double* gt, const int* gt_offsets,
                             int G_total, int max_gt_per_image, int B, int N, int C, double img_height, double img_width,
                             int matching_type, double pos_iou_threshold, double neg_iou_limit,
                             int coords, int normalize_coords, int border_pixels, int background_id,
                             float* y_encoded_f32, double* y_encoded_f64, int* match_gt,
                             void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!anchors || !variances || !gt_offsets || B <= 0 || N <= 0 || C < 1 || G_total < 0) return SSDHIP_E_BADARG;
    if (G_total > 0 && !gt) return SSDHIP_E_BADARG;
    if (!y_encoded_f32 && !y_encoded_f64 && !match_gt) return SSDHIP_E_BADARG;
    if (coords < 0 || coords > 2 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (matching_type != 0 && matching_type != 1) return SSDHIP_E_BADARG;
    if (background_id < 0 || background_id >= C) return SSDHIP_E_BADARG;
    if (max_gt_per_image < 0 || max_gt_per_image > ENC_MAX_GT || max_gt_per_image > G_total + 0) return SSDHIP_E_BADARG;
    const EncodeWs lay = encode_ws_layout(B, N, G_total);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;

    EncodeParams p;
    p.B = B; p.N = N; p.C = C; p.L = C + 12;
    p.tiles = (N + ENC_THREADS - 1) / ENC_THREADS;
    p.max_gt = max_gt_per_image > 0 ? max_gt_per_image : 1;
    p.matching_multi = matching_type == 1;
    p.coords = coords; p.normalize = normalize_coords ? 1 : 0; p.border = border_pixels; p.background_id = background_id;
    p.img_h = img_height; p.img_w = img_width; p.pos_thr = pos_iou_threshold; p.neg_limit = neg_iou_limit;

    unsigned char* wsb = static_cast<unsigned char*>(ws);
    double* part_val = reinterpret_cast<double*>(wsb + lay.part_val);
    int* part_col = reinterpret_cast<int*>(wsb + lay.part_col);
    int* matches = reinterpret_cast<int*>(wsb + lay.matches);
    if (G_total > 0) {
        hipLaunchKernelGGL(rowmax_kernel, dim3(p.tiles, B), dim3(ENC_THREADS), (size_t)p.max_gt * sizeof(GtBox), stream, p, anchors, gt,
                           gt_offsets, part_val, part_col);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
        const size_t m_lds = (size_t)p.max_gt * (sizeof(GtBox) + sizeof(double) + 3 * sizeof(int)) + (size_t)((N + 31) / 32) * sizeof(u32) + 16;
        if (m_lds > 150 * 1024) return SSDHIP_E_BADARG;
        if (m_lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(match_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)m_lds) != hipSuccess)
            return SSDHIP_E_LAUNCH;
        hipLaunchKernelGGL(match_kernel, dim3(B), dim3(MATCH_THREADS), m_lds, stream, p, anchors, gt, gt_offsets, part_val, part_col, matches);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    // F: tile = anchors whose staged rows fit ~36 KiB of LDS next to the GT table (float32 rows unless the float64 copy is asked for)
    const bool want64 = y_encoded_f64 != nullptr;
    const size_t esz = want64 ? sizeof(double) : sizeof(float);
    int TA = 256;
    while (TA > 64 && (size_t)TA * p.L * esz > 36 * 1024) TA >>= 1;
    const size_t fin_lds = (((size_t)TA * p.L + 4) * esz + 15) / 16 * 16 + (size_t)p.max_gt * (sizeof(GtBox) + sizeof(int));
    if (fin_lds > 150 * 1024) return SSDHIP_E_BADARG;
    const void* fk = want64 ? reinterpret_cast<const void*>(finalize_kernel<double>) : reinterpret_cast<const void*>(finalize_kernel<float>);
    if (fin_lds > 48 * 1024 && hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    if (want64)
        hipLaunchKernelGGL(finalize_kernel<double>, dim3((N + TA - 1) / TA, B), dim3(TA), fin_lds, stream, p, anchors, variances, gt,
                           gt_offsets, matches, y_encoded_f32, y_encoded_f64, match_gt);
    else
        hipLaunchKernelGGL(finalize_kernel<float>, dim3((N + TA - 1) / TA, B), dim3(TA), fin_lds, stream, p, anchors, variances, gt,
                           gt_offsets, matches, y_encoded_f32, y_encoded_f64, match_gt);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    return SSDHIP_OK;
}
