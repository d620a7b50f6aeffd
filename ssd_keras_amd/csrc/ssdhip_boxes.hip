// ssdhip_boxes.hip -- the reference's public box utilities as stand-alone gfx950 kernels.
//
// Replaces, one C-ABI entry per callable (include/ssdhip.h):
//   convert_coordinates / convert_coordinates2   bounding_box_utils/bounding_box_utils.py:24-87, 89-117
//   intersection_area / intersection_area_ / iou bounding_box_utils/bounding_box_utils.py:119-383
//   match_bipartite_greedy / match_multi         ssd_encoder_decoder/matching_utils.py:22-116
//   greedy_nms / _greedy_nms / _greedy_nms2      ssd_encoder_decoder/ssd_output_decoder.py:27-109
// (inside the encoder / decoder hot path the same arithmetic is fused into E1-E3 / K3-K5; these entries serve callers
// that use the utilities on their own: the Evaluator, the augmentation box filters, user code).
//
// Dtype rules are NumPy's, because they decide where the reference rounds:
//   * convert_coordinates evaluates its right-hand sides in the INPUT's dtype and stores into a float64 copy;
//   * iou / intersection_area convert 'centroids' inputs that way first (-> float64 everywhere); for 'corners' / 'minmax'
//     the box areas are computed in each operand's own dtype, everything else in the promoted dtype, which is also the
//     result dtype (float32 only when both inputs are float32).
// Compiled with -ffp-contract=off: one IEEE rounding per written operation, same order as the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

constexpr int BX_THREADS = 256;

__device__ __forceinline__ double bx_border(int border) {
    return border == SSDHIP_BORDER_INCLUDE ? 1.0 : (border == SSDHIP_BORDER_EXCLUDE ? -1.0 : 0.0);
}

// ======================================================================================
// convert_coordinates: out = float64 copy of in [rows, L]; columns start..start+3 converted
// ======================================================================================
template <typename T>
__device__ __forceinline__ T cc_value(int conv, int k, T a, T b, T c, T e, T d) {
    const T two = (T)2;
    switch (conv) {
        case SSDHIP_MINMAX2CENTROIDS:    // (xmin,xmax,ymin,ymax) -> (cx,cy,w,h)   :61-65
            return k == 0 ? (a + b) / two : k == 1 ? (c + e) / two : k == 2 ? (b - a) + d : (e - c) + d;
        case SSDHIP_CENTROIDS2MINMAX:    // :66-70
            return k == 0 ? a - c / two : k == 1 ? a + c / two : k == 2 ? b - e / two : b + e / two;
        case SSDHIP_CORNERS2CENTROIDS:   // (xmin,ymin,xmax,ymax) -> (cx,cy,w,h)   :71-75
            return k == 0 ? (a + c) / two : k == 1 ? (b + e) / two : k == 2 ? (c - a) + d : (e - b) + d;
        case SSDHIP_CENTROIDS2CORNERS:   // :76-80
            return k == 0 ? a - c / two : k == 1 ? b - e / two : k == 2 ? a + c / two : b + e / two;
        default:                         // minmax2corners == corners2minmax: swap the middle pair (:81-83)
            return k == 0 ? a : k == 1 ? c : k == 2 ? b : e;
    }
}

template <typename T>
__global__ __launch_bounds__(BX_THREADS) void convert_kernel(const T* __restrict__ in, double* __restrict__ out,
                                                             long long total, int L, int start, int conv, int border) {
    const T d = (T)bx_border(border);
    for (long long i = (long long)blockIdx.x * BX_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * BX_THREADS) {
        const int col = (int)(i % L);
        const int k = col - start;
        T v = in[i];
        if (k >= 0 && k < 4) {
            const T* r = in + (i - k);
            v = cc_value<T>(conv, k, r[0], r[1], r[2], r[3], d);
        }
        out[i] = (double)v;
    }
}

// ======================================================================================
// iou / intersection_area
// ======================================================================================
// One operand box prepared the way the reference prepares it: corner view in the result dtype R + its area.
template <typename T, typename R>
__device__ __forceinline__ PxBox<R> public_box(const T* __restrict__ p, int coords, int border) {
    const T v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    PxBox<R> r;
    if (coords == SSDHIP_CENTROIDS) {           // convert_coordinates(..., 'centroids2corners') in T, stored as float64 (:334-337)
        const T two = (T)2;
        r.x0 = (R)(v0 - v2 / two);
        r.y0 = (R)(v1 - v3 / two);
        r.x1 = (R)(v0 + v2 / two);
        r.y1 = (R)(v1 + v3 / two);
        r.area = box_area<R>(r.x0, r.y0, r.x1, r.y1, (R)bx_border(border));
    } else {
        T x0, y0, x1, y1;
        if (coords == SSDHIP_MINMAX) { x0 = v0; x1 = v1; y0 = v2; y1 = v3; }
        else { x0 = v0; y0 = v1; x1 = v2; y1 = v3; }
        r.area = (R)box_area<T>(x0, y0, x1, y1, (T)bx_border(border));     // areas in the operand's own dtype (:371-378)
        r.x0 = (R)x0; r.y0 = (R)y0; r.x1 = (R)x1; r.y1 = (R)y1;
    }
    return r;
}

// op 0: iou (intersection with d = 0, the :345 quirk); op 1: intersection_area (side lengths see d, :214-224)
template <typename R>
__device__ __forceinline__ R pair_value(const PxBox<R>& a, const PxBox<R>& b, int op, R d) {
    if (op == 0) return iou_px<R>(a, b);
    const R ix0 = np_maximum<R>(a.x0, b.x0), iy0 = np_maximum<R>(a.y0, b.y0);
    const R ix1 = np_minimum<R>(a.x1, b.x1), iy1 = np_minimum<R>(a.y1, b.y1);
    return clamp0<R>((ix1 - ix0) + d) * clamp0<R>((iy1 - iy0) + d);
}

// outer product: out[i, j], grid (ceil(n / 256), min(m, 65535))
template <typename T1, typename T2, typename R>
__global__ __launch_bounds__(BX_THREADS) void iou_outer_kernel(const T1* __restrict__ b1, int m, const T2* __restrict__ b2, int n,
                                                               int coords, int border, int op, R* __restrict__ out) {
    const int j = blockIdx.x * BX_THREADS + threadIdx.x;
    if (j >= n) return;
    const PxBox<R> q = public_box<T2, R>(b2 + (size_t)j * 4, coords, border);
    const R d = (R)bx_border(border);
    for (int i = blockIdx.y; i < m; i += gridDim.y) {
        const PxBox<R> p = public_box<T1, R>(b1 + (size_t)i * 4, coords, border);      // wave-uniform address: scalar loads
        out[(size_t)i * n + j] = pair_value<R>(p, q, op, d);
    }
}

// element-wise with NumPy broadcasting: out[i] for i < max(m, n); an operand with one row is repeated
template <typename T1, typename T2, typename R>
__global__ __launch_bounds__(BX_THREADS) void iou_elem_kernel(const T1* __restrict__ b1, int m, const T2* __restrict__ b2, int n,
                                                              int coords, int border, int op, R* __restrict__ out) {
    const int len = m > n ? m : n;
    const R d = (R)bx_border(border);
    for (int i = blockIdx.x * BX_THREADS + threadIdx.x; i < len; i += gridDim.x * BX_THREADS) {
        const PxBox<R> p = public_box<T1, R>(b1 + (size_t)(m == 1 ? 0 : i) * 4, coords, border);
        const PxBox<R> q = public_box<T2, R>(b2 + (size_t)(n == 1 ? 0 : i) * 4, coords, border);
        out[i] = pair_value<R>(p, q, op, d);
    }
}

template <typename T1, typename T2, typename R>
static int launch_iou(const void* b1, int m, const void* b2, int n, int coords, int mode, int border, int op, void* out,
                      hipStream_t stream) {
    if (mode == 0) {
        const dim3 grid((n + BX_THREADS - 1) / BX_THREADS, m < 65535 ? m : 65535);
        hipLaunchKernelGGL((iou_outer_kernel<T1, T2, R>), grid, dim3(BX_THREADS), 0, stream, (const T1*)b1, m, (const T2*)b2, n,
                           coords, border, op, (R*)out);
    } else {
        const int len = m > n ? m : n;
        int blocks = (len + BX_THREADS - 1) / BX_THREADS;
        if (blocks > 65535) blocks = 65535;
        hipLaunchKernelGGL((iou_elem_kernel<T1, T2, R>), dim3(blocks), dim3(BX_THREADS), 0, stream, (const T1*)b1, m,
                           (const T2*)b2, n, coords, border, op, (R*)out);
    }
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// ======================================================================================
// matching_utils
// ======================================================================================
// (value, index) order of np.argmax: larger value first, then the lower index
__device__ __forceinline__ bool bx_better(double v, int c, double bv, int bc) { return v > bv || (v == bv && c < bc); }

__device__ __forceinline__ void bx_wave_argmax(double& v, int& c) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oc = __shfl_xor(c, off);
        if (bx_better(ov, oc, v, c)) { v = ov; c = oc; }
    }
}

// block-wide argmax of (v, c); every thread receives the winner.  wv/wc: LDS, blockDim.x / 64 entries.
__device__ __forceinline__ void bx_block_argmax(double& v, int& c, double* wv, int* wc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    bx_wave_argmax(v, c);
    __syncthreads();                                   // previous readers of wv/wc are done
    if (lane == 0) { wv[wave] = v; wc[wave] = c; }
    __syncthreads();
    v = wv[0]; c = wc[0];
    for (int w = 1; w < nw; ++w) if (bx_better(wv[w], wc[w], v, c)) { v = wv[w]; c = wc[w]; }
}

constexpr int BM_THREADS = 1024;
constexpr int BM_MAX_ROWS = 4096;

// match_bipartite_greedy (:22-79) for an arbitrary weight matrix [m, n]: m rounds of
//   "largest entry of the matrix -> record (row, col) -> set that row and that column to 0".
// The matrix is never modified: removed rows / columns are LDS flags whose entries count as 0 exactly like the
// reference's in-place zeroing; each row caches its (max, first argmax) and is re-scanned only when the column it
// pointed to is taken while its maximum was positive.  One workgroup (the rounds are sequential).
__global__ __launch_bounds__(BM_THREADS) void bipartite_generic_kernel(const double* __restrict__ w, int m, int n,
                                                                       int* __restrict__ matches) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* rowval = reinterpret_cast<double*>(smem_raw);            // [m]
    int* rowcol = reinterpret_cast<int*>(rowval + m);                // [m]
    int* rowgone = rowcol + m;                                       // [m]
    u32* colgone = reinterpret_cast<u32*>(rowgone + m);              // [(n + 31) / 32]
    __shared__ double wv[BM_THREADS / 64];
    __shared__ int wc[BM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = BM_THREADS / 64;

    for (int i = tid; i < (n + 31) / 32; i += BM_THREADS) colgone[i] = 0;
    // initial row maxima: one wave per row
    for (int r = wave; r < m; r += nw) {
        double bv = -__builtin_inf();
        int bc = 0x7fffffff;
        for (int c = lane; c < n; c += 64) {
            const double v = w[(size_t)r * n + c];
            if (bx_better(v, c, bv, bc)) { bv = v; bc = c; }
        }
        bx_wave_argmax(bv, bc);
        if (lane == 0) { rowval[r] = bv; rowcol[r] = bc; rowgone[r] = 0; matches[r] = 0; }
    }
    __syncthreads();

    for (int round = 0; round < m; ++round) {
        // np.argmax over rows of each row's maximum (:63-68); a removed row is all zeros -> (0, column 0)
        double bv = -__builtin_inf();
        int br = 0x7fffffff;
        for (int r = tid; r < m; r += BM_THREADS) {
            const double v = rowgone[r] ? 0.0 : rowval[r];
            if (bx_better(v, r, bv, br)) { bv = v; br = r; }
        }
        bx_block_argmax(bv, br, wv, wc);
        const int col = rowgone[br] ? 0 : rowcol[br];
        __syncthreads();
        if (tid == 0) {
            matches[br] = col;
            rowgone[br] = 1;
            colgone[col >> 5] |= 1u << (col & 31);
        }
        __syncthreads();
        // column `col` now reads 0 in every row
        for (int r = 0; r < m; ++r) {
            if (rowgone[r]) continue;                                    // uniform across the block
            const double v = rowval[r];
            const int c = rowcol[r];
            if (c == col && v > 0.0) {                                   // lost its maximum: re-scan with removed columns = 0
                double nv = -__builtin_inf();
                int nc = 0x7fffffff;
                for (int k = tid; k < n; k += BM_THREADS) {
                    const double x = ((colgone[k >> 5] >> (k & 31)) & 1u) ? 0.0 : w[(size_t)r * n + k];
                    if (bx_better(x, k, nv, nc)) { nv = x; nc = k; }
                }
                bx_block_argmax(nv, nc, wv, wc);
                if (tid == 0) { rowval[r] = nv; rowcol[r] = nc; }
            } else if (bx_better(0.0, col, v, c)) {                      // the new zero beats a non-positive maximum
                if (tid == 0) { rowval[r] = 0.0; rowcol[r] = col; }
            }
        }
        __syncthreads();
    }
}

// match_multi (:81-116), pass 1: per column the first argmax over rows and whether it reaches the threshold
__global__ __launch_bounds__(BX_THREADS) void multi_flag_kernel(const double* __restrict__ w, int m, int n, double threshold,
                                                                int* __restrict__ best_row, int* __restrict__ tile_count) {
    const int c = blockIdx.x * BX_THREADS + threadIdx.x;
    int flag = 0;
    if (c < n) {
        double bv = w[c];
        int br = 0;
        for (int r = 1; r < m; ++r) {
            const double v = w[(size_t)r * n + c];
            if (v > bv) { bv = v; br = r; }
        }
        flag = bv >= threshold;
        best_row[c] = flag ? br : -1;
    }
    const int cnt = __syncthreads_count(flag);
    if (threadIdx.x == 0) tile_count[blockIdx.x] = cnt;
}

// pass 2: ordered compaction (np.nonzero order = ascending column)
__global__ __launch_bounds__(BX_THREADS) void multi_compact_kernel(const int* __restrict__ best_row, const int* __restrict__ tile_count,
                                                                   int n, int* __restrict__ out_gt, int* __restrict__ out_col,
                                                                   int* __restrict__ out_count) {
    __shared__ int wsum[BX_THREADS / 64];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        int s = 0;
        for (int t = lane; t < (int)blockIdx.x; t += 64) s += tile_count[t];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) base_s = s;
    }
    const int c = blockIdx.x * BX_THREADS + tid;
    const int br = c < n ? best_row[c] : -1;
    const u64 mask = __ballot(br >= 0);
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int off = base_s;
    for (int w2 = 0; w2 < wave; ++w2) off += wsum[w2];
    if (br >= 0) {
        const int o = off + __popcll(mask & lanemask_lt());
        out_gt[o] = br;
        out_col[o] = c;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        int tot = base_s;
        for (int w2 = 0; w2 < BX_THREADS / 64; ++w2) tot += wsum[w2];
        *out_count = tot;
    }
}

// ======================================================================================
// greedy NMS over rows of an arbitrary table (:27-109): one workgroup per segment (batch item)
// ======================================================================================
struct NmsRowsParams {
    int row_len, score_col, box_col, coords, border, n_segments;
    double iou_threshold;
};

__global__ __launch_bounds__(BX_THREADS) void nms_rows_kernel(NmsRowsParams p, const double* __restrict__ rows,
                                                              const int* __restrict__ seg_off, PxBox<double>* __restrict__ box_ws,
                                                              double* __restrict__ score_ws, unsigned char* __restrict__ alive,
                                                              int* __restrict__ kept_idx, int* __restrict__ kept_count) {
    __shared__ double wv[BX_THREADS / 64];
    __shared__ int wc[BX_THREADS / 64];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int r0 = seg_off[seg], n = seg_off[seg + 1] - r0;
    for (int i = tid; i < n; i += BX_THREADS) {
        const double* row = rows + (size_t)(r0 + i) * p.row_len;
        box_ws[r0 + i] = public_box<double, double>(row + p.box_col, p.coords, p.border);
        score_ws[r0 + i] = row[p.score_col];
        alive[r0 + i] = 1;
    }
    __syncthreads();
    int kept = 0;
    // first maximum of the scores still in the pool (np.argmax on an order-preserving pool: lowest row wins ties)
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += BX_THREADS) {
        const double s = score_ws[r0 + i];
        if (bx_better(s, i, bv, bi)) { bv = s; bi = i; }
    }
    bx_block_argmax(bv, bi, wv, wc);
    while (bi != 0x7fffffff) {
        if (tid == 0) kept_idx[r0 + kept] = bi;
        ++kept;
        const PxBox<double> mx = box_ws[r0 + bi];
        const int cur = bi;
        bv = -__builtin_inf();
        bi = 0x7fffffff;
        for (int i = tid; i < n; i += BX_THREADS) {
            if (!alive[r0 + i]) continue;
            if (i == cur) { alive[r0 + i] = 0; continue; }
            const double v = iou_px<double>(box_ws[r0 + i], mx);
            if (!(v <= p.iou_threshold)) { alive[r0 + i] = 0; continue; }      // keep `similarities <= iou_threshold`
            const double s = score_ws[r0 + i];
            if (bx_better(s, i, bv, bi)) { bv = s; bi = i; }
        }
        // a thread without candidates holds (-inf, INT_MAX), which loses against every real (score, row)
        bx_block_argmax(bv, bi, wv, wc);
    }
    if (tid == 0) kept_count[seg] = kept;
}

// ======================================================================================
// BoxFilter (data_generator/object_detection_2d_image_boxes_validation_utils.py:147-232), batched: one thread per box
// ======================================================================================
struct BoxFilterParams {
    int check_overlap, check_min_area, check_degenerate, criterion, border;   // criterion 0 'center_point', 1 'iou', 2 'area'
    double min_area, lower, upper;
};

__global__ __launch_bounds__(BX_THREADS) void box_filter_kernel(BoxFilterParams q, const double* __restrict__ boxes,
                                                                const int* __restrict__ box_image, const double* __restrict__ image_hw,
                                                                int G, unsigned char* __restrict__ keep) {
    const int g = blockIdx.x * BX_THREADS + threadIdx.x;
    if (g >= G) return;
    const double xmin = boxes[(size_t)g * 4], ymin = boxes[(size_t)g * 4 + 1], xmax = boxes[(size_t)g * 4 + 2], ymax = boxes[(size_t)g * 4 + 3];
    const double H = image_hw[(size_t)box_image[g] * 2], W = image_hw[(size_t)box_image[g] * 2 + 1];
    bool ok = true;
    if (q.check_degenerate) ok = ok && (xmax > xmin) && (ymax > ymin);                          // :170-172
    if (q.check_min_area) ok = ok && ((xmax - xmin) * (ymax - ymin) >= q.min_area);             // :174-176
    if (q.check_overlap) {
        const double d = bx_border(q.border);
        if (q.criterion == 1) {                                                                 // 'iou' :187-192: iou(image, box)
            PxBox<double> im, bb;
            im.x0 = 0.0; im.y0 = 0.0; im.x1 = W; im.y1 = H;
            im.area = box_area<double>(im.x0, im.y0, im.x1, im.y1, d);
            bb.x0 = xmin; bb.y0 = ymin; bb.x1 = xmax; bb.y1 = ymax;
            bb.area = box_area<double>(xmin, ymin, xmax, ymax, d);
            const double v = iou_px<double>(im, bb);
            ok = ok && (v > q.lower) && (v <= q.upper);
        } else if (q.criterion == 2) {                                                          // 'area' :193-214
            const double box_area_ = ((xmax - xmin) + d) * ((ymax - ymin) + d);
            const double cy0 = np_minimum<double>(np_maximum<double>(ymin, 0.0), H - 1.0), cy1 = np_minimum<double>(np_maximum<double>(ymax, 0.0), H - 1.0);
            const double cx0 = np_minimum<double>(np_maximum<double>(xmin, 0.0), W - 1.0), cx1 = np_minimum<double>(np_maximum<double>(xmax, 0.0), W - 1.0);
            const double inter = ((cx1 - cx0) + d) * ((cy1 - cy0) + d);
            const bool lo = q.lower == 0.0 ? (inter > q.lower * box_area_) : (inter >= q.lower * box_area_);
            ok = ok && lo && (inter <= q.upper * box_area_);
        } else {                                                                                // 'center_point' :215-220
            const double cy = (ymin + ymax) / 2.0, cx = (xmin + xmax) / 2.0;
            ok = ok && (cy >= 0.0) && (cy <= H - 1.0) && (cx >= 0.0) && (cx <= W - 1.0);
        }
    }
    keep[g] = ok ? 1 : 0;
}

static inline size_t bx_align(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace ssdhip

using namespace ssdhip;

extern "C" int ssdhip_convert_coordinates(const void* in, int in_dtype, double* out, long long n_rows, int row_len,
                                          int start_index, int conversion, int border_pixels, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!in || !out || n_rows < 0 || row_len < 4 || start_index < 0 || start_index + 4 > row_len) return SSDHIP_E_BADARG;
    if (conversion < 0 || conversion > SSDHIP_SWAP_MINMAX_CORNERS || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (in_dtype != SSDHIP_F32 && in_dtype != SSDHIP_F64) return SSDHIP_E_BADARG;
    const long long total = n_rows * row_len;
    if (total == 0) return SSDHIP_OK;
    long long blocks = (total + BX_THREADS - 1) / BX_THREADS;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (in_dtype == SSDHIP_F32)
        hipLaunchKernelGGL(convert_kernel<float>, dim3((unsigned)blocks), dim3(BX_THREADS), 0, stream, (const float*)in, out, total,
                           row_len, start_index, conversion, border_pixels);
    else
        hipLaunchKernelGGL(convert_kernel<double>, dim3((unsigned)blocks), dim3(BX_THREADS), 0, stream, (const double*)in, out, total,
                           row_len, start_index, conversion, border_pixels);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_iou_result_dtype(int dtype1, int dtype2, int coords) {
    return (coords != SSDHIP_CENTROIDS && dtype1 == SSDHIP_F32 && dtype2 == SSDHIP_F32) ? SSDHIP_F32 : SSDHIP_F64;
}

extern "C" int ssdhip_box_overlap(int op, const void* boxes1, int dtype1, int m, const void* boxes2, int dtype2, int n,
                                  int coords, int mode, int border_pixels, void* out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!boxes1 || !boxes2 || !out || m < 0 || n < 0) return SSDHIP_E_BADARG;
    if ((op != 0 && op != 1) || (mode != 0 && mode != 1) || coords < 0 || coords > 2 || border_pixels < 0 || border_pixels > 2)
        return SSDHIP_E_BADARG;
    if ((dtype1 != SSDHIP_F32 && dtype1 != SSDHIP_F64) || (dtype2 != SSDHIP_F32 && dtype2 != SSDHIP_F64)) return SSDHIP_E_BADARG;
    if (mode == 1 && m != n && m != 1 && n != 1) return SSDHIP_E_BADARG;     // NumPy could not broadcast these either
    if (m == 0 || n == 0) return SSDHIP_OK;
    const bool f1 = dtype1 == SSDHIP_F32, f2 = dtype2 == SSDHIP_F32;
    const bool r32 = ssdhip_iou_result_dtype(dtype1, dtype2, coords) == SSDHIP_F32;
    if (r32) return launch_iou<float, float, float>(boxes1, m, boxes2, n, coords, mode, border_pixels, op, out, stream);
    if (f1 && f2) return launch_iou<float, float, double>(boxes1, m, boxes2, n, coords, mode, border_pixels, op, out, stream);
    if (f1) return launch_iou<float, double, double>(boxes1, m, boxes2, n, coords, mode, border_pixels, op, out, stream);
    if (f2) return launch_iou<double, float, double>(boxes1, m, boxes2, n, coords, mode, border_pixels, op, out, stream);
    return launch_iou<double, double, double>(boxes1, m, boxes2, n, coords, mode, border_pixels, op, out, stream);
}

extern "C" int ssdhip_match_bipartite_greedy(const double* weight_matrix, int m, int n, int* matches, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!weight_matrix || !matches || m < 0 || n < 0) return SSDHIP_E_BADARG;
    if (m == 0) return SSDHIP_OK;
    if (n == 0 || m > BM_MAX_ROWS) return SSDHIP_E_BADARG;
    const size_t lds = (size_t)m * (sizeof(double) + 2 * sizeof(int)) + (size_t)((n + 31) / 32) * sizeof(u32) + 16;
    if (lds > 150 * 1024) return SSDHIP_E_BADARG;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(bipartite_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(bipartite_generic_kernel, dim3(1), dim3(BM_THREADS), lds, stream, weight_matrix, m, n, matches);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" size_t ssdhip_match_multi_workspace_bytes(int m, int n) {
    if (m < 0 || n < 0) return 0;
    const size_t tiles = (size_t)(n + BX_THREADS - 1) / BX_THREADS;
    return bx_align((size_t)(n > 0 ? n : 1) * sizeof(int)) + bx_align((tiles > 0 ? tiles : 1) * sizeof(int));
}

extern "C" int ssdhip_match_multi(const double* weight_matrix, int m, int n, double threshold, int* gt_idx, int* anchor_idx,
                                  int* count, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!weight_matrix || !gt_idx || !anchor_idx || !count || m <= 0 || n < 0) return SSDHIP_E_BADARG;
    if (!ws || ws_bytes < ssdhip_match_multi_workspace_bytes(m, n)) return SSDHIP_E_WORKSPACE;
    if (n == 0) return zero_async(count, sizeof(int), stream) == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
    const int tiles = (n + BX_THREADS - 1) / BX_THREADS;
    int* best_row = static_cast<int*>(ws);
    int* tile_count = reinterpret_cast<int*>(static_cast<unsigned char*>(ws) + bx_align((size_t)n * sizeof(int)));
    hipLaunchKernelGGL(multi_flag_kernel, dim3(tiles), dim3(BX_THREADS), 0, stream, weight_matrix, m, n, threshold, best_row,
                       tile_count);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(multi_compact_kernel, dim3(tiles), dim3(BX_THREADS), 0, stream, best_row, tile_count, n, gt_idx, anchor_idx,
                       count);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" size_t ssdhip_greedy_nms_workspace_bytes(int n_rows_total) {
    const size_t n = n_rows_total > 0 ? (size_t)n_rows_total : 1;
    return bx_align(n * sizeof(PxBox<double>)) + bx_align(n * sizeof(double)) + bx_align(n);
}

extern "C" int ssdhip_greedy_nms(const double* rows, int n_rows_total, int row_len, int score_col, int box_col,
                                 const int* seg_offsets, int n_segments, double iou_threshold, int coords, int border_pixels,
                                 int* kept_idx, int* kept_count, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!seg_offsets || !kept_idx || !kept_count || n_segments < 0 || n_rows_total < 0) return SSDHIP_E_BADARG;
    if (row_len < 5 || score_col < 0 || score_col >= row_len || box_col < 0 || box_col + 4 > row_len) return SSDHIP_E_BADARG;
    if (coords < 0 || coords > 2 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (n_rows_total > 0 && !rows) return SSDHIP_E_BADARG;
    if (!ws || ws_bytes < ssdhip_greedy_nms_workspace_bytes(n_rows_total)) return SSDHIP_E_WORKSPACE;
    if (n_segments == 0) return SSDHIP_OK;
    const size_t n = n_rows_total > 0 ? (size_t)n_rows_total : 1;
    unsigned char* base = static_cast<unsigned char*>(ws);
    PxBox<double>* box_ws = reinterpret_cast<PxBox<double>*>(base);
    double* score_ws = reinterpret_cast<double*>(base + bx_align(n * sizeof(PxBox<double>)));
    unsigned char* alive = base + bx_align(n * sizeof(PxBox<double>)) + bx_align(n * sizeof(double));
    NmsRowsParams p;
    p.row_len = row_len; p.score_col = score_col; p.box_col = box_col; p.coords = coords; p.border = border_pixels;
    p.n_segments = n_segments; p.iou_threshold = iou_threshold;
    hipLaunchKernelGGL(nms_rows_kernel, dim3(n_segments), dim3(BX_THREADS), 0, stream, p, rows, seg_offsets, box_ws, score_ws, alive,
                       kept_idx, kept_count);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_box_filter(const double* boxes, const int* box_image, const double* image_hw, int G, int n_images,
                                 int check_overlap, int check_min_area, int check_degenerate, int overlap_criterion,
                                 double lower, double upper, double min_area, int border_pixels, unsigned char* keep, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (G < 0 || n_images < 0 || overlap_criterion < 0 || overlap_criterion > 2 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (G == 0) return SSDHIP_OK;
    if (!boxes || !box_image || !image_hw || !keep) return SSDHIP_E_BADARG;
    BoxFilterParams q;
    q.check_overlap = check_overlap ? 1 : 0; q.check_min_area = check_min_area ? 1 : 0; q.check_degenerate = check_degenerate ? 1 : 0;
    q.criterion = overlap_criterion; q.border = border_pixels; q.min_area = min_area; q.lower = lower; q.upper = upper;
    hipLaunchKernelGGL(box_filter_kernel, dim3((G + BX_THREADS - 1) / BX_THREADS), dim3(BX_THREADS), 0, stream, q, boxes, box_image,
                       image_hw, G, keep);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
