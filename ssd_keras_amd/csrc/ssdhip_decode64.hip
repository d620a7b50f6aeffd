// ssdhip_decode64.hip -- the decoders' all-float64 flow: float64 predictions in, float64 arithmetic throughout.
//
// Replaces (reference pierluigiferrari/ssd_keras), for a float64 `y_pred`:
//   ssd_encoder_decoder/ssd_output_decoder.py  decode_detections :111-226 (decode :172-198 runs in the input's dtype, so
//                                              every product, exp and the centroids->corners conversion is float64),
//                                              decode_detections_fast :228-333, decode_detections_debug :342-467,
//                                              _greedy_nms* :77-109
//
// The model emits float32, so this is the path of hand-built tensors (y_encoded templates, float64 test fixtures), not the
// hot one: a float64 score does not fit the float32 path's one-word sortable key [score 32 | anchor 20], so the three
// stages are restated in their simplest exact form instead of templating K3-K5 (csrc/ssdhip_decode.hip):
//   D3 scan64_kernel   one thread per (image, anchor): decode the box (pixel corners + area, float64), append
//                      (score, anchor) to the (image, class) candidate lists (one atomic per candidate);
//   D4 nms64_kernel    one workgroup per (image, class): repeated block arg-max of (score desc, anchor asc) over the
//                      candidates still alive == the reference's np.argmax on an order-preserving pool, then the IoU
//                      test `similarities <= iou_threshold` with IEEE division; stops at `cap` survivors;
//   D5 topk64_kernel   one workgroup per image: class-major concatenation, top-k by rank counting
//                      (score desc, position asc), rows out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_decode64.h"

namespace ssdhip {

constexpr int D64_THREADS = 256;

struct D64Params {
    int B, N, C, L, G;
    int class_agnostic, semantics, coords, border, thr_inclusive, no_nms;
    double conf_thresh, iou_thresh, img_w, img_h;
    int top_k, cap_store, out_rows;
};

// ---- D3 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(D64_THREADS) void scan64_kernel(const double* __restrict__ y, D64Params p,
                                                             PxBox<double>* __restrict__ boxes, double* __restrict__ cand_score,
                                                             int* __restrict__ cand_idx, int* __restrict__ cand_count,
                                                             unsigned short* __restrict__ cls_out) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * D64_THREADS + threadIdx.x;
    if (a >= p.N) return;
    const double* row = y + ((size_t)b * p.N + a) * (size_t)p.L;
    const int C = p.C;
    const double o0 = row[C], o1 = row[C + 1], o2 = row[C + 2], o3 = row[C + 3];
    const double a_0 = row[C + 4], a_1 = row[C + 5], a_2 = row[C + 6], a_3 = row[C + 7];
    const double v0 = row[C + 8], v1 = row[C + 9], v2 = row[C + 10], v3 = row[C + 11];
    double x0, y0, x1, y1;
    if (p.coords == SSDHIP_CENTROIDS) {
        double cx, cy;
        if (p.semantics == SSDHIP_SEM_DEBUG) {                     // ssd_output_decoder.py:400: (d * a_wh) * var + a_c
            cx = (o0 * a_2) * v0 + a_0;
            cy = (o1 * a_3) * v1 + a_1;
        } else {                                                   // :177-178: d * (var * a_wh) + a_c
            cx = o0 * (v0 * a_2) + a_0;
            cy = o1 * (v1 * a_3) + a_1;
        }
        const double w = det_exp64(o2 * v2) * a_2;                 // :175-176
        const double h = det_exp64(o3 * v3) * a_3;
        const double hw = w / 2.0, hh = h / 2.0;
        x0 = cx - hw; y0 = cy - hh; x1 = cx + hw; y1 = cy + hh;    // bounding_box_utils.py:76-80
    } else if (p.coords == SSDHIP_MINMAX) {                        // anchors (xmin,xmax,ymin,ymax), :181-186
        const double aw = a_1 - a_0, ah = a_3 - a_2;
        x0 = (o0 * v0) * aw + a_0; x1 = (o1 * v1) * aw + a_1;
        y0 = (o2 * v2) * ah + a_2; y1 = (o3 * v3) * ah + a_3;
    } else {                                                       // corners, :187-191
        const double aw = a_2 - a_0, ah = a_3 - a_1;
        x0 = (o0 * v0) * aw + a_0; y0 = (o1 * v1) * ah + a_1;
        x1 = (o2 * v2) * aw + a_2; y1 = (o3 * v3) * ah + a_3;
    }
    PxBox<double> bx;
    bx.x0 = x0 * p.img_w; bx.y0 = y0 * p.img_h; bx.x1 = x1 * p.img_w; bx.y1 = y1 * p.img_h;    // :196-198 (x * 1.0 is exact)
    const double d = p.border == SSDHIP_BORDER_INCLUDE ? 1.0 : (p.border == SSDHIP_BORDER_EXCLUDE ? -1.0 : 0.0);
    bx.area = box_area<double>(bx.x0, bx.y0, bx.x1, bx.y1, d);
    boxes[(size_t)b * p.N + a] = bx;

    const double t = p.conf_thresh;
    if (p.class_agnostic) {                                        // first argmax / max over ALL classes, :291-293
        double best = row[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const double s = row[c];
            if (s > best) { best = s; bi = c; }
        }
        for (int c = 0; c < C; ++c) if (row[c] != row[c]) { best = row[c]; bi = c; break; }   // np.argmax: first NaN wins
        cls_out[(size_t)b * p.N + a] = (unsigned short)bi;
        if (bi != 0 && (p.thr_inclusive ? (best >= t) : (best > t))) {
            const int slot = atomicAdd(&cand_count[b], 1);
            cand_score[(size_t)b * p.N + slot] = best;
            cand_idx[(size_t)b * p.N + slot] = a;
        }
    } else {
        for (int c = 1; c < C; ++c) {
            const double s = row[c];
            if (s > t) {                                           // :209
                const size_t w = (size_t)b * p.G + (c - 1);
                const int slot = atomicAdd(&cand_count[w], 1);
                cand_score[w * p.N + slot] = s;
                cand_idx[w * p.N + slot] = a;
            }
        }
    }
}

// ---- D4 ------------------------------------------------------------------------------------------
struct Best64 {
    double s;
    int a, i;            // anchor index, position in the candidate list (i < 0: none)
};

__device__ __forceinline__ bool better64(double s, int a, const Best64& o) {
    return o.i < 0 || s > o.s || (s == o.s && a < o.a);            // np.argmax: the first maximum == the lowest anchor index
}

__device__ __forceinline__ Best64 block_best64(Best64 m, Best64* red) {
    for (int off = 32; off > 0; off >>= 1) {
        Best64 o;
        o.s = __shfl_down(m.s, off);
        o.a = __shfl_down(m.a, off);
        o.i = __shfl_down(m.i, off);
        if (o.i >= 0 && better64(o.s, o.a, m)) m = o;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                               // `red` may still be read from the previous round
    if (lane == 0) red[wave] = m;
    __syncthreads();
    Best64 r = red[0];
    for (int w = 1; w < D64_THREADS / 64; ++w) {
        const Best64 o = red[w];
        if (o.i >= 0 && better64(o.s, o.a, r)) r = o;
    }
    return r;
}

__global__ __launch_bounds__(D64_THREADS) void nms64_kernel(D64Params p, const PxBox<double>* __restrict__ boxes,
                                                            const double* __restrict__ cand_score, const int* __restrict__ cand_idx,
                                                            const int* __restrict__ cand_count, unsigned char* __restrict__ alive,
                                                            int* __restrict__ kept_idx, double* __restrict__ kept_score,
                                                            int* __restrict__ kept_count) {
    __shared__ Best64 red[D64_THREADS / 64];
    const int work = blockIdx.x, tid = threadIdx.x;
    const int b = work / p.G;
    const int n = cand_count[work];
    const double* sc = cand_score + (size_t)work * p.N;
    const int* ix = cand_idx + (size_t)work * p.N;
    unsigned char* al = alive + (size_t)work * p.N;
    const PxBox<double>* img_boxes = boxes + (size_t)b * p.N;
    const int cap_eff = min(p.cap_store, n);
    const double thr = p.iou_thresh;

    Best64 m;
    m.s = 0.0; m.a = 0; m.i = -1;
    for (int i = tid; i < n; i += D64_THREADS) {
        al[i] = 1;
        const double s = sc[i];
        const int a = ix[i];
        if (better64(s, a, m)) { m.s = s; m.a = a; m.i = i; }
    }
    Best64 best = block_best64(m, red);
    int K = 0;
    while (best.i >= 0 && K < cap_eff) {
        if (tid == 0) {
            kept_idx[(size_t)work * p.cap_store + K] = best.a;
            kept_score[(size_t)work * p.cap_store + K] = best.s;
        }
        ++K;
        if (K >= cap_eff) break;
        const PxBox<double> mx = img_boxes[best.a];
        m.s = 0.0; m.a = 0; m.i = -1;
        for (int i = tid; i < n; i += D64_THREADS) {
            if (!al[i]) continue;
            if (i == best.i) { al[i] = 0; continue; }
            const int a = ix[i];
            if (!p.no_nms) {
                const double v = iou_px<double>(img_boxes[a], mx);
                if (!(v <= thr)) { al[i] = 0; continue; }          // keep `similarities <= iou_threshold` (:91)
            }
            const double s = sc[i];
            if (better64(s, a, m)) { m.s = s; m.a = a; m.i = i; }
        }
        best = block_best64(m, red);
    }
    if (tid == 0) kept_count[work] = K;
}

// ---- D5 ------------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(D64_THREADS) void topk64_kernel(D64Params p, const PxBox<double>* __restrict__ boxes,
                                                             const int* __restrict__ kept_idx, const double* __restrict__ kept_score,
                                                             const int* __restrict__ kept_count, const unsigned short* __restrict__ cls_map,
                                                             double* __restrict__ flat_score, int* __restrict__ flat_pos,
                                                             OutT* __restrict__ out, int* __restrict__ out_count, int* __restrict__ out_idx) {
    extern __shared__ int offs[];                                  // G + 1 class-major offsets
    __shared__ int wave_tot[D64_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.G;
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < G; ++g) { offs[g] = run; run += kept_count[b * G + g]; }
        offs[G] = run;
    }
    __syncthreads();
    const int T = offs[G];
    int rows = p.top_k > 0 ? min(T, p.top_k) : T;
    rows = min(rows, p.out_rows);
    double* fs = flat_score + (size_t)b * G * p.cap_store;
    int* fp = flat_pos + (size_t)b * G * p.cap_store;
    for (int g = 0; g < G; ++g) {
        const int kc = offs[g + 1] - offs[g];
        for (int r = tid; r < kc; r += D64_THREADS) {
            fs[offs[g] + r] = kept_score[((size_t)b * G + g) * p.cap_store + r];
            fp[offs[g] + r] = g * p.cap_store + r;
        }
    }
    __syncthreads();
    OutT* img_out = out + (size_t)b * p.out_rows * 6;
    int* img_idx = out_idx ? out_idx + (size_t)b * p.out_rows : nullptr;
    int base = 0;
    for (int e0 = 0; e0 < T && rows > 0; e0 += D64_THREADS) {
        const int e = e0 + tid;
        bool sel = false;
        if (e < T) {
            sel = true;
            if (T > rows) {                                        // rank by (score desc, class-major position asc)
                const double s = fs[e];
                int rank = 0;
                for (int j = 0; j < T; ++j) {
                    const double o = fs[j];
                    rank += (int)(o > s || (o == s && j < e));
                }
                sel = rank < rows;
            }
        }
        const u64 mk = __ballot(sel);
        if (lane == 0) wave_tot[wave] = __popcll(mk);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (sel) {
            const int row = off + __popcll(mk & lanemask_lt());
            const int pos = fp[e];
            const int g = pos / p.cap_store;
            const int a = kept_idx[(size_t)b * G * p.cap_store + pos];
            const PxBox<double> bx = boxes[(size_t)b * p.N + a];
            OutT* r = img_out + (size_t)row * 6;
            r[0] = (OutT)(p.class_agnostic ? (int)cls_map[(size_t)b * p.N + a] : g + 1);
            r[1] = (OutT)fs[e];
            r[2] = (OutT)bx.x0; r[3] = (OutT)bx.y0; r[4] = (OutT)bx.x1; r[5] = (OutT)bx.y1;
            if (img_idx) img_idx[row] = a;
        }
        for (int w = 0; w < D64_THREADS / 64; ++w) base += wave_tot[w];
        __syncthreads();
    }
    for (int i = rows * 6 + tid; i < p.out_rows * 6; i += D64_THREADS) img_out[i] = (OutT)0;
    if (img_idx) for (int i = rows + tid; i < p.out_rows; i += D64_THREADS) img_idx[i] = -1;
    if (tid == 0) out_count[b] = rows;
}

// ---- host ----------------------------------------------------------------------------------------
struct D64Ws {
    size_t boxes, cand_count, kept_count, cls, cand_score, cand_idx, alive, kept_idx, kept_score, flat_score, flat_pos, total;
};

static inline size_t d64_align(size_t v) { return (v + 255) / 256 * 256; }

static int d64_cap_store(int N, int top_k, int nms_cap) {
    int cap = N;
    if (nms_cap > 0 && nms_cap < cap) cap = nms_cap;
    if (top_k > 0 && top_k < cap) cap = top_k;       // members of the global top-k are within a class's first top_k survivors
    return cap < 1 ? 1 : cap;
}

static D64Ws d64_layout(int B, int N, int C, int top_k, int nms_cap, int class_agnostic) {
    const size_t G = class_agnostic ? 1 : (size_t)(C - 1);
    const size_t cap = (size_t)d64_cap_store(N, top_k, nms_cap);
    D64Ws w;
    size_t o = 0;
    w.boxes = o;      o = d64_align(o + (size_t)B * N * sizeof(PxBox<double>));
    w.cand_count = o; o = d64_align(o + (size_t)B * G * sizeof(int));
    w.kept_count = o; o = d64_align(o + (size_t)B * G * sizeof(int));
    w.cls = o;        o = d64_align(o + (size_t)B * N * sizeof(unsigned short));
    w.cand_score = o; o = d64_align(o + (size_t)B * G * N * sizeof(double));
    w.cand_idx = o;   o = d64_align(o + (size_t)B * G * N * sizeof(int));
    w.alive = o;      o = d64_align(o + (size_t)B * G * N);
    w.kept_idx = o;   o = d64_align(o + (size_t)B * G * cap * sizeof(int));
    w.kept_score = o; o = d64_align(o + (size_t)B * G * cap * sizeof(double));
    w.flat_score = o; o = d64_align(o + (size_t)B * G * cap * sizeof(double));
    w.flat_pos = o;   o = d64_align(o + (size_t)B * G * cap * sizeof(int));
    w.total = o;
    return w;
}

size_t decode64_workspace_bytes(int B, int N, int C, int top_k, int nms_cap, int class_agnostic) {
    return d64_layout(B, N, C, top_k, nms_cap, class_agnostic).total;
}

int decode64_run(int stages, const double* y_pred, int B, int N, int C, double conf_thresh, double iou_thresh, int top_k,
                 int nms_cap, int class_agnostic, int semantics, int coords, int normalize_coords, double img_height,
                 double img_width, int border_pixels, void* out, int out_dtype, int out_rows, int* out_count,
                 int* out_anchor_idx, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (semantics == SSDHIP_SEM_KERAS) return SSDHIP_E_BADARG;      // the Keras layers are float32 graphs
    if (C > 65535) return SSDHIP_E_BADARG;
    const D64Ws lay = d64_layout(B, N, C, top_k, nms_cap, class_agnostic);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;
    D64Params p;
    p.B = B; p.N = N; p.C = C; p.L = C + 12; p.G = class_agnostic ? 1 : C - 1;
    p.class_agnostic = class_agnostic ? 1 : 0;
    p.semantics = semantics; p.coords = coords; p.border = border_pixels;
    p.thr_inclusive = class_agnostic ? 1 : 0;                       // ssd_output_decoder.py:325 (>=) vs :209 (>)
    p.no_nms = !(iou_thresh < __builtin_inf()) && iou_thresh == iou_thresh ? 1 : 0;
    p.conf_thresh = conf_thresh; p.iou_thresh = iou_thresh;
    p.img_w = normalize_coords ? img_width : 1.0;
    p.img_h = normalize_coords ? img_height : 1.0;
    p.top_k = top_k; p.cap_store = d64_cap_store(N, top_k, nms_cap); p.out_rows = out_rows;

    unsigned char* base = static_cast<unsigned char*>(ws);
    PxBox<double>* boxes = reinterpret_cast<PxBox<double>*>(base + lay.boxes);
    int* cand_count = reinterpret_cast<int*>(base + lay.cand_count);
    int* kept_count = reinterpret_cast<int*>(base + lay.kept_count);
    unsigned short* cls_map = reinterpret_cast<unsigned short*>(base + lay.cls);
    double* cand_score = reinterpret_cast<double*>(base + lay.cand_score);
    int* cand_idx = reinterpret_cast<int*>(base + lay.cand_idx);
    unsigned char* alive = base + lay.alive;
    int* kept_idx = reinterpret_cast<int*>(base + lay.kept_idx);
    double* kept_score = reinterpret_cast<double*>(base + lay.kept_score);
    double* flat_score = reinterpret_cast<double*>(base + lay.flat_score);
    int* flat_pos = reinterpret_cast<int*>(base + lay.flat_pos);

    if (stages & 1) {
        if (zero_async(cand_count, (size_t)B * p.G * sizeof(int), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
        hipLaunchKernelGGL(scan64_kernel, dim3((N + D64_THREADS - 1) / D64_THREADS, B), dim3(D64_THREADS), 0, stream, y_pred, p, boxes,
                           cand_score, cand_idx, cand_count, cls_map);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    if (stages & 2) {
        hipLaunchKernelGGL(nms64_kernel, dim3(B * p.G), dim3(D64_THREADS), 0, stream, p, boxes, cand_score, cand_idx, cand_count, alive,
                           kept_idx, kept_score, kept_count);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    if (stages & 4) {
        const size_t lds = (size_t)(p.G + 1) * sizeof(int);
        if (out_dtype == SSDHIP_F32)
            hipLaunchKernelGGL(topk64_kernel<float>, dim3(B), dim3(D64_THREADS), lds, stream, p, boxes, kept_idx, kept_score, kept_count,
                               cls_map, flat_score, flat_pos, static_cast<float*>(out), out_count, out_anchor_idx);
        else
            hipLaunchKernelGGL(topk64_kernel<double>, dim3(B), dim3(D64_THREADS), lds, stream, p, boxes, kept_idx, kept_score, kept_count,
                               cls_map, flat_score, flat_pos, static_cast<double*>(out), out_count, out_anchor_idx);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    return SSDHIP_OK;
}

}  // namespace ssdhip
