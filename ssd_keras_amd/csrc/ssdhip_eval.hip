// ssdhip_eval.hip -- Evaluator.match_predictions on gfx950 (MI355X): SURVEY section 8f row 1.
//
// Replaces the per-class body of eval_utils/average_precision_evaluator.py:604-725: sort the class's predictions by
// descending confidence, and for each -- in that order -- find the same-image, same-class ground truth box of highest
// IoU (bounding_box_utils.iou, 'corners', element-wise, the evaluator's border_pixels); below the matching threshold it is a
// false positive; a neutral ('difficult') best match is skipped; otherwise it is a true positive iff no earlier
// prediction claimed that ground truth box, else a duplicate = false positive.  The reference walks up to ~184 k
// predictions per class in a Python loop.
//
// The loop is only sequential through "was this box claimed already", and a prediction's target box does not depend on
// the other predictions.  So:
//   V1 evm_target_kernel   one thread per prediction: best box + overlap (NumPy's mixed-dtype rules: ground truth float64,
//                          the prediction float32 as the reference stores it -- its area is evaluated in float32), and an
//                          atomicMax of the prediction's sort key [confidence bits | inverted index] on that box: the winner
//                          is exactly the first claimant in descending-confidence (stable) order;
//   sort                   rocPRIM radix sort of the 64-bit keys, descending == argsort(-confidence, kind='mergesort');
//   V2 evm_assign_kernel   in sorted order: true positive iff the prediction's key is its box's winning key;
//   scan                   rocPRIM inclusive sums -> cumulative true / false positives.
// rocPRIM is used for the two plain library primitives only (header-only, compiled into libssdhip.so).
#include <cstring>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

constexpr int EVM_THREADS = 256;

struct EvmParams {
    int P, n_images, G, border;
    double thr;
};

// key: larger = earlier in the reference's order (confidence descending; ties: lower original index first)
__device__ __forceinline__ u64 evm_key(float conf, int p) { return ((u64)float_key(conf) << 32) | (u64)(0xffffffffu - (u32)p); }

// All classes in one pass (round 6): the predictions lie class by class (slot 0 first), `pred_class[p]` is the slot.  One 64-bit key
// still orders everything: [127 - slot : 7 | confidence bits : 32 | inverted index : 25] sorted descending = slot ascending, inside a
// slot confidence descending, ties by the lower index -- each slot's stretch of the sorted array is that class's
// argsort(-confidence, kind='mergesort').  Up to 128 slots and 2^25 predictions per call.
constexpr int EVM_MULTI_IDX_BITS = 25;
__device__ __forceinline__ u64 evm_key_multi(float conf, int p, int slot) {
    return ((u64)(127u - (u32)slot) << 57) | ((u64)float_key(conf) << EVM_MULTI_IDX_BITS) | (u64)(((1u << EVM_MULTI_IDX_BITS) - 1u) - (u32)p);
}

__global__ __launch_bounds__(EVM_THREADS) void evm_target_kernel(EvmParams q, const float* __restrict__ pred,
                                                                 const int* __restrict__ pred_image, const int* __restrict__ pred_class,
                                                                 const double* __restrict__ gt,
                                                                 const int* __restrict__ gt_off, const unsigned char* __restrict__ neutral,
                                                                 u64* __restrict__ keys, int* __restrict__ idx, int* __restrict__ target,
                                                                 u64* __restrict__ winner) {
    const int p = blockIdx.x * EVM_THREADS + threadIdx.x;
    if (p >= q.P) return;
    const float* r = pred + (size_t)p * 5;
    const u64 key = pred_class ? evm_key_multi(r[0], p, pred_class[p]) : evm_key(r[0], p);
    keys[p] = key;
    idx[p] = p;
    const int img = pred_image[p];
    int t = -1;                                        // -1: false positive (no box of this class in the image / below threshold)
    if (img >= 0 && img < q.n_images) {
        const int g0 = gt_off[img], g1 = gt_off[img + 1];
        if (g1 > g0) {
            // boxes2 = the prediction, float32: its area is computed in float32 and then promoted (bounding_box_utils.py:371-378)
            const float d32 = q.border == SSDHIP_BORDER_INCLUDE ? 1.f : (q.border == SSDHIP_BORDER_EXCLUDE ? -1.f : 0.f);
            const double d = (double)d32;
            PxBox<double> pb;
            pb.x0 = (double)r[1]; pb.y0 = (double)r[2]; pb.x1 = (double)r[3]; pb.y1 = (double)r[4];
            pb.area = (double)box_area<float>(r[1], r[2], r[3], r[4], d32);
            double best = 0.0;
            int best_g = -1;
            for (int g = g0; g < g1; ++g) {
                PxBox<double> gb;
                gb.x0 = gt[(size_t)g * 4]; gb.y0 = gt[(size_t)g * 4 + 1]; gb.x1 = gt[(size_t)g * 4 + 2]; gb.y1 = gt[(size_t)g * 4 + 3];
                gb.area = box_area<double>(gb.x0, gb.y0, gb.x1, gb.y1, d);
                const double v = iou_px<double>(gb, pb);           // boxes1 = ground truth, boxes2 = prediction
                if (best_g < 0 || v > best) { best = v; best_g = g; }   // np.argmax: first maximum
            }
            if (!(best < q.thr)) {                                  // `if gt_match_overlap < matching_iou_threshold: false positive`
                if (neutral && neutral[best_g]) t = -2;             // neutral best match: neither true nor false positive
                else { t = best_g; atomicMax(&winner[best_g], key); }
            }
        }
    }
    target[p] = t;
}

__global__ __launch_bounds__(EVM_THREADS) void evm_assign_kernel(int P, const u64* __restrict__ sorted_keys, const int* __restrict__ order,
                                                                 const int* __restrict__ target, const u64* __restrict__ winner,
                                                                 int* __restrict__ tp, int* __restrict__ fp) {
    const int s = blockIdx.x * EVM_THREADS + threadIdx.x;
    if (s >= P) return;
    const int t = target[order[s]];
    int is_tp = 0, is_fp = 0;
    if (t == -1) is_fp = 1;
    else if (t >= 0) { if (winner[t] == sorted_keys[s]) is_tp = 1; else is_fp = 1; }
    tp[s] = is_tp;
    fp[s] = is_fp;
}

// cumulative counts of the all-classes call: the chip-wide running sums minus what the earlier slots contributed
__global__ __launch_bounds__(EVM_THREADS) void evm_segment_cum_kernel(int P, const u64* __restrict__ sorted_keys, const int* __restrict__ class_start,
                                                                      const int* __restrict__ run_tp, const int* __restrict__ run_fp,
                                                                      int* __restrict__ cum_tp, int* __restrict__ cum_fp) {
    const int s = blockIdx.x * EVM_THREADS + threadIdx.x;
    if (s >= P) return;
    const int slot = 127 - (int)(sorted_keys[s] >> 57);
    const int base = class_start[slot];
    cum_tp[s] = run_tp[s] - (base > 0 ? run_tp[base - 1] : 0);
    cum_fp[s] = run_fp[s] - (base > 0 ? run_fp[base - 1] : 0);
}

struct EvmWs {
    size_t keys, keys_sorted, idx, target, winner, run_tp, run_fp, tmp, tmp_bytes, total;
};

static inline size_t evm_align(size_t v) { return (v + 255) / 256 * 256; }

static EvmWs evm_layout(int P, int G) {
    const size_t p = P > 0 ? (size_t)P : 1, g = G > 0 ? (size_t)G : 1;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, (u64*)nullptr, (u64*)nullptr, (int*)nullptr, (int*)nullptr, p, 0, 64,
                                         (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, p, rocprim::plus<int>(), (hipStream_t)0);
    EvmWs w;
    size_t o = 0;
    w.keys = o;        o = evm_align(o + p * sizeof(u64));
    w.keys_sorted = o; o = evm_align(o + p * sizeof(u64));
    w.idx = o;         o = evm_align(o + p * sizeof(int));
    w.target = o;      o = evm_align(o + p * sizeof(int));
    w.winner = o;      o = evm_align(o + g * sizeof(u64));
    w.run_tp = o;      o = evm_align(o + p * sizeof(int));
    w.run_fp = o;      o = evm_align(o + p * sizeof(int));
    w.tmp = o;
    w.tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    const size_t floor_bytes = p * 32 + (1u << 20);    // never below a generous bound (the size query needs a device to answer)
    if (w.tmp_bytes < floor_bytes) w.tmp_bytes = floor_bytes;
    o = evm_align(o + w.tmp_bytes);
    w.total = o;
    return w;
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" size_t ssdhip_match_predictions_workspace_bytes(int P, int G) {
    if (P < 0 || G < 0) return 0;
    return evm_layout(P, G).total;
}

extern "C" int ssdhip_match_predictions(const float* pred, const int* pred_image, int P, const double* gt_boxes, const int* gt_offsets,
                                        const unsigned char* gt_neutral, int n_images, int G, double matching_iou_threshold,
                                        int border_pixels, int* order, int* true_pos, int* false_pos, int* cum_true_pos,
                                        int* cum_false_pos, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0 || G < 0 || n_images < 0 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (P == 0) return SSDHIP_OK;
    if (!pred || !pred_image || !gt_offsets || !order || !true_pos || !false_pos || !cum_true_pos || !cum_false_pos) return SSDHIP_E_BADARG;
    if (G > 0 && !gt_boxes) return SSDHIP_E_BADARG;
    const EvmWs lay = evm_layout(P, G);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;
    unsigned char* base = static_cast<unsigned char*>(ws);
    u64* keys = reinterpret_cast<u64*>(base + lay.keys);
    u64* keys_sorted = reinterpret_cast<u64*>(base + lay.keys_sorted);
    int* idx = reinterpret_cast<int*>(base + lay.idx);
    int* target = reinterpret_cast<int*>(base + lay.target);
    u64* winner = reinterpret_cast<u64*>(base + lay.winner);
    void* tmp = base + lay.tmp;
    size_t tmp_bytes = lay.tmp_bytes;

    if (zero_async(winner, (size_t)(G > 0 ? G : 1) * sizeof(u64), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
    EvmParams q;
    q.P = P; q.n_images = n_images; q.G = G; q.border = border_pixels; q.thr = matching_iou_threshold;
    const int blocks = (P + EVM_THREADS - 1) / EVM_THREADS;
    hipLaunchKernelGGL(evm_target_kernel, dim3(blocks), dim3(EVM_THREADS), 0, stream, q, pred, pred_image, (const int*)nullptr, gt_boxes,
                       gt_offsets, gt_neutral, keys, idx, target, winner);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    if (rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys, keys_sorted, idx, order, (size_t)P, 0, 64, stream) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(evm_assign_kernel, dim3(blocks), dim3(EVM_THREADS), 0, stream, P, keys_sorted, order, target, winner, true_pos,
                       false_pos);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    tmp_bytes = lay.tmp_bytes;
    if (rocprim::inclusive_scan(tmp, tmp_bytes, true_pos, cum_true_pos, (size_t)P, rocprim::plus<int>(), stream) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    tmp_bytes = lay.tmp_bytes;
    if (rocprim::inclusive_scan(tmp, tmp_bytes, false_pos, cum_false_pos, (size_t)P, rocprim::plus<int>(), stream) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    return SSDHIP_OK;
}

// Evaluator.match_predictions for ALL classes in one call (round 6; eval_utils/average_precision_evaluator.py:604-725 is the body of a
// loop over the classes, :601).  The predictions of every class concatenated slot by slot (slot = position in the caller's class list):
// pred [P,5] float32 [conf, xmin, ymin, xmax, ymax], pred_segment [P] = slot * n_images + image index, pred_class [P] = slot,
// class_start [n_slots + 1] (device): slot c owns [class_start[c], class_start[c+1]).  Ground truth as CSR over the n_slots * n_images
// segments: gt_offsets [n_segments + 1], gt_boxes [G,4] float64, gt_neutral [G] or NULL.  Outputs [P] in the concatenated order, each
// slot's stretch sorted by descending confidence: order (index into the concatenated input), true / false positive flags and their
// PER-SLOT running sums.  Same arithmetic per prediction as ssdhip_match_predictions; n_slots <= 128, P < 2^25.
extern "C" int ssdhip_match_predictions_multi(const float* pred, const int* pred_segment, const int* pred_class, int P,
                                              const double* gt_boxes, const int* gt_offsets, const unsigned char* gt_neutral, int n_segments,
                                              int G, const int* class_start, int n_slots, double matching_iou_threshold, int border_pixels,
                                              int* order, int* true_pos, int* false_pos, int* cum_true_pos, int* cum_false_pos, void* ws,
                                              size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0 || G < 0 || n_segments < 0 || n_slots < 1 || n_slots > 128 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (P >= (1 << EVM_MULTI_IDX_BITS)) return SSDHIP_E_BADARG;
    if (P == 0) return SSDHIP_OK;
    if (!pred || !pred_segment || !pred_class || !gt_offsets || !class_start || !order || !true_pos || !false_pos || !cum_true_pos ||
        !cum_false_pos)
        return SSDHIP_E_BADARG;
    if (G > 0 && !gt_boxes) return SSDHIP_E_BADARG;
    const EvmWs lay = evm_layout(P, G);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;
    unsigned char* base = static_cast<unsigned char*>(ws);
    u64* keys = reinterpret_cast<u64*>(base + lay.keys);
    u64* keys_sorted = reinterpret_cast<u64*>(base + lay.keys_sorted);
    int* idx = reinterpret_cast<int*>(base + lay.idx);
    int* target = reinterpret_cast<int*>(base + lay.target);
    u64* winner = reinterpret_cast<u64*>(base + lay.winner);
    int* run_tp = reinterpret_cast<int*>(base + lay.run_tp);
    int* run_fp = reinterpret_cast<int*>(base + lay.run_fp);
    void* tmp = base + lay.tmp;
    size_t tmp_bytes = lay.tmp_bytes;
    if (zero_async(winner, (size_t)(G > 0 ? G : 1) * sizeof(u64), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
    EvmParams q;
    q.P = P; q.n_images = n_segments; q.G = G; q.border = border_pixels; q.thr = matching_iou_threshold;
    const int blocks = (P + EVM_THREADS - 1) / EVM_THREADS;
    hipLaunchKernelGGL(evm_target_kernel, dim3(blocks), dim3(EVM_THREADS), 0, stream, q, pred, pred_segment, pred_class, gt_boxes, gt_offsets,
                       gt_neutral, keys, idx, target, winner);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    if (rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, keys, keys_sorted, idx, order, (size_t)P, 0, 64, stream) != hipSuccess)
        return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(evm_assign_kernel, dim3(blocks), dim3(EVM_THREADS), 0, stream, P, keys_sorted, order, target, winner, true_pos,
                       false_pos);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    tmp_bytes = lay.tmp_bytes;
    if (rocprim::inclusive_scan(tmp, tmp_bytes, true_pos, run_tp, (size_t)P, rocprim::plus<int>(), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
    tmp_bytes = lay.tmp_bytes;
    if (rocprim::inclusive_scan(tmp, tmp_bytes, false_pos, run_fp, (size_t)P, rocprim::plus<int>(), stream) != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(evm_segment_cum_kernel, dim3(blocks), dim3(EVM_THREADS), 0, stream, P, keys_sorted, class_start, run_tp, run_fp,
                       cum_true_pos, cum_false_pos);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
