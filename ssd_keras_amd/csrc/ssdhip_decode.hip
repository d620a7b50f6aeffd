// ssdhip_decode.hip -- prediction decoding for SSD on gfx950 (MI355X).
//
// Replaces (reference pierluigiferrari/ssd_keras):
//   ssd_encoder_decoder/ssd_output_decoder.py  decode_detections :111-226, decode_detections_fast :228-333,
//                                              decode_detections_debug :342-467, _greedy_nms* :77-109
//   keras_layers/keras_layer_DecodeDetections.py :109-265, keras_layer_DecodeDetectionsFast.py :111-248
//
// Three kernels per call, all enqueued on the caller's stream:
//   K3 scan_kernel   grid (anchor tiles, B).  One coalesced pass over y_pred: each workgroup copies a
//                    contiguous tile of rows into LDS, decodes the boxes (float32 'corners', one float4
//                    per anchor) and appends every (class, anchor) pair over the confidence threshold to
//                    that (image, class)'s candidate list as one sortable 64-bit key
//                    [score bits | inverted anchor index].  Slots are reserved with one global atomic per
//                    (workgroup, class); wave ballots give the positions inside the workgroup.
//   K4 nms_kernel    one workgroup per (image, class) [or per image when class-agnostic], mapped so that
//                    all classes of an image run on the same XCD (shared L2 for its boxes).  Repeats:
//                    radix-select the next <= M best keys -> bitonic sort in LDS -> greedy NMS in batches
//                    of 64 (each wave tests the batch against a quarter of the kept list, 64x64 in-batch
//                    suppression masks by wave ballot, scalar resolve), until `cap` survivors or no
//                    candidates are left.  IoU in float64 with IEEE division, exactly the reference's
//                    operation order.
//   K5 topk_kernel   one workgroup per image: global top-k over the per-class survivors (radix select on
//                    [score | class-major position]), optional sort, zero padding, final rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_tile.h"
#include "ssdhip_heads.h"
#include "ssdhip_decode64.h"

// In-kernel phase timers, compiled only into the profiling build (tools/prof_build.sh, -DSSDHIP_PROFILE).
#ifdef SSDHIP_PROFILE
__device__ unsigned long long g_prof[64];
#define PROF_DECL long long _pt = clock64(); long long _pa[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(i) { const long long _t = clock64(); _pa[i] += _t - _pt; _pt = _t; }
#define PROF_FLUSH(base) if (threadIdx.x == 0) { for (int _i = 0; _i < 12; ++_i) if (_pa[_i]) atomicAdd(&g_prof[(base) + _i], (unsigned long long)_pa[_i]); }
extern "C" int ssdhip_profile_read(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH(base)
#endif

namespace ssdhip {

constexpr int IDX_BITS = 20;                 // anchor index field of a candidate key: N <= 2^20
constexpr u32 IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr int DIGIT_BITS = 13;               // radix-select digit / K5 histogram resolution
constexpr int NBINS = 1 << DIGIT_BITS;       // 8192 LDS counters = 32 KiB
constexpr int NMS_BIN_SHIFT = 14;            // K4 score histogram: 2^14 float32 ulps per bin ...
constexpr int NMS_NBINS = 4096;              // ... 4096 bins = 16 KiB (9 binades above the threshold)
#ifndef SSDHIP_MAX_CHUNK
#define SSDHIP_MAX_CHUNK 512
#endif
#ifndef SSDHIP_NMS_MINWAVES
#define SSDHIP_NMS_MINWAVES 3
#endif
constexpr int MAX_CHUNK = SSDHIP_MAX_CHUNK;   // candidates sorted + staged per round in K4
constexpr int KEPT_LDS = 256;                // survivors whose boxes are cached in LDS
constexpr int KC_KEYS = 9216;                // K4: candidate lists up to this long are read in one trip per pass (KC_KEYS / T loads per thread)
constexpr int TOPK_SORT_MAX = 4096;          // rows K5 can return sorted

struct DecodeParams {
    int B, N, C, L, G;          // L = C + 12, G = groups per image (C-1, or 1 when class-agnostic)
    int class_agnostic, semantics, coords, border;
    int thr_inclusive;          // '>=' instead of '>'
    float thr_eff;              // float32 threshold with the same outcome as the reference's compare (see host code)
    u32 thr_key;                // float_key(thr_eff): origin of the score histograms
    int iou_f32;                // NMS arithmetic in float32 (reference: float32 input + 'corners')
    int fast_ok;                // division-free IoU test usable (0 < iou_thresh < inf)
    int px_f32;                 // pixel boxes rounded to float32 before NMS (the Keras layer's float32 scaling)
    int no_nms;                 // iou_thresh == +inf: the reference skipped NMS (decode_detections_fast with a falsy threshold)
    float filter_kE;            // POL_NUMPY64: |R32| > filter_kE * S decides a pair in float32 (inf: never)
    double iou_thresh, img_w, img_h;   // img_w/img_h = 1 when !normalize_coords
    int top_k, cap, cap_store, out_rows, sorted;
    int no_dma;                 // scan_kernel: tile_copy_f32 instead of the LDS-DMA copy (SSDHIP_SCAN_DMA=0, A/B)
};

// monotone score-key -> histogram bin: 2^SHIFT float32 ulps per bin starting at the threshold, clamped to NB-1
template <int SHIFT, int NB>
__device__ __forceinline__ int bin_of(u32 skey, u32 thr_key) {
    const u32 v = skey >> SHIFT, base = thr_key >> SHIFT;
    if (v <= base) return 0;
    const u32 d = v - base;
    return d > (u32)(NB - 1) ? NB - 1 : (int)d;
}

// ======================================================================================
// K3
// ======================================================================================
template <int SEM>
__device__ __forceinline__ float decode_center(float off, float var, float a_wh, float a_c) {
    if (SEM == SSDHIP_SEM_KERAS) return (off * var) * a_wh + a_c;      // keras_layer_DecodeDetections.py:124-125
    if (SEM == SSDHIP_SEM_DEBUG) return (off * a_wh) * var + a_c;      // ssd_output_decoder.py:400
    return off * (var * a_wh) + a_c;                                   // ssd_output_decoder.py:177-178
}

// The body shared by scan_kernel (tile copied from y_pred) and scan_heads_kernel (tile built from the head outputs): `tile`
// holds rows [a0, a0 + na) of image b as [C+12] floats in LDS, one thread per row; wave_cnt: LDS, (blockDim.x / 64) * G ints.
__device__ __forceinline__ void scan_tile_body(const float* tile, int* wave_cnt, const DecodeParams& p, const int b, const int a0,
                                               const int na, float4* __restrict__ boxes, u64* __restrict__ cand,
                                               int* __restrict__ cand_count, unsigned short* __restrict__ cls_out) {
    const int TA = blockDim.x;
    const int L = p.L;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = TA >> 6;
    PROF_DECL
    PROF_MARK(0)

    const bool active = tid < na;
    const float* row = tile + (size_t)tid * L;
    const int C = p.C;

    // ---- decode this thread's box (float32, the reference's operation order) ----
    int fast_cls = 0;
    float fast_conf = 0.f;
    if (active) {
        const float o0 = row[C], o1 = row[C + 1], o2 = row[C + 2], o3 = row[C + 3];
        const float a_0 = row[C + 4], a_1 = row[C + 5], a_2 = row[C + 6], a_3 = row[C + 7];
        const float v0 = row[C + 8], v1 = row[C + 9], v2 = row[C + 10], v3 = row[C + 11];
        float4 box;
        if (p.coords == SSDHIP_CENTROIDS) {
            float cx, cy;
            if (p.semantics == SSDHIP_SEM_KERAS) {
                cx = decode_center<SSDHIP_SEM_KERAS>(o0, v0, a_2, a_0);
                cy = decode_center<SSDHIP_SEM_KERAS>(o1, v1, a_3, a_1);
            } else if (p.semantics == SSDHIP_SEM_DEBUG) {
                cx = decode_center<SSDHIP_SEM_DEBUG>(o0, v0, a_2, a_0);
                cy = decode_center<SSDHIP_SEM_DEBUG>(o1, v1, a_3, a_1);
            } else {
                cx = decode_center<SSDHIP_SEM_NUMPY>(o0, v0, a_2, a_0);
                cy = decode_center<SSDHIP_SEM_NUMPY>(o1, v1, a_3, a_1);
            }
            const float w = det_expf(o2 * v2) * a_2;
            const float h = det_expf(o3 * v3) * a_3;
            const float hw = w / 2.0f, hh = h / 2.0f;          // == 0.5f*w exactly
            box = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);   // bounding_box_utils.py:76-80
        } else if (p.coords == SSDHIP_MINMAX) {                    // anchors (xmin,xmax,ymin,ymax), :181-186
            const float aw = a_1 - a_0, ah = a_3 - a_2;
            const float t0 = (o0 * v0) * aw + a_0, t1 = (o1 * v1) * aw + a_1;
            const float t2 = (o2 * v2) * ah + a_2, t3 = (o3 * v3) * ah + a_3;
            box = make_float4(t0, t2, t1, t3);
        } else {                                                   // corners, :187-191
            const float aw = a_2 - a_0, ah = a_3 - a_1;
            box = make_float4((o0 * v0) * aw + a_0, (o1 * v1) * ah + a_1, (o2 * v2) * aw + a_2, (o3 * v3) * ah + a_3);
        }
        boxes[(size_t)b * p.N + a0 + tid] = box;
        if (p.class_agnostic) {                                    // first argmax / max over ALL classes, :291-293
            float best = row[0];
            int bi = 0;
            for (int c = 1; c < C; ++c) {
                const float s = row[c];
                if (s > best) { best = s; bi = c; }
            }
            for (int c = 0; c < C; ++c) if (row[c] != row[c]) { best = row[c]; bi = c; break; }   // np.argmax: first NaN wins
            fast_cls = bi;
            fast_conf = best;
            cls_out[(size_t)b * p.N + a0 + tid] = (unsigned short)bi;
        }
    }
    PROF_MARK(1)

    // ---- pass 1: candidates per (wave, group).  One LDS read + one v_cmp (its SGPR result IS the ballot) + one
    //      s_bcnt1 per class; the count is parked in lane j of a VGPR, one LDS store per 64 classes ----
    const int G = p.G;
    const float t = p.thr_eff;
    const bool incl = p.thr_inclusive != 0;
    const bool fast_pred = active && (fast_cls != 0) && (incl ? (fast_conf >= t) : (fast_conf > t));
    for (int gb = 0; gb < G; gb += 64) {
        const int ng = min(64, G - gb);
        int mycnt = 0;
        if (p.class_agnostic) {
            mycnt = __popcll(__ballot(fast_pred));
        } else {
            const float* sc = row + gb + 1;
            for (int j0 = 0; j0 < ng; j0 += 8) {                  // eight scores per step: their LDS reads are in flight together
                float s8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) s8[u] = sc[min(j0 + u, ng - 1)];      // clamped index: branch-free, masked below
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool pr = active && (j0 + u < ng) && (incl ? (s8[u] >= t) : (s8[u] > t));
                    const int cnt = __popcll(__ballot(pr));
                    mycnt = lane == j0 + u ? cnt : mycnt;
                }
            }
        }
        if (lane < ng) wave_cnt[wave * G + gb + lane] = mycnt;
    }
    __syncthreads();
    PROF_MARK(2)
    // ---- one global atomic per (workgroup, group) reserves the slots ----
    for (int g = tid; g < G; g += TA) {
        int tot = 0;
        for (int w = 0; w < nwaves; ++w) tot += wave_cnt[w * G + g];
        int base = 0;
        if (tot) base = atomicAdd(&cand_count[b * G + g], tot);
        for (int w = 0; w < nwaves; ++w) {
            const int c = wave_cnt[w * G + g];
            wave_cnt[w * G + g] = base;
            base += c;
        }
    }
    __syncthreads();
    PROF_MARK(3)
    // ---- pass 2: write the keys (slot = wave base of the class + rank of the lane among the class's candidates) ----
    const u32 inv_idx = IDX_MASK - (u32)(a0 + tid);
    for (int gb = 0; gb < G; gb += 64) {
        const int ng = min(64, G - gb);
        const int mybase = lane < ng ? wave_cnt[wave * G + gb + lane] : 0;
        const float* sc = row + gb + 1;
        for (int j0 = 0; j0 < ng; j0 += 8) {
            float s8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] = p.class_agnostic ? fast_conf : sc[min(j0 + u, ng - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                const float s = s8[u];
                const bool pr = (j < ng) && (p.class_agnostic ? fast_pred : (active && (incl ? (s >= t) : (s > t))));
                const u64 m = __ballot(pr);
                if (m == 0) continue;
                const int base_j = __builtin_amdgcn_readlane(mybase, j);
                if (pr) {
                    const int slot = base_j + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                    cand[((size_t)b * G + gb + j) * p.N + slot] = ((u64)float_key(s) << IDX_BITS) | inv_idx;
                }
            }
        }
    }
    PROF_MARK(4)
    PROF_FLUSH(16)
}

__global__ __launch_bounds__(256) void scan_kernel(const float* __restrict__ y, DecodeParams p,
                                                   float4* __restrict__ boxes, u64* __restrict__ cand,
                                                   int* __restrict__ cand_count, unsigned short* __restrict__ cls_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int TA = blockDim.x;                       // anchors per tile = threads per block
    const int b = blockIdx.y;
    const int a0 = blockIdx.x * TA;
    const int na = min(TA, p.N - a0);
    // ---- coalesced tile copy: rows [a0, a0+na) are one contiguous run of na*L floats.  Where the runs are 16-byte aligned they come in by
    //      LDS-DMA, every load of the workgroup in flight at once (round 5; tile_copy_f32 keeps one 16-byte load in flight per thread) ----
    const float* tile;
    const size_t total_bytes = (size_t)gridDim.y * p.N * (size_t)p.L * sizeof(float);
    if (!((uintptr_t)y & 15) && !(((size_t)p.N * p.L) & 3) && !((TA * p.L) & 3) && total_bytes < 0x7fffff00ull && !(TA & 63) && !p.no_dma) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
        const tile_i32x4 rs = tile_rsrc(y, (u32)total_bytes);
        const u32 off = (u32)(((size_t)b * p.N + a0) * (size_t)p.L * sizeof(float));
        const int nch = (na * p.L + 3) >> 2;                       // the last chunk may run into the next rows (or read zeros past the end)
        for (int q0 = wave * 64; q0 < nch; q0 += TA) {
            const int q = q0 + lane;
            tile_dma16(q < nch ? off + (u32)q * 16u : TILE_OOB, rs, lds0 + (u32)q0 * 16u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile = reinterpret_cast<const float*>(smem_raw);
    } else {
        tile = tile_copy_f32(reinterpret_cast<float*>(smem_raw), y + ((size_t)b * p.N + a0) * (size_t)p.L, na * p.L, threadIdx.x, TA);
    }
    int* wave_cnt = reinterpret_cast<int*>(smem_raw + (((size_t)TA * p.L + 4) * sizeof(float) + 15) / 16 * 16);
    __syncthreads();
    scan_tile_body(tile, wave_cnt, p, b, a0, na, boxes, cand, cand_count, cls_out);
}

// K3 for WIDE rows (C + 12 > 64 floats: SSD512 / COCO's 93).  scan_kernel stages whole rows: 64 rows x 93 floats = 24 KB per wave, six
// one-wave workgroups per CU -- too few waves to hide the copy -> ballots -> atomic -> store chain (143 us = 1.0 TB/s at batch 16,
// r02k).  Here a wave stages its 64 rows one 32-column WINDOW at a time (8.4 KB per wave, 20 waves per CU) and finishes a window's
// classes before the next one comes in: ballots and counts, ONE returning atomic per (wave, class) from the lane that holds the class,
// then the keys.  The windows are counted from the end of the row, so the last one always holds the 12 box columns.  Same candidate
// lists as scan_kernel up to the order of their entries (K4 sorts by key).  Per-class semantics only (the class-agnostic mode keeps
// scan_kernel).
constexpr int SW_WAVES = 4, SW_CW = 32, SW_PITCH = SW_CW + 1;
__global__ __launch_bounds__(SW_WAVES * 64) void scan_wide_kernel(const float* __restrict__ y, DecodeParams p, float4* __restrict__ boxes,
                                                                  u64* __restrict__ cand, int* __restrict__ cand_count) {
    __shared__ float win_all[SW_WAVES][64 * SW_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int a0 = ((int)blockIdx.x * SW_WAVES + wave) * 64;
    if (a0 >= p.N) return;                                   // whole wave (no workgroup barrier in this kernel)
    const int na = min(64, p.N - a0);
    float* win = win_all[wave];
    const int L = p.L, C = p.C, G = p.G;
    const float* src = y + ((size_t)b * p.N + a0) * (size_t)L;
    const float t = p.thr_eff;
    const bool incl = p.thr_inclusive != 0;
    const bool active = lane < na;
    const u32 inv_idx = IDX_MASK - (u32)(a0 + lane);
    const int nwin = (L + SW_CW - 1) / SW_CW;
    for (int wi = nwin - 1; wi >= 0; --wi) {
        // window wi covers columns [c_lo, c_hi): the last one is [L - 32, L), the one before [L - 64, L - 32), ... the first may be short
        const int c_hi = L - (nwin - 1 - wi) * SW_CW, c_lo = max(0, c_hi - SW_CW), cw = c_hi - c_lo;
        // copy: two rows of the window per instruction (each a contiguous run of cw floats)
        {
            const int c = lane & 31, rh = lane >> 5;
#pragma unroll
            for (int r0 = 0; r0 < 64; r0 += 16) {             // eight loads in flight per trip
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + 2 * u + rh;
                    v[u] = (r < na && c < cw) ? src[(size_t)r * L + c_lo + c] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) win[(r0 + 2 * u + rh) * SW_PITCH + c] = v[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const float* row = win + lane * SW_PITCH;
        if (wi == nwin - 1 && active) {                      // the box columns: the last 12 of the row
            const float* q = row + (cw - 12);
            const float o0 = q[0], o1 = q[1], o2 = q[2], o3 = q[3];
            const float a_0 = q[4], a_1 = q[5], a_2 = q[6], a_3 = q[7];
            const float v0 = q[8], v1 = q[9], v2 = q[10], v3 = q[11];
            float4 box;
            if (p.coords == SSDHIP_CENTROIDS) {
                float cx, cy;
                if (p.semantics == SSDHIP_SEM_KERAS) {
                    cx = decode_center<SSDHIP_SEM_KERAS>(o0, v0, a_2, a_0);
                    cy = decode_center<SSDHIP_SEM_KERAS>(o1, v1, a_3, a_1);
                } else if (p.semantics == SSDHIP_SEM_DEBUG) {
                    cx = decode_center<SSDHIP_SEM_DEBUG>(o0, v0, a_2, a_0);
                    cy = decode_center<SSDHIP_SEM_DEBUG>(o1, v1, a_3, a_1);
                } else {
                    cx = decode_center<SSDHIP_SEM_NUMPY>(o0, v0, a_2, a_0);
                    cy = decode_center<SSDHIP_SEM_NUMPY>(o1, v1, a_3, a_1);
                }
                const float w = det_expf(o2 * v2) * a_2;
                const float h = det_expf(o3 * v3) * a_3;
                const float hw = w / 2.0f, hh = h / 2.0f;
                box = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
            } else if (p.coords == SSDHIP_MINMAX) {
                const float aw = a_1 - a_0, ah = a_3 - a_2;
                const float t0 = (o0 * v0) * aw + a_0, t1 = (o1 * v1) * aw + a_1;
                const float t2 = (o2 * v2) * ah + a_2, t3 = (o3 * v3) * ah + a_3;
                box = make_float4(t0, t2, t1, t3);
            } else {
                const float aw = a_2 - a_0, ah = a_3 - a_1;
                box = make_float4((o0 * v0) * aw + a_0, (o1 * v1) * ah + a_1, (o2 * v2) * aw + a_2, (o3 * v3) * ah + a_3);
            }
            boxes[(size_t)b * p.N + a0 + lane] = box;
        }
        // classes of this window: columns [max(c_lo, 1), min(c_hi, C)) -> groups g = column - 1; lane j keeps the count of class j
        const int k_lo = max(c_lo, 1), k_hi = min(c_hi, C), nk = k_hi - k_lo;
        if (nk > 0) {
            int mycnt = 0;
            for (int j = 0; j < nk; ++j) {
                const float sc = row[k_lo - c_lo + j];
                const bool pr = active && (incl ? (sc >= t) : (sc > t));
                const int cnt = __popcll(__ballot(pr));
                mycnt = lane == j ? cnt : mycnt;
            }
            int mybase = 0;
            if (lane < nk && mycnt) mybase = atomicAdd(&cand_count[b * G + (k_lo - 1) + lane], mycnt);
            for (int j = 0; j < nk; ++j) {
                const float sc = row[k_lo - c_lo + j];
                const bool pr = active && (incl ? (sc >= t) : (sc > t));
                const u64 m = __ballot(pr);
                if (m == 0) continue;
                const int base_j = __builtin_amdgcn_readlane(mybase, j);
                if (pr) {
                    const int slot = base_j + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                    cand[((size_t)b * G + (k_lo - 1) + j) * p.N + slot] = ((u64)float_key(sc) << IDX_BITS) | inv_idx;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();                    // the window is rewritten by the next trip
    }
}

// K3 fused with the prediction assembly: the [C+12]-float rows are built in LDS straight from the predictor
// heads' bf16 outputs (bias, softmax, anchors: head_build_rows) and decoded / thresholded at once -- y_pred (36.9 MB at
// SSD300 / batch 32, a quarter of it the constant anchor + variance columns) is neither written nor read back.
__global__ __launch_bounds__(256) void scan_heads_kernel(HeadParams hp, const float* __restrict__ anchors_var, DecodeParams p,
                                                         float4* __restrict__ boxes, u64* __restrict__ cand,
                                                         int* __restrict__ cand_count, unsigned short* __restrict__ cls_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.y;
    int l, a0, na;
    head_tile_of(hp, (int)blockIdx.x, l, a0, na);
    float* rows = reinterpret_cast<float*>(smem_raw);
    hbf16_t* cl = reinterpret_cast<hbf16_t*>(smem_raw + (size_t)hp.TA * p.L * sizeof(float));
    int* wave_cnt = reinterpret_cast<int*>(smem_raw + (head_tile_lds(hp.TA, p.C) + 15) / 16 * 16);
    head_build_rows(hp, anchors_var, l, b, a0, na, rows, cl, threadIdx.x, blockDim.x);
    scan_tile_body(rows, wave_cnt, p, b, hp.anchor_off[l] + a0, na, boxes, cand, cand_count, cls_out);
}

// ======================================================================================
// block-wide helpers (256 threads)
// ======================================================================================
// Exact k-th largest (k >= 1) of {key : key < upper (if has_upper)}: radix select, one pass per DB-bit digit.
// Only the fallback when a histogram bin overflows the chunk.  `hist` = 2^DB LDS counters, `red` = 260 LDS ints.
template <int KEY_BITS, int DB, int T, typename KeyF>
__device__ u64 block_select_kth(KeyF key_at, int n, u64 upper, bool has_upper, int k, u32* hist, int* red) {
    const int tid = threadIdx.x;
    constexpr int NB = 1 << DB;
    u64 prefix = 0, pmask = 0;
    constexpr int NPASS = (KEY_BITS + DB - 1) / DB;
    for (int pass = 0; pass < NPASS; ++pass) {
        const int shift = (NPASS - 1 - pass) * DB;
        for (int i = tid; i < NB; i += T) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += T) {
            const u64 key = key_at(i);
            if (has_upper && !(key < upper)) continue;
            if ((key & pmask) != prefix) continue;
            atomicAdd(&hist[(u32)(key >> shift) & (NB - 1)], 1u);
        }
        __syncthreads();
        block_find_digit<NB / T>(hist, k, red, red + 256);
        const int d = red[256];
        k -= red[257];
        prefix |= (u64)d << shift;
        pmask |= (u64)(NB - 1) << shift;
        __syncthreads();
    }
    return prefix;
}

// ======================================================================================
// K4
// ======================================================================================
// A box as the NMS loop holds it in LDS: pixel corners + area in the dtype the reference's IoU works in (float64,
// or float32 on the 'corners' flow), computed once per box with the reference's operations.
template <typename F>
struct __attribute__((aligned(16))) NBox {
    F x0, y0, x1, y1, area;
    u32 idx;                      // anchor index
    u32 ok;                       // all five values finite -> the division-free test below is valid for this box
};

template <typename F>
__device__ __forceinline__ bool is_finite(F v) { return v - v == (F)0; }

// px_f32: the Keras layer scales the box by the image size in float32 (keras_layer_DecodeDetections.py:140-149)
// before NMS sees it; the NumPy decoder multiplies the float64 copy (ssd_output_decoder.py:196-198).
template <typename F>
__device__ __forceinline__ NBox<F> make_nbox(const float4 bx, u32 idx, F W, F H, F d, bool px_f32) {
    NBox<F> r;
    if (px_f32) {                 // float32 product == correctly rounded exact product (W, H exact in float32: host checks)
        r.x0 = (F)(float)((double)bx.x * (double)W); r.y0 = (F)(float)((double)bx.y * (double)H);
        r.x1 = (F)(float)((double)bx.z * (double)W); r.y1 = (F)(float)((double)bx.w * (double)H);
    } else {
        r.x0 = (F)bx.x * W; r.y0 = (F)bx.y * H; r.x1 = (F)bx.z * W; r.y1 = (F)bx.w * H;   // exact in double
    }
    r.area = box_area<F>(r.x0, r.y0, r.x1, r.y1, d);
    r.idx = idx;
    r.ok = (is_finite(r.x0) && is_finite(r.y0) && is_finite(r.x1) && is_finite(r.y1) && is_finite(r.area)) ? 1u : 0u;
    return r;
}

// "IoU(a,b) is NOT <= thr" (the reference's suppression test, ssd_output_decoder.py:91) -> 1 suppressed, 0 kept,
// 2 undecided (caller evaluates exact_suppresses()).
//  * float64 flow: inter and uni are computed exactly as the reference computes them; instead of dividing, the sign
//    of R = inter - thr*uni (one fma, correctly rounded, so sign(r) == sign(R)) decides, which is the same answer as
//    fl(inter/uni) <= thr whenever |R| > 2^-51 * thr*uni, i.e. the exact ratio is more than 2 ulp away from thr
//    (rounding is monotone).  Needs finite boxes, uni > 0 (not tiny) and 0 < thr < inf (fast_ok); two zero-area boxes
//    (0/0) are decided directly; everything else -- non-finite boxes, knife edges -- takes the IEEE division.
//    float64 add/mul/max/fma issue at the float32 (non-packed) rate on CDNA4, so this costs ~16 full-rate VALU ops
//    per pair and no division.
//  * float32 flow ('corners' + float32 input): the reference itself works in float32 -> evaluate it directly.
__device__ __forceinline__ int nms_test(const NBox<double>& a, const NBox<double>& b, double thr, int fast_ok) {
    const double ix0 = fmax(a.x0, b.x0), iy0 = fmax(a.y0, b.y0);
    const double ix1 = fmin(a.x1, b.x1), iy1 = fmin(a.y1, b.y1);
    const double iw = fmax(ix1 - ix0, 0.0), ih = fmax(iy1 - iy0, 0.0);
    const double inter = iw * ih;
    const double uni = (a.area + b.area) - inter;
    const double tu = thr * uni;
    const double r = fma(-thr, uni, inter);
    const bool fin = (a.ok & b.ok & (u32)fast_ok) != 0u;
    const bool zero = uni == 0.0 && inter == 0.0;            // 0/0 = NaN: "not <= thr" (two zero-area boxes)
    const bool dec = uni > 0x1p-900 && fabs(r) > 0x1p-50 * tu;
    return fin ? (zero ? 1 : (dec ? (r > 0.0 ? 1 : 0) : 2)) : 2;
}
__device__ __forceinline__ int nms_test(const NBox<float>& a, const NBox<float>& b, float thr, int) {
    const PxBox<float> pa = {a.x0, a.y0, a.x1, a.y1, a.area}, pb = {b.x0, b.y0, b.x1, b.y1, b.area};
    return (iou_px<float>(pa, pb) <= thr) ? 0 : 1;
}

template <typename F>
__device__ __forceinline__ bool exact_suppresses(const NBox<F>& a, const NBox<F>& b, F thr) {
    const PxBox<F> pa = {a.x0, a.y0, a.x1, a.y1, a.area}, pb = {b.x0, b.y0, b.x1, b.y1, b.area};
    return !(iou_px<F>(pa, pb) <= thr);
}

// ---------------------------------------------------------------------------------------------------------------
// K4 (second generation).  One workgroup of T threads (W = T/64 waves) per (image, class).  Per round: histogram-select
// the next <= MAX_CHUNK best keys -> bitonic sort in LDS -> stage the chunk's boxes as 32-byte FBox records -> greedy
// NMS in batches of 64 (lane = candidate):
//   phase A  the batch against the survivors so far; the kept list is striped over the waves, each kept box is ONE
//            broadcast LDS read (b128 + b64) and ~15 float32 VALU instructions per 64 pairs;
//   phase B  the batch against itself: only the 2016 pairs i > j, row j folded with row 63-j into one 64-lane step;
//   resolve  scalar walk over the lanes whose row is non-zero (rows fetched by readlane).
// Three pair-test policies, one per arithmetic the reference family uses:
//   POL_NUMPY64  ssd_output_decoder.py:77-92 with float64 boxes (float32 input routed through convert_coordinates, SURVEY
//                A.3): the test "IoU <= thr" is decided by a float32 evaluation of R = inter - thr*union with a rigorous
//                error bound E = kE * S (S >= max(4 max|coord|^2, |area|) of the pair; derivation in DESIGN.md 4.1);
//                |R32| <= E -- knife edges, non-finite boxes -- falls through to the float64 sign test / IEEE division
//                of the first generation (nms_test / exact_suppresses), so results stay bit-identical to the reference;
//   POL_NUMPY32  the same loop in float32 ('corners' input stays float32 in the reference): exact float32 IoU;
//   POL_TF32     tf.image.non_max_suppression as the Keras layers call it (keras_layer_DecodeDetections.py:195-199):
//                float32 IoU of min/max-normalised corners, 0 when either area is <= 0, suppressed when IoU > float32(thr)
//                (tensorflow/core/kernels/non_max_suppression_op.cc; un-pinned dependency, restated from its source).
// ---------------------------------------------------------------------------------------------------------------
enum { POL_NUMPY64 = 0, POL_NUMPY32 = 1, POL_TF32 = 2 };

struct __attribute__((aligned(16))) FBox {
    float x0, y0, x1, y1;        // pixel corners, float32
    float area;                  // POL_TF32: +inf marks a box that neither suppresses nor is suppressed (area <= 0 or NaN)
    float sc;                    // POL_NUMPY64: error scale S of the float32 filter, rounded up (inf: never decided in float32)
    u32 idx;                     // anchor index
    u32 pad;
};

struct NmsK {                    // wave-uniform constants of a launch
    double W, H, d, thr;
    float Wf, Hf, df, thr32, kE;
    float thr_lo, thr_hi;        // POL_TF32: float32(thr) * (1 -+ 2^-20), the constants of the division-free pair test
    int fast_ok, no_nms;
};

// v_max_f32 / v_min_f32 without the canonicalising self-max hipcc adds to fmaxf() on loaded values (IEEE mode).  NaN
// operands never reach a decision through them: POL_NUMPY64 gives such boxes S = inf, POL_TF32 stores them as inert boxes.
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int POL>
__device__ __forceinline__ FBox make_fbox(const float4 bx, u32 idx, const NmsK& k) {
    FBox r;
    r.idx = idx; r.pad = 0u; r.sc = 0.f;
    if (POL == POL_NUMPY64) {
        const double x0 = (double)bx.x * k.W, y0 = (double)bx.y * k.H, x1 = (double)bx.z * k.W, y1 = (double)bx.w * k.H;   // exact
        const double area = box_area<double>(x0, y0, x1, y1, k.d);
        const double m = fmax(fmax(fabs(x0), fabs(y0)), fmax(fabs(x1), fabs(y1)));
        const double S = fmax(fmax(4.0 * m * m, fabs(area)), 0x1p-60) * (1.0 + 0x1p-20);
        // finite everywhere and far from float32 overflow (2 S must stay finite), else the float32 filter never decides
        const bool fin = is_finite(x0) && is_finite(y0) && is_finite(x1) && is_finite(y1) && is_finite(area) && S < 0x1p120;
        r.x0 = (float)x0; r.y0 = (float)y0; r.x1 = (float)x1; r.y1 = (float)y1; r.area = (float)area;
        r.sc = fin ? (float)S : __builtin_inff();
    } else if (POL == POL_NUMPY32) {
        r.x0 = bx.x * k.Wf; r.y0 = bx.y * k.Hf; r.x1 = bx.z * k.Wf; r.y1 = bx.w * k.Hf;         // ssd_output_decoder.py:196-198 in float32
        r.area = box_area<float>(r.x0, r.y0, r.x1, r.y1, k.df);
    } else {
        // the layer scales in float32 (keras_layer_DecodeDetections.py:140-149): rounded exact product (W, H exact in float32)
        const float ax = (float)((double)bx.x * k.W), ay = (float)((double)bx.y * k.H);
        const float bxx = (float)((double)bx.z * k.W), by = (float)((double)bx.w * k.H);
        const float x0 = bxx < ax ? bxx : ax, x1 = ax < bxx ? bxx : ax;      // std::min / std::max of the TF kernel
        const float y0 = by < ay ? by : ay, y1 = ay < by ? by : ay;
        const float area = (y1 - y0) * (x1 - x0);
        if (area > 0.f) { r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1; r.area = area; }
        else { r.x0 = 0.f; r.y0 = 0.f; r.x1 = 0.f; r.y1 = 0.f; r.area = __builtin_inff(); }   // IoU 0 or NaN with everything: inert
    }
    return r;
}

// supp: the pair is decided "b suppresses a"; und (POL_NUMPY64 only): the float32 filter cannot decide
template <int POL>
__device__ __forceinline__ void pair_test(const FBox& a, const FBox& b, const NmsK& k, bool& supp, bool& und) {
    if (POL == POL_TF32) {
        // fl(inter / uni) > thr  <=  inter >= fl(uni * c_hi),  c_hi = fl(thr (1 + 2^-20));   fl(inter / uni) <= thr  <=  inter <= fl(uni * c_lo),
        // c_lo = fl(thr (1 - 2^-20))  (thr > 0; uni > 0 or +inf: inter <= either area in float32 as well, rounding is monotone; the
        // roundings of the constant, of the product and of the quotient are each below 2^-23 relative, so the exact ratio is beyond
        // thr (1 +- 2^-21)); what lies in between -- and every NaN -- takes the division.  inter and uni are the TF kernel's own float32
        // values, except that the HEIGHT is not clamped: a negative one makes inter <= 0 and uni >= area_a + area_b > 0, which both
        // tests read as "not suppressed", the right answer for an empty intersection (0 * inf = NaN: the division path, which clamps).
        // 15 VALU instructions per 64 pairs (17 in rounds 2-4: three products, the second clamp).
        const float ix0 = vmaxf(a.x0, b.x0), iy0 = vmaxf(a.y0, b.y0);
        const float ix1 = vminf(a.x1, b.x1), iy1 = vminf(a.y1, b.y1);
        const float iw = vmaxf(ix1 - ix0, 0.f), ih = iy1 - iy0;
        const float inter = iw * ih;
        const float uni = (a.area + b.area) - inter;
        const bool yes = inter >= uni * k.thr_hi, no = inter <= uni * k.thr_lo;
        supp = k.fast_ok && yes;
        und = !k.fast_ok || !(yes || no);
        return;
    }
    if (POL == POL_NUMPY32) {
        const PxBox<float> pa = {a.x0, a.y0, a.x1, a.y1, a.area}, pb = {b.x0, b.y0, b.x1, b.y1, b.area};
        supp = !(iou_px<float>(pa, pb) <= k.thr32);
        und = false;
        return;
    }
    const float ix0 = vmaxf(a.x0, b.x0), iy0 = vmaxf(a.y0, b.y0);
    const float ix1 = vminf(a.x1, b.x1), iy1 = vminf(a.y1, b.y1);
    const float iw = vmaxf(ix1 - ix0, 0.f), ih = vmaxf(iy1 - iy0, 0.f);
    const float inter = iw * ih;
    const float uni = (a.area + b.area) - inter;
    // POL_NUMPY64
    const float r = __builtin_fmaf(-k.thr32, uni, inter);
    const float E = k.kE * vmaxf(a.sc, b.sc);
    // the sign of R answers "IoU <= thr" only for a positive union (negative areas: border_pixels 'exclude', inverted boxes)
    const bool dec = __builtin_fabsf(r) > E && uni > E;
    supp = dec && r > 0.f;
    und = !dec;
}

// the exact evaluation of a pair the fast test left undecided
template <int POL>
__device__ __forceinline__ bool exact_pair(const FBox& a, const FBox& b, const float4 a4, const float4 b4, const NmsK& k) {
    if (POL == POL_TF32) {
        const float ix0 = a.x0 < b.x0 ? b.x0 : a.x0, iy0 = a.y0 < b.y0 ? b.y0 : a.y0;        // std::max / std::min as the TF kernel
        const float ix1 = b.x1 < a.x1 ? b.x1 : a.x1, iy1 = b.y1 < a.y1 ? b.y1 : a.y1;
        const float dw = ix1 - ix0, dh = iy1 - iy0;
        const float inter = (dw < 0.f ? 0.f : dw) * (dh < 0.f ? 0.f : dh);
        return inter / ((a.area + b.area) - inter) > k.thr32;
    }
    // POL_NUMPY64: the first generation's float64 evaluation (sign test, then IEEE division for what is left)
    const NBox<double> A = make_nbox<double>(a4, 0u, k.W, k.H, k.d, false), B = make_nbox<double>(b4, 0u, k.W, k.H, k.d, false);
    const int c = nms_test(A, B, k.thr, k.fast_ok);
    return c == 1 || (c == 2 && exact_suppresses<double>(A, B, k.thr));
}

// Bitonic sort, descending, in place, of P (a power of two, >= 2) u64 keys in LDS by T threads.  Compare-exchange steps at
// distance <= 64 stay inside the 128-key segment the wave owns, so only the few long-distance steps need a block barrier.
template <int T>
__device__ void block_bitonic_desc(u64* a, int P) {
    const int tid = threadIdx.x;
    for (int kk = 2; kk <= P; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 x = a[i], y = a[l];
                const bool desc = (i & kk) == 0;
                if ((x < y) == desc) { a[i] = y; a[l] = x; }
            }
            const int nj = j > 1 ? (j >> 1) : kk;             // distance of the next step (the next stage opens at kk)
            if (j > 64 || nj > 64) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        }
    }
    __syncthreads();
}

// Bitonic sort, descending, of T * PER u64 keys held PER per thread (element e = tid * PER + r): compare-exchange partners
// at distance < PER sit in the same thread, at distance < 64 * PER in the same wave (two 32-bit lane shuffles per key), and only
// the last log2(T / 64) distances of the last stages go through LDS (`xch`: T * PER u64, two barriers per such step).
template <int T, int PER>
__device__ __forceinline__ void block_bitonic_desc_regs(u64 (&v)[PER], u64* xch) {
    static_assert(PER == 1 || PER == 2 || PER == 4, "PER");
    const int tid = threadIdx.x;
    constexpr int N = T * PER;
    for (int kk = 2; kk <= N; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j < PER) {                                    // both elements in this thread
#pragma unroll
                for (int r = 0; r < PER; ++r) {
#pragma unroll
                    for (int jj = 1; jj < PER; jj <<= 1) {
                        if (jj == j && (r & jj) == 0) {
                            const bool desc = (((tid * PER + r) & kk) == 0);
                            const u64 a = v[r], b = v[r | jj];
                            if ((a < b) == desc) { v[r] = b; v[r | jj] = a; }
                        }
                    }
                }
            } else if (j < 64 * PER) {                        // partner lane = lane ^ (j / PER)
                const int m = j / PER;
                const bool low = (tid & m) == 0;              // this thread holds the lower-indexed element of each pair
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const bool desc = (((tid * PER + r) & kk) == 0);
                    const u32 olo = (u32)__shfl_xor((int)(u32)v[r], m), ohi = (u32)__shfl_xor((int)(u32)(v[r] >> 32), m);
                    const u64 o = ((u64)ohi << 32) | olo;
                    const bool take_max = low == desc;
                    v[r] = ((o > v[r]) == take_max) ? o : v[r];   // one 64-bit compare (keys are distinct; equal zero pads: either)
                }
            } else {                                          // another wave: exchange through LDS
                __syncthreads();
#pragma unroll
                for (int r = 0; r < PER; ++r) xch[tid * PER + r] = v[r];
                __syncthreads();
                const bool low = ((tid * PER) & j) == 0;
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const int e = tid * PER + r;
                    const bool desc = ((e & kk) == 0);
                    const u64 o = xch[e ^ j];
                    const bool take_max = low == desc;
                    v[r] = ((o > v[r]) == take_max) ? o : v[r];
                }
            }
        }
    }
}

// K4's one-pass read of a candidate list into registers: KC buffer loads per thread, all in flight together, ONE 32-bit
// offset register (the step rides in the scalar soffset; past-the-end slots come back as 0 from the buffer unit's range check
// and are mapped to the "consumed" key ~0).  The record count goes through an opaque scalar so that the loads are not hoisted
// out of the round loop (the list is loop invariant; hoisting would keep 2 * KC registers alive through the NMS phases).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int T, int KP>
__device__ __forceinline__ void load_keys_cached(const u64* keys, int n, int tid, int first, u64 (&kc)[KP]) {
#if defined(__HIP_DEVICE_COMPILE__)
    int nn = n;
    asm volatile("" : "+s"(nn));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(keys), 0, nn * 8, 0x00020000);
#pragma unroll
    for (int u = 0; u < KP; ++u) {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, tid * 8, (first + u) * T * 8, 0);
        const u64 key = ((u64)v.y << 32) | v.x;
        kc[u] = key ? key : ~0ull;
    }
#endif
}

// A value every lane holds alike (read from LDS after a barrier) moved to scalar registers: the compiler cannot prove the uniformity
// and would keep it in VGPRs -- 80 of them is all the 512-thread kernel has.
__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 uni64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}

#ifndef SSDHIP_NMS_WAVES512
#define SSDHIP_NMS_WAVES512 6      // minimum waves per SIMD the 512-thread variant is compiled for (tools/prof_build.sh sweeps it)
#endif
// FULL = false (the 512-thread form): the kernel has no exact-selection fallback.  A class whose score histogram cannot deliver a chunk
// (more than MAX_CHUNK near-equal scores in the bin at the cut and fewer than M / 4 above it -- not met once on any measured workload)
// is handed back with kept_count = -1 and redone by a FULL launch (512 threads as well: the round-4 kernel, scratch and all -- it only
// ever multiplies for the handed-back classes) with redo_only = 1, whose other workgroups return at once.  Inlined,
// the fallback's five radix passes made the compiler park loop invariants in scratch at the top of EVERY workgroup: 48 bytes per lane,
// 15.7 MB of scratch stores per launch -- the "1.39 x" HBM traffic of rounds 2-4 (profiles/r04zz_decode_pmc_traffic.json).
template <int POL, int T, bool FULL>
__global__ __launch_bounds__(T, (T >= 512 ? SSDHIP_NMS_WAVES512 : 3)) void nms_kernel(DecodeParams p, const float4* __restrict__ boxes,
                                               const u64* __restrict__ cand, const int* __restrict__ cand_count,
                                               const int* __restrict__ work_order,
                                               u64* __restrict__ kept, int* __restrict__ kept_count, int redo_only) {
    constexpr int W = T / 64;
    // pairs tested per step in phase A / folded rows per step in phase B, and whether phase A prefetches the next step's
    // survivors: eight-wave workgroups trade unrolling (registers) for waves per SIMD
    constexpr int KC = KC_KEYS / T;
    constexpr int KP = 9;                    // keys of one trip over the cached list (KC = 18 or 36: two or four trips)
    static_assert(KC % KP == 0, "KC must be a multiple of the trip size");
    // (round 6: with the dense phase B the 512-thread kernel has room for four survivor records per phase-A step again -- 80 VGPRs, no
    //  scratch; tamed heads 82.7 -> 80.2 us, SSD512 sparse 81.4 -> 78.0, r06v.  The float64-fallback policy keeps two: four spill 14
    //  registers there.  One per step with a prefetch, or two with one, are 5-15 % SLOWER: more, shorter steps.)
    constexpr int UA = T >= 512 ? (POL == POL_NUMPY64 ? 2 : 4) : 4, UB = T >= 512 ? 2 : 4;
    constexpr bool PREFETCH = T < 512;
    // XCD-aware work mapping: hardware places block x on XCD x%8; give every XCD a contiguous range of
    // (image, class) work items so that all classes of an image share one L2.
    const int total_work = p.B * p.G;
    const int per_xcd = (total_work + 7) >> 3;
    int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || work >= total_work) return;
    if (work_order) work = work_order[work];
    if (redo_only && kept_count[work] != -1) return;
    const int b = work / p.G;

    __shared__ __attribute__((aligned(16))) u64 keybuf[MAX_CHUNK + 8];     // chunk keys, sorted in place
    // the keys of the NEXT bins down (same histogram, same pass over the candidate list): a class that needs a second chunk -- one in
    // two on distinct confidences -- takes it from LDS instead of reading, histogramming and filtering its candidate list again
    __shared__ __attribute__((aligned(16))) u64 keybuf2[MAX_CHUNK + 8];
    __shared__ int fill2;
    constexpr size_t SCRATCH = sizeof(FBox) * MAX_CHUNK > NMS_NBINS * sizeof(u32) ? sizeof(FBox) * MAX_CHUNK : NMS_NBINS * sizeof(u32);
    __shared__ __attribute__((aligned(16))) unsigned char scratch[SCRATCH];   // score histogram, then the chunk's boxes
    __shared__ __attribute__((aligned(16))) float4 cf4[POL == POL_NUMPY64 ? MAX_CHUNK : 1];    // normalised corners (exact fallback)
    __shared__ FBox kb[KEPT_LDS + 8 * W + 8];                                    // survivors (reads may run 8W past K: masked)
    __shared__ __attribute__((aligned(16))) float4 kf4[POL == POL_NUMPY64 ? KEPT_LDS : 1];
    __shared__ u64 maskrow[64];
    __shared__ u64 supp_a[W];
    __shared__ int red[264];
    __shared__ int fill;
    u32* hist = reinterpret_cast<u32*>(scratch);
    FBox* cb = reinterpret_cast<FBox*>(scratch);
    u64* sorted = keybuf;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = cand_count[work];
    const u64* keys = cand + (size_t)work * p.N;
    u64* kept_out = kept + (size_t)work * p.cap_store;
    const float4* img_boxes = boxes + (size_t)b * p.N;
    NmsK k;
    k.W = p.img_w; k.H = p.img_h;
    k.d = p.border == SSDHIP_BORDER_INCLUDE ? 1.0 : (p.border == SSDHIP_BORDER_EXCLUDE ? -1.0 : 0.0);
    k.thr = p.iou_thresh;
    k.Wf = (float)p.img_w; k.Hf = (float)p.img_h; k.df = (float)k.d; k.thr32 = (float)p.iou_thresh;
    k.kE = p.filter_kE;
    k.thr_lo = k.thr32 * 0.99999905f;
    k.thr_hi = k.thr32 * 1.00000095f;
    k.fast_ok = p.fast_ok; k.no_nms = p.no_nms;
    const int cap_eff = min(p.cap_store, n);
    if (!FULL && n > KC * T) {               // lists beyond the register-cached length (SSD512's dense classes): the FULL kernel's
        if (tid == 0) kept_count[work] = -1; // eight-loads-at-a-time passes; their addresses would cost this kernel registers it lacks
        return;
    }

    int K = 0, consumed = 0;
    int pending2 = 0;                        // keys of the second chunk waiting in keybuf2
    u64 upper = ~0ull;                       // keys >= upper are consumed; a real key is never all ones (its score field is a
    bool has_upper = false;                  // float key, whose all-ones value is a NaN that cannot pass the threshold)
    PROF_DECL
    bool finished = false;
    while (consumed < n && K < cap_eff && !finished) {
        const int remaining = n - consumed;
        int want = 2 * (cap_eff - K);        // survivors still wanted, x2 headroom for suppressed candidates
        int M = 128;
        while (M < want && M < MAX_CHUNK) M <<= 1;
        int m;
        u64 cutoff = 0;
        bool by_bin = false;
        int bin_cut = 0, bin_cut2 = 0, m2 = 0;
        if (pending2) {                      // the bins right below the last chunk are already in LDS
            m = pending2;
            pending2 = 0;
            for (int i = tid; i < MAX_CHUNK + 8; i += T) keybuf[i] = keybuf2[i];
            __syncthreads();
            PROF_MARK(0)
            PROF_MARK(1)
        } else {
        // A candidate list of up to KC * T keys (every SSD300 class list) is read in trips of KP keys per thread and pass (histogram,
        // collection): all of a trip's loads are in flight together.  Longer lists (SSD512) take eight loads at a time.  (One trip of
        // KC keys, as in rounds 2-4, or keeping the keys in registers from the first pass to the second, spills: 2 KC registers
        // against a budget of 80 at six waves per SIMD -- 15 MB of scratch stores per launch, profiles/r04zz_decode_pmc_traffic.json.)
        const bool cached = !FULL || n <= KC * T;
        if (remaining <= MAX_CHUNK) {
            m = remaining;                   // take everything that is left
        } else {
            // histogram of the remaining keys by score bin
            for (int i = tid; i < NMS_NBINS; i += T) hist[i] = 0;
            __syncthreads();
            if (cached) {
#pragma unroll 1
                for (int part = 0; part < KC / KP; ++part) {
                    u64 kc[KP];
                    load_keys_cached<T, KP>(keys, n, tid, part * KP, kc);
#pragma unroll
                    for (int u = 0; u < KP; ++u)
                        if (kc[u] < upper) atomicAdd(&hist[bin_of<NMS_BIN_SHIFT, NMS_NBINS>((u32)(kc[u] >> IDX_BITS), p.thr_key)], 1u);
                }
            } else {
                for (int i0 = tid; i0 < n; i0 += 8 * T) {
                    u64 k8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int i = i0 + u * T; k8[u] = i < n ? keys[i] : ~0ull; }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (k8[u] < upper) atomicAdd(&hist[bin_of<NMS_BIN_SHIFT, NMS_NBINS>((u32)(k8[u] >> IDX_BITS), p.thr_key)], 1u);
                }
            }
            __syncthreads();
            // highest bin d such that the bins above it hold <= MAX_CHUNK keys (and d included would not fit); and the same question
            // again for the bins below that cut: the second chunk
            block_find_digit2<NMS_NBINS / T>(hist, MAX_CHUNK + 1, MAX_CHUNK + 1, red, red + 256, -1, remaining);
            bin_cut = uni32(red[256]) + 1;
            m = uni32(red[257]);
            bin_cut2 = uni32(red[258]) + 1;
            m2 = uni32(red[259]) - m;
            __syncthreads();
            if (m >= (M >> 2)) {
                by_bin = true;
            } else {                         // one bin holds too many near-equal scores: exact selection of the M best
                m2 = 0;
                if (!FULL) {                 // (uniform: every thread leaves here)
                    if (tid == 0) kept_count[work] = -1;
                    return;
                }
#ifdef SSDHIP_PROFILE
                if (tid == 0) { atomicAdd(&g_prof[15], 1ull); if (m == 0) atomicAdd(&g_prof[11], 1ull); }
#endif
                m = M;
                if (FULL) cutoff = uni64(block_select_kth<32 + IDX_BITS, 12, T>([&](int i) { return keys[i]; }, n, upper, has_upper, M, hist, red));
            }
        }
        PROF_MARK(0)
        for (int i = tid; i < MAX_CHUNK + 8; i += T) { keybuf[i] = 0ull; keybuf2[i] = 0ull; }     // unused sort slots hold 0 (below every key)
        if (tid == 0) { fill = 0; fill2 = 0; }
        __syncthreads();
        if (cached) {
#pragma unroll 1
            for (int part = 0; part < KC / KP; ++part) {
                u64 kc[KP];
                load_keys_cached<T, KP>(keys, n, tid, part * KP, kc);
#pragma unroll
                for (int u = 0; u < KP; ++u) {
                    const u64 key = kc[u];
                    if (!(key < upper)) continue;
                    const int bin = bin_of<NMS_BIN_SHIFT, NMS_NBINS>((u32)(key >> IDX_BITS), p.thr_key);
                    const bool take = by_bin ? (bin >= bin_cut) : (key >= cutoff);
                    if (take) keybuf[atomicAdd(&fill, 1)] = key;
                    else if (m2 > 0 && bin >= bin_cut2) keybuf2[atomicAdd(&fill2, 1)] = key;
                }
            }
        } else {
            for (int i0 = tid; i0 < n; i0 += 8 * T) {
                u64 k8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * T; k8[u] = i < n ? keys[i] : ~0ull; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const u64 key = k8[u];
                    if (!(key < upper)) continue;
                    const int bin = bin_of<NMS_BIN_SHIFT, NMS_NBINS>((u32)(key >> IDX_BITS), p.thr_key);
                    const bool take = by_bin ? (bin >= bin_cut) : (key >= cutoff);
                    if (take) keybuf[atomicAdd(&fill, 1)] = key;
                    else if (m2 > 0 && bin >= bin_cut2) keybuf2[atomicAdd(&fill2, 1)] = key;
                }
            }
        }
        pending2 = m2;
        __syncthreads();
        PROF_MARK(1)
        }
        {
            constexpr int PER = MAX_CHUNK / T;
            u64 v[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) v[r] = keybuf[tid * PER + r];
            block_bitonic_desc_regs<T, PER>(v, reinterpret_cast<u64*>(scratch));
#pragma unroll
            for (int r = 0; r < PER; ++r) keybuf[tid * PER + r] = v[r];
            __syncthreads();
        }
        PROF_MARK(2)
        upper = uni64(sorted[m - 1]);
        has_upper = true;
        consumed += m;
        // stage the chunk's boxes (independent gathers, one latency exposure per round)
        for (int i = tid; i < m; i += T) {
            const u32 idx = IDX_MASK - (u32)(sorted[i] & IDX_MASK);
            const float4 f4 = img_boxes[idx];
            cb[i] = make_fbox<POL>(f4, idx, k);
            if (POL == POL_NUMPY64) cf4[i] = f4;
        }
        __syncthreads();
        PROF_MARK(3)

        for (int base = 0; base < m && K < cap_eff && !finished; base += 64) {
            const int nb = min(64, m - base);
            const bool valid = lane < nb;
            const FBox me = cb[base + (valid ? lane : 0)];
            float4 me4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (POL == POL_NUMPY64) me4 = cf4[base + (valid ? lane : 0)];
            // ---- phase A: against the survivors of earlier batches, striped over the waves, four per step ----
            bool supp = false;
            if (!k.no_nms) {
                const int Kl = min(K, KEPT_LDS);
                FBox o[UA];
#pragma unroll
                for (int u = 0; u < UA; ++u) o[u] = kb[wave + u * W];
                for (int j0 = wave; j0 < Kl; j0 += UA * W) {
                    FBox nx[PREFETCH ? UA : 1];                  // PREFETCH: the next step's survivors are on their way while this
                    if (PREFETCH) {                              // step's are tested (kb is padded: the reads stay in bounds)
#pragma unroll
                        for (int u = 0; u < UA; ++u) nx[u] = kb[j0 + UA * W + u * W];
                    }
                    bool s_any = false, u_any = false;
                    bool uu[UA];
#pragma unroll
                    for (int u = 0; u < UA; ++u) {
                        const int j = j0 + u * W;
                        bool s, un;
                        pair_test<POL>(me, o[u], k, s, un);
                        const bool in = j < Kl;
                        s_any |= in && s;
                        uu[u] = in && un;
                        u_any |= uu[u];
                    }
                    supp |= s_any;
                    if (POL != POL_NUMPY32 && __ballot(u_any && valid && !supp) != 0ull) {
#pragma unroll
                        for (int u = 0; u < UA; ++u)
                            if (uu[u] && !supp) supp = exact_pair<POL>(me, o[u], me4, POL == POL_NUMPY64 ? kf4[j0 + u * W] : me4, k);
                    }
                    if (__ballot(valid && !supp) == 0ull) break;        // the whole batch is already suppressed
#pragma unroll
                    for (int u = 0; u < UA; ++u) o[u] = PREFETCH ? nx[u] : kb[j0 + UA * W + u * W];
                }
                for (int j = KEPT_LDS + wave; j < K; j += W) {          // survivors beyond the LDS cache (uncapped decodes)
                    const u32 idx = IDX_MASK - (u32)(kept_out[j] & IDX_MASK);
                    const float4 o4 = img_boxes[idx];
                    const FBox o = make_fbox<POL>(o4, idx, k);
                    bool s, un;
                    pair_test<POL>(me, o, k, s, un);
                    if (POL != POL_NUMPY32 && un && !supp) s = exact_pair<POL>(me, o, me4, o4, k);
                    supp |= s;
                    if (__ballot(valid && !supp) == 0ull) break;
                }
            }
            const u64 sa = __ballot(supp && valid);
            if (lane == 0) supp_a[wave] = sa;
            __syncthreads();
            // candidates already suppressed by an earlier survivor are dead: their rows of the in-batch matrix are never read
            // by the resolve loop, so phase B skips them (on densely overlapping boxes most of a batch dies in phase A)
            u64 sall_v = 0ull;
#pragma unroll
            for (int w = 0; w < W; ++w) sall_v |= supp_a[w];
            const u64 sall = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(sall_v >> 32)) << 32) |
                             (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)sall_v);      // scalar: the skips below are s_cbranch
            PROF_MARK(4)
            // ---- round 6: the rest of the batch runs on the candidates phase A left ALIVE, renumbered densely (order kept).  On real
            //      score distributions two thirds of a batch die against the earlier survivors: their rows of the in-batch matrix are
            //      never read and nothing they could suppress matters either, so the triangle shrinks from nb (nb - 1) / 2 pairs to
            //      na (na - 1) / 2 (na ~ nb / 3: a ninth), and the resolve walks na lanes.  `dl` = the batch lane of the candidate with
            //      dense rank `lane`: every lane pushes its own index to its rank (alive first, dead behind them) through the LDS
            //      crossbar -- no memory, no barrier.
            const u64 nbmask = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
            const u64 live0 = nbmask & ~sall;                                       // scalar
            const int na = __popcll(live0);
            int dl;
            {
                const bool mine = (live0 >> lane) & 1ull;
                const int below = __popcll(live0 & lanemask_lt());                  // alive lanes below this one
                const int dest = mine ? below : na + (lane - below);                // a bijection of the 64 lanes
                dl = __builtin_amdgcn_ds_permute(dest << 2, lane);
            }
            // ---- phase B: pairs i > j among the na alive candidates.  Row j (lanes j+1..na-1) and row na-1-j (lanes na-j..na-1)
            //      together fill na-1 lanes: folded row f handles rows f and na-1-f in one step.  Bit i of maskrow[j] = "j suppresses i",
            //      i and j DENSE ranks. ----
            const int nfold = (na + 1) >> 1;
            for (int f0 = wave * UB; f0 < nfold; f0 += UB * W) {     // UB folded rows per step: their LDS reads overlap
                bool s4[UB], un4[UB], act4[UB];
                int ii4[UB], jj4[UB];
                FBox bi[UB], bj[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int f = f0 + u;
                    const int jA = f, jB = na - 1 - f;
                    const bool row = f < nfold;
                    const bool isA = lane > jA && lane < na;
                    const bool isB = lane < f && jB != jA;                   // pair (na - f + lane, jB)
                    const int id = isA ? lane : (isB ? na - f + lane : 0);   // dense rank of the suppressed side
                    const int jd = row ? (isA ? jA : jB) : 0;                // ... of the suppressing side (one of two values per step)
                    ii4[u] = __builtin_amdgcn_ds_bpermute((row ? id : 0) << 2, dl);      // -> batch lanes
                    jj4[u] = __builtin_amdgcn_ds_bpermute(jd << 2, dl);
                    act4[u] = !k.no_nms && row && (isA || isB);
                }
                if (!k.no_nms) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) { bi[u] = cb[base + ii4[u]]; bj[u] = cb[base + jj4[u]]; }
#pragma unroll
                    for (int u = 0; u < UB; ++u) pair_test<POL>(bi[u], bj[u], k, s4[u], un4[u]);
                    if (POL != POL_NUMPY32) {
                        bool un_any = false;
#pragma unroll
                        for (int u = 0; u < UB; ++u) un_any |= un4[u] && act4[u];
                        if (__ballot(un_any) != 0ull) {
#pragma unroll
                            for (int u = 0; u < UB; ++u)
                                if (un4[u] && act4[u])
                                    s4[u] = exact_pair<POL>(bi[u], bj[u], POL == POL_NUMPY64 ? cf4[base + ii4[u]] : me4,
                                                            POL == POL_NUMPY64 ? cf4[base + jj4[u]] : me4, k);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < UB; ++u) s4[u] = false;
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int f = f0 + u;
                    if (f >= nfold) break;
                    const int jA = f, jB = na - 1 - f;
                    const u64 ball = __ballot(act4[u] && s4[u]);
                    const u64 rowA = ball & ~((2ull << jA) - 1ull);
                    const u64 rowB = (jB == jA || f == 0) ? 0ull : ((ball & ((1ull << f) - 1ull)) << (na - f));
                    if (lane == 0) {
                        maskrow[jA] = rowA;
                        if (jB != jA) maskrow[jB] = rowB;
                    }
                }
            }
            __syncthreads();
            PROF_MARK(5)
            // resolve, in dense ranks.  Candidates are taken in order; a candidate is kept iff no earlier KEPT candidate suppresses it.
            // Only lanes whose row is non-zero can change anything, so the scalar loop visits just those ("conflict
            // lanes"); everything still alive at the end is kept.  Rows sit one per lane and are fetched by readlane.
            const u64 myrow = lane < na ? maskrow[lane] : 0ull;
            const u32 row_lo = (u32)myrow, row_hi = (u32)(myrow >> 32);
            u64 alive = na == 64 ? ~0ull : ((1ull << na) - 1ull);
            const u64 cmask = __ballot(myrow != 0ull);
            u64 pending = alive & cmask;
            while (pending) {
                const int j = __ffsll((long long)pending) - 1;
                const u64 mj = ((u64)(u32)__builtin_amdgcn_readlane((int)row_hi, j) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)row_lo, j);
                alive &= ~mj;
                pending = alive & cmask & ~((2ull << j) - 1ull);
            }
            int cnt = __popcll(alive);
            while (K + cnt > cap_eff) {                     // cap reached inside this batch: drop the lowest-scored extras
                alive &= ~(1ull << (63 - __clzll((long long)alive)));
                --cnt;
            }
            const bool keep_me = (alive >> lane) & 1ull;   // lane = dense rank; its candidate sits at batch lane dl
            if (wave == 0 && keep_me) {
                const int pos = K + __popcll(alive & lanemask_lt());
                kept_out[pos] = sorted[base + dl];
                if (pos < KEPT_LDS) {
                    kb[pos] = cb[base + dl];
                    if (POL == POL_NUMPY64) kf4[pos] = cf4[base + dl];
                }
            }
            K += cnt;
            // NumPy flows: a kept box whose area is NaN makes every later IoU NaN ("not <= thr"): nothing after it can survive
            if (POL != POL_TF32 && !k.no_nms) {
                const float a_me = keep_me ? cb[base + dl].area : 0.f;
                if (__ballot(keep_me && a_me != a_me)) finished = true;
            }
            __syncthreads();
            PROF_MARK(6)
        }
    }
    PROF_FLUSH(0)
#ifdef SSDHIP_PROFILE
    if (tid == 0) { atomicAdd(&g_prof[12], 1ull); atomicAdd(&g_prof[13], (unsigned long long)consumed); atomicAdd(&g_prof[14], (unsigned long long)K); }
#endif
    if (tid == 0) kept_count[work] = K;
}

// ======================================================================================
// K5
// ======================================================================================
template <typename OutT>
__device__ __forceinline__ void write_row(OutT* out, int* out_idx, int row, int cls, u64 key, const float4* img_boxes,
                                          const DecodeParams& p) {
    const u32 idx = IDX_MASK - (u32)(key & IDX_MASK);
    const float s = key_float((u32)(key >> IDX_BITS));
    const float4 bx = img_boxes[idx];
    OutT* r = out + (size_t)row * 6;
    r[0] = (OutT)cls;
    r[1] = (OutT)s;
    if (p.iou_f32) {             // float32 flow end to end ('corners' + float32 input)
        r[2] = (OutT)(bx.x * (float)p.img_w);
        r[3] = (OutT)(bx.y * (float)p.img_h);
        r[4] = (OutT)(bx.z * (float)p.img_w);
        r[5] = (OutT)(bx.w * (float)p.img_h);
    } else {
        r[2] = (OutT)((double)bx.x * p.img_w);
        r[3] = (OutT)((double)bx.y * p.img_h);
        r[4] = (OutT)((double)bx.z * p.img_w);
        r[5] = (OutT)((double)bx.w * p.img_h);
    }
    if (out_idx) out_idx[row] = (int)idx;
}

// entry e of the class-major concatenation of an image's survivor lists -> group (offs[g] <= e < offs[g+1])
__device__ __forceinline__ int find_group(const int* offs, int G, int e) {
    int lo = 0, hi = G;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// K5 (second generation): one workgroup of TOPK_THREADS per image.  The survivors' composite keys [score (32) | inverted class-major
// position (32)] order them by (score desc, class asc, NMS rank asc) -- tf.nn.top_k's order on the layer's class-major padded array.
//   nothing to cut and no order asked (NumPy semantics, T <= top_k): rows leave in class-major order, the reference's own;
//   otherwise: 8192-bin score histogram -> the bin holding the rows-th best -> collect every key of that bin and above (>= rows
//   keys; more only when scores tie or crowd inside the boundary bin) -> bitonic sort in LDS -> the first `rows` keys, in order.
//   Exact however the scores tie: the keys are unique.  Only when the collected set outgrows the LDS sort buffer (`sort_cap` keys)
//   the first generation's exact radix select over the 64-bit keys runs instead.
constexpr int TOPK_THREADS = 1024;

template <typename OutT>
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(DecodeParams p, const float4* __restrict__ boxes,
                                                            const u64* __restrict__ kept, const int* __restrict__ kept_count,
                                                            const unsigned short* __restrict__ cls_map,
                                                            OutT* __restrict__ out, int* __restrict__ out_count,
                                                            int* __restrict__ out_idx, int sort_cap) {
    constexpr int T5 = TOPK_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* buf = reinterpret_cast<u64*>(smem_raw);                                          // >= 32 KiB: histogram, then keys
    u32* hist = reinterpret_cast<u32*>(smem_raw);
    int* offs = reinterpret_cast<int*>(smem_raw + (size_t)sort_cap * sizeof(u64));        // G+1 ints
    __shared__ int red[260];
    __shared__ int fill;
    __shared__ int wave_tot[T5 / 64];
    __shared__ u32 tie_lo, tie_hi;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.G;
    const float4* img_boxes = boxes + (size_t)b * p.N;
    const u64* img_kept = kept + (size_t)b * G * p.cap_store;
    OutT* img_out = out + (size_t)b * p.out_rows * 6;
    int* img_idx = out_idx ? out_idx + (size_t)b * p.out_rows : nullptr;
    const unsigned short* img_cls = p.class_agnostic ? cls_map + (size_t)b * p.N : nullptr;

    // class-major offsets: block scan of the survivor counts
    {
        int run = 0;
        for (int g0 = 0; g0 < G; g0 += T5) {
            const int g = g0 + tid;
            const int c = g < G ? kept_count[b * G + g] : 0;
            int inc = c;                                         // inclusive scan inside the wave
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
            if (lane == 63) wave_tot[wave] = inc;
            __syncthreads();
            int before = run;
            for (int w = 0; w < wave; ++w) before += wave_tot[w];
            if (g < G) offs[g] = before + inc - c;
            for (int w = 0; w < T5 / 64; ++w) run += wave_tot[w];
            __syncthreads();
        }
        if (tid == 0) offs[G] = run;
    }
    PROF_DECL
    __syncthreads();
    PROF_MARK(0)
    const int T = offs[G];
    int rows = p.top_k > 0 ? min(T, p.top_k) : T;
    rows = min(rows, p.out_rows);

    auto comp_of = [&](int g, int r) -> u64 {
        const u64 key = img_kept[(size_t)g * p.cap_store + r];
        const u32 pos = (u32)g * (u32)p.cap_store + (u32)r;
        return ((key >> IDX_BITS) << 32) | (u64)(0xffffffffu - pos);
    };
    auto comp_at = [&](int e) -> u64 {
        const int g = find_group(offs, G, e);
        return comp_of(g, e - offs[g]);
    };
    auto emit = [&](int row, u64 comp) {
        const u32 pos = 0xffffffffu - (u32)comp;
        const int g = (int)(pos / (u32)p.cap_store);
        const u64 key = img_kept[(size_t)g * p.cap_store + (pos - (u32)g * (u32)p.cap_store)];
        const int cls = p.class_agnostic ? (int)img_cls[IDX_MASK - (u32)(key & IDX_MASK)] : g + 1;
        write_row<OutT>(img_out, img_idx, row, cls, key, img_boxes, p);
    };

    const bool need_cut = T > rows;
    if (rows > 0 && !need_cut && !p.sorted) {
        // the reference's order when nothing is cut: class ascending, NMS order inside a class
        for (int e = tid; e < T; e += T5) emit(e, comp_at(e));
    } else if (rows > 0) {
        // up to PL * T5 survivors (every SSD300 image: 20 classes x 200) are read ONCE, PL independent loads per thread, and
        // stay in registers for the histogram and the collection
        constexpr int PL = 4;
        const bool in_regs = T <= PL * T5;
        u64 cm[PL];
#pragma unroll
        for (int u = 0; u < PL; ++u) { const int e = tid + u * T5; cm[u] = (in_regs && e < T) ? comp_at(e) : 0ull; }
        int bin_cut = 0, c = T, above = 0;
        if (need_cut) {
            for (int i = tid; i < NBINS; i += T5) hist[i] = 0;
            __syncthreads();
            if (in_regs) {
#pragma unroll
                for (int u = 0; u < PL; ++u)
                    if (cm[u]) atomicAdd(&hist[bin_of<DIGIT_BITS, NBINS>((u32)(cm[u] >> 32), p.thr_key)], 1u);
            } else {
                for (int e = tid; e < T; e += T5)
                    atomicAdd(&hist[bin_of<DIGIT_BITS, NBINS>((u32)(comp_at(e) >> 32), p.thr_key)], 1u);
            }
            __syncthreads();
            block_find_digit<NBINS / T5>(hist, rows, red, red + 256);
            bin_cut = red[256];
            above = red[257];
            c = above + (int)hist[bin_cut];                       // keys in the boundary bin and above
            __syncthreads();
        }
        PROF_MARK(1)
        // Many EQUAL scores at the cut (a saturated softmax: hundreds or thousands of survivors at exactly 1.0): if every key of the
        // boundary bin carries the same score, nothing needs sorting there -- the composite key orders equal scores by class-major
        // position, which is the order the survivors are enumerated in (e = tid + u T5).  The rows above the bin are ranked by
        // counting among themselves (fewer than `rows` of them), the ties take the remaining rows in enumeration order: a block
        // scan instead of a 4096-key bitonic sort (42 -> ~15 us on the random-init workload).
        bool done = false;
        if (need_cut && in_regs && c > 512 && above <= 512) {
            if (tid == 0) { tie_lo = 0xffffffffu; tie_hi = 0u; fill = 0; }
            __syncthreads();
            bool in_bin[PL];
            u32 lo = 0xffffffffu, hi = 0u;
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                in_bin[u] = cm[u] != 0ull && bin_of<DIGIT_BITS, NBINS>((u32)(cm[u] >> 32), p.thr_key) == bin_cut;
                if (in_bin[u]) { lo = min(lo, (u32)(cm[u] >> 32)); hi = max(hi, (u32)(cm[u] >> 32)); }
            }
            for (int off = 32; off > 0; off >>= 1) {          // one LDS atomic per wave, not two per key (thousands on one address)
                lo = min(lo, (u32)__shfl_xor((int)lo, off));
                hi = max(hi, (u32)__shfl_xor((int)hi, off));
            }
            if (lane == 0) { atomicMin(&tie_lo, lo); atomicMax(&tie_hi, hi); }
            __syncthreads();
            if (tie_lo == tie_hi) {                               // block-uniform
                done = true;
                // (a) the keys above the bin: collected, ranked by counting
                int Pc = 128;
                while (Pc < above) Pc <<= 1;
                for (int i = tid; i < Pc; i += T5) buf[i] = 0ull;
                u32* rank = reinterpret_cast<u32*>(buf + T5);
                if (tid < Pc) rank[tid] = 0u;
                __syncthreads();
#pragma unroll
                for (int u = 0; u < PL; ++u)
                    if (cm[u] != 0ull && !in_bin[u] && bin_of<DIGIT_BITS, NBINS>((u32)(cm[u] >> 32), p.thr_key) > bin_cut)
                        buf[atomicAdd(&fill, 1)] = cm[u];
                __syncthreads();
                if (above > 0) {
                    const int parts = T5 / Pc, len = Pc / parts;
                    const int ki = tid & (Pc - 1), part = tid / Pc;
                    const u64 mine = buf[ki];
                    u32 r = 0;
                    for (int j = part * len; j < part * len + len; j += 8) {
                        u64 o8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) o8[u] = buf[j + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) r += (u32)(o8[u] > mine);
                    }
                    if (mine != 0ull && r) atomicAdd(&rank[ki], r);
                    __syncthreads();
                    if (tid < Pc && mine != 0ull) emit((int)rank[tid], mine);
                }
                // (b) the ties, in enumeration order
                int base = above;
#pragma unroll
                for (int u = 0; u < PL; ++u) {
                    const u64 m = __ballot(in_bin[u]);
                    __syncthreads();                              // wave_tot is reused
                    if (lane == 0) wave_tot[wave] = __popcll(m);
                    __syncthreads();
                    int off = base;
                    for (int w = 0; w < wave; ++w) off += wave_tot[w];
                    const int row = off + __popcll(m & lanemask_lt());
                    if (in_bin[u] && row < rows) emit(row, cm[u]);
                    for (int w = 0; w < T5 / 64; ++w) base += wave_tot[w];
                }
                PROF_MARK(2)
                PROF_MARK(3)
            }
        }
        u64 cutoff = 0;
        bool by_bin = true;
        if (!done && c > sort_cap) {
            // the boundary bin is too crowded for the sort buffer: exact selection of the rows-th largest key instead
            cutoff = block_select_kth<64, DIGIT_BITS, T5>(comp_at, T, 0, false, rows, hist, red);
            by_bin = false;
            c = rows;
        }
        if (done) {
        } else if (c <= sort_cap) {
            int P = 2;
            while (P < c) P <<= 1;
            if (P < T5) P = T5;                                   // the register sorts take T5 or 4 * T5 keys
            else if (P > T5 && P < 4 * T5) P = 4 * T5;
            for (int i = tid; i < P; i += T5) buf[i] = 0ull;
            if (tid == 0) fill = 0;
            __syncthreads();
            if (in_regs) {
#pragma unroll
                for (int u = 0; u < PL; ++u) {
                    const u64 v = cm[u];
                    const bool take = v != 0ull && (by_bin ? (bin_of<DIGIT_BITS, NBINS>((u32)(v >> 32), p.thr_key) >= bin_cut) : (v >= cutoff));
                    if (take) buf[atomicAdd(&fill, 1)] = v;
                }
            } else {
                for (int e = tid; e < T; e += T5) {
                    const u64 v = comp_at(e);
                    const bool take = by_bin ? (bin_of<DIGIT_BITS, NBINS>((u32)(v >> 32), p.thr_key) >= bin_cut) : (v >= cutoff);
                    if (take) buf[atomicAdd(&fill, 1)] = v;
                }
            }
            __syncthreads();
            PROF_MARK(2)
            if (P == T5 && c <= 512) {
                // the usual case, a few hundred keys around the cut: rank by counting -- T5 / Pc threads share a key, each compares it
                // with a slice of the list (LDS broadcast reads), the partial ranks meet in an LDS counter; no barrier-per-step sort
                int Pc = 128;
                while (Pc < c) Pc <<= 1;
                const int parts = T5 / Pc, len = Pc / parts;          // Pc in {128..512}: parts in {8..2}, len = Pc^2 / T5 >= 16
                u32* rank = reinterpret_cast<u32*>(buf + T5);         // buf holds >= 4096 u64: ranks live behind the first T5 keys
                if (tid < Pc) rank[tid] = 0u;
                __syncthreads();
                const int ki = tid & (Pc - 1), part = tid / Pc;
                const u64 mine = buf[ki];
                u32 r = 0;
                for (int j = part * len; j < part * len + len; j += 8) {
                    u64 o8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) o8[u] = buf[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) r += (u32)(o8[u] > mine);
                }
                if (mine != 0ull && r) atomicAdd(&rank[ki], r);
                __syncthreads();
                PROF_MARK(3)
                if (tid < Pc && mine != 0ull && (int)rank[tid] < rows) emit((int)rank[tid], mine);
            } else if (P == T5) {                                 // up to 1024 keys
            } else if (P == 4 * T5) {                             // many equal scores at the cut (saturated softmax)
                u64 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = buf[tid * 4 + r];
                block_bitonic_desc_regs<T5, 4>(v, buf);
                PROF_MARK(3)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tid * 4 + r < rows) emit(tid * 4 + r, v[r]);
            } else {                                              // more than 4096 keys: the sort runs in LDS
                block_bitonic_desc<T5>(buf, P);
                PROF_MARK(3)
                for (int r = tid; r < rows; r += T5) emit(r, buf[r]);
            }
        } else {
            // more rows than the sort buffer holds (an uncapped NumPy-semantics decode that still cuts): class-major compaction of
            // the keys >= cutoff by a block scan
            int base = 0;
            for (int e0 = 0; e0 < T; e0 += T5) {
                const int e = e0 + tid;
                u64 cm2 = 0;
                bool sel = false;
                if (e < T) { cm2 = comp_at(e); sel = cm2 >= cutoff; }
                const u64 m = __ballot(sel);
                if (lane == 0) wave_tot[wave] = __popcll(m);
                __syncthreads();
                int off = base;
                for (int w = 0; w < wave; ++w) off += wave_tot[w];
                if (sel) emit(off + __popcll(m & lanemask_lt()), cm2);
                for (int w = 0; w < T5 / 64; ++w) base += wave_tot[w];
                __syncthreads();
            }
        }
    }
    PROF_MARK(4)
    // zero padding
    for (int i = rows * 6 + tid; i < p.out_rows * 6; i += T5) img_out[i] = (OutT)0;
    if (img_idx) for (int i = rows + tid; i < p.out_rows; i += T5) img_idx[i] = -1;
    if (tid == 0) out_count[b] = rows;
    PROF_MARK(5)
    PROF_FLUSH(32)
}

// ======================================================================================
// host side
// ======================================================================================
struct DecodeWs {
    size_t boxes, cand_count, kept_count, cls, cand, kept, total;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int cap_store_for(int N, int top_k, int nms_cap) {
    int cap = N;                                   // uncapped: every candidate may survive
    if (nms_cap > 0) cap = nms_cap < cap ? nms_cap : cap;
    if (top_k > 0) cap = top_k < cap ? top_k : cap;        // members of the global top-k are within a class's first top_k survivors
    return cap < 1 ? 1 : cap;
}

static DecodeWs decode_ws_layout(int B, int N, int C, int top_k, int nms_cap, int class_agnostic) {
    const size_t G = class_agnostic ? 1 : (size_t)(C - 1);
    const size_t cap = (size_t)cap_store_for(N, top_k, nms_cap);
    DecodeWs w;
    size_t o = 0;
    w.boxes = o;      o = align_up(o + (size_t)B * N * sizeof(float4), 256);
    w.cand_count = o; o = align_up(o + (size_t)B * G * sizeof(int), 256);
    w.kept_count = o; o = align_up(o + (size_t)B * G * sizeof(int), 256);
    w.cls = o;        o = align_up(o + (class_agnostic ? (size_t)B * N * sizeof(unsigned short) : 0), 256);
    w.cand = o;       o = align_up(o + (size_t)B * G * N * sizeof(u64), 256);
    w.kept = o;       o = align_up(o + (size_t)B * G * cap * sizeof(u64), 256);
    w.total = o;
    return w;
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" size_t ssdhip_decode_workspace_bytes(int B, int N, int C, int top_k, int nms_cap, int class_agnostic, int in_dtype) {
    if (B <= 0 || N <= 0 || C < 2 || (in_dtype != SSDHIP_F32 && in_dtype != SSDHIP_F64)) return 0;
    if (in_dtype == SSDHIP_F64) return decode64_workspace_bytes(B, N, C, top_k, nms_cap, class_agnostic);
    return decode_ws_layout(B, N, C, top_k, nms_cap, class_agnostic).total;
}

struct HeadSource {              // non-null: K3 reads the predictor heads instead of y_pred
    const HeadParams* hp;
    const float* anchors_var;
    int tiles;
};

static int decode_run(const HeadSource* heads, int stages, const void* y_pred, int in_dtype, int B, int N, int C,
                      double conf_thresh, double iou_thresh, int top_k, int nms_cap,
                      int class_agnostic, int semantics,
                      int coords, int normalize_coords, double img_height, double img_width,
                      int border_pixels,
                      void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx,
                      void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if ((!y_pred && !heads) || !out || !out_count || B <= 0 || N <= 0 || C < 2 || out_rows <= 0) return SSDHIP_E_BADARG;
    if (in_dtype != SSDHIP_F32 && in_dtype != SSDHIP_F64) return SSDHIP_E_BADARG;
    if (out_dtype != SSDHIP_F32 && out_dtype != SSDHIP_F64) return SSDHIP_E_BADARG;
    if (N > (1 << IDX_BITS) || C > 1025) return SSDHIP_E_BADARG;
    if (coords < 0 || coords > 2 || border_pixels < 0 || border_pixels > 2) return SSDHIP_E_BADARG;
    if (semantics < 0 || semantics > 2) return SSDHIP_E_BADARG;
    if (in_dtype == SSDHIP_F64) {                                   // the all-float64 flow: csrc/ssdhip_decode64.hip
        if (heads || !y_pred) return SSDHIP_E_BADARG;
        return decode64_run(stages, static_cast<const double*>(y_pred), B, N, C, conf_thresh, iou_thresh, top_k, nms_cap,
                            class_agnostic, semantics, coords, normalize_coords, img_height, img_width, border_pixels, out, out_dtype,
                            out_rows, out_count, out_anchor_idx, ws, ws_bytes, stream);
    }
    if (semantics == SSDHIP_SEM_KERAS && coords != SSDHIP_CENTROIDS) return SSDHIP_E_BADARG;  // as the layer (:81-82)
    const int sorted = semantics == SSDHIP_SEM_KERAS;
    if (sorted && (top_k <= 0 || top_k > TOPK_SORT_MAX)) return SSDHIP_E_BADARG;
    const DecodeWs lay = decode_ws_layout(B, N, C, top_k, nms_cap, class_agnostic);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;

    DecodeParams p;
    p.B = B; p.N = N; p.C = C; p.L = C + 12; p.G = class_agnostic ? 1 : C - 1;
    p.no_dma = 0;
    p.class_agnostic = class_agnostic ? 1 : 0;
    p.semantics = semantics; p.coords = coords; p.border = border_pixels;
    // dtype flow of the reference: float32 predictions stay float32 only on the 'corners' path (no convert_coordinates)
    p.iou_f32 = (semantics != SSDHIP_SEM_KERAS && coords == SSDHIP_CORNERS) ? 1 : 0;
    p.thr_inclusive = (class_agnostic && semantics != SSDHIP_SEM_KERAS) ? 1 : 0;   // ssd_output_decoder.py:325 vs layer :180
    {
        // The reference compares float32 scores with the python-float threshold either in float32 (Keras layer,
        // 'corners' flow) or after widening the score to float64.  Both are reproduced by ONE float32 compare
        // against thr_eff:  (double)s >  t  <=>  s >  largest float <= t ;  (double)s >= t  <=>  s >= smallest float >= t.
        const bool cmp_f32 = (semantics == SSDHIP_SEM_KERAS) || p.iou_f32;
        float te = (float)conf_thresh;
        if (!cmp_f32 && conf_thresh == conf_thresh && te - te == 0.f) {
            if (!p.thr_inclusive && (double)te > conf_thresh) te = nextafterf(te, -INFINITY);
            if (p.thr_inclusive && (double)te < conf_thresh) te = nextafterf(te, INFINITY);
        }
        p.thr_eff = te;
        union { float f; unsigned u; } cv;
        cv.f = te;
        const unsigned key = (cv.u & 0x80000000u) ? ~cv.u : (cv.u | 0x80000000u);
        p.thr_key = key;
    }
    p.iou_thresh = iou_thresh;
    p.img_w = normalize_coords ? img_width : 1.0;
    p.img_h = normalize_coords ? img_height : 1.0;
    p.fast_ok = (iou_thresh > 0.0 && iou_thresh < 1e300) ? 1 : 0;
    if (semantics == SSDHIP_SEM_KERAS) p.fast_ok = ((float)iou_thresh >= 1e-30f && (float)iou_thresh < 1e30f) ? 1 : 0;
    p.px_f32 = (semantics == SSDHIP_SEM_KERAS) ? 1 : 0;
    p.no_nms = (iou_thresh == (double)INFINITY) ? 1 : 0;
    // error bound of the float32 evaluation of R = inter - thr*union: <= (6.25 + 18.4 thr) 2^-24 S; twice that
    p.filter_kE = (iou_thresh > 0.0 && iou_thresh <= 16.0) ? (float)((16.0 + 40.0 * iou_thresh) * 0x1p-24 * 1.001) : INFINITY;
    if (p.px_f32 && ((double)(float)p.img_w != p.img_w || (double)(float)p.img_h != p.img_h)) return SSDHIP_E_BADARG;
    p.top_k = top_k; p.cap = nms_cap; p.cap_store = cap_store_for(N, top_k, nms_cap);
    p.out_rows = out_rows; p.sorted = sorted;

    unsigned char* base = static_cast<unsigned char*>(ws);
    float4* boxes = reinterpret_cast<float4*>(base + lay.boxes);
    int* cand_count = reinterpret_cast<int*>(base + lay.cand_count);
    int* kept_count = reinterpret_cast<int*>(base + lay.kept_count);
    unsigned short* cls_map = reinterpret_cast<unsigned short*>(base + lay.cls);
    u64* cand = reinterpret_cast<u64*>(base + lay.cand);
    u64* kept = reinterpret_cast<u64*>(base + lay.kept);

    if (stages & 1) {
    if (zero_async(cand_count, (size_t)B * p.G * sizeof(int), stream) != hipSuccess) return SSDHIP_E_LAUNCH;

    // K3: small tiles (<= 24 KiB of LDS) keep many workgroups in flight per CU to hide the load -> atomic -> store chain
    int TA = 256;
    while (TA > 64 && (size_t)TA * p.L * sizeof(float) > 24 * 1024) TA >>= 1;
    size_t k3_lds = align_up(((size_t)TA * p.L + 4) * sizeof(float), 16) + (size_t)(TA / 64) * p.G * sizeof(int);
    k3_lds = k3_lds > align_up((size_t)TA * p.L * sizeof(float), 1024) ? k3_lds : align_up((size_t)TA * p.L * sizeof(float), 1024);   // the DMA copy writes whole 1 KiB pieces
    { const char* e = getenv("SSDHIP_SCAN_DMA"); p.no_dma = (e && e[0] == '0') ? 1 : 0; }                                            // A/B switch
    if (k3_lds > 160 * 1024) return SSDHIP_E_BADARG;
    if (heads) {
        const int HT = heads->hp->TA;                 // one thread per row of the head tile
        const size_t lds = (head_tile_lds(HT, C) + 15) / 16 * 16 + (size_t)((HT + 63) / 64) * p.G * sizeof(int);
        if (lds > 160 * 1024) return SSDHIP_E_BADARG;
        hipLaunchKernelGGL(scan_heads_kernel, dim3(heads->tiles, B), dim3(HT < 64 ? 64 : HT), lds, stream, *heads->hp,
                           heads->anchors_var, p, boxes, cand, cand_count, cls_map);
    } else if (p.L > 64 && !p.class_agnostic) {
        // rows wider than 64 floats (SSD512 / COCO): the windowed kernel keeps 20 waves per CU in flight instead of 6
        dim3 g3((N + SW_WAVES * 64 - 1) / (SW_WAVES * 64), B);
        hipLaunchKernelGGL(scan_wide_kernel, g3, dim3(SW_WAVES * 64), 0, stream, static_cast<const float*>(y_pred), p, boxes, cand, cand_count);
    } else {
        dim3 g3((N + TA - 1) / TA, B);
        hipLaunchKernelGGL(scan_kernel, g3, dim3(TA), k3_lds, stream, static_cast<const float*>(y_pred), p, boxes, cand,
                           cand_count, cls_map);
    }
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }

    if (stages & 2) {
    const int work = B * p.G;
    const int g4 = ((work + 7) / 8) * 8;
    const int* order = nullptr;
    // workgroup size of K4: 512 threads measured 4-8 % faster than 256 for the layer semantics, equal otherwise (profiles/r02e);
    // SSDHIP_NMS_T overrides (A/B timing)
    static const int nms_env = []() { const char* e = getenv("SSDHIP_NMS_T"); return e ? atoi(e) : 0; }();
    const int nms_t = nms_env ? nms_env : (semantics == SSDHIP_SEM_KERAS ? 512 : 256);
#define SSDHIP_LAUNCH_NMS(POL)                                                                                                            \
    do {                                                                                                                                  \
        if (nms_t >= 512) {                                                                                                               \
            hipLaunchKernelGGL((nms_kernel<POL, 512, false>), dim3(g4), dim3(512), 0, stream, p, boxes, cand, cand_count, order, kept, kept_count, 0); \
            hipLaunchKernelGGL((nms_kernel<POL, 512, true>), dim3(g4), dim3(512), 0, stream, p, boxes, cand, cand_count, order, kept, kept_count, 1); \
        } else {                                                                                                                          \
            hipLaunchKernelGGL((nms_kernel<POL, 256, true>), dim3(g4), dim3(256), 0, stream, p, boxes, cand, cand_count, order, kept, kept_count, 0); \
        }                                                                                                                                 \
    } while (0)
    if (semantics == SSDHIP_SEM_KERAS) SSDHIP_LAUNCH_NMS(POL_TF32);
    else if (p.iou_f32) SSDHIP_LAUNCH_NMS(POL_NUMPY32);
    else SSDHIP_LAUNCH_NMS(POL_NUMPY64);
#undef SSDHIP_LAUNCH_NMS
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }

    if (stages & 4) {
    // LDS sort buffer: every survivor of an image when that fits 128 KiB, never less than the 8192-counter histogram it aliases
    size_t sort_cap = 4096;
    while (sort_cap < (size_t)p.G * (size_t)p.cap_store && sort_cap < 16384) sort_cap <<= 1;
    const size_t k5_lds = sort_cap * sizeof(u64) + align_up((size_t)(p.G + 1) * sizeof(int), 16);
    const void* fn = out_dtype == SSDHIP_F32 ? reinterpret_cast<const void*>(topk_kernel<float>) : reinterpret_cast<const void*>(topk_kernel<double>);
    if (k5_lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k5_lds) != hipSuccess) return SSDHIP_E_LAUNCH;
    if (out_dtype == SSDHIP_F32)
        hipLaunchKernelGGL(topk_kernel<float>, dim3(B), dim3(TOPK_THREADS), k5_lds, stream, p, boxes, kept, kept_count, cls_map,
                           static_cast<float*>(out), out_count, out_anchor_idx, (int)sort_cap);
    else
        hipLaunchKernelGGL(topk_kernel<double>, dim3(B), dim3(TOPK_THREADS), k5_lds, stream, p, boxes, kept, kept_count, cls_map,
                           static_cast<double*>(out), out_count, out_anchor_idx, (int)sort_cap);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    return SSDHIP_OK;
}

extern "C" int ssdhip_decode_detections(const void* y_pred, int in_dtype, int B, int N, int C,
                                        double conf_thresh, double iou_thresh, int top_k, int nms_cap,
                                        int class_agnostic, int semantics,
                                        int coords, int normalize_coords, double img_height, double img_width,
                                        int border_pixels,
                                        void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx,
                                        void* ws, size_t ws_bytes, void* stream) {
    return decode_run(nullptr, 7, y_pred, in_dtype, B, N, C, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics, coords,
                      normalize_coords, img_height, img_width, border_pixels, out, out_dtype, out_rows, out_count,
                      out_anchor_idx, ws, ws_bytes, stream);
}

extern "C" int ssdhip_decode_stages(int stages, const void* y_pred, int in_dtype, int B, int N, int C,
                                    double conf_thresh, double iou_thresh, int top_k, int nms_cap,
                                    int class_agnostic, int semantics,
                                    int coords, int normalize_coords, double img_height, double img_width,
                                    int border_pixels,
                                    void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (stages <= 0 || stages > 7) return SSDHIP_E_BADARG;
    return decode_run(nullptr, stages, y_pred, in_dtype, B, N, C, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics,
                      coords, normalize_coords, img_height, img_width, border_pixels, out, out_dtype, out_rows, out_count,
                      out_anchor_idx, ws, ws_bytes, stream);
}

static int decode_from_heads_any(int src_f32, int n_layers, const void* const* conf_h, const void* const* loc_h,
                                 const void* const* conf_bias_h, const void* const* loc_bias_h, const int* n_anchors_h,
                                 const int* n_boxes_h, const int* conf_stride_h, const int* loc_stride_h, const float* anchors_var,
                                 int B, int N, int C, double conf_thresh, double iou_thresh, int top_k, int nms_cap, int class_agnostic,
                                 int semantics, int coords, int normalize_coords, double img_height, double img_width, int border_pixels,
                                 void* out, int out_dtype, int out_rows, int* out_count, int* out_anchor_idx, void* ws, size_t ws_bytes,
                                 void* stream) {
    if (!anchors_var) return SSDHIP_E_BADARG;
    HeadParams hp;
    HeadSource src;
    src.hp = &hp; src.anchors_var = anchors_var; src.tiles = 0;
    size_t tile_lds = 60 * 1024;
#if defined(SSDHIP_PROFILE)
    if (const char* e = getenv("SSDHIP_HEADS_LDS_KB")) { const int v = atoi(e); if (v >= 8 && v <= 120) tile_lds = (size_t)v * 1024; }   // tile sweep
#endif
    const int rc = head_fill_params(hp, n_layers, conf_h, loc_h, conf_bias_h, loc_bias_h, n_anchors_h, n_boxes_h, conf_stride_h,
                                    loc_stride_h, N, C, tile_lds, &src.tiles, src_f32);
    if (rc != SSDHIP_OK) return rc;
    if (hp.TA < 64) return SSDHIP_E_BADARG;          // C too large for a one-thread-per-row tile
    return decode_run(&src, 7, nullptr, SSDHIP_F32, B, N, C, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics,
                      coords, normalize_coords, img_height, img_width, border_pixels, out, out_dtype, out_rows, out_count,
                      out_anchor_idx, ws, ws_bytes, stream);
}

// DecodeDetections straight from the predictor heads (no y_pred): the arguments of ssdhip_assemble_predictions_strided_bf16
// followed by those of ssdhip_decode_detections (in_dtype is implied: the rows are built in float32).
extern "C" int ssdhip_decode_from_heads(int n_layers, const void* const* conf_h, const void* const* loc_h,
                                        const void* const* conf_bias_h, const void* const* loc_bias_h,
                                        const int* n_anchors_h, const int* n_boxes_h, const int* conf_stride_h,
                                        const int* loc_stride_h, const float* anchors_var, int B, int N, int C,
                                        double conf_thresh, double iou_thresh, int top_k, int nms_cap, int class_agnostic,
                                        int semantics, int coords, int normalize_coords, double img_height, double img_width,
                                        int border_pixels, void* out, int out_dtype, int out_rows, int* out_count,
                                        int* out_anchor_idx, void* ws, size_t ws_bytes, void* stream) {
    return decode_from_heads_any(0, n_layers, conf_h, loc_h, conf_bias_h, loc_bias_h, n_anchors_h, n_boxes_h, conf_stride_h, loc_stride_h,
                                 anchors_var, B, N, C, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics, coords,
                                 normalize_coords, img_height, img_width, border_pixels, out, out_dtype, out_rows, out_count,
                                 out_anchor_idx, ws, ws_bytes, stream);
}

// ... and from FLOAT32 head outputs that carry their bias already (the reference-precision path: ssdhip_conv2d_x3_nhwc_f16 with out_f32).
extern "C" int ssdhip_decode_from_heads_f32(int n_layers, const void* const* conf_h, const void* const* loc_h, const int* n_anchors_h,
                                            const int* n_boxes_h, const int* conf_stride_h, const int* loc_stride_h,
                                            const float* anchors_var, int B, int N, int C, double conf_thresh, double iou_thresh,
                                            int top_k, int nms_cap, int class_agnostic, int semantics, int coords, int normalize_coords,
                                            double img_height, double img_width, int border_pixels, void* out, int out_dtype,
                                            int out_rows, int* out_count, int* out_anchor_idx, void* ws, size_t ws_bytes, void* stream) {
    return decode_from_heads_any(1, n_layers, conf_h, loc_h, nullptr, nullptr, n_anchors_h, n_boxes_h, conf_stride_h, loc_stride_h,
                                 anchors_var, B, N, C, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics, coords,
                                 normalize_coords, img_height, img_width, border_pixels, out, out_dtype, out_rows, out_count,
                                 out_anchor_idx, ws, ws_bytes, stream);
}
