// ssdhip_loss.hip -- the SSD multibox loss (forward + backward) on gfx950 (MI355X).
//
// Replaces SSDLoss.compute_loss (reference keras_loss_function/keras_ssd_loss.py:98-211, with smooth_L1_loss
// :53-75 and log_loss :77-96), float32 like the TensorFlow graph.  Hard-negative mining is global over the
// batch the call is given (:179-183): the k = min(max(ratio*n_pos, n_neg_min), #non-zero negative losses)
// largest negative classification losses of the flattened (B*N) array are kept; among equal losses at the
// k-th place the lowest flat index wins (tf.nn.top_k).
//
// Forward on the caller's stream:
//   L1 anchor_kernel  grid (anchor tiles, B): tiles of y_true / y_pred rows copied coalesced into LDS; per anchor the
//                     log loss (only where y_true != 0), smooth L1, positive / negative weights; writes cls_loss[B,N], neg_all[B,N] and the tile's four partial sums (float64).
//                     No atomics anywhere in the sums: L2b and L4 add the partials in a fixed order, so the loss is
//                     bit-reproducible from run to run.
//   L2 sel_*_kernel   k, then a radix select (11+11+10 bits of the order-preserving float key) for the k-th largest negative
//                     loss: a chip-wide histogram of the top 11 bits (L2a), the pivot digit (L2b), a chip-wide compaction
//                     of that digit's [key | index] pairs (L2c), and one workgroup finishing on the short list (L2d) --
//                     including, only if ties straddle the cut, the flat-index limit.
//   L3 keep_kernel    grid-stride over B*N: keep mask + per-image sum of the kept negative losses.
//   L4 total_kernel   B threads: (pos_cls + neg_cls + alpha*loc) / max(1, n_pos) * B.
// Backward = one kernel with the same LDS tiling writing d loss / d y_pred coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_tile.h"

namespace ssdhip {

constexpr int LOSS_THREADS = 256;
constexpr int SEL_THREADS = 1024;
constexpr int SELG_THREADS = 256;
constexpr int SELG_MAX_BLOCKS = 256;
constexpr int SELC_ITEMS = 8;
constexpr int KEEP_BLOCKS = 64;
constexpr int SEL_BINS = 2048;

struct LossWs {
    size_t sums, counts, hist, sel, cls, neg, list, part, keep_part, total;
    // sums: per-image positive class loss [B] | loc loss [B] | (unused [B]) | n_pos; counts[1]: list length; hist: L2a bins;
    // sel: select state; list: L2c pairs; part: L1's per-tile partial sums [4][B][tiles]; keep_part: L3's [B][KEEP_BLOCKS]
    int tiles;
};

struct SelectResult {
    int k;
    unsigned int thresh_key;      // float_key of the k-th largest negative loss
    int tie_limit;                // elements == thresh are kept iff flat index < tie_limit
    int n_neg_losses;
    float n_pos;
    float thresh;
    int digit;                    // L2b -> L2c/L2d: top 11 key bits of the threshold
    int want;                     //                 how many of that digit's values are kept
};

static inline size_t lalign(size_t v) { return (v + 255) / 256 * 256; }

// anchors per L1 / backward tile: both row tiles ([TA][C+12] of y_true and y_pred) within 64 KB of LDS
static int loss_tile(int L) {
    int TA = 256;
    while (TA > 64 && 2 * ((size_t)TA * L + 8) * sizeof(float) > 64 * 1024) TA >>= 1;
    return TA;
}

static LossWs loss_ws_layout(int B, int N, int C) {
    LossWs w;
    const int TA = loss_tile(C + 12);
    w.tiles = (N + TA - 1) / TA;
    size_t o = 0;
    w.sums = o;   o = lalign(o + (size_t)(3 * B + 1) * sizeof(double));
    w.counts = o; o = lalign(o + 4 * sizeof(int));
    w.hist = o;   o = lalign(o + SEL_BINS * sizeof(u32));
    w.sel = o;    o = lalign(o + sizeof(SelectResult));
    w.cls = o;    o = lalign(o + (size_t)B * N * sizeof(float));
    w.neg = o;    o = lalign(o + (size_t)B * N * sizeof(float));
    w.list = o;   o = lalign(o + (size_t)B * N * sizeof(u64));
    w.part = o;   o = lalign(o + (size_t)4 * B * w.tiles * sizeof(double));
    w.keep_part = o; o = lalign(o + (size_t)B * KEEP_BLOCKS * sizeof(double));
    w.total = o;
    return w;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ======================================================================================
// L1
// ======================================================================================
__global__ __launch_bounds__(LOSS_THREADS) void anchor_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                              int B, int N, int C, float* __restrict__ cls_out,
                                                              float* __restrict__ neg_out, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[4][LOSS_THREADS / 64];
    const int TA = blockDim.x, L = C + 12;
    const int b = blockIdx.y, a0 = blockIdx.x * TA, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = min(TA, N - a0);
    const size_t off = ((size_t)b * N + a0) * (size_t)L;
    float* lds = reinterpret_cast<float*>(smem_raw);
    const size_t half = ((size_t)TA * L + 4 + 3) / 4 * 4;
    const float* yt = tile_copy_f32(lds, y_true + off, na * L, tid, TA);
    const float* yp = tile_copy_f32(lds + half, y_pred + off, na * L, tid, TA);
    __syncthreads();
    double s_poscls = 0.0, s_loc = 0.0, s_npos = 0.0;
    int nonzero = 0;
    if (tid < na) {
        const float* t = yt + (size_t)tid * L;
        const float* q = yp + (size_t)tid * L;
        float cls = 0.f, pos = t[1];
        for (int c = 0; c < C; ++c) {                                  // log_loss (:93-95)
            const float tc = t[c];
            if (c >= 1) pos = fmaxf(pos, tc);                           // positives = max(y_true[1:C]) (:140)
            if (tc != 0.f) cls += tc * logf(fmaxf(q[c], 1e-15f));
        }
        cls = -cls;
        float loc = 0.f;
        for (int k = 0; k < 4; ++k) {                                   // smooth_L1_loss (:72-75)
            const float d = t[C + k] - q[C + k];
            const float ad = fabsf(d);
            loc += ad < 1.0f ? 0.5f * (d * d) : ad - 0.5f;
        }
        const float neg = cls * t[0];                                   // neg_class_loss_all (:150)
        cls_out[(size_t)b * N + a0 + tid] = cls;
        neg_out[(size_t)b * N + a0 + tid] = neg;
        s_poscls = (double)(cls * pos);
        s_loc = (double)(loc * pos);
        s_npos = (double)pos;
        nonzero = neg != 0.f;
    }
    s_poscls = wave_sum(s_poscls); s_loc = wave_sum(s_loc); s_npos = wave_sum(s_npos);
    const double s_nz = wave_sum((double)nonzero);
    if (lane == 0) { red[0][wave] = s_poscls; red[1][wave] = s_loc; red[2][wave] = s_npos; red[3][wave] = s_nz; }
    __syncthreads();
    if (tid == 0) {                                                     // per-tile partial sums, reduced in a fixed order by L2b
        double a = 0, l = 0, n = 0, z = 0;
        for (int w = 0; w < TA / 64; ++w) { a += red[0][w]; l += red[1][w]; n += red[2][w]; z += red[3][w]; }
        const size_t tiles = gridDim.x, slot = (size_t)b * tiles + blockIdx.x, plane = (size_t)B * tiles;
        part[slot] = a; part[plane + slot] = l; part[2 * plane + slot] = n; part[3 * plane + slot] = z;
    }
}

// ======================================================================================
// L2: k and the k-th largest negative loss.  Four short launches; only the two streaming passes use the whole chip.
// ======================================================================================
// L2a: histogram of the top 11 key bits over all B*N values, privatised in LDS per workgroup
__global__ __launch_bounds__(SELG_THREADS) void sel_hist_kernel(const float* __restrict__ neg_all, int total, u32* __restrict__ ghist) {
    __shared__ u32 hist[SEL_BINS];
    const int tid = threadIdx.x;
    for (int i = tid; i < SEL_BINS; i += SELG_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = blockIdx.x * SELG_THREADS + tid; i < total; i += gridDim.x * SELG_THREADS)
        atomicAdd(&hist[float_key(neg_all[i]) >> 21], 1u);
    __syncthreads();
    for (int i = tid; i < SEL_BINS; i += SELG_THREADS) {
        const u32 c = hist[i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

// L2b: k (:166-177) and the top-11-bit digit the k-th largest value falls in
__global__ __launch_bounds__(SEL_THREADS) void sel_pivot_kernel(const u32* __restrict__ ghist, int neg_pos_ratio, int n_neg_min,
                                                                const double* __restrict__ part, int tiles, double* __restrict__ sums,
                                                                int B, SelectResult* __restrict__ res, float* __restrict__ stats) {
    __shared__ u32 hist[SEL_BINS];
    __shared__ int sh_out[2];
    __shared__ int wave_cnt[SEL_THREADS / 64];
    __shared__ double tot[2][SEL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // L1's per-tile partial sums -> per-image sums (one wave each) and the two batch totals: fixed order, no atomics
    const size_t plane = (size_t)B * tiles;
    for (int job = wave; job < 2 * B; job += SEL_THREADS / 64) {         // job = q * B + b, q = 0 positive class loss, 1 loc loss
        double v = 0.0;
        for (int t = lane; t < tiles; t += 64) v += part[(size_t)job * tiles + t];
        v = wave_sum(v);
        if (lane == 0) sums[job] = v;
    }
    double np_ = 0.0, nz_ = 0.0;
    for (size_t i = tid; i < plane; i += SEL_THREADS) { np_ += part[2 * plane + i]; nz_ += part[3 * plane + i]; }
    np_ = wave_sum(np_); nz_ = wave_sum(nz_);
    if (lane == 0) { tot[0][wave] = np_; tot[1][wave] = nz_; }
    __syncthreads();
    np_ = 0.0; nz_ = 0.0;
    for (int w = 0; w < SEL_THREADS / 64; ++w) { np_ += tot[0][w]; nz_ += tot[1][w]; }
    const float n_pos = (float)np_;
    const int n_neg_losses = (int)nz_;
    int k = neg_pos_ratio * (int)n_pos;                                  // tf.to_int32(n_positive) truncates (:166)
    k = k > n_neg_min ? k : n_neg_min;
    k = k < n_neg_losses ? k : n_neg_losses;
    if (k > 0) {
        for (int i = tid; i < SEL_BINS; i += SEL_THREADS) hist[i] = ghist[i];
        __syncthreads();
        block_find_digit<SEL_BINS / SEL_THREADS>(hist, k, wave_cnt, sh_out);
    }
    if (tid == 0) {
        sums[3 * B] = np_;
        res->k = k; res->thresh_key = 0; res->tie_limit = 0x7fffffff; res->n_neg_losses = n_neg_losses;
        res->n_pos = n_pos; res->thresh = 0.f;
        res->digit = k > 0 ? sh_out[0] : 0;
        res->want = k > 0 ? k - sh_out[1] : 0;
        stats[0] = n_pos; stats[1] = (float)n_neg_losses; stats[2] = (float)k; stats[3] = 0.f;
    }
}

// L2c: the values of that digit, as [key | flat index] pairs, appended to a list (order does not matter): each workgroup
// takes SELC_ITEMS * 256 values, counts its matches, reserves its slice of the list with one atomic
__global__ __launch_bounds__(SELG_THREADS) void sel_compact_kernel(const float* __restrict__ neg_all, int total,
                                                                   const SelectResult* __restrict__ res, int* __restrict__ list_count,
                                                                   u64* __restrict__ list) {
    __shared__ int wave_tot[SELG_THREADS / 64];
    __shared__ int base_sh;
    if (res->k <= 0) return;
    const u32 digit = (u32)res->digit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * (SELG_THREADS * SELC_ITEMS) + tid;
    u32 keys[SELC_ITEMS];
    u32 hit = 0;
#pragma unroll
    for (int j = 0; j < SELC_ITEMS; ++j) {
        const int i = i0 + j * SELG_THREADS;
        keys[j] = i < total ? float_key(neg_all[i]) : 0u;
        if (i < total && (keys[j] >> 21) == digit) hit |= 1u << j;
    }
    const int cnt = __popc(hit);
    int incl = cnt;                                                     // inclusive prefix over the lanes of this wave
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < SELG_THREADS / 64; ++w) t += wave_tot[w];
        base_sh = t ? atomicAdd(list_count, t) : 0;
    }
    __syncthreads();
    int pos = base_sh + incl - cnt;
    for (int w = 0; w < wave; ++w) pos += wave_tot[w];
#pragma unroll
    for (int j = 0; j < SELC_ITEMS; ++j)
        if (hit & (1u << j)) list[pos++] = ((u64)keys[j] << 32) | (u64)(u32)(i0 + j * SELG_THREADS);
}

// L2d: one workgroup finishes on the list: the remaining 21 key bits, then -- only if ties straddle the cut -- the flat
// index limit: the want-th smallest index among the ties, found by the same radix select on the inverted index.
__global__ __launch_bounds__(SEL_THREADS) void sel_finish_kernel(const u64* __restrict__ list, const int* __restrict__ list_count,
                                                                 SelectResult* __restrict__ res, float* __restrict__ stats) {
    __shared__ u32 hist[SEL_BINS];
    __shared__ int sh_out[2];
    __shared__ int wave_cnt[SEL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (res->k <= 0) return;
    const int n = *list_count;
    int want = res->want;
    u32 prefix = (u32)res->digit << 21, pmask = 0x7ffu << 21;
    {
        const int shifts[2] = {10, 0};
        const u32 masks[2] = {0x7ffu, 0x3ffu};
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = tid; i < SEL_BINS; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += SEL_THREADS) {
                const u32 key = (u32)(list[i] >> 32);
                if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & masks[pass]], 1u);
            }
            __syncthreads();
            block_find_digit<SEL_BINS / SEL_THREADS>(hist, want, wave_cnt, sh_out);
            want -= sh_out[1];
            prefix |= (u32)sh_out[0] << shifts[pass];
            pmask |= masks[pass] << shifts[pass];
            __syncthreads();
        }
    }
    // `want` of the elements equal to the threshold are kept; if that is not all of them, the lowest flat indices win
    int eq_local = 0;
    for (int i = tid; i < n; i += SEL_THREADS) eq_local += (u32)(list[i] >> 32) == prefix;
    eq_local = (int)wave_sum((double)eq_local);
    if (lane == 0) wave_cnt[wave] = eq_local;
    __syncthreads();
    int eq_total = 0;
    for (int w = 0; w < SEL_THREADS / 64; ++w) eq_total += wave_cnt[w];
    __syncthreads();
    int tie_limit = 0x7fffffff;
    if (eq_total != want) {
        u32 ip = 0, im = 0;
        int w2 = want;
        const int shifts[3] = {21, 10, 0};
        const u32 masks[3] = {0x7ffu, 0x7ffu, 0x3ffu};
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = tid; i < SEL_BINS; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += SEL_THREADS) {
                const u64 e = list[i];
                const u32 inv = ~(u32)e;
                if ((u32)(e >> 32) == prefix && (inv & im) == ip) atomicAdd(&hist[(inv >> shifts[pass]) & masks[pass]], 1u);
            }
            __syncthreads();
            block_find_digit<SEL_BINS / SEL_THREADS>(hist, w2, wave_cnt, sh_out);
            w2 -= sh_out[1];
            ip |= (u32)sh_out[0] << shifts[pass];
            im |= masks[pass] << shifts[pass];
            __syncthreads();
        }
        tie_limit = (int)(~ip) + 1;                                     // one past the want-th tie
    }
    if (tid == 0) {
        res->thresh_key = prefix; res->tie_limit = tie_limit; res->thresh = key_float(prefix);
        stats[3] = res->thresh;
    }
}

// ======================================================================================
// L3 / L4
// ======================================================================================
__global__ __launch_bounds__(LOSS_THREADS) void keep_kernel(const float* __restrict__ cls, const float* __restrict__ neg_all,
                                                            int B, int N, const SelectResult* __restrict__ res,
                                                            unsigned char* __restrict__ keep, double* __restrict__ keep_part) {
    __shared__ double red[LOSS_THREADS / 64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = res->k, tie_limit = res->tie_limit;
    const u32 tk = res->thresh_key;
    double s = 0.0;
    for (int n = blockIdx.x * LOSS_THREADS + tid; n < N; n += gridDim.x * LOSS_THREADS) {
        const int flat = b * N + n;
        bool kp = false;
        if (k > 0) {
            const u32 key = float_key(neg_all[flat]);
            kp = key > tk || (key == tk && flat < tie_limit);
        }
        keep[flat] = kp ? 1 : 0;
        if (kp) s += (double)cls[flat];                                 // classification_loss * negatives_keep (:190)
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        double a = 0;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) a += red[w];
        keep_part[(size_t)b * KEEP_BLOCKS + blockIdx.x] = a;              // summed in a fixed order by L4
    }
}

__global__ void total_kernel(const double* __restrict__ sums, const double* __restrict__ keep_part, int keep_blocks, int B, float alpha,
                             float* __restrict__ loss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float n_pos = (float)sums[3 * B];
    double neg = 0.0;
    for (int i = 0; i < keep_blocks; ++i) neg += keep_part[(size_t)b * KEEP_BLOCKS + i];
    const float cls = (float)sums[b] + (float)neg;                        // class_loss = pos + neg (:200)
    const float tot = (cls + alpha * (float)sums[B + b]) / fmaxf(1.0f, n_pos);
    loss[b] = tot * (float)B;                                             // :208-209
}

// ======================================================================================
// backward
// ======================================================================================
__global__ __launch_bounds__(LOSS_THREADS) void backward_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                                const unsigned char* __restrict__ keep,
                                                                const float* __restrict__ stats, const float* __restrict__ grad_out,
                                                                int B, int N, int C, float alpha, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int TA = blockDim.x, L = C + 12;
    const int b = blockIdx.y, a0 = blockIdx.x * TA, tid = threadIdx.x;
    const int na = min(TA, N - a0);
    const size_t off = ((size_t)b * N + a0) * (size_t)L;
    float* lds = reinterpret_cast<float*>(smem_raw);
    const size_t half = ((size_t)TA * L + 4 + 3) / 4 * 4;
    const float* yt = tile_copy_f32(lds, y_true + off, na * L, tid, TA);
    float* yp = tile_copy_f32(lds + half, y_pred + off, na * L, tid, TA);
    __syncthreads();
    const float scale = grad_out[b] * (float)B / fmaxf(1.0f, stats[0]);
    if (tid < na) {
        const float* t = yt + (size_t)tid * L;
        float* q = yp + (size_t)tid * L;                                  // overwritten with the gradient row
        float pos = t[1];
        for (int c = 2; c < C; ++c) pos = fmaxf(pos, t[c]);
        const float w_cls = (pos + (keep[(size_t)b * N + a0 + tid] ? 1.0f : 0.0f)) * scale;
        for (int c = 0; c < C; ++c) {
            const float pc = q[c];
            q[c] = (pc >= 1e-15f && t[c] != 0.f) ? -(t[c] / pc) * w_cls : 0.f;
        }
        const float w_loc = alpha * pos * scale;
        for (int k = 0; k < 4; ++k) {
            const float d = t[C + k] - q[C + k];
            const float dl = fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f);
            q[C + k] = -dl * w_loc;
        }
        for (int k = 4; k < 12; ++k) q[C + k] = 0.f;
    }
    __syncthreads();
    const int total = na * L;
    for (int i = tid; i < total; i += TA) grad[off + i] = yp[i];
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" size_t ssdhip_loss_workspace_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C < 2) return 0;
    return loss_ws_layout(B, N, C).total;
}

extern "C" int ssdhip_loss_forward(const float* y_true, const float* y_pred, int B, int N, int C,
                                   int neg_pos_ratio, int n_neg_min, float alpha,
                                   float* loss_per_item, float* stats, unsigned char* keep_mask,
                                   void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!y_true || !y_pred || !loss_per_item || !stats || !keep_mask || B <= 0 || N <= 0 || C < 2) return SSDHIP_E_BADARG;
    if ((long long)B * N > 0x7ffffff0LL) return SSDHIP_E_BADARG;
    const LossWs lay = loss_ws_layout(B, N, C);
    if (!ws || ws_bytes < lay.total) return SSDHIP_E_WORKSPACE;
    unsigned char* base = static_cast<unsigned char*>(ws);
    double* sums = reinterpret_cast<double*>(base + lay.sums);
    int* counts = reinterpret_cast<int*>(base + lay.counts);
    SelectResult* sel = reinterpret_cast<SelectResult*>(base + lay.sel);
    float* cls = reinterpret_cast<float*>(base + lay.cls);
    float* neg = reinterpret_cast<float*>(base + lay.neg);
    u32* ghist = reinterpret_cast<u32*>(base + lay.hist);
    u64* list = reinterpret_cast<u64*>(base + lay.list);
    double* part = reinterpret_cast<double*>(base + lay.part);
    double* keep_part = reinterpret_cast<double*>(base + lay.keep_part);
    if (hipMemsetAsync(base, 0, lay.sel, stream) != hipSuccess) return SSDHIP_E_LAUNCH;    // list length + L2a bins (the sums are plain stores)

    const int L = C + 12;
    const int TA = loss_tile(L);
    const size_t lds = 2 * (((size_t)TA * L + 4 + 3) / 4 * 4) * sizeof(float) + 16;
    if (lds > 150 * 1024) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(anchor_kernel, dim3(lay.tiles, B), dim3(TA), lds, stream, y_true, y_pred, B, N, C, cls, neg, part);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const int total = B * N;
    int sel_blocks = (total + SELG_THREADS - 1) / SELG_THREADS;
    if (sel_blocks > SELG_MAX_BLOCKS) sel_blocks = SELG_MAX_BLOCKS;
    const int compact_blocks = (total + SELG_THREADS * SELC_ITEMS - 1) / (SELG_THREADS * SELC_ITEMS);
    hipLaunchKernelGGL(sel_hist_kernel, dim3(sel_blocks), dim3(SELG_THREADS), 0, stream, neg, total, ghist);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(sel_pivot_kernel, dim3(1), dim3(SEL_THREADS), 0, stream, ghist, neg_pos_ratio, n_neg_min, part, lay.tiles, sums, B, sel,
                       stats);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(sel_compact_kernel, dim3(compact_blocks), dim3(SELG_THREADS), 0, stream, neg, total, sel, counts + 1, list);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(sel_finish_kernel, dim3(1), dim3(SEL_THREADS), 0, stream, list, counts + 1, sel, stats);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    int gx = (N + LOSS_THREADS - 1) / LOSS_THREADS;
    if (gx > KEEP_BLOCKS) gx = KEEP_BLOCKS;
    hipLaunchKernelGGL(keep_kernel, dim3(gx, B), dim3(LOSS_THREADS), 0, stream, cls, neg, B, N, sel, keep_mask, keep_part);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(total_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, sums, keep_part, gx, B, alpha, loss_per_item);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    return SSDHIP_OK;
}

extern "C" int ssdhip_loss_backward(const float* y_true, const float* y_pred, const unsigned char* keep_mask,
                                    const float* stats, const float* grad_out, int B, int N, int C, float alpha,
                                    float* grad_y_pred, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!y_true || !y_pred || !keep_mask || !stats || !grad_out || !grad_y_pred || B <= 0 || N <= 0 || C < 2) return SSDHIP_E_BADARG;
    const int L = C + 12;
    const int TA = loss_tile(L);
    const size_t lds = 2 * (((size_t)TA * L + 4 + 3) / 4 * 4) * sizeof(float) + 16;
    if (lds > 150 * 1024) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(backward_kernel, dim3((N + TA - 1) / TA, B), dim3(TA), lds, stream, y_true, y_pred, keep_mask, stats, grad_out,
                       B, N, C, alpha, grad_y_pred);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    return SSDHIP_OK;
}
