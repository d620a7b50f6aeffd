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
//   L2 sel_pass*      k, then a radix select (8+12+12 bits of the order-preserving float key) for the k-th largest negative
//                     loss: level 1's histogram comes out of L1 itself; two chip-wide passes over neg_all build levels 2 and 3,
//                     each finding the previous level's digit redundantly in every block.  No single-workgroup step, no list.
//   L3 keep_kernel    the last digit and -- only if ties straddle the cut -- the flat-index limit (from per-block level-3
//                     counts), then grid-stride over B*N: keep mask + per-image sum of the kept negative losses.
//   L4 (folded into L3: the last keep block of an image to arrive): (pos_cls + neg_cls + alpha*loc) / max(1, n_pos) * B.
// Backward = one kernel with the same LDS tiling writing d loss / d y_pred coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ssdhip.h"
#include "ssdhip_math.h"
#include "ssdhip_tile.h"

namespace ssdhip {

constexpr int LOSS_THREADS = 256;
constexpr int KEEP_BLOCKS = 64;
constexpr int L1_BINS = 256, L1_SHARDS = 8;         // level 1: the top 8 key bits, counted by L1 itself into 8 global copies (block & 7)
constexpr int SEL_BINS = 4096;                      // levels 2 and 3: 12 bits each
constexpr int SELP_THREADS = 256;                   // a select pass block: 256 threads x 8 values = SELP_CHUNK consecutive flat indices
constexpr int SELP_ITEMS = 8;
constexpr int SELP_CHUNK = SELP_THREADS * SELP_ITEMS;

struct LossWs {
    size_t sums, hist, sel, cls, neg, part, keep_part, bcol, total;
    // sums: per-image positive class loss [B] | loc loss [B] | (unused [B]) | n_pos; hist: level 1's bins [L1_SHARDS][L1_BINS], then
    // levels 2 and 3 [2][SEL_BINS] (zeroed per call); sel: select state; part: L1's per-tile partial sums [4][B][tiles];
    // keep_part: L3's [B][KEEP_BLOCKS]; bcol: level-3 bin counts per pass block [nblk][SEL_BINS] u16 (where among the ties the cut falls)
    int tiles, nblk;
};

struct SelectResult {
    int k;
    unsigned int thresh_key;      // float_key of the k-th largest negative loss
    int tie_limit;                // elements == thresh are kept iff flat index < tie_limit
    int n_neg_losses;
    float n_pos;
    float thresh;
    int digit;                    // level 1: top 8 key bits of the threshold
    int want;                     //          how many of that digit's values are kept
    int digit2, want2;            // level 2: the next 12 bits, how many of the 20-bit prefix's values are kept
};

static inline size_t lalign(size_t v) { return (v + 255) / 256 * 256; }

// anchors per L1 / backward tile: both row tiles ([TA][C+12] of y_true and y_pred) within 64 KB of LDS
static int loss_tile(int L) {
    int TA = 256;
#if defined(SSDHIP_PROFILE)
    if (const char* e = getenv("SSDHIP_LOSS_TA")) { const int v = atoi(e); if (v == 64 || v == 128 || v == 256) TA = v; }   // tile sweep (tools/time_loss.py)
#endif
    while (TA > 64 && 2 * ((size_t)TA * L + 8) * sizeof(float) > 64 * 1024) TA >>= 1;
    return TA;
}

static LossWs loss_ws_layout(int B, int N, int C) {
    LossWs w;
    const int TA = loss_tile(C + 12);
    w.tiles = (N + TA - 1) / TA;
    w.nblk = (int)(((long long)B * N + SELP_CHUNK - 1) / SELP_CHUNK);
    size_t o = 0;
    w.hist = o;   o = lalign(o + ((size_t)L1_SHARDS * L1_BINS + 2 * SEL_BINS + (size_t)B) * sizeof(u32));   // first: one zero-fill covers the bins and the B arrival counters of L3
    w.sums = o;   o = lalign(o + (size_t)(3 * B + 1) * sizeof(double));
    w.sel = o;    o = lalign(o + sizeof(SelectResult));
    w.cls = o;    o = lalign(o + (size_t)B * N * sizeof(float));
    w.neg = o;    o = lalign(o + (size_t)B * N * sizeof(float));
    w.part = o;   o = lalign(o + (size_t)4 * B * ((N + 63) / 64) * sizeof(double));      // the streaming L1 writes per 64 anchors, the tiled one per TA >= 64
    w.keep_part = o; o = lalign(o + (size_t)B * KEEP_BLOCKS * sizeof(double));
    w.bcol = o;   o = lalign(o + (size_t)w.nblk * SEL_BINS * sizeof(unsigned short));
    w.total = o;
    return w;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// hist[bin] += 1 for every active lane, with the lanes of a wave that share a bin combined into one LDS atomic: mined losses
// cluster (every clipped loss is -log(1e-15); a saturated background scores the same few values), and 64 lanes on one LDS
// address serialise.  Up to two rounds of "everybody who shares the first remaining lane's bin", then plain atomics.
__device__ __forceinline__ void hist_add_aggregated(u32* hist, u32 bin, bool active) {
    u64 todo = __ballot(active);
#pragma unroll 1
    for (int round = 0; round < 2 && todo; ++round) {
        const int leader = (int)__builtin_ctzll(todo);
        const u32 b0 = (u32)__shfl((int)bin, leader);
        const u64 same = __ballot(active && bin == b0) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (u32)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> (threadIdx.x & 63)) & 1ull) atomicAdd(&hist[bin], 1u);
}

// ======================================================================================
// L1
// ======================================================================================
__global__ __launch_bounds__(LOSS_THREADS) void anchor_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                              int B, int N, int C, float* __restrict__ cls_out,
                                                              float* __restrict__ neg_out, double* __restrict__ part,
                                                              u32* __restrict__ ghist1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[4][LOSS_THREADS / 64];
    __shared__ u32 hist[L1_BINS];                                       // level 1 of the hard-negative select: top 8 key bits of neg_all
    const int TA = blockDim.x, L = C + 12;
    const int b = blockIdx.y, a0 = blockIdx.x * TA, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = min(TA, N - a0);
    const size_t off = ((size_t)b * N + a0) * (size_t)L;
    float* lds = reinterpret_cast<float*>(smem_raw);
    const size_t half = ((size_t)TA * L + 4 + 3) / 4 * 4;
    for (int i = tid; i < L1_BINS; i += TA) hist[i] = 0u;
    const float* yt = tile_copy_f32(lds, y_true + off, na * L, tid, TA);
    const float* yp = tile_copy_f32(lds + half, y_pred + off, na * L, tid, TA);
    __syncthreads();
    double s_poscls = 0.0, s_loc = 0.0, s_npos = 0.0;
    int nonzero = 0;
    u32 bin = 0;
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
        bin = float_key(neg) >> 24;
    }
    hist_add_aggregated(hist, bin, tid < na);
    s_poscls = wave_sum(s_poscls); s_loc = wave_sum(s_loc); s_npos = wave_sum(s_npos);
    const double s_nz = wave_sum((double)nonzero);
    if (lane == 0) { red[0][wave] = s_poscls; red[1][wave] = s_loc; red[2][wave] = s_npos; red[3][wave] = s_nz; }
    __syncthreads();
    u32* gshard = ghist1 + ((blockIdx.x + blockIdx.y) & (L1_SHARDS - 1)) * L1_BINS;     // 8 copies: an eighth of the contention on a hot bin
    for (int i = tid; i < L1_BINS; i += TA) {                           // integer atomics: the order does not show in the result
        const u32 c = hist[i];
        if (c) atomicAdd(&gshard[i], c);
    }
    if (tid == 0) {                                                     // per-tile partial sums, reduced in a fixed order by L2
        double a = 0, l = 0, n = 0, z = 0;
        for (int w = 0; w < TA / 64; ++w) { a += red[0][w]; l += red[1][w]; n += red[2][w]; z += red[3][w]; }
        const size_t tiles = gridDim.x, slot = (size_t)b * tiles + blockIdx.x, plane = (size_t)B * tiles;
        part[slot] = a; part[plane + slot] = l; part[2 * plane + slot] = n; part[3 * plane + slot] = z;
    }
}

// ======================================================================================
// L2: k and the k-th largest negative loss by a three-level radix select (8 + 12 + 12 bits of the order-preserving float key).
// Every level is a CHIP-WIDE histogram pass over neg_all (1.1 MB at SSD300 / batch 32: L2-resident); the digit of level l is found
// from level l's bins by EVERY block of the next pass (a suffix scan over the bins: cheaper than a launch of its own).  Nothing runs on
// one workgroup, and nothing depends on how the values are distributed: when every mined loss ties (random-init predictions: all
// clipped at -log(1e-15)), the first two generations' single-workgroup finish over a 279 k-entry list took 0.28 ms.
// Ties straddling the cut ("the lowest flat indices win", tf.nn.top_k) are resolved from the per-block level-3 bin counts the
// last pass leaves behind: which block holds the want-th tie is a scan over nblk numbers, where inside it a scan over its 2048 values.
// ======================================================================================
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot, int& total) {   // blockDim.x == 256; all threads call it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    __syncthreads();                                                    // wave_tot may still be read from a previous call
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, t = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wave_tot[w]; t += wave_tot[w]; }
    total = t;
    return base + incl - v;
}

// L2a: k (:166-177) + the level-1 digit (every block), per-image sums and the select state (block 0); level-2 histogram of the
// values whose top 8 bits equal that digit
__global__ __launch_bounds__(SELP_THREADS) void sel_pass2_kernel(const float* __restrict__ neg_all, int total, u32* __restrict__ ghist,
                                                                 int neg_pos_ratio, int n_neg_min, const double* __restrict__ part,
                                                                 int tiles, double* __restrict__ sums, int B,
                                                                 SelectResult* __restrict__ res, float* __restrict__ stats) {
    __shared__ u32 hist[SEL_BINS];
    __shared__ int sh_out[2];
    __shared__ int wave_cnt[SELP_THREADS / 64];
    __shared__ double tot[2][SELP_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t plane = (size_t)B * tiles;
    // L1's per-tile partial sums -> per-image sums, one wave each: fixed order, no atomics.  Spread over the blocks (a job is one
    // dependent memory round trip: all 2 B of them on one block's four waves were 16 us of this kernel).
    for (int job = blockIdx.x * (SELP_THREADS / 64) + wave; job < 2 * B; job += gridDim.x * (SELP_THREADS / 64)) {   // job = q * B + b, q = 0 positive class loss, 1 loc loss
        double v = 0.0;
        for (int t = lane; t < tiles; t += 64) v += part[(size_t)job * tiles + t];
        v = wave_sum(v);
        if (lane == 0) sums[job] = v;
    }
    // n_pos and the number of non-zero negative losses: the same order in every block, identical k everywhere.  Four loads per
    // plane in flight per trip (a plain accumulate loop waits for one memory round trip per element).
    double np_ = 0.0, nz_ = 0.0;
    {
        const double* pn = part + 2 * plane;
        const double* pz = part + 3 * plane;
        for (size_t i0 = tid; i0 < plane; i0 += 4 * SELP_THREADS) {
            double a[4], z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = i0 + (size_t)u * SELP_THREADS;
                a[u] = i < plane ? pn[i] : 0.0;
                z[u] = i < plane ? pz[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { np_ += a[u]; nz_ += z[u]; }
        }
    }
    u32 l1 = 0;                                                         // level 1: this thread's bin, summed over the 8 copies
#pragma unroll
    for (int sh = 0; sh < L1_SHARDS; ++sh) l1 += ghist[sh * L1_BINS + tid];
    static_assert(L1_BINS == SELP_THREADS, "one level-1 bin per thread");
    hist[tid] = l1;
    np_ = wave_sum(np_); nz_ = wave_sum(nz_);
    if (lane == 0) { tot[0][wave] = np_; tot[1][wave] = nz_; }
    __syncthreads();
    np_ = 0.0; nz_ = 0.0;
    for (int w = 0; w < SELP_THREADS / 64; ++w) { np_ += tot[0][w]; nz_ += tot[1][w]; }
    const float n_pos = (float)np_;
    const int n_neg_losses = (int)nz_;
    int k = neg_pos_ratio * (int)n_pos;                                  // tf.to_int32(n_positive) truncates (:166)
    k = k > n_neg_min ? k : n_neg_min;
    k = k < n_neg_losses ? k : n_neg_losses;
    if (k > 0) block_find_digit<L1_BINS / SELP_THREADS>(hist, k, wave_cnt, sh_out);
    const int digit = k > 0 ? sh_out[0] : 0, want = k > 0 ? k - sh_out[1] : 0;
    if (blockIdx.x == 0 && tid == 0) {
        sums[3 * B] = np_;
        res->k = k; res->thresh_key = 0; res->tie_limit = 0x7fffffff; res->n_neg_losses = n_neg_losses;
        res->n_pos = n_pos; res->thresh = 0.f;
        res->digit = digit; res->want = want; res->digit2 = 0; res->want2 = 0;
        stats[0] = n_pos; stats[1] = (float)n_neg_losses; stats[2] = (float)k; stats[3] = 0.f;
    }
    if (k <= 0) return;
    __syncthreads();
    for (int i = tid; i < SEL_BINS; i += SELP_THREADS) hist[i] = 0u;
    __syncthreads();
    const int i0 = blockIdx.x * SELP_CHUNK + tid;
    u32 keys[SELP_ITEMS];
#pragma unroll
    for (int j = 0; j < SELP_ITEMS; ++j) {                               // all eight loads in flight together
        const int i = i0 + j * SELP_THREADS;
        keys[j] = i < total ? float_key(neg_all[i]) : 0u;
    }
#pragma unroll
    for (int j = 0; j < SELP_ITEMS; ++j)
        hist_add_aggregated(hist, (keys[j] >> 12) & 0xfffu, i0 + j * SELP_THREADS < total && (keys[j] >> 24) == (u32)digit);
    __syncthreads();
    u32* g2 = ghist + L1_SHARDS * L1_BINS;
    for (int i = tid; i < SEL_BINS; i += SELP_THREADS) {
        const u32 c = hist[i];
        if (c) atomicAdd(&g2[i], c);
    }
}

// L2b: the level-2 digit (every block); level-3 histogram (the last 12 bits) of the values with that 20-bit prefix, to the global bins
// and, per block, to bcol[block][SEL_BINS]
__global__ __launch_bounds__(SELP_THREADS) void sel_pass3_kernel(const float* __restrict__ neg_all, int total, u32* __restrict__ ghist,
                                                                 SelectResult* __restrict__ res, unsigned short* __restrict__ bcol) {
    __shared__ u32 hist[SEL_BINS];
    __shared__ int sh_out[2];
    __shared__ int wave_cnt[SELP_THREADS / 64];
    const int tid = threadIdx.x;
    const int k = res->k;
    if (k <= 0) return;
    const u32* g2 = ghist + L1_SHARDS * L1_BINS;
    const int i0 = blockIdx.x * SELP_CHUNK + tid;
    u32 keys[SELP_ITEMS];
#pragma unroll
    for (int j = 0; j < SELP_ITEMS; ++j) {                               // requested before the digit search: both latencies overlap
        const int i = i0 + j * SELP_THREADS;
        keys[j] = i < total ? float_key(neg_all[i]) : 0u;
    }
    for (int i = tid; i < SEL_BINS; i += SELP_THREADS) hist[i] = g2[i];
    __syncthreads();
    const int want1 = res->want;
    block_find_digit<SEL_BINS / SELP_THREADS>(hist, want1, wave_cnt, sh_out);
    const int digit2 = sh_out[0], want2 = want1 - sh_out[1];
    const u32 prefix20 = ((u32)res->digit << 12) | (u32)digit2;          // key >> 12
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) { res->digit2 = digit2; res->want2 = want2; }
    for (int i = tid; i < SEL_BINS; i += SELP_THREADS) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SELP_ITEMS; ++j)
        hist_add_aggregated(hist, keys[j] & 0xfffu, i0 + j * SELP_THREADS < total && (keys[j] >> 12) == prefix20);
    __syncthreads();
    u32* g3 = ghist + L1_SHARDS * L1_BINS + SEL_BINS;
    for (int i = tid; i < SEL_BINS; i += SELP_THREADS) {
        const u32 c = hist[i];
        bcol[(size_t)blockIdx.x * SEL_BINS + i] = (unsigned short)c;     // <= SELP_CHUNK = 2048
        if (c) atomicAdd(&g3[i], c);
    }
}

// ======================================================================================
// L3 / L4
// ======================================================================================
// L3: the last digit, the threshold key and -- only if ties straddle the cut -- the flat-index limit (every block, redundantly),
// then the keep mask + per-image partial sums of the kept negative losses
__global__ __launch_bounds__(LOSS_THREADS) void keep_kernel(const float* __restrict__ cls, const float* __restrict__ neg_all,
                                                            int B, int N, const u32* __restrict__ ghist, int nblk,
                                                            const unsigned short* __restrict__ bcol, SelectResult* __restrict__ res,
                                                            float* __restrict__ stats, unsigned char* __restrict__ keep,
                                                            double* __restrict__ keep_part, u32* __restrict__ arrived,
                                                            const double* __restrict__ sums, float alpha, float* __restrict__ loss) {
    __shared__ double red[LOSS_THREADS / 64];
    __shared__ u32 hist[SEL_BINS];
    __shared__ int sh_out[2];
    __shared__ int wave_cnt[LOSS_THREADS / 64];
    __shared__ int sh_found[2];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = res->k;
    u32 tk = 0;
    int tie_limit = 0x7fffffff;
    if (k > 0) {
        const u32* g3 = ghist + L1_SHARDS * L1_BINS + SEL_BINS;
        for (int i = tid; i < SEL_BINS; i += LOSS_THREADS) hist[i] = g3[i];
        __syncthreads();
        const int want2 = res->want2;
        block_find_digit<SEL_BINS / LOSS_THREADS>(hist, want2, wave_cnt, sh_out);
        const int digit3 = sh_out[0], want3 = want2 - sh_out[1];         // want3 of the values equal to the threshold are kept
        const int eq_total = (int)hist[digit3];
        tk = ((u32)res->digit << 24) | ((u32)res->digit2 << 12) | (u32)digit3;
        if (want3 < eq_total) {                                          // the lowest flat indices among the ties win
            // the pass block that holds the want3-th tie ...
            int before = 0, jstar = -1, rank = 0;
            for (int j0 = 0; j0 < nblk && jstar < 0; j0 += LOSS_THREADS) {
                const int j = j0 + tid;
                const int c = j < nblk ? (int)bcol[(size_t)j * SEL_BINS + digit3] : 0;
                int tot;
                const int excl = block_exclusive_scan(c, wave_cnt, tot);
                if (tid == 0) sh_found[0] = -1;
                __syncthreads();
                if (c > 0 && before + excl < want3 && want3 <= before + excl + c) { sh_found[0] = j; sh_found[1] = want3 - (before + excl); }
                __syncthreads();
                jstar = sh_found[0]; rank = sh_found[1];
                before += tot;
            }
            // ... and the rank-th tie inside it, in flat order (a thread takes 8 consecutive values)
            const int f0 = jstar * SELP_CHUNK + tid * SELP_ITEMS, total = B * N;
            int c = 0;
            u32 hit = 0;
#pragma unroll
            for (int j = 0; j < SELP_ITEMS; ++j)
                if (f0 + j < total && float_key(neg_all[f0 + j]) == tk) { hit |= 1u << j; ++c; }
            int tot;
            const int excl = block_exclusive_scan(c, wave_cnt, tot);
            if (c > 0 && excl < rank && rank <= excl + c) {
                int need = rank - excl;
#pragma unroll
                for (int j = 0; j < SELP_ITEMS; ++j)
                    if ((hit >> j) & 1u) { if (--need == 0) sh_found[0] = f0 + j + 1; }   // one past the want3-th tie
            }
            __syncthreads();
            tie_limit = sh_found[0];
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            res->thresh_key = tk; res->tie_limit = tie_limit; res->thresh = key_float(tk);
            stats[3] = key_float(tk);
        }
    }
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
        keep_part[(size_t)b * KEEP_BLOCKS + blockIdx.x] = a;              // summed in a fixed order below
        // L4 folded in (round 4: one launch and its gap less): the LAST block of the image to arrive adds the image's partial sums in
        // block order -- the order does not depend on who is last -- and writes the image's loss.  Release: the partial sum above is in L2
        // before the counter moves; acquire: the last block reads the others' sums with device-scope loads.
        __threadfence();
        const u32 before = atomicAdd(&arrived[b], 1u);
        if (before == gridDim.x - 1) {
            __threadfence();
            const float n_pos = (float)sums[3 * B];
            double neg = 0.0;
            for (u32 i = 0; i < gridDim.x; ++i)
                neg += __hip_atomic_load(&keep_part[(size_t)b * KEEP_BLOCKS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float cls_loss = (float)sums[b] + (float)neg;               // class_loss = pos + neg (:200)
            const float tot = (cls_loss + alpha * (float)sums[B + b]) / fmaxf(1.0f, n_pos);
            loss[b] = tot * (float)B;                                         // :208-209
        }
    }
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


// ======================================================================================
// L1 and backward, streaming form (round 5).  The tiled kernels above move 74 MB (111 MB backward) at 2.6-3 TB/s: a workgroup loads
// its tile with one 16-byte load in flight per thread, works, and leaves; the latency is covered only by the other workgroups of the
// CU, and batching the loads through VGPRs made it worse (profiles/r01l_loss_copy_variants.txt).  Here every WAVE is a stream of its
// own: tiles of 64 anchors (one per lane, whole [C+12]-rows) arrive in its private LDS double buffer by LDS-DMA (ssdhip_tile.h) while
// it works on the previous tile -- no barrier inside the loop, a tile in flight per wave at all times.  A persistent grid: wave w
// takes tiles w, w + W, w + 2W ... (neighbouring waves read neighbouring rows at the same time).  Needs 16-byte aligned tiles
// (N (C+12) % 4 == 0, true for every reference geometry with an even number of anchors) and the tensors below 2 GiB; anything else
// runs the tiled kernels.
// ======================================================================================
struct StreamPlan {
    int ok, G, nw, grid, tiles64;     // G: 1 KiB loads per array and tile; nw: waves per workgroup
    size_t lds;
};

static StreamPlan stream_plan(const void* y_true, const void* y_pred, int B, int N, int C, int extra_lds_per_stage) {
    StreamPlan p = {0, 0, 0, 0, (N + 63) / 64, 0};
    const int L = C + 12;
    const long long bytes = (long long)B * N * L * 4;
    if (bytes >= 0x7fffff00LL || ((long long)N * L) % 4 != 0 || N % 4 != 0) return p;
    if (((uintptr_t)y_true | (uintptr_t)y_pred) & 15u) return p;
    if (const char* e = getenv("SSDHIP_LOSS_STREAM")) { if (e[0] == '0') return p; }    // A/B switch: the tiled kernels
    p.G = (64 * L * 4 + 1023) / 1024;
    if (2 * p.G + 2 > 60) return p;                                     // vmcnt counts to 63
    const size_t stage = (size_t)2 * p.G * 1024 + extra_lds_per_stage;
    p.nw = 4;
    while (p.nw > 1 && p.nw * 2 * stage > 150 * 1024) p.nw >>= 1;
    if (p.nw < 4 || p.nw * 2 * stage > 150 * 1024) return p;            // rows too long for four streams per CU (C + 12 > 40 floats, e.g. COCO's 93): the tiled kernels keep more waves in flight there (not measured)
    p.lds = p.nw * 2 * stage;
    const long long WT = (long long)B * p.tiles64;
    int cus = 256;
    long long grid = (WT + p.nw - 1) / p.nw;
    if (grid > cus) grid = cus;
    const long long per = (WT + grid * p.nw - 1) / (grid * p.nw);      // tiles per wave; then the smallest grid that still does it in `per`
    grid = (WT + per * p.nw - 1) / (per * p.nw);
    p.grid = (int)grid;
    p.ok = 1;
    return p;
}

static bool stream_big_lds(const void* fn, int which, size_t lds) {     // the > 64 KB opt-in is per device and per kernel
    if (lds <= 64 * 1024) return true;
    static int granted[2][64] = {{0}};                                  // bytes opted in so far; -1: refused
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return false;
    int& g = granted[which][devid];
    if (g >= 0 && (size_t)g < lds) {
        // the dynamic part only: the kernel's static LDS comes out of the same 160 KB
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) g = (int)lds;
        else { g = -1; (void)hipGetLastError(); }                       // a refusal must not surface as the next launch's error
    }
    return g >= 0 && (size_t)g >= lds;
}

// the wave's next tile -> stage `lds_dst`: G loads per array, always (a short last tile of an image and the lanes past its end write zeros)
__device__ __forceinline__ void stream_issue(int t, int tiles64, int N, int L, int G, tile_i32x4 rt, tile_i32x4 rp, u32 lds_dst, int lane) {
    const int b = t / tiles64, a0 = (t - b * tiles64) * 64;
    const int na = min(64, N - a0);
    const u32 tile_off = (u32)(((size_t)b * N + a0) * (size_t)L * 4);
    const int nch = (na * L) >> 2;                                      // 16-byte chunks: na L % 4 == 0 (N L % 4 == 0, a0 = 64 i)
    const u32 arr = (u32)G * 1024u;
    for (int j = 0; j < G; ++j) {
        const int c = j * 64 + lane;
        const u32 voff = c < nch ? tile_off + (u32)c * 16u : TILE_OOB;
        tile_dma16(voff, rt, lds_dst + (u32)j * 1024u);
        tile_dma16(voff, rp, lds_dst + arr + (u32)j * 1024u);
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void anchor_stream_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                                     int B, int N, int C, int tiles64, int G,
                                                                     float* __restrict__ cls_out, float* __restrict__ neg_out,
                                                                     double* __restrict__ part, u32* __restrict__ ghist1) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    __shared__ u32 hist[L1_BINS];
    const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = C + 12;
    const u32 arr = (u32)G * 1024u, stage = 2u * arr;
    unsigned char* mine = smem_raw + (size_t)wave * 2 * stage;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + (u32)wave * 2u * stage;
    const u32 total_bytes = (u32)((size_t)B * N * (size_t)L * 4);
    const tile_i32x4 rt = tile_rsrc(y_true, total_bytes), rp = tile_rsrc(y_pred, total_bytes);
    const int WT = B * tiles64, W = gridDim.x * nw, gw = blockIdx.x * nw + wave;
    for (int i = tid; i < L1_BINS; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    if (gw < WT) stream_issue(gw, tiles64, N, L, G, rt, rp, lds0, lane);
    int par = 0;
    for (int t = gw; t < WT; t += W, par ^= 1) {
        const bool more = t + W < WT;
        if (more) stream_issue(t + W, tiles64, N, L, G, rt, rp, lds0 + (u32)(par ^ 1) * stage, lane);
        tile_wait_vmcnt(more ? 2 * G : 0);                              // everything older than the tile just requested has landed
        const int b = t / tiles64, ti = t - b * tiles64, a0 = ti * 64;
        const int na = min(64, N - a0);
        const float* tb = reinterpret_cast<const float*>(mine + (size_t)par * stage);
        const float* qb = reinterpret_cast<const float*>(mine + (size_t)par * stage + arr);
        double s_poscls = 0.0, s_loc = 0.0, s_npos = 0.0;
        int nonzero = 0;
        u32 bin = 0;
        if (lane < na) {
            const float* tr = tb + (size_t)lane * L;
            const float* q = qb + (size_t)lane * L;
            float cls = 0.f, pos = tr[1];
            int c = 0;
            for (; c + 4 <= C; c += 4) {                               // log_loss (:93-95); four classes' LDS reads in flight together
                float tc[4], qc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { tc[u] = tr[c + u]; qc[u] = q[c + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c + u >= 1) pos = fmaxf(pos, tc[u]);            // positives = max(y_true[1:C]) (:140)
                    if (tc[u] != 0.f) cls += tc[u] * logf(fmaxf(qc[u], 1e-15f));
                }
            }
            for (; c < C; ++c) {
                const float tc = tr[c];
                if (c >= 1) pos = fmaxf(pos, tc);
                if (tc != 0.f) cls += tc * logf(fmaxf(q[c], 1e-15f));
            }
            cls = -cls;
            float loc = 0.f;
            for (int k = 0; k < 4; ++k) {                               // smooth_L1_loss (:72-75)
                const float d = tr[C + k] - q[C + k];
                const float ad = fabsf(d);
                loc += ad < 1.0f ? 0.5f * (d * d) : ad - 0.5f;
            }
            const float neg = cls * tr[0];                              // neg_class_loss_all (:150)
            cls_out[(size_t)b * N + a0 + lane] = cls;
            neg_out[(size_t)b * N + a0 + lane] = neg;
            s_poscls = (double)(cls * pos);
            s_loc = (double)(loc * pos);
            s_npos = (double)pos;
            nonzero = neg != 0.f;
            bin = float_key(neg) >> 24;
        }
        hist_add_aggregated(hist, bin, lane < na);
        s_poscls = wave_sum(s_poscls); s_loc = wave_sum(s_loc); s_npos = wave_sum(s_npos);
        const double s_nz = wave_sum((double)nonzero);
        if (lane == 0) {                                                // per-tile partial sums, reduced in a fixed order by L2
            const size_t slot = (size_t)b * tiles64 + ti, plane = (size_t)B * tiles64;
            part[slot] = s_poscls; part[plane + slot] = s_loc; part[2 * plane + slot] = s_npos; part[3 * plane + slot] = s_nz;
        }
    }
    __syncthreads();
    u32* gshard = ghist1 + (blockIdx.x & (L1_SHARDS - 1)) * L1_BINS;
    for (int i = tid; i < L1_BINS; i += blockDim.x) {
        const u32 c = hist[i];
        if (c) atomicAdd(&gshard[i], c);
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void backward_stream_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                                       const unsigned char* __restrict__ keep,
                                                                       const float* __restrict__ stats, const float* __restrict__ grad_out,
                                                                       int B, int N, int C, int tiles64, int G, float alpha,
                                                                       float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = C + 12;
    const u32 arr = (u32)G * 1024u, stage = 2u * arr + 256u;            // + the tile's 64 keep bytes (a 4-byte DMA load on 16 lanes) and its image's grad_out
    unsigned char* mine = smem_raw + (size_t)wave * 2 * stage;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + (u32)wave * 2u * stage;
    const u32 total_bytes = (u32)((size_t)B * N * (size_t)L * 4);
    const tile_i32x4 rt = tile_rsrc(y_true, total_bytes), rp = tile_rsrc(y_pred, total_bytes), rk = tile_rsrc(keep, (u32)((size_t)B * N));
    const tile_i32x4 rg = tile_rsrc(grad_out, (u32)B * 4u);
    const int WT = B * tiles64, W = gridDim.x * nw, gw = blockIdx.x * nw + wave;
    // No load of hipcc's own may follow the first DMA load: its s_waitcnt would drain ours with it (hipcc counts only what it issued).
    // stats[0] is read -- and waited for, the empty asm pins that -- up here; grad_out[b] comes with the tile.
    float n_pos = fmaxf(1.0f, stats[0]);
    asm volatile("" : "+v"(n_pos) :: "memory");
    auto issue = [&](int t, u32 dst) {
        stream_issue(t, tiles64, N, L, G, rt, rp, dst, lane);
        const int b = t / tiles64, a0 = (t - b * tiles64) * 64;
        const int nk = (min(64, N - a0) + 3) >> 2;                     // N % 4 == 0: whole dwords inside the mask
        // under an EXEC mask, not with out-of-range offsets: a lane past the end would write ZEROS at dst + 4 lane -- 256 bytes, over the
        // other word here and into the next stage.  Both masks are never empty, so each instruction issues (vmcnt counts instructions).
        if (lane < nk) tile_dma4((u32)((size_t)b * N + a0) + (u32)lane * 4u, rk, dst + 2u * arr);
        if (lane == 0) tile_dma4((u32)b * 4u, rg, dst + 2u * arr + 128u);
    };
    if (gw < WT) issue(gw, lds0);
    int par = 0;
    for (int t = gw; t < WT; t += W, par ^= 1) {
        const bool more = t + W < WT;
        if (more) issue(t + W, lds0 + (u32)(par ^ 1) * stage);
        tile_wait_vmcnt(more ? 2 * G + 2 : 0);
        const int b = __builtin_amdgcn_readfirstlane(t / tiles64), ti = t - b * tiles64, a0 = ti * 64;
        const int na = min(64, N - a0);
        const float* tb = reinterpret_cast<const float*>(mine + (size_t)par * stage);
        float* qb = reinterpret_cast<float*>(mine + (size_t)par * stage + arr);
        const unsigned char* kb = mine + (size_t)par * stage + 2 * arr;
        const float scale = *reinterpret_cast<const float*>(kb + 128) * (float)B / n_pos;    // the tiled kernel's order of operations
        if (lane < na) {
            const float* tr = tb + (size_t)lane * L;
            float* q = qb + (size_t)lane * L;                          // overwritten with the gradient row
            float pos = tr[1];
            int c = 2;
            for (; c + 4 <= C; c += 4) {                               // four LDS reads in flight together
                float tc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) tc[u] = tr[c + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) pos = fmaxf(pos, tc[u]);
            }
            for (; c < C; ++c) pos = fmaxf(pos, tr[c]);
            const float w_cls = (pos + (kb[lane] ? 1.0f : 0.0f)) * scale;
            for (c = 0; c + 4 <= C; c += 4) {
                float tc[4], pc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { tc[u] = tr[c + u]; pc[u] = q[c + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float g = 0.f;
                    if (tc[u] != 0.f && pc[u] >= 1e-15f) g = -(tc[u] / pc[u]) * w_cls;       // a branch: few classes of few lanes divide
                    q[c + u] = g;
                }
            }
            for (; c < C; ++c) {
                const float pc = q[c];
                q[c] = (pc >= 1e-15f && tr[c] != 0.f) ? -(tr[c] / pc) * w_cls : 0.f;
            }
            const float w_loc = alpha * pos * scale;
            for (int k = 0; k < 4; ++k) {
                const float d = tr[C + k] - q[C + k];
                const float dl = fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f);
                q[C + k] = -dl * w_loc;
            }
            for (int k = 4; k < 12; ++k) q[C + k] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();                                // one wave: its LDS operations execute in order
        const int nch = (na * L) >> 2;
        float4* gdst = reinterpret_cast<float4*>(grad + ((size_t)b * N + a0) * (size_t)L);
        const float4* gsrc = reinterpret_cast<const float4*>(qb);
        for (int c = lane; c < nch; c += 64) gdst[c] = gsrc[c];
    }
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
    SelectResult* sel = reinterpret_cast<SelectResult*>(base + lay.sel);
    float* cls = reinterpret_cast<float*>(base + lay.cls);
    float* neg = reinterpret_cast<float*>(base + lay.neg);
    u32* ghist = reinterpret_cast<u32*>(base + lay.hist);
    double* part = reinterpret_cast<double*>(base + lay.part);
    double* keep_part = reinterpret_cast<double*>(base + lay.keep_part);
    unsigned short* bcol = reinterpret_cast<unsigned short*>(base + lay.bcol);
    if (zero_async(ghist, ((size_t)L1_SHARDS * L1_BINS + 2 * SEL_BINS + (size_t)B) * sizeof(u32), stream) != hipSuccess) return SSDHIP_E_LAUNCH;    // the three levels' bins + L3's arrival counters (sums are plain stores)

    const int L = C + 12;
    const int TA = loss_tile(L);
    const size_t lds = 2 * (((size_t)TA * L + 4 + 3) / 4 * 4) * sizeof(float) + 16;
    if (lds > 140 * 1024) return SSDHIP_E_BADARG;
    const StreamPlan sp = stream_plan(y_true, y_pred, B, N, C, 0);
    int part_tiles = lay.tiles;
    if (sp.ok && stream_big_lds(reinterpret_cast<const void*>(anchor_stream_kernel), 0, sp.lds)) {
        hipLaunchKernelGGL(anchor_stream_kernel, dim3(sp.grid), dim3(64 * sp.nw), sp.lds, stream, y_true, y_pred, B, N, C, sp.tiles64, sp.G,
                           cls, neg, part, ghist);
        part_tiles = sp.tiles64;
    } else {
        hipLaunchKernelGGL(anchor_kernel, dim3(lay.tiles, B), dim3(TA), lds, stream, y_true, y_pred, B, N, C, cls, neg, part, ghist);
    }
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const int total = B * N;
    hipLaunchKernelGGL(sel_pass2_kernel, dim3(lay.nblk), dim3(SELP_THREADS), 0, stream, neg, total, ghist, neg_pos_ratio, n_neg_min, part,
                       part_tiles, sums, B, sel, stats);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    hipLaunchKernelGGL(sel_pass3_kernel, dim3(lay.nblk), dim3(SELP_THREADS), 0, stream, neg, total, ghist, sel, bcol);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    // every keep block repeats the last digit search: a few fat blocks per image (about a chip's worth in all), not one per 256 anchors
    int gx = (N + 4 * LOSS_THREADS - 1) / (4 * LOSS_THREADS);
    while (gx > 1 && gx * B > 512) --gx;
    if (gx > KEEP_BLOCKS) gx = KEEP_BLOCKS;
    hipLaunchKernelGGL(keep_kernel, dim3(gx, B), dim3(LOSS_THREADS), 0, stream, cls, neg, B, N, ghist, lay.nblk, bcol, sel, stats, keep_mask,
                       keep_part, ghist + L1_SHARDS * L1_BINS + 2 * SEL_BINS, sums, alpha, loss_per_item);
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
    const StreamPlan sp = stream_plan(y_true, y_pred, B, N, C, 256);
    if (sp.ok && !((uintptr_t)grad_y_pred & 15u) && !((uintptr_t)keep_mask & 3u) &&
        stream_big_lds(reinterpret_cast<const void*>(backward_stream_kernel), 1, sp.lds)) {
        hipLaunchKernelGGL(backward_stream_kernel, dim3(sp.grid), dim3(64 * sp.nw), sp.lds, stream, y_true, y_pred, keep_mask, stats, grad_out,
                           B, N, C, sp.tiles64, sp.G, alpha, grad_y_pred);
    } else {
        hipLaunchKernelGGL(backward_kernel, dim3((N + TA - 1) / TA, B), dim3(TA), lds, stream, y_true, y_pred, keep_mask, stats, grad_out,
                           B, N, C, alpha, grad_y_pred);
    }
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    return SSDHIP_OK;
}
