// ssdhip_augment.hip -- the DECISIONS of the original-SSD augmentation chain for a whole batch in one launch, gfx950 (MI355X).
//
// Replaces, for a device-resident batch, the per-image Python loop over (reference data_generator/)
//   data_augmentation_chain_original_ssd.py:103-162  SSDExpand        (RandomPatch, prob 0.5, canvas 1-4 x the image)
//   data_augmentation_chain_original_ssd.py:29-101   SSDRandomCrop    (RandomPatchInf: rounds of 50 candidate patches, IoU-validated)
//   object_detection_2d_geometric_ops.py:202-262     RandomFlip, :86-148 ResizeRandomInterp + Resize's label arithmetic
//   object_detection_2d_patch_sampling_ops.py:24-339 PatchCoordinateGenerator, CropPad (label shift, centre-point BoxFilter, clipping)
//   object_detection_2d_image_boxes_validation_utils.py:79-322  BoxFilter / ImageValidator
// i.e. everything of SSDDataAugmentation.__call__ (:208-280) behind the photometric part that is not a pixel operation.  Round 4 ran
// it per image on the host with one GPU round trip per sampling round (733 images/s for a pipeline whose two pixel kernels take 0.4 ms
// per batch of 32); here one WAVE per image walks the chain:
//   * the random numbers are NumPy's: the image's MT19937 state (np.random.RandomState(seed).get_state(), after the photometric draws the
//     host still makes) lives in LDS, and uniform / randint / choice consume it exactly as numpy/random/mtrand does (53-bit doubles from
//     two words, masked rejection sampling for bounded integers, searchsorted over the normalised cumulative weights) -- the decisions,
//     and the stream position afterwards, are those of the per-image chain under the same seed, bit for bit (tests/test_image_ops.py);
//   * a candidate patch is validated by the lanes in parallel (lane = ground truth box, IoU in the reference's float64 expression) and the
//     search stops at the first valid trial, which is where the reference's loop stops consuming random numbers;
//   * the label arithmetic (shift, centre-point filter, clipping, mirroring, rounding to the network input, degenerate-box filter) runs in
//     float64, exact for int64 and float64 label arrays alike.
// Outputs per image: the geometry (canvas, crop window, flip, interpolation mode) the gather launch needs, the surviving labels, and the
// generator state.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

constexpr int AUG_MAXG = 64;                              // ground truth boxes per image (one lane each)

struct AugParams {
    int H, W;                                             // the batch's image size
    double exp_prob, exp_min, exp_max;                    // SSDExpand: RandomPatch(prob), PatchCoordinateGenerator(min_scale, max_scale, scale_uniformly)
    double crop_prob, crop_min, crop_max, ar_min, ar_max; // SSDRandomCrop: RandomPatchInf(prob), generator scales / aspect ratios
    int n_trials, n_bounds;
    double cdf[8], lower[8], upper[8];                    // BoundGenerator: cumulative weights / cdf[-1] (as np.random.choice builds them), bounds
    double flip_prob;
    int n_modes, modes[8], out_h, out_w;                  // ResizeRandomInterp
    int max_rounds;                                       // safety net of the crop loop (the reference loops without one)
};

struct Mt {                                               // numpy's rk_state: 624 key words in LDS, the position in a register
    u32* key;
    int pos;
    // Round 6: a window of 64 TEMPERED outputs in a register, lane i = output `base + i`.  A draw is then one v_readlane with a scalar
    // index instead of a dependent LDS round trip + the tempering ALU chain -- the batch-serial walk of ssd_augment_stream_kernel is a
    // chain of ~100-cycle draws otherwise (1.2 ms for a batch of 32).  base < 0: no window loaded.
    u32 win;
    int base;
};

__device__ __forceinline__ void mt_twist(u32* key, int lane) {
    constexpr u32 UP = 0x80000000u, LO = 0x7fffffffu, MAG = 0x9908b0dfu;
    for (int base = 0; base < 623; base += 64) {
        const int i = base + lane;
        u32 nv = 0u;
        if (i < 623) {
            const u32 y = (key[i] & UP) | (key[i + 1] & LO);
            const u32 c = key[i < 227 ? i + 397 : i - 227];
            nv = c ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
        }
        __syncthreads();                                  // every lane has read before any lane writes (one wave per block)
        if (i < 623) key[i] = nv;
        __syncthreads();
    }
    if (lane == 0) {
        const u32 y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
    }
    __syncthreads();
}

__device__ __forceinline__ u32 mt_u32(Mt& s, int lane) {
    if (s.pos == 624) { mt_twist(s.key, lane); s.pos = 0; s.base = -1; }
    if (s.base < 0 || s.pos - s.base >= 64) {            // (the position only moves forward inside a block of 624)
        s.base = s.pos;
        const int i = s.pos + lane;
        u32 y = s.key[i < 624 ? i : 623];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        s.win = y;
    }
    const int k = __builtin_amdgcn_readfirstlane(s.pos - s.base);
    ++s.pos;
    return (u32)__builtin_amdgcn_readlane((int)s.win, k);
}
__device__ __forceinline__ double mt_double(Mt& s, int lane) {           // rk_double / random_sample
    const u32 a = mt_u32(s, lane) >> 5, b = mt_u32(s, lane) >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);   // (x 2^-53: exact, the same double as NumPy's division)
}
__device__ __forceinline__ double mt_uniform(Mt& s, int lane, double lo, double hi) { return lo + (hi - lo) * mt_double(s, lane); }
// np.random.randint(lo, hi) for int64 results: masked rejection sampling on 32-bit words (numpy/random/_bounded_integers: use_masked)
__device__ __forceinline__ long long mt_randint(Mt& s, int lane, long long lo, long long hi) {
    const unsigned long long rng = (unsigned long long)(hi - 1 - lo);
    if (rng == 0ull) return lo;
    unsigned long long mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (rng == 0xffffffffull) return lo + (long long)mt_u32(s, lane);
    if (rng > 0xffffffffull) {                                          // (never in this chain: extents are image sizes)
        unsigned long long v;
        do { const unsigned long long hi32 = mt_u32(s, lane), lo32 = mt_u32(s, lane); v = ((hi32 << 32) | lo32) & mask; } while (v > rng);
        return lo + (long long)v;
    }
    u32 v;
    do { v = mt_u32(s, lane) & (u32)mask; } while (v > (u32)rng);
    return lo + (long long)v;
}
// PatchCoordinateGenerator._position (object_detection_2d_patch_sampling_ops.py:163-178)
__device__ __forceinline__ int aug_position(Mt& s, int lane, int extent, int size) {
    const int room = extent - size;
    return (int)(room >= 0 ? mt_randint(s, lane, 0, (long long)room + 1) : mt_randint(s, lane, room, 1));
}

// The geometric decisions of ONE image (SSDExpand, SSDRandomCrop, RandomFlip, ResizeRandomInterp + the label arithmetic) by one wave on
// the generator state `s`: what SSDDataAugmentation.__call__ does behind the photometric part (:208-280).
__device__ __forceinline__ void aug_decide_image(const AugParams& p, Mt& s, const int lane, const int b, const double* __restrict__ lab_in,
                                                 const int* __restrict__ n_in, int* __restrict__ geo, double* __restrict__ lab_out,
                                                 int* __restrict__ n_out) {
    const int n = n_in[b];
    bool alive = lane < n;
    double cls = 0.0, x0 = 0.0, y0 = 0.0, x1 = 0.0, y1 = 0.0;
    if (alive) {
        const double* r = lab_in + ((size_t)b * AUG_MAXG + lane) * 5;
        cls = r[0]; x0 = r[1]; y0 = r[2]; x1 = r[3]; y1 = r[4];
    }
    int H = p.H, W = p.W;
    int g[12] = {0, 0, 0, H, W, 0, 0, 0, H, W, 0, 0};

    // ---- SSDExpand: RandomPatch(prob), one trial, no validator: the image on a canvas of 1 .. 4 times its size ----------------------
    if (!(mt_uniform(s, lane, 0.0, 1.0) < (1.0 - p.exp_prob))) {
        const double factor = mt_uniform(s, lane, p.exp_min, p.exp_max);
        const int h = (int)(factor * (double)H), w = (int)(factor * (double)W);
        const int top = aug_position(s, lane, H, h), left = aug_position(s, lane, W, w);
        y0 -= (double)top; y1 -= (double)top; x0 -= (double)left; x1 -= (double)left;      // CropPad: labels -= (top, left); no filter, no clip
        g[0] = 1; g[1] = top; g[2] = left; g[3] = h; g[4] = w;
        H = h; W = w;
    }
    g[8] = H; g[9] = W;

    // ---- SSDRandomCrop: RandomPatchInf -- rounds until a valid patch or the "leave it" draw -------------------------------------------
    for (int round = 0; round < p.max_rounds; ++round) {
        if (mt_uniform(s, lane, 0.0, 1.0) < (1.0 - p.crop_prob)) break;                    // unaltered
        const double u = mt_double(s, lane);                                               // BoundGenerator: np.random.choice(n, p=weights)
        int bi = 0;
        while (bi < p.n_bounds - 1 && !(u < p.cdf[bi])) ++bi;                               // searchsorted(cdf, u, side='right')
        const double lower = p.lower[bi], upper = p.upper[bi];
        bool found = false;
        for (int t = 0; t < p.n_trials && !found; ++t) {
            const int h = (int)(mt_uniform(s, lane, p.crop_min, p.crop_max) * (double)H);
            const int w = (int)(mt_uniform(s, lane, p.crop_min, p.crop_max) * (double)W);
            const int top = aug_position(s, lane, H, h), left = aug_position(s, lane, W, w);
            const double ar = (double)w / (double)h;
            if (!(p.ar_min <= ar && ar <= p.ar_max)) continue;
            // ImageValidator('iou', bounds, n_boxes_min = 1, border 'half'): iou(patch, box shifted into the patch's frame) in (lower, upper]
            PxBox<double> im, bb;
            im.x0 = 0.0; im.y0 = 0.0; im.x1 = (double)w; im.y1 = (double)h;
            im.area = box_area<double>(im.x0, im.y0, im.x1, im.y1, 0.0);
            bb.x0 = x0 - (double)left; bb.y0 = y0 - (double)top; bb.x1 = x1 - (double)left; bb.y1 = y1 - (double)top;
            bb.area = box_area<double>(bb.x0, bb.y0, bb.x1, bb.y1, 0.0);
            const double v = iou_px<double>(im, bb);
            if (__ballot(alive && v > lower && v <= upper) == 0ull) continue;
            // the cut: CropPad(clip_boxes=True, box_filter = centre point inside the patch)
            x0 = bb.x0; y0 = bb.y0; x1 = bb.x1; y1 = bb.y1;
            const double cy = (y0 + y1) / 2.0, cx = (x0 + x1) / 2.0;
            alive = alive && cy >= 0.0 && cy <= (double)h - 1.0 && cx >= 0.0 && cx <= (double)w - 1.0;
            const double ymax = (double)(h - 1), xmax = (double)(w - 1);
            y0 = y0 < 0.0 ? 0.0 : (y0 > ymax ? ymax : y0); y1 = y1 < 0.0 ? 0.0 : (y1 > ymax ? ymax : y1);
            x0 = x0 < 0.0 ? 0.0 : (x0 > xmax ? xmax : x0); x1 = x1 < 0.0 ? 0.0 : (x1 > xmax ? xmax : x1);
            g[5] = 1; g[6] = top; g[7] = left; g[8] = h; g[9] = w;
            H = h; W = w;
            found = true;
        }
        if (found) break;
    }

    // ---- RandomFlip('horizontal', prob) ----------------------------------------------------------------------------------------------
    if (!(mt_uniform(s, lane, 0.0, 1.0) < (1.0 - p.flip_prob))) {
        const double nx0 = (double)W - x1, nx1 = (double)W - x0;
        x0 = nx0; x1 = nx1;
        g[10] = 1;
    }

    // ---- ResizeRandomInterp: mode = np.random.choice(modes); Resize's label arithmetic + the degenerate-box filter -----------------------
    g[11] = p.modes[(int)mt_randint(s, lane, 0, p.n_modes)];
    {
        const double sy = (double)p.out_h / (double)H, sx = (double)p.out_w / (double)W;
        y0 = __builtin_rint(y0 * sy); y1 = __builtin_rint(y1 * sy);
        x0 = __builtin_rint(x0 * sx); x1 = __builtin_rint(x1 * sx);
        alive = alive && x1 > x0 && y1 > y0;
    }

    const u64 keep = __ballot(alive);
    if (alive) {
        const int pos = __popcll(keep & lanemask_lt());
        double* r = lab_out + ((size_t)b * AUG_MAXG + pos) * 5;
        r[0] = cls; r[1] = x0; r[2] = y0; r[3] = x1; r[4] = y1;
    }
    if (lane == 0) {
        n_out[b] = __popcll(keep);
        for (int i = 0; i < 12; ++i) geo[(size_t)b * 12 + i] = g[i];
    }
}

// one wave per image, every image on its OWN generator state (augment_batch(seeds=...), round 5)
__global__ __launch_bounds__(64) void ssd_augment_decide_kernel(AugParams p, const u32* __restrict__ mt_in, const double* __restrict__ lab_in,
                                                                const int* __restrict__ n_in, int* __restrict__ geo,
                                                                double* __restrict__ lab_out, int* __restrict__ n_out,
                                                                u32* __restrict__ mt_out) {
    __shared__ u32 key[624];
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    for (int i = lane; i < 624; i += 64) key[i] = mt_in[(size_t)b * 625 + i];
    Mt s;
    s.key = key;
    s.pos = (int)mt_in[(size_t)b * 625 + 624];
    s.base = -1; s.win = 0u;
    __syncthreads();
    aug_decide_image(p, s, lane, b, lab_in, n_in, geo, lab_out, n_out);
    if (lane == 0) mt_out[(size_t)b * 625 + 624] = (u32)s.pos;
    __syncthreads();
    for (int i = lane; i < 624; i += 64) mt_out[(size_t)b * 625 + i] = key[i];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the reference's OWN contract -- ONE generator for the whole batch.  Its generator loop calls the chain image after image on the
// global np.random stream (object_detection_2d_data_generator.py:1050-1089 -> data_augmentation_chain_original_ssd.py:208-280), so the
// draws of image i + 1 start where image i's ended: inherently serial, and cheap (a few dozen draws per image).  ONE wave walks the batch
// in order on that one stream and takes EVERY decision, the photometric ones included (SSDPhotometricDistortions :146-208: which of the
// two sequences, and per op "does it fire" + its parameter), writing them as the per-image programs ssdhip_image_program runs; the
// geometric half is aug_decide_image.  Out: the programs, the geometry, the labels and the generator state behind the LAST image, which
// the host writes back into np.random -- the stream continues exactly as if the reference's loop had run.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct AugPhoto {                                         // RandomBrightness / Contrast / Saturation / Hue: prob, uniform(lo, hi); swap: prob 0
    double prob[4], lo[4], hi[4], swap_prob;
};
constexpr int AUG_PROG = 16;                              // steps of a program (include/ssdhip.h: SSDHIP_IMG_PROG)
enum { OP_END = 0, OP_TO_F32 = 1, OP_TO_U8 = 2, OP_BRIGHTNESS = 3, OP_CONTRAST = 4, OP_SATURATION = 5, OP_HUE = 6, OP_RGB2HSV = 7, OP_HSV2RGB = 8 };

__global__ __launch_bounds__(64) void ssd_augment_stream_kernel(AugParams p, AugPhoto ph, int B, const u32* __restrict__ mt_in,
                                                                const double* __restrict__ lab_in, const int* __restrict__ n_in,
                                                                int* __restrict__ prog_ops, double* __restrict__ prog_args,
                                                                int* __restrict__ geo, double* __restrict__ lab_out, int* __restrict__ n_out,
                                                                u32* __restrict__ mt_out) {
    __shared__ u32 key[624];
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < 624; i += 64) key[i] = mt_in[i];
    Mt s;
    s.key = key;
    s.pos = (int)mt_in[624];
    s.base = -1; s.win = 0u;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        // ---- SSDPhotometricDistortions.__call__: `if np.random.choice(2)` picks the sequence, every random op draws uniform(0, 1) and,
        //      when it fires (>= 1 - prob), its parameter; RandomChannelSwap(prob 0) draws and never fires ---------------------------------
        int ops[AUG_PROG];
        double args[AUG_PROG];
        int k = 0;
        auto push = [&](int op, double a) { ops[k] = op; args[k] = a; ++k; };
        auto maybe = [&](int which, int op) {            // RandomX.draw(): the firing draw, then uniform(lower, upper)
            if (mt_uniform(s, lane, 0.0, 1.0) >= (1.0 - ph.prob[which])) push(op, mt_uniform(s, lane, ph.lo[which], ph.hi[which]));
        };
        const bool contrast_first = mt_randint(s, lane, 0, 2) != 0;
        push(OP_TO_F32, 0.0);
        maybe(0, OP_BRIGHTNESS);
        if (contrast_first) maybe(1, OP_CONTRAST);
        push(OP_TO_U8, 0.0); push(OP_RGB2HSV, 0.0); push(OP_TO_F32, 0.0);
        maybe(2, OP_SATURATION);
        maybe(3, OP_HUE);
        push(OP_TO_U8, 0.0); push(OP_HSV2RGB, 0.0);
        if (!contrast_first) { push(OP_TO_F32, 0.0); maybe(1, OP_CONTRAST); push(OP_TO_U8, 0.0); }
        (void)mt_uniform(s, lane, 0.0, 1.0);             // RandomChannelSwap(prob = 0.0): uniform(0, 1) >= 1.0 never holds
        for (; k < AUG_PROG; ) push(OP_END, 0.0);
        if (lane == 0)
            for (int i = 0; i < AUG_PROG; ++i) { prog_ops[(size_t)b * AUG_PROG + i] = ops[i]; prog_args[(size_t)b * AUG_PROG + i] = args[i]; }
        // ---- the geometric half on the same stream ------------------------------------------------------------------------------------
        aug_decide_image(p, s, lane, b, lab_in, n_in, geo, lab_out, n_out);
    }
    if (lane == 0) mt_out[624] = (u32)s.pos;
    __syncthreads();
    for (int i = lane; i < 624; i += 64) mt_out[i] = key[i];
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The tap tables of the gather launch (ssdhip_image_resize_gather_u8) built ON THE DEVICE from the decisions above: per image and axis,
// cv2.resize's source indices and float64 weights for the drawn interpolation mode (data_generator/_image_ops.py axis_taps: the same
// expressions in the same order -- nearest, bilinear, bicubic a = -0.75, the true area filter when both axes shrink and cv2's area-mode
// bilinear variant otherwise, Lanczos-4 -- incl. NumPy's summation order where a row of weights is normalised), composed with the
// recorded geometry (resize <- flip <- crop window <- expansion canvas; -1 = a canvas pixel, filled with the background colour).
// Round 5: the host built these per image in NumPy and uploaded 1.8 MB per batch (2.5 ms of a 4.5 ms batch).
// ---------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double aug_np_sum(const double* w, int n) {          // numpy's pairwise_sum for one contiguous row (n <= 64)
    if (n < 8) {
        double r = 0.0;                                                          // (-0.0 + w0 in NumPy; the weights are never -0.0)
        for (int i = 0; i < n; ++i) r += w[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = w[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += w[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += w[i];
    return res;
}

// grid (blocks over positions, 2 axes, B); axis 0 = columns (x), 1 = rows (y)
__global__ __launch_bounds__(128) void aug_taps_kernel(const int* __restrict__ geo, int H, int W, int out_h, int out_w, int n_taps,
                                                       int* __restrict__ ix, double* __restrict__ wx, int* __restrict__ iy,
                                                       double* __restrict__ wy) {
    const int b = (int)blockIdx.z, axis = (int)blockIdx.y;
    const int n_dst = axis == 0 ? out_w : out_h;
    const int i = (int)blockIdx.x * 128 + (int)threadIdx.x;
    if (i >= n_dst) return;
    const int* g = geo + (size_t)b * 12;
    const int Hc = g[8], Wc = g[9], interp = g[11];
    const int n_src = axis == 0 ? Wc : Hc;
    const bool area_linear = interp == 3 && !(Wc >= out_w && Hc >= out_h);       // _area_linear
    const double scale = (double)n_src / (double)n_dst;
    int idx[64];
    double w[64];
    int T = 0;
    if (interp == 3 && area_linear) {
        double sxd = floor((double)i * scale);
        float fx = (float)(((double)i + 1.0) - (sxd + 1.0) * (1.0 / scale));
        fx = fx <= 0.f ? 0.f : fx - floorf(fx);
        double fxd = (double)fx;
        if (sxd >= (double)(n_src - 1)) { sxd = (double)(n_src - 1); fxd = 0.0; }
        const long long s0 = (long long)sxd;
        idx[0] = (int)(s0 < 0 ? 0 : (s0 > n_src - 1 ? n_src - 1 : s0));
        idx[1] = (int)(s0 + 1 < 0 ? 0 : (s0 + 1 > n_src - 1 ? n_src - 1 : s0 + 1));
        w[0] = 1.0 - fxd; w[1] = fxd;
        T = 2;
    } else if (interp == 0) {
        const double f = floor((double)i * scale);
        idx[0] = (int)(f < (double)(n_src - 1) ? f : (double)(n_src - 1));
        w[0] = 1.0;
        T = 1;
    } else if (interp == 3 && scale >= 1.0) {
        const double lo = (double)i * scale, hi = ((double)i + 1.0) * scale;
        const long long first = (long long)floor(lo);
        T = (int)ceil(scale) + 1;
        T = T > 64 ? 64 : T;
        for (int t = 0; t < T; ++t) {
            const long long cell = first + t;
            const double a = (double)cell + 1.0 < hi ? (double)cell + 1.0 : hi;   // np.minimum(cells + 1.0, hi)
            const double c = (double)cell > lo ? (double)cell : lo;               // np.maximum(cells, lo)
            const double d = a - c;
            w[t] = d < 0.0 ? 0.0 : d;
            idx[t] = (int)(cell < 0 ? 0 : (cell > n_src - 1 ? n_src - 1 : cell));
        }
        const double sum = aug_np_sum(w, T);
        for (int t = 0; t < T; ++t) w[t] = w[t] / sum;
    } else {
        const double center = ((double)i + 0.5) * scale - 0.5;
        const double base = floor(center);
        const double frac = center - base;
        int off0;
        if (interp == 1 || interp == 3) {
            off0 = 0; T = 2;
            w[0] = 1.0 - frac; w[1] = frac;
        } else if (interp == 2) {
            off0 = -1; T = 4;
            const double a = -0.75;
            for (int t = 0; t < 4; ++t) {
                const double d = fabs(frac - (double)(off0 + t));
                const double near = ((a + 2) * d - (a + 3)) * d * d + 1;
                const double far = ((a * d - 5 * a) * d + 8 * a) * d - 4 * a;
                w[t] = d <= 1.0 ? near : (d < 2.0 ? far : 0.0);
            }
        } else {
            off0 = -3; T = 8;
            const double pi = 3.141592653589793;
            for (int t = 0; t < 8; ++t) {
                const double d = frac - (double)(off0 + t);
                const bool inside = fabs(d) >= 1e-12 && fabs(d) < 4.0;
                const double sf = inside ? d : 1.0;
                const double val = 4 * sin(pi * sf) * sin(pi * sf / 4) / (pi * pi * sf * sf);
                w[t] = inside ? val : (fabs(d) < 1e-12 ? 1.0 : 0.0);
            }
            const double sum = aug_np_sum(w, 8);
            for (int t = 0; t < 8; ++t) w[t] = w[t] / sum;
        }
        for (int t = 0; t < T; ++t) {
            const long long c = (long long)base + off0 + t;
            idx[t] = (int)(c < 0 ? 0 : (c > n_src - 1 ? n_src - 1 : c));
        }
    }
    // the recorded geometry: position j of the final (pre-resize) image <- flip <- crop window <- expansion canvas <- the image
    int* oi = (axis == 0 ? ix : iy) + ((size_t)b * n_dst + i) * n_taps;
    double* ow = (axis == 0 ? wx : wy) + ((size_t)b * n_dst + i) * n_taps;
    for (int t = 0; t < n_taps; ++t) {
        int src = 0;
        double wt = 0.0;
        if (t < T) {
            int j = idx[t];
            if (axis == 0 && g[10]) j = Wc - 1 - j;                               // RandomFlip('horizontal')
            if (g[5]) j += axis == 0 ? g[7] : g[6];                               // the crop window's corner on the canvas
            if (g[0]) {                                                           // the canvas: the image sits at (-top, -left)
                j += axis == 0 ? g[2] : g[1];
                if (j < 0 || j >= (axis == 0 ? W : H)) j = -1;
            }
            src = j;
            wt = w[t];
        }
        oi[t] = src;
        ow[t] = wt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the gather launch's PLANS with cv2.resize's own 8-bit arithmetic (ssdhip_image_resize_gather_cv_u8), built on the device from
// the decisions: per image the dispatch of cv::resize (copy / nearest / fixed-point linear, cubic, Lanczos-4 / area-mode bilinear /
// ResizeArea / ResizeAreaFast) and per axis the tap indices + table values of data_generator/_image_ops.py resize_plan -- the same
// float32 / float64 operations in the same order (this file is compiled without contraction), Lanczos' sine and cosine by the same
// fixed Horner chain -- composed with the recorded geometry as aug_taps_kernel does.
// ---------------------------------------------------------------------------------------------------------------------------------------
enum { PK_NEAREST = 0, PK_LINEAR = 1, PK_KERNEL = 2, PK_AREA = 3, PK_AREA_FAST = 4, PK_AREA_FAST2 = 5, PK_COPY = 6 };
__device__ const double AUG_SIN_C[12] = {1.0, 0.16666666666666666, 0.008333333333333333, 0.0001984126984126984, 2.7557319223985893e-06, 2.505210838544172e-08, 1.6059043836821613e-10, 7.647163731819816e-13, 2.8114572543455206e-15, 8.22063524662433e-18, 1.9572941063391263e-20, 3.8681701706306835e-23};       // 1 / (2 k + 1)!
__device__ const double AUG_COS_C[12] = {1.0, 0.5, 0.041666666666666664, 0.001388888888888889, 2.48015873015873e-05, 2.755731922398589e-07, 2.08767569878681e-09, 1.1470745597729725e-11, 4.779477332387385e-14, 1.5619206968586225e-16, 4.110317623312165e-19, 8.896791392450574e-22};       // 1 / (2 k)!

__device__ __forceinline__ void aug_sincos_near_minus_pi(double y, double& sn, double& cs) {   // _image_ops.sincos_near_minus_pi
    const double t = y + 3.1415926535897932384626433832795;
    const double t2 = t * t;
    double s = AUG_SIN_C[11];
    for (int k = 10; k >= 0; --k) s = AUG_SIN_C[k] - s * t2;
    double c = AUG_COS_C[11];
    for (int k = 10; k >= 0; --k) c = AUG_COS_C[k] - c * t2;
    sn = -(s * t);
    cs = -c;
}
__device__ __forceinline__ double aug_to_short(float coef) {                  // saturate_cast<short>(coef * 2048): nearest-even, clamped
    const double r = __builtin_rint((double)(coef * 2048.f));
    return r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
}

struct AugPlan { int kind, area, tx, ty, interp, area_mode; };
__device__ __forceinline__ AugPlan aug_plan_of(int Hc, int Wc, int out_h, int out_w, int interp, int n_taps) {
    AugPlan q;
    q.area = 1; q.tx = 1; q.ty = 1; q.interp = interp; q.area_mode = 0;
    if (Hc == out_h && Wc == out_w) { q.kind = PK_COPY; return q; }
    if (interp == 0) { q.kind = PK_NEAREST; return q; }
    const double scale_x = 1.0 / ((double)out_w / (double)Wc), scale_y = 1.0 / ((double)out_h / (double)Hc);
    const int isx = (int)__builtin_rint(scale_x), isy = (int)__builtin_rint(scale_y);
    const double eps = 2.220446049250313e-16;
    const bool fast = fabs(scale_x - (double)isx) < eps && fabs(scale_y - (double)isy) < eps;
    if (interp == 1 && fast && isx == 2 && isy == 2) interp = 3;
    if (interp == 3 && scale_x >= 1.0 && scale_y >= 1.0) {
        if (fast) { q.kind = (isx == 2 && isy == 2) ? PK_AREA_FAST2 : PK_AREA_FAST; q.area = isx * isy; q.tx = isx; q.ty = isy; }
        else {
            q.kind = PK_AREA;
            const int bx = (int)ceil(scale_x) + 2, by = (int)ceil(scale_y) + 2;
            q.tx = bx < n_taps ? bx : n_taps; q.ty = by < n_taps ? by : n_taps;
        }
        q.interp = 3;
        return q;
    }
    q.interp = interp;
    q.area_mode = interp == 3;
    if (interp == 1 || interp == 3) { q.kind = PK_LINEAR; q.tx = q.ty = 2; }
    else { q.kind = PK_KERNEL; q.tx = q.ty = interp == 2 ? 4 : 8; }
    return q;
}

// grid (blocks over positions, 2 axes, B); axis 0 = columns (x), 1 = rows (y); plan [B][4] = kind, area, taps per column, taps per row
__global__ __launch_bounds__(128) void aug_plan_kernel(const int* __restrict__ geo, int H, int W, int out_h, int out_w, int n_taps,
                                                       int* __restrict__ plan, int* __restrict__ ix, double* __restrict__ wx,
                                                       int* __restrict__ iy, double* __restrict__ wy) {
    const int b = (int)blockIdx.z, axis = (int)blockIdx.y;
    const int n_dst = axis == 0 ? out_w : out_h;
    const int i = (int)blockIdx.x * 128 + (int)threadIdx.x;
    const int* g = geo + (size_t)b * 12;
    const int Hc = g[8], Wc = g[9];
    const AugPlan q = aug_plan_of(Hc, Wc, out_h, out_w, g[11], n_taps);
    if (axis == 0 && i == 0) { plan[b * 4] = q.kind; plan[b * 4 + 1] = q.area; plan[b * 4 + 2] = q.tx; plan[b * 4 + 3] = q.ty; }
    if (i >= n_dst) return;
    const int n_src = axis == 0 ? Wc : Hc;
    const double inv = (double)n_dst / (double)n_src, scale = 1.0 / inv;
    int idx[64];
    double w[64];
    int T = 1;
    if (q.kind == PK_COPY) { idx[0] = i; w[0] = 1.0; }
    else if (q.kind == PK_NEAREST) {
        const double f = floor((double)i * scale);
        idx[0] = (int)(f < (double)(n_src - 1) ? f : (double)(n_src - 1));
        w[0] = 1.0;
    } else if (q.kind == PK_AREA_FAST || q.kind == PK_AREA_FAST2) {
        const int is = axis == 0 ? q.tx : q.ty;
        T = is > 64 ? 64 : is;
        for (int t = 0; t < T; ++t) { idx[t] = i * is + t; w[t] = 1.0; }
    } else if (q.kind == PK_AREA) {                                         // computeResizeAreaTab, the entries of destination i
        const double f1 = (double)i * scale, f2 = f1 + scale;
        const double room = (double)n_src - f1;
        const double cell = scale < room ? scale : room;
        long long s2 = (long long)floor(f2), s1 = (long long)ceil(f1);
        s2 = s2 < n_src - 1 ? s2 : n_src - 1;
        s1 = s1 < s2 ? s1 : s2;
        T = 0;
        if ((double)s1 - f1 > 1e-3) { idx[T] = (int)(s1 - 1); w[T] = (double)(float)(((double)s1 - f1) / cell); ++T; }
        for (long long sx = s1; sx < s2 && T < 63; ++sx) { idx[T] = (int)sx; w[T] = (double)(float)(1.0 / cell); ++T; }
        if (f2 - (double)s2 > 1e-3) {
            double m = f2 - (double)s2;
            m = m < 1.0 ? m : 1.0;
            m = m < cell ? m : cell;
            idx[T] = (int)s2; w[T] = (double)(float)(m / cell); ++T;
        }
        if (T == 0) { idx[0] = 0; w[0] = 0.0; T = 1; }
        for (int t = 0; t < T; ++t) idx[t] = idx[t] < 0 ? 0 : (idx[t] > n_src - 1 ? n_src - 1 : idx[t]);
    } else {
        long long sx;
        float fx;
        if (q.area_mode) {
            sx = (long long)floor((double)i * scale);
            fx = (float)(((double)i + 1.0) - ((double)sx + 1.0) * inv);
            fx = fx <= 0.f ? 0.f : fx - floorf(fx);
        } else {
            fx = (float)(((double)i + 0.5) * scale - 0.5);
            const float fl = floorf(fx);
            sx = (long long)fl;
            fx = fx - fl;
        }
        int off0;
        float co[8];
        if (q.kind == PK_LINEAR) {
            if (axis == 0) {                                                // only the x loop resets the pair at the borders
                if (sx < 0) { sx = 0; fx = 0.f; }
                else if (sx >= n_src - 1) { sx = n_src - 1; fx = 0.f; }
            }
            off0 = 0; T = 2;
            co[0] = 1.f - fx; co[1] = fx;
        } else if (q.interp == 2) {                                         // interpolateCubic, A = -0.75, float32 operation by operation
            off0 = -1; T = 4;
            const float A = -0.75f, u = fx + 1.f, v = 1.f - fx;
            co[0] = ((A * u - 5.f * A) * u + 8.f * A) * u - 4.f * A;
            co[1] = ((A + 2.f) * fx - (A + 3.f)) * fx * fx + 1.f;
            co[2] = ((A + 2.f) * v - (A + 3.f)) * v * v + 1.f;
            co[3] = 1.f - co[0] - co[1] - co[2];
        } else {                                                            // interpolateLanczos4
            off0 = -3; T = 8;
            if (fx < 1.1920928955078125e-07f) {
                for (int t = 0; t < 8; ++t) co[t] = 0.f;
                co[3] = 1.f;
            } else {
                const double r = 0.70710678118654752440084436210485;
                const double cs[8][2] = {{1, 0}, {-r, -r}, {0, 1}, {r, -r}, {-1, 0}, {r, r}, {0, -1}, {-r, r}};
                const float x3 = fx + 3.f;
                double s0, c0;
                aug_sincos_near_minus_pi(-((double)x3) * 3.1415926535897932384626433832795 * 0.25, s0, c0);
                float total = 0.f;
                for (int t = 0; t < 8; ++t) {
                    const double y = -((double)(x3 - (float)t)) * 3.1415926535897932384626433832795 * 0.25;
                    co[t] = (float)((cs[t][0] * s0 + cs[t][1] * c0) / (y * y));
                }
                for (int t = 0; t < 8; ++t) total = total + co[t];
                total = 1.f / total;
                for (int t = 0; t < 8; ++t) co[t] = co[t] * total;
            }
        }
        for (int t = 0; t < T; ++t) {
            const long long c = sx + off0 + t;
            idx[t] = (int)(c < 0 ? 0 : (c > n_src - 1 ? n_src - 1 : c));
            w[t] = aug_to_short(co[t]);
        }
    }
    // the recorded geometry: position j of the final (pre-resize) image <- flip <- crop window <- expansion canvas <- the image
    int* oi = (axis == 0 ? ix : iy) + ((size_t)b * n_dst + i) * n_taps;
    double* ow = (axis == 0 ? wx : wy) + ((size_t)b * n_dst + i) * n_taps;
    for (int t = 0; t < n_taps; ++t) {
        int j = idx[t < T ? t : T - 1];
        if (axis == 0 && g[10]) j = Wc - 1 - j;                                   // RandomFlip('horizontal')
        if (g[5]) j += axis == 0 ? g[7] : g[6];                                   // the crop window's corner on the canvas
        if (g[0]) {                                                               // the canvas: the image sits at (-top, -left)
            j += axis == 0 ? g[2] : g[1];
            if (j < 0 || j >= (axis == 0 ? W : H)) j = -1;
        }
        oi[t] = j;
        ow[t] = t < T ? w[t] : 0.0;
    }
}

}  // namespace ssdhip

using namespace ssdhip;

static int aug_params_from(const ssdhip_augment_params* q, AugParams& p);

// The whole batch on ONE generator stream, photometric decisions included (round 6; the reference's own generator semantics).
// mt_state / mt_state_out [625]; programs_ops [B][16] int32 + programs_args [B][16] float64: the per-image programs of
// ssdhip_image_program; the rest as ssdhip_ssd_augment_decide.  photo: probabilities and uniform ranges of RandomBrightness,
// RandomContrast, RandomSaturation, RandomHue (in that order); RandomChannelSwap must have probability 0 (SSDPhotometricDistortions).
extern "C" int ssdhip_ssd_augment_decide_stream(const ssdhip_augment_params* q, const ssdhip_augment_photo* photo, int B,
                                                const unsigned int* mt_state, const double* labels, const int* n_labels, int* programs_ops,
                                                double* programs_args, int* geometry, double* labels_out, int* n_labels_out,
                                                unsigned int* mt_state_out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!q || !photo || B <= 0 || !mt_state || !labels || !n_labels || !programs_ops || !programs_args || !geometry || !labels_out ||
        !n_labels_out || !mt_state_out)
        return SSDHIP_E_BADARG;
    if (photo->swap_prob != 0.0) return SSDHIP_E_BADARG;
    AugParams p;
    const int rc = aug_params_from(q, p);
    if (rc != SSDHIP_OK) return rc;
    AugPhoto ph;
    for (int i = 0; i < 4; ++i) {
        if (!(photo->prob[i] >= 0.0 && photo->prob[i] <= 1.0)) return SSDHIP_E_BADARG;
        ph.prob[i] = photo->prob[i]; ph.lo[i] = photo->lower[i]; ph.hi[i] = photo->upper[i];
    }
    ph.swap_prob = 0.0;
    hipLaunchKernelGGL(ssd_augment_stream_kernel, dim3(1), dim3(64), 0, stream, p, ph, B, mt_state, labels, n_labels, programs_ops,
                       programs_args, geometry, labels_out, n_labels_out, mt_state_out);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_ssd_augment_decide(const ssdhip_augment_params* q, int B, const unsigned int* mt_state, const double* labels,
                                         const int* n_labels, int* geometry, double* labels_out, int* n_labels_out,
                                         unsigned int* mt_state_out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!q || B <= 0 || !mt_state || !labels || !n_labels || !geometry || !labels_out || !n_labels_out || !mt_state_out) return SSDHIP_E_BADARG;
    AugParams p;
    const int rc = aug_params_from(q, p);
    if (rc != SSDHIP_OK) return rc;
    hipLaunchKernelGGL(ssd_augment_decide_kernel, dim3((unsigned)B), dim3(64), 0, stream, p, mt_state, labels, n_labels, geometry,
                       labels_out, n_labels_out, mt_state_out);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

static int aug_params_from(const ssdhip_augment_params* q, AugParams& p) {
    if (q->img_height <= 0 || q->img_width <= 0 || q->out_height <= 0 || q->out_width <= 0) return SSDHIP_E_BADARG;
    if (q->n_bounds < 1 || q->n_bounds > 8 || q->n_modes < 1 || q->n_modes > 8 || q->n_trials < 1) return SSDHIP_E_BADARG;
    if (!(q->expand_min_scale >= 1.0) || !(q->expand_max_scale > q->expand_min_scale)) return SSDHIP_E_BADARG;       // a canvas, never a crop
    if (!(q->crop_min_scale > 0.0) || !(q->crop_max_scale <= 1.0) || !(q->crop_max_scale > q->crop_min_scale)) return SSDHIP_E_BADARG;
    p.H = q->img_height; p.W = q->img_width;
    p.exp_prob = q->expand_prob; p.exp_min = q->expand_min_scale; p.exp_max = q->expand_max_scale;
    p.crop_prob = q->crop_prob; p.crop_min = q->crop_min_scale; p.crop_max = q->crop_max_scale;
    p.ar_min = q->crop_min_aspect_ratio; p.ar_max = q->crop_max_aspect_ratio;
    p.n_trials = q->n_trials; p.n_bounds = q->n_bounds;
    for (int i = 0; i < 8; ++i) { p.cdf[i] = q->bound_cdf[i]; p.lower[i] = q->bound_lower[i]; p.upper[i] = q->bound_upper[i]; p.modes[i] = q->interpolation_modes[i]; }
    p.flip_prob = q->flip_prob; p.n_modes = q->n_modes; p.out_h = q->out_height; p.out_w = q->out_width;
    p.max_rounds = q->max_rounds > 0 ? q->max_rounds : 100000;
    return SSDHIP_OK;
}

extern "C" int ssdhip_augment_plans(const int* geometry_dev, int B, int H, int W, int out_h, int out_w, int n_taps, int* plan_dev,
                                    int* ix_dev, double* wx_dev, int* iy_dev, double* wy_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!geometry_dev || !plan_dev || !ix_dev || !wx_dev || !iy_dev || !wy_dev || B <= 0 || B > 65535 || H <= 0 || W <= 0 || out_h <= 0 ||
        out_w <= 0)
        return SSDHIP_E_BADARG;
    if (n_taps < 8 || n_taps > 64) return SSDHIP_E_BADARG;                       // Lanczos-4 needs eight
    const int n = out_h > out_w ? out_h : out_w;
    hipLaunchKernelGGL(aug_plan_kernel, dim3((unsigned)((n + 127) / 128), 2, (unsigned)B), dim3(128), 0, stream, geometry_dev, H, W, out_h,
                       out_w, n_taps, plan_dev, ix_dev, wx_dev, iy_dev, wy_dev);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_augment_taps(const int* geometry_dev, int B, int H, int W, int out_h, int out_w, int n_taps, int* ix_dev,
                                   double* wx_dev, int* iy_dev, double* wy_dev, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!geometry_dev || !ix_dev || !wx_dev || !iy_dev || !wy_dev || B <= 0 || B > 65535 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return SSDHIP_E_BADARG;
    if (n_taps < 8 || n_taps > 64) return SSDHIP_E_BADARG;                       // Lanczos-4 needs eight
    const int n = out_h > out_w ? out_h : out_w;
    hipLaunchKernelGGL(aug_taps_kernel, dim3((unsigned)((n + 127) / 128), 2, (unsigned)B), dim3(128), 0, stream, geometry_dev, H, W, out_h,
                       out_w, n_taps, ix_dev, wx_dev, iy_dev, wy_dev);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
