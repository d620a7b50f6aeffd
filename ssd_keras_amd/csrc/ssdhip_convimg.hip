// ssdhip_convimg.hip -- 3x3 'same' convolution with ANY dilation on small maps (H W <= 384 pixels: fc6 of models/keras_ssd300.py:298,
// Conv2D(1024, (3, 3), dilation_rate=(6, 6)) on the 19 x 19 x 512 map behind pool5) + bias + ReLU, gfx950, bf16 NHWC, float32 accumulation.
//
// Why its own kernel.  The slab kernel (ssdhip_convh.hip) turns a tap into ONE byte displacement by walking a padded position grid; with
// dilation 6 that grid is 1.7 x the map.  The implicit-GEMM kernels (ssdhip_conv.hip) gather every tap's pixels again: 16 KB from L2 into
// LDS per MFLOP against the slab kernel's 4.7 -- fc6 runs at a third of the MFMA peak there, bound by what a CU pulls from L2.
// Here a tile is ONE IMAGE x 128 output channels:
//   * the 64-channel slice of the WHOLE image sits in LDS (H W rows of 144 bytes: 128 data + 16 pad, double buffered; the pad makes the
//     16-byte fragment reads of 32 consecutive pixels conflict-free without a swizzle term, as in ssdhip_conv64.hip);
//   * a tap is a per-lane ADDRESS: pixel (h, w) reads row (h + d dh, w + d dw) of the slice, or the slice's ZERO ROW when that lies outside
//     the map -- 9 taps x 3 pixel blocks = 27 address registers per lane, computed once; no border masks, no padded grid;
//   * the filters of a (slice, tap) step stream through a three-stage ring of 16 KB stages, requested two steps ahead (slab kernel);
//   * SSD300's fc6 at batch 32 is 32 images x 8 channel tiles = 256 tiles for 256 CUs; the 8 tiles of an image share an XCD (its L2 holds
//     the image once).
// Eight waves = 2 (64 output channels) x 4 (96 pixels = three 32-pixel MFMA blocks; 361 pixels pad to 384): six MFMAs per two filter and
// three pixel fragments, 0.83 KB of LDS reads per MFMA (slab kernel: 1 KB).  One barrier per (slice, tap) step.
// K order: slices outer, taps, 16-channel blocks -- the order of every other convolution kernel of the library, so the result is
// bit-identical to theirs (tests/test_conv_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CI_THREADS = 512;
constexpr int CI_PX = 384;                               // pixels of a tile (12 blocks of 32)
constexpr int CI_ROW = 144;                              // bytes of a pixel row in LDS
constexpr int CI_SLAB = (CI_PX + 1) * CI_ROW;            // + the zero row
[[maybe_unused]] constexpr int CI_ZERO = CI_PX * CI_ROW;                  // offset of the zero row inside a slab buffer
constexpr int CI_WST = 128 * 128;                        // a filter stage: [BM co][64 ci] bf16 (BM = 128; the 64-channel form uses half of it)
constexpr int CI_NW = 3;                                 // ring stages
constexpr int CI_W0 = 2 * CI_SLAB;                       // LDS: slab 0 | slab 1 | ring | dump
constexpr int CI_DUMP = CI_W0 + CI_NW * CI_WST;          // 1 KB that absorbs the dummy requests (every wave issues the same count per step)
constexpr int CI_BIAS = CI_DUMP + 1024;                  // the tile's bias values as float32
constexpr int CI_LDS = CI_BIAS + 512;
static_assert(CI_LDS <= 160 * 1024, "LDS budget");
static_assert(CI_SLAB % 16 == 0, "slab buffers are 16-byte aligned");

// In-kernel phase timers, profiling build only (tools/prof_build.sh): shader cycles of wave 0 in (prologue, wait + barrier, the rest of
// the steps, epilogue), summed over workgroups; read back with ssdhip_profile_read_convimg.
#ifdef SSDHIP_PROFILE
__device__ unsigned long long g_profi[8];
#define CI_PROF_DECL long long _pt = clock64(); long long _pa[4] = {0, 0, 0, 0};
#define CI_PROF_MARK(i) { const long long _t = clock64(); _pa[i] += _t - _pt; _pt = _t; }
#define CI_PROF_FLUSH if (tid == 0) { for (int _i = 0; _i < 4; ++_i) atomicAdd(&g_profi[_i], (unsigned long long)_pa[_i]); atomicAdd(&g_profi[4], 1ull); }
#else
#define CI_PROF_DECL
#define CI_PROF_MARK(i)
#define CI_PROF_FLUSH
#endif

struct ConvImgParams {
    const bf16_t* x;             // [B, H, W, Cin]
    const bf16_t* w;             // [Cout, 3, 3, Cin]
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [B, H, W, Cout]
    int B, H, W, Cin, Cout, dil, relu;
    int n_cot;                   // Cout / BM
    int xcd_map;                 // the channel tiles of an image on one XCD (needs n_cot == 8 or a tile count that is a multiple of 8 n_cot)
    int n_pxt;                   // pixel tiles per image (1 x 1 layers at stride 1 only, 128 pixels each with NPB = 1; 1: the whole image).
                                 // Round 6, fourth session: a 1 x 1 step has no taps to reuse the slice over, so what a tile costs is what it
                                 // pulls from L2 -- its pixels' rows + its channels' filters, Cin deep.  conv6_1 (1024 -> 256 on 19 x 19) as 32
                                 // images x 4 tiles of 64 channels x the whole image moved 867 KB per tile on half the chip's CUs; as 32 x 2 x 3
                                 // tiles of 128 channels x 128 pixels it moves 512 KB per tile on 192 CUs.
    int Ho, Wo, stride, pad;     // output map, stride and zero padding (the 3x3 'same' form: Ho = H, Wo = W, stride 1, pad = dil)
    // X3 (the reference-precision form, models/precise.py): x rows hold xC = 2 C float16 channels [hi | lo], the K loop walks Cin = 3 C
    // filter channels [w hi | w lo | w hi] and slice j reads x slice j < xslices ? j : j - xslices (hi, hi, lo); float32 bias, the
    // accumulator scaled by oscale, the activated float32 value split into the next layer's pair (or written as float32)
    int xC, xslices, out_f32;
    const float* bias32;
    float oscale;
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 ci_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ci_f32x2 __attribute__((ext_vector_type(2)));
typedef short ci_s16x2 __attribute__((ext_vector_type(2)));
typedef u32 ci_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32 ci_pack2(float a, float b) {
    const ci_f32x2 v = {a, b};
    return __builtin_bit_cast(u32, __builtin_convertvector(v, ci_bf16x2));
}
__device__ __forceinline__ u32 ci_pkmax_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(ci_s16x2, a), __builtin_bit_cast(ci_s16x2, b)));
}
// one wave-wide 1 KiB LDS-DMA load: lane L writes 16 bytes at lds_dst + 16 L from base(rsrc) + voff (zeros if out of range)
__device__ __forceinline__ void ci_bload(u32 voff, i32x4 rsrc, u32 lds_dst) {
    u32 keep;
    lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ i32x4 ci_rsrc(const void* base, u32 num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r.x = (int)(u32)a;
    r.y = (int)((u32)(a >> 32) & 0xffffu);
    r.z = (int)num_records;
    r.w = 0x00020000;
    return r;
}
#endif

// BM: output channels of a tile.  128: a wave owns 64 channels (two 32-channel MFMA blocks) x 96 pixels.  64: a wave owns ONE block x 96
// pixels -- twice the tiles (conv5_x: 32 images x 8 = 256 instead of 128 for 256 CUs) for 1.33 KB of LDS reads per MFMA instead of 0.83.
// KS: filter size, 3 (nine taps per 64-channel slice, any dilation / stride / padding: a tap is an address) or 1 (round 6: fc7, conv6_1 --
// ONE step per slice; the implicit-GEMM form moved 2.5 x the bytes per FLOP of a 3x3 layer from L2 into LDS and ran fc7 at a quarter of the
// peak).  NPB: 32-pixel MFMA blocks of a wave, 3 (output maps up to 384 pixels) or 1 (up to 128: the strided conv6_2, 19 x 19 -> 10 x 10,
// would idle in three quarters of a 384-pixel tile).
typedef _Float16 ci_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32 ci_split2(float a, float b, u32& lo_out) {          // hi = fl16(v), lo = fl16(v - hi), two values per word
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    lo_out = (u32)__builtin_bit_cast(unsigned short, la) | ((u32)__builtin_bit_cast(unsigned short, lb) << 16);
    return (u32)__builtin_bit_cast(unsigned short, ha) | ((u32)__builtin_bit_cast(unsigned short, hb) << 16);
}

template <int BM, int KS, int NPB, bool X3 = false>
__global__ __launch_bounds__(CI_THREADS, 1) void conv_image_kernel(ConvImgParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[CI_LDS];
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NCB = BM / 64;                         // 32-channel MFMA blocks of a wave
    constexpr int WCH = BM / 2;                          // output channels of a wave
    constexpr int NFP = BM / 64;                         // filter pieces (8 rows x 128 B) a wave requests per step
    constexpr int NT = KS * KS;                          // taps = steps per slice

    const int wm = wave & 1, wn = wave >> 1;             // WCH output channels x 32 NPB pixels per wave
    const int r31 = lane & 31, khalf = lane >> 5;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo;

    // ---- tile: image b, channel tile ct ---------------------------------------------------------------------------------------
    int b, ct;
    const int tpi = p.n_cot * p.n_pxt;                   // tiles per image
    if (p.xcd_map) {                                     // blockIdx & 7 = the XCD (round-robin dispatch): an image's tiles share it
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        ct = j % tpi;
        b = xcd + 8 * (j / tpi);
    } else {
        b = (int)blockIdx.x / tpi;
        ct = (int)blockIdx.x - b * tpi;
    }
    if (b >= p.B) return;
    const int pt = ct / p.n_cot;                         // pixel tile (0 unless n_pxt > 1: 1 x 1, stride 1 -- slab row r is pixel px0 + r)
    ct -= pt * p.n_cot;
    const int px0 = pt * (128 * NPB);
    const int npx = (KS == 1 && p.n_pxt > 1) ? min(HW - px0, 128 * NPB) : HW;     // input pixels the tile's slab holds
    const int co0 = ct * BM;
    const int n_slices = p.Cin >> 6, n_steps = n_slices * NT;
    constexpr int SR = (KS == 1 && NPB == 1) ? 3 : 7;    // slab request rounds of a slice (8 pieces each): 9 x 128 / 64 = 18 pieces | 9 x 384 / 64 = 54

    CI_PROF_DECL
    // ---- zero rows, descriptors ---------------------------------------------------------------------------------------------------
    if (tid < 2 * (CI_ROW / 4)) reinterpret_cast<u32*>(lds + (tid / (CI_ROW / 4)) * CI_SLAB + CI_ZERO)[tid % (CI_ROW / 4)] = 0u;
    // the bias of the tile's channels, fetched now and read from LDS in the epilogue (per-value global loads there cost 6 us per tile)
    if (tid >= 128 && tid < 128 + BM) {
        if constexpr (X3) reinterpret_cast<float*>(lds + CI_BIAS)[tid - 128] = p.bias32 ? p.bias32[co0 + tid - 128] : 0.f;
        else reinterpret_cast<float*>(lds + CI_BIAS)[tid - 128] = p.bias ? __uint_as_float((u32)p.bias[co0 + tid - 128] << 16) : 0.f;
    }
    const int XC = X3 ? p.xC : p.Cin;                    // channels of an x row
    const i32x4 rx = ci_rsrc(p.x + ((size_t)b * HW + px0) * XC, (u32)((size_t)npx * XC * 2));
    const i32x4 rw = ci_rsrc(p.w + (size_t)co0 * NT * p.Cin, (u32)((size_t)BM * NT * p.Cin * 2));

    // ---- request plan.  Every wave issues exactly NFP + 1 LDS-DMA pieces per step, in this order: [NFP filter pieces of step i + 2 | one slab
    //      piece of the next slice]; a request that has nothing to fetch goes out of range into the dump area.  So `s_waitcnt vmcnt(NFP + 2)`
    //      at the top of step i leaves the requests of step i - 1 and the slab piece of step i - 2 in flight: the filters of step i
    //      (issued at step i - 2) and every slab piece issued up to step i - 3 have landed -- a slice's pieces go out at its taps 0 .. 6
    //      (9 HW / 64 <= 54 pieces, eight per tap), three steps and more ahead of the next slice's first step.  (Waiting for the previous
    //      step's slab piece -- the first version -- exposed a memory round trip per step.)
    //      slab: slot n = 64 piece + lane -> pixel n / 9, 16-byte chunk n % 9 (8 = the row's pad: not fetched).
    //      filters of (slice s, tap t): [BM co][64 ci], piece = 8 rows; row r, chunk c at position c ^ ((r >> 1) & 7) ---------------------
    u32 wrel[NFP];                                       // byte offset of the lane's slot in filter piece NFP wave + i, (slice 0, tap 0)
#pragma unroll
    for (int i = 0; i < NFP; ++i) {
        const int row = (NFP * wave + i) * 8 + (lane >> 3);
        const int j = (lane & 7) ^ ((row >> 1) & 7);
        wrel[i] = (u32)((row * NT * p.Cin) * 2 + j * 16);
    }
    // slab piece (t 8 + wave) of `slice` into buffer `buf` (or a dummy when the slice does not exist / the piece lies beyond the map)
    auto issue_slab = [&](const int slice, const int t, const int buf) {
        const int piece = t * 8 + wave, n = piece * 64 + lane;
        const int px = n / 9, c = n - 9 * px;
        const bool ok = (slice < n_slices) & (px < npx) & (c < 8);
        const bool any = (slice < n_slices) & (piece * 64 < 9 * npx);         // wave-uniform: the piece holds at least one slot of the map
        const int xs = (X3 && slice >= p.xslices) ? slice - p.xslices : slice;      // X3: the x slice this K slice multiplies (hi, hi, lo)
        ci_bload(ok ? (u32)((px * XC + xs * 64) * 2 + c * 16) : OOB, rx, any ? lds0 + buf * CI_SLAB + piece * 1024 : lds0 + CI_DUMP);
    };
    // filters of step `wstep` = (slice, tap) into ring stage `stage`
    auto issue_filter_piece = [&](const int i, const int wstep, const int stage) {
        const bool wok = wstep < n_steps;
        const int ws = wstep / NT, wt = wstep - NT * ws;
        const u32 wo = (u32)((wt * p.Cin + ws * 64) * 2);
        ci_bload(wok ? wrel[i] + wo : OOB, rw, wok ? lds0 + CI_W0 + stage * CI_WST + (NFP * wave + i) * 1024 : lds0 + CI_DUMP);
    };
    auto issue_filters = [&](const int wstep, const int stage) {
#pragma unroll
        for (int i = 0; i < NFP; ++i) issue_filter_piece(i, wstep, stage);
    };

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------------
    u32 abase[4];                                        // filter stage 0: row = channel, chunk (2 kk + khalf) ^ ((row >> 1) & 7); + 32 rows for the second block
    {
        const int row = wm * WCH + r31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) abase[kk] = (u32)(CI_W0 + row * 128 + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4));
    }
    u32 baddr[NT][NPB];                                  // current slab buffer: row of the tap's source pixel (or the zero row) + the lane's K half
#pragma unroll
    for (int pi = 0; pi < NPB; ++pi) {
        const int q = px0 + wn * (32 * NPB) + pi * 32 + r31;   // output pixel (of the image)
        const int h = q / p.Wo, w = q - h * p.Wo;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int hh = h * p.stride - p.pad + p.dil * (t / KS), ww = w * p.stride - p.pad + p.dil * (t % KS);
            const bool ok = (q < HWo) & (hh >= 0) & (hh < p.H) & (ww >= 0) & (ww < p.W);
            baddr[t][pi] = (u32)((ok ? (hh * p.W + ww - px0) * CI_ROW : CI_ZERO) + khalf * 16);
        }
    }

    f32x16 acc[NCB][NPB];
#pragma unroll
    for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
        for (int pi = 0; pi < NPB; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;

    // ---- prologue: the whole slab of slice 0 (its nine piece rounds), filters of steps 0 and 1 -------------------------------------
#pragma unroll
    for (int t = 0; t < SR; ++t) issue_slab(0, t, 0);
    issue_filters(0, 0);
    issue_filters(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- K loop: slices outer (runtime), the nine taps unrolled (the tap selects registers and ring stages at compile time) --------------
    // Fragment reads run one 16-channel block ahead of the MFMAs through a two-slot register ring (the sched_barriers pin the order: hipcc
    // left to itself sinks every read next to its use).  INSIDE a slice the pixel fragments of a step's first block are read before the
    // step's barrier (the slab does not change); the filter fragments, and the first pixel fragments of a NEW slice, only behind it (the
    // last pieces of a slab go out three steps before the slice begins: `vmcnt(4)` + that barrier are what makes them visible).
    bf16x8 fa[2][NCB], fb[2][NPB];
    auto read_b = [&](const int t, const int kk, const int slot) {
#pragma unroll
        for (int pi = 0; pi < NPB; ++pi) fb[slot][pi] = *reinterpret_cast<const bf16x8*>(lds + baddr[t][pi] + kk * 32);
    };
    // `stage`: the ring stage of the step's filters -- (9 s + t) % 3 == t % 3 at compile time for 3x3, a running counter for 1x1
    auto read_a = [&](const u32 stage_off, const int kk, const int slot) {
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci) fa[slot][ci] = *reinterpret_cast<const bf16x8*>(lds + abase[kk] + stage_off + ci * (32 * 128));
    };
    CI_PROF_MARK(0)
    int stg = 0;                                         // (KS == 1) ring stage of the current step
    for (int s = 0; s < n_slices; ++s) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // the filters of this step (and the slab of this slice) have landed for this wave's share; the barrier makes that true for
            // all waves and tells that everybody has finished the previous step (whose filter stage and, behind a slice boundary, whose
            // slab buffer the requests below overwrite).
            // 3x3: a step requests [NFP filter pieces | one slab round]; NFP + 2 in flight = the previous step's + the slab round before.
            // 1x1: a step requests [7 slab rounds of the NEXT slice | NFP filter pieces of step + 2]: everything but the previous step's
            //      filter pieces (the most recent NFP requests) has to be here -- the slab of this slice was requested one step ago.
            CI_PROF_MARK(2)
            if constexpr (KS == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NFP + 2) : "memory");   // lgkmcnt: this wave's fragment reads of the previous step have RETURNED
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NFP) : "memory");
            __builtin_amdgcn_s_barrier();                                 // before anybody requests over the stage / buffer they came from
            CI_PROF_MARK(1)
            const u32 stage_off = (u32)((KS == 3 ? (t % CI_NW) : stg) * CI_WST);
            const int stage_p2 = KS == 3 ? (t + 2) % CI_NW : (stg == 0 ? 2 : stg - 1);      // the stage of step + 2
            read_a(stage_off, 0, 0);
            if (t == 0) read_b(0, 0, 0);                                 // a new slice: its slab is complete and visible only behind this barrier
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                if (kk < 3) { read_a(stage_off, kk + 1, (kk + 1) & 1); read_b(t, kk + 1, (kk + 1) & 1); }
                else if (t < NT - 1) read_b(t + 1, 0, 0);                // (last tap of a slice: after the buffer flip below)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pi = 0; pi < NPB; ++pi) {
                    if constexpr (X3) acc[0][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ci_f16x8, fa[kk & 1][0]), __builtin_bit_cast(ci_f16x8, fb[kk & 1][pi]), acc[0][pi], 0, 0, 0);
                    else acc[0][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][0], fb[kk & 1][pi], acc[0][pi], 0, 0, 0);
                }
                // the step's requests go out a few at a time BETWEEN MFMA blocks: an LDS-DMA instruction holds its wave
                // for 60-120 cycles, and issued together behind the barrier they left the matrix pipe idle in both waves of the SIMD at once
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KS == 3) {                                 // order: filters, slab
                    if (kk < NFP) issue_filter_piece(kk, s * NT + t + 2, stage_p2);
                    if (kk == NFP) issue_slab(s + 1, t, (s + 1) & 1);
                } else if constexpr (SR == 7) {                          // order: the seven slab rounds, then the filters
                    if (kk < 3) { issue_slab(s + 1, 2 * kk, (s + 1) & 1); issue_slab(s + 1, 2 * kk + 1, (s + 1) & 1); }
                    else {
                        issue_slab(s + 1, 6, (s + 1) & 1);
#pragma unroll
                        for (int i = 0; i < NFP; ++i) issue_filter_piece(i, s + 2, stage_p2);
                    }
                } else {                                                 // 128-pixel slabs: three rounds, then the filters
                    if (kk < 3) issue_slab(s + 1, kk, (s + 1) & 1);
                    else {
#pragma unroll
                        for (int i = 0; i < NFP; ++i) issue_filter_piece(i, s + 2, stage_p2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NCB == 2) {
#pragma unroll
                    for (int pi = 0; pi < NPB; ++pi) {
                        if constexpr (X3) acc[1][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ci_f16x8, fa[kk & 1][1]), __builtin_bit_cast(ci_f16x8, fb[kk & 1][pi]), acc[1][pi], 0, 0, 0);
                        else acc[1][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][1], fb[kk & 1][pi], acc[1][pi], 0, 0, 0);
                    }
                }
            }
            if constexpr (KS == 1) stg = stg == CI_NW - 1 ? 0 : stg + 1;
        }
        // the next slice lives in the other slab buffer
        const u32 flip = (s & 1) ? (u32)(-CI_SLAB) : (u32)CI_SLAB;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pi = 0; pi < NPB; ++pi) baddr[t][pi] += flip;
    }

    // ---- epilogue: bias, one rounding, ReLU on the rounded pair; 16-byte stores from the accumulator layout (ssdhip_conv64.hip) -----------
    CI_PROF_MARK(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy requests of the last two steps
    if constexpr (X3) {
        // ---- reference-precision epilogue: float32 straight from the registers (a lane holds 4 consecutive channels of a pixel per
        //      accumulator quad): scale, float32 bias, activation on float32, then the next layer's (hi, lo) pair or the float32 value ----
#pragma unroll
        for (int ci = 0; ci < NCB; ++ci)
#pragma unroll
            for (int pi = 0; pi < NPB; ++pi) {
                const int q = px0 + wn * (32 * NPB) + pi * 32 + r31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chl = wm * WCH + ci * 32 + 8 * g + 4 * khalf;         // channel inside the tile
                    const float4 b4 = *reinterpret_cast<const float4*>(lds + CI_BIAS + chl * 4);
                    const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[ci][pi][4 * g + e] * p.oscale + bv[e];
                        if (p.relu) t = t > 0.f ? t : (t != t ? t : 0.f);
                        v[e] = t;
                    }
                    if (q >= HWo) continue;
                    const size_t pix = (size_t)b * HWo + q;
                    if (p.out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + pix * p.Cout + co0 + chl) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        u32 l0, l1;
                        const u32 h0 = ci_split2(v[0], v[1], l0), h1 = ci_split2(v[2], v[3], l1);
                        bf16_t* row = p.y + pix * (2 * p.Cout) + co0 + chl;
                        *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(row + p.Cout) = make_uint2(l0, l1);
                    }
                }
            }
        CI_PROF_MARK(3)
        CI_PROF_FLUSH
        return;
    }
    const u32 floor16 = p.relu ? 0u : 0x80008000u;
    const size_t img = (size_t)HWo * p.Cout * 2;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.y) + (size_t)b * img, 0, (int)img, 0x00020000);
#pragma unroll
    for (int ci = 0; ci < NCB; ++ci) {
        float bv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t4 = *reinterpret_cast<const float4*>(lds + CI_BIAS + (wm * WCH + ci * 32 + 8 * g + 4 * khalf) * 4);
            bv[4 * g] = t4.x; bv[4 * g + 1] = t4.y; bv[4 * g + 2] = t4.z; bv[4 * g + 3] = t4.w;
        }
#pragma unroll
        for (int pi = 0; pi < NPB; ++pi) {
            const int q = px0 + wn * (32 * NPB) + pi * 32 + r31;
            const u32 voff = (u32)((q * p.Cout + co0 + wm * WCH + ci * 32) * 2 + khalf * 16) | (q < HWo ? 0u : OOB);
            u32 lo[4], hi[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                lo[g] = ci_pkmax_i16(ci_pack2(acc[ci][pi][4 * g] + bv[4 * g], acc[ci][pi][4 * g + 1] + bv[4 * g + 1]), floor16);
                hi[g] = ci_pkmax_i16(ci_pack2(acc[ci][pi][4 * g + 2] + bv[4 * g + 2], acc[ci][pi][4 * g + 3] + bv[4 * g + 3]), floor16);
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const auto sl = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
                const auto sh = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
                __builtin_amdgcn_raw_buffer_store_b128(ci_u32x4{sl[0], sh[0], sl[1], sh[1]}, ry, voff + pr * 32, 0, 0);
            }
        }
    }
    CI_PROF_MARK(3)
    CI_PROF_FLUSH
#endif
}

}  // namespace ssdhip

using namespace ssdhip;

// Channel tile and pixel tiles of a launch.  Whole-image tiles: 128 channels where that already gives three quarters of a chip's worth of
// tiles (fc6 / fc7 at batch 32: 256), else 64 (conv5_x, conv6_2 at batch 32: 256 tiles).  A 1 x 1 layer at stride 1 on a map of more than
// 128 pixels may instead run on 128-pixel tiles (ConvImgParams::n_pxt): the cheaper of the two by rounds of `n_cu` tiles x bytes a tile
// moves from L2 (its slab rows + its filters, Kc channels deep).  SSDHIP_CONVIMG_BM forces the channel tile, SSDHIP_CONVIMG_PXT=0 the
// whole image (A/B runs).
static void ci_pick_tiles(int B, int HW, int Kc, int Cout, int ksize, int stride, int padding, int HWo, int& bm, int& n_pxt) {
    bm = ((Cout % 128) == 0 && (long long)B * (Cout / 128) >= 192) ? 128 : 64;
    if (const char* e = getenv("SSDHIP_CONVIMG_BM")) { const int v = atoi(e); if (v == 64 || (v == 128 && (Cout % 128) == 0)) bm = v; }
    n_pxt = 1;
    const char* e = getenv("SSDHIP_CONVIMG_PXT");
    if (ksize != 1 || stride != 1 || padding != 0 || HWo <= 128 || (e && atoi(e) == 0) || getenv("SSDHIP_CONVIMG_BM")) return;
    int n_cu = 256;
    { int dev = 0, n = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) n_cu = n; }
    auto cost = [&](long long tiles, long long px, long long co) { return ((tiles + n_cu - 1) / n_cu) * (px + co) * Kc * 2; };
    long long best = cost((long long)B * (Cout / bm), HW, bm);
    const int npt = (HWo + 127) / 128;
    for (int co = 128; co >= 64; co -= 64) {
        if (Cout % co) continue;
        const long long c = cost((long long)B * (Cout / co) * npt, 128, co);
        if (c < best) { best = c; bm = co; n_pxt = npt; }
    }
}

// The general entry (round 6): k x k convolution, k in {1, 3}, of maps with H W <= 384 input pixels and at most 384 output pixels, one
// image per tile: x [B, H, W, Cin] bf16, weight [Cout, k, k, Cin] bf16, bias [Cout] bf16 or NULL, y [B, Ho, Wo, Cout] bf16 with
// Ho = (H + 2 padding - dilation (k - 1) - 1) / stride + 1 (Wo alike); Cin % 64 == 0, Cout % 64 == 0, 1 <= dilation <= 16,
// 1 <= stride <= 4, 0 <= padding <= dilation (k / 2).  Replaces Conv2D(Cout, (k, k), strides, padding, dilation_rate, activation) of
// models/keras_ssd300.py:298-313 -- fc6 (3x3, dilation 6), fc7 and conv6_1 (1x1), conv6_2 (ZeroPadding2D + 3x3 stride 2 'valid') --
// with the K order (and hence the bits) of ssdhip_conv2d_nhwc_bf16.  SSDHIP_E_BADARG for other geometries.
extern "C" int ssdhip_conv2d_image_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin,
                                             int Cout, int ksize, int stride, int padding, int dilation, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SSDHIP_E_BADARG;
    if ((ksize != 1 && ksize != 3) || stride < 1 || stride > 4 || dilation < 1 || dilation > 16 || padding < 0 ||
        padding > dilation * (ksize / 2))
        return SSDHIP_E_BADARG;
    if ((Cin % 64) || (Cout % 64) || (long long)H * W > CI_PX) return SSDHIP_E_BADARG;
    const int Ho = (H + 2 * padding - dilation * (ksize - 1) - 1) / stride + 1, Wo = (W + 2 * padding - dilation * (ksize - 1) - 1) / stride + 1;
    if (H + 2 * padding < dilation * (ksize - 1) + 1 || W + 2 * padding < dilation * (ksize - 1) + 1 || Ho < 1 || Wo < 1 ||
        (long long)Ho * Wo > CI_PX)
        return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    if ((long long)H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7ffff000LL || 128LL * 9 * Cin * 2 >= 0x7ffff000LL) return SSDHIP_E_BADARG;
    ConvImgParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = padding;
    p.xC = Cin; p.xslices = Cin / 64; p.out_f32 = 0; p.bias32 = nullptr; p.oscale = 1.f;
    int bm;
    ci_pick_tiles(B, H * W, Cin, Cout, ksize, stride, padding, Ho * Wo, bm, p.n_pxt);
    p.n_cot = Cout / bm;
    const dim3 grid((unsigned)(B * p.n_cot * p.n_pxt)), block(CI_THREADS);
    p.xcd_map = (B % 8 == 0) ? 1 : 0;                    // b = xcd + 8 (j / tiles per image) covers 0 .. B - 1 exactly when B is a multiple of 8
    // one 32-pixel block per wave; a 1 x 1 tile of that form requests 128 slab rows per slice, so a STRIDED 1 x 1 layer whose input
    // has more pixels than that stays on the three-block form whatever its output size
    const bool small = (Ho * Wo <= 128 && (ksize == 3 || H * W <= 128)) || p.n_pxt > 1;
#define CI_LAUNCH(BM_, KS_, NPB_) hipLaunchKernelGGL((conv_image_kernel<BM_, KS_, NPB_>), grid, block, 0, stream, p)
    if (ksize == 3) {
        if (bm == 128) { if (small) CI_LAUNCH(128, 3, 1); else CI_LAUNCH(128, 3, 3); }
        else { if (small) CI_LAUNCH(64, 3, 1); else CI_LAUNCH(64, 3, 3); }
    } else {
        if (bm == 128) { if (small) CI_LAUNCH(128, 1, 1); else CI_LAUNCH(128, 1, 3); }
        else { if (small) CI_LAUNCH(64, 1, 1); else CI_LAUNCH(64, 1, 3); }
    }
#undef CI_LAUNCH
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// The reference-precision (X3) form of the same kernel (round 6; models/precise.py): x [B, H, W, 2 C] float16 = [hi | lo], weight
// [Cout, k, k, 3 C] float16 = [w hi | w lo | w hi] (x3_pack_weight), bias float32 [Cout] or NULL, y [B, Ho, Wo, 2 Cout] float16 pairs
// or, with out_f32, [B, Ho, Wo, Cout] float32.  One K loop over 3 C channels, float32 accumulation, the K order of
// ssdhip_conv2d_x3_nhwc_f16 (bit-identical to it).  C % 64 == 0, Cout % 64 == 0; geometry limits as ssdhip_conv2d_image_nhwc_bf16.
extern "C" int ssdhip_conv2d_image_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C,
                                               int Cout, int ksize, int stride, int padding, int dilation, int relu, int out_f32,
                                               float oscale, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return SSDHIP_E_BADARG;
    if ((ksize != 1 && ksize != 3) || stride < 1 || stride > 4 || dilation < 1 || dilation > 16 || padding < 0 ||
        padding > dilation * (ksize / 2))
        return SSDHIP_E_BADARG;
    if ((C % 64) || (Cout % 64) || (long long)H * W > CI_PX) return SSDHIP_E_BADARG;
    const int Ho = (H + 2 * padding - dilation * (ksize - 1) - 1) / stride + 1, Wo = (W + 2 * padding - dilation * (ksize - 1) - 1) / stride + 1;
    if (H + 2 * padding < dilation * (ksize - 1) + 1 || W + 2 * padding < dilation * (ksize - 1) + 1 || Ho < 1 || Wo < 1 ||
        (long long)Ho * Wo > CI_PX)
        return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 3)) return SSDHIP_E_BADARG;
    if ((long long)H * W * 3 * (C > Cout ? C : Cout) * 4 >= 0x7ffff000LL || 128LL * 9 * 3 * C * 2 >= 0x7ffff000LL) return SSDHIP_E_BADARG;
    ConvImgParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = nullptr;
    p.y = static_cast<bf16_t*>(y);
    p.B = B; p.H = H; p.W = W; p.Cin = 3 * C; p.Cout = Cout; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = padding;
    p.xC = 2 * C; p.xslices = C / 64; p.out_f32 = out_f32 ? 1 : 0; p.bias32 = bias; p.oscale = oscale;
    int bm;
    ci_pick_tiles(B, H * W, 3 * C, Cout, ksize, stride, padding, Ho * Wo, bm, p.n_pxt);
    p.n_cot = Cout / bm;
    const dim3 grid((unsigned)(B * p.n_cot * p.n_pxt)), block(CI_THREADS);
    p.xcd_map = (B % 8 == 0) ? 1 : 0;
    const bool small = (Ho * Wo <= 128 && (ksize == 3 || H * W <= 128)) || p.n_pxt > 1;
#define CI_LAUNCH3(BM_, KS_, NPB_) hipLaunchKernelGGL((conv_image_kernel<BM_, KS_, NPB_, true>), grid, block, 0, stream, p)
    if (ksize == 3) {
        if (bm == 128) { if (small) CI_LAUNCH3(128, 3, 1); else CI_LAUNCH3(128, 3, 3); }
        else { if (small) CI_LAUNCH3(64, 3, 1); else CI_LAUNCH3(64, 3, 3); }
    } else {
        if (bm == 128) { if (small) CI_LAUNCH3(128, 1, 1); else CI_LAUNCH3(128, 1, 3); }
        else { if (small) CI_LAUNCH3(64, 1, 1); else CI_LAUNCH3(64, 1, 3); }
    }
#undef CI_LAUNCH3
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// 3x3 'same' convolution (padding = dilation) of maps with H W <= 384 pixels: the round-4 entry, now the general one at stride 1 and
// padding = dilation (fc6, conv5_x; the data gradient of the same layers in the training step).
extern "C" int ssdhip_conv3x3_image_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin,
                                              int Cout, int dilation, int relu, void* stream_) {
    return ssdhip_conv2d_image_nhwc_bf16(x, weight, bias, y, B, H, W, Cin, Cout, 3, 1, dilation, dilation, relu, stream_);
}

#ifdef SSDHIP_PROFILE
// profiling build only: out[0..3] = wave 0's cycles in (prologue, wait + barrier, steps, epilogue) summed over workgroups, out[4] = workgroups
extern "C" int ssdhip_profile_read_convimg(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_profi), sizeof(g_profi)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_profi), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
