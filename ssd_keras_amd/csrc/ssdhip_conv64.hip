// ssdhip_conv64.hip -- 3x3 'same' convolution for the Cin = 64 layers (conv1_2, conv2_1 of models/keras_ssd300.py:275-279
// and twins) + bias + ReLU [+ 2x2/2 max-pool], gfx950, bf16 NHWC, float32 accumulation.
//
// Why its own kernel.  On these layers the implicit-GEMM kernel of ssdhip_conv.hip moves 216 KB from L2 into LDS per
// 128-pixel x 64-channel tile (nine tap tiles of activations + nine weight tiles) for 72 MFMAs per wave, and pays nine
// barriers: it runs at ~19 % of the MFMA peak there.  With Cin = 64 the WHOLE filter bank of a 64-channel output slice is
// 9 x 64 x 64 x 2 B = 72 KB -- it fits in LDS next to the activations.  So:
//   * persistent workgroups (one per CU): the weight slice is loaded ONCE and stays resident -- in LDS in the first version, since
//     round 3 in the multiplying waves' REGISTERS (WREG: 36 fragments = 144 VGPRs per wave), which halves the LDS read traffic;
//   * per tile only the activation HALO (the (2 RP + 2) x (CC + 2) pixels a 2 RP x CC tile reads through its nine taps,
//     ~23 KB) comes in -- by LDS-DMA from a dedicated LOADER wave, three tiles ahead, while the current tile is multiplied; the
//     nine taps are nine constant byte displacements into that halo (rows are padded to 144 B so a displacement does not change
//     the bank pattern: no swizzle term, every fragment address is base register + immediate);
//   * image borders need no per-tap masks: out-of-image halo pixels are out-of-range buffer offsets, which the buffer unit
//     turns into zeros in LDS;
//   * ONE barrier per tile (9x fewer), L2 -> LDS traffic 9x lower.
// One multiplying wave per SIMD (the workgroup owns the CU's LDS), so the K loop is software pipelined by hand: the 36 (tap, k16)
// steps of a tile are fully unrolled with the fragment reads running four steps ahead of the MFMAs through a register ring.
// The epilogue (round 4) leaves from the accumulator layout: bias, one bf16 rounding, ReLU as v_pk_max_i16 on the rounded pair,
// v_permlane32_swap to 16-byte runs, buffer stores -- no LDS transpose (with one wave per SIMD every epilogue cycle is MFMA idle
// time: the in-kernel phase timers of profiles/r04p_* put the staged epilogue at 40 % of a tile).
//
// Tile geometry and numerics are those of conv_igemm4_pool_kernel (ssdhip_conv.hip): 2-D tiles of RP row pairs x CC columns;
// pooled epilogue = 2x2 max on the float32 accumulators (vertical in-lane, horizontal by DPP), then bias, one bf16 rounding, ReLU.
// Results are bit-identical to the implicit-GEMM kernel (same K order per output: taps outer, channels inner; rounding and ReLU
// commute).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "ssdhip.h"
#include "ssdhip_math.h"

#ifndef SSDHIP_C64_ABLATE
#define SSDHIP_C64_ABLATE 0      // profiling builds only (tools/prof_build.sh): 1 no global stores, 2 no epilogue, 4 no fragment reads (wrong results)
#endif

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// In-kernel phase timers of the Cin = 64 kernels, profiling build only (tools/prof_build.sh): shader cycles of wave 0 (multiplier) and
// wave 4 (loader / first producer) per phase, summed over tiles and workgroups; read back with ssdhip_profile_read_c64.
#ifdef SSDHIP_PROFILE
__device__ unsigned long long g_prof64[32];
#define C64_PROF_DECL long long _pt = clock64(); long long _pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int _pn = 0;
#define C64_PROF_MARK(i) { const long long _t = clock64(); _pa[i] += _t - _pt; _pt = _t; }
#define C64_PROF_TILE ++_pn;
#define C64_PROF_FLUSH(base) if (lane == 0) { for (int _i = 0; _i < 8; ++_i) atomicAdd(&g_prof64[(base) + _i], (unsigned long long)_pa[_i]); atomicAdd(&g_prof64[(base) + 8], (unsigned long long)_pn); }
#else
#define C64_PROF_DECL
#define C64_PROF_MARK(i)
#define C64_PROF_TILE
#define C64_PROF_FLUSH(base)
#endif

#ifndef SSDHIP_C64_NLOAD
#define SSDHIP_C64_NLOAD 1       // loader waves of the non-fused kernel (each issues every NLOAD-th LDS-DMA piece of a halo); 2 measured equal (r03zb)
#endif
constexpr int C64_NLOAD = SSDHIP_C64_NLOAD;
constexpr int C64_THREADS = (4 + C64_NLOAD) * 64;        // four multiplying waves (one per SIMD) + the loader waves
constexpr int C64_NPROD = 3;                             // FRONT: producer waves
constexpr int C64_FRONT_THREADS = (4 + C64_NPROD) * 64;  // FRONT: four multiplying waves + the producers
constexpr int C64_WBYTES = 9 * 64 * 128;                 // resident weight slice: [tap][co][64 ci] bf16, 128-byte rows
constexpr int C64_STAGE = 0;                             // (rounds 2-3: a per-wave LDS stage for the output transpose; the epilogue now stores from the accumulator layout)
constexpr int c64_halo_bytes(int cs) { return ((9 * (2 * (64 >> cs) + 2) * ((1 << cs) + 2) + 63) / 64) * 1024; }
constexpr int C64_PATCH1 = 1600;                         // FRONT: one producer's copy of the 3-channel input patch of a tile's halo (13 x 20 px x 3 ch bf16 = 1560 B)
constexpr int C64_PATCH = ((C64_NPROD * C64_PATCH1 + 1023) / 1024) * 1024;
constexpr int C64_WREG_BYTES = 1024;                     // WREG: the filters live in registers; this region only holds the float32 bias of the slice
constexpr int c64_lds_bytes(int cs, int nb, bool front = false, bool wreg = false) {
    return (wreg ? C64_WREG_BYTES : C64_WBYTES) + nb * c64_halo_bytes(cs) + C64_STAGE + (front ? C64_PATCH : 0);
}

struct C64Params {
    const bf16_t* x;             // [B, H, W, 64]
    const bf16_t* w;             // [Cout, 3, 3, 64]
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [B, H, W, Cout]  or pooled [B, Ho, Wo, Cout]
    int B, H, W, Cout, relu;
    int HT, WT, n_slices, tiles; // tiles per image grid: HT x WT, tiles = B * HT * WT
    int Ho, Wo;                  // pooled map (POOL)
    int x_bytes, w_bytes;
    const bf16_t* x3;            // FRONT: [B, H, W, 3] image; x is unused
    const bf16_t* w1;            // FRONT: [64, 3, 3, 3] first-layer filters
    const bf16_t* b1;            // FRONT: [64] first-layer bias or null
    int prio;                    // FRONT: raise the multiplying waves' issue priority over the producers'
    int xcd_pairs;               // slice / tile-sequence mapping that keeps a tile's n_slices workgroups on one XCD (needs G % (8 n_slices) == 0)
    bf16_t* y2;                  // KEEP (training, round 6): the full-resolution activation [B, H, W, Cout] beside the pooled y
};

__device__ __forceinline__ u32 c64_f2bf_rn(float f) {
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 c64_bf16x2 __attribute__((ext_vector_type(2)));
typedef float c64_f32x2 __attribute__((ext_vector_type(2)));
// two float32 -> packed bf16, round to nearest even: one v_cvt_pk_bf16_f32 (the integer formulation of ssdhip_conv.hip costs
// ~6 VALU operations per value, and with one wave per SIMD every epilogue instruction is MFMA idle time)
__device__ __forceinline__ u32 c64_pack2(float a, float b) {
    const c64_f32x2 v = {a, b};
    return __builtin_bit_cast(u32, __builtin_convertvector(v, c64_bf16x2));
}
// max(v, v of lane ^ 1) in ONE VALU instruction (DPP quad_perm [1,0,3,2] on the first operand).  The s_nop covers the
// VALU-write -> DPP-read hazard, which hipcc does not pad inside an asm statement; the operand always comes from a VALU op.
__device__ __forceinline__ float c64_max_with_lane_xor1(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
typedef short c64_s16x2 __attribute__((ext_vector_type(2)));
typedef u32 c64_u32x2 __attribute__((ext_vector_type(2)));
// v_pk_max_i16 on two packed bf16 values (see the epilogue)
__device__ __forceinline__ u32 c64_pkmax_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(c64_s16x2, a), __builtin_bit_cast(c64_s16x2, b)));
}
typedef u32 c64_u32x4 __attribute__((ext_vector_type(4)));
// The epilogue's stores.  In the accumulator layout the lane (pixel, khalf) holds channels 8 g + 4 khalf + (0 .. 3) of the wave's 32 as
// (lo[g], hi[g]) for g = 0 .. 3: 8-byte runs.  v_permlane32_swap trades them between the two lanes of a pixel (lane, lane + 32) so that
// the khalf = 0 lane ends up with all 16 bytes of g = 0 / 2 and the khalf = 1 lane with those of g = 1 / 3: two 16-byte stores per
// lane instead of four 8-byte ones -- the CU's vector memory pipe (these stores + the loader's LDS-DMA pieces) is what a tile of the
// non-fused kernels waits for.  voff = the lane's pixel offset + 16 khalf (or out of range), soff = the tile origin.
__device__ __forceinline__ void c64_store_runs(const u32 (&lo)[4], const u32 (&hi)[4], __amdgpu_buffer_rsrc_t ry, u32 voff, u32 soff) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const auto sl = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
        const auto sh = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
        if (!(SSDHIP_C64_ABLATE & 1)) __builtin_amdgcn_raw_buffer_store_b128(c64_u32x4{sl[0], sh[0], sl[1], sh[1]}, ry, voff + pr * 32, soff, 0);
    }
}
__device__ __forceinline__ float c64_relu(float v) { return v <= 0.f ? 0.f : v; }     // NaN stays NaN, -0 -> +0 (as ssdhip_conv.hip)

// one wave-wide 1 KiB LDS-DMA load (see ssdhip_conv.hip: inline asm so hipcc does not drain vmcnt before aliasing ds_reads)
__device__ __forceinline__ void c64_bload(u32 voff, i32x4 rsrc, u32 lds_dst, u32 soff = 0) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

// one two-byte buffer load (zero-extended), not tracked by hipcc's wait-count insertion: the caller counts vmcnt
__device__ __forceinline__ u32 c64_load_u16(u32 voff, i32x4 rsrc, u32 soff) {
    u32 v;
    asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}

__device__ __forceinline__ i32x4 c64_rsrc(const void* base, int num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r.x = (int)(u32)a;
    r.y = (int)((u32)(a >> 32) & 0xffffu);
    r.z = num_records;
    r.w = 0x00020000;
    return r;
}

// FRONT: the 64-channel input map is never read from memory -- it is conv1_1 (3 -> 64 channels, models/keras_ssd300.py:274) of the
// 3-channel image, recomputed per tile by the fifth wave straight into the halo buffer (see the producer section below).
// WREG: each multiplying wave keeps the 36 filter fragments of its 32 output channels in REGISTERS (144 VGPRs, loaded once per launch)
// instead of reading them from LDS every step.  With the filters in LDS a step reads 3 KB per wave for two MFMAs -- 12 KB per 64 MFMA
// cycles per CU, 75 % of what ds_read_b128 delivers (256 B/clk), and the producers' gathers and halo stores come on top: the fused
// block was LDS-bound.  With WREG a step reads the two pixel fragments only (50 %).
// KEEP (POOL only; the training step's Conv2D(relu) -> MaxPooling2D pairs): the pooled epilogue ALSO stores the full-resolution activation
// the backward pass needs -- the accumulators survive the first epilogue -- instead of a second launch that reads the 368 MB map back to pool it.
template <int CS, bool POOL, int NB, bool FRONT, bool WREG, bool KEEP = false>
__device__ __forceinline__ void conv64_body(const C64Params& p, unsigned char* lds) {
    static_assert(!KEEP || (POOL && !FRONT), "KEEP is a variant of the pooled, un-fused kernel");
    constexpr int CC = 1 << CS, RP = 64 >> CS;           // tile: RP row pairs x CC columns = 128 pixels
    constexpr int HC = CC + 2, HR = 2 * RP + 2;          // halo columns / rows
    constexpr int HPX = HR * HC;                         // halo pixels, one 144-byte LDS row each (128 data + 16 pad)
    constexpr int SLOTS = 9 * HPX;                       // 16-byte slots of a halo buffer
    constexpr int NP = (SLOTS + 63) / 64;                // 1 KiB LDS-DMA pieces per halo
    constexpr int HB = NP * 1024;                        // NB halo buffers: loads run NB - 1 tiles ahead of the MFMAs
    static_assert(HB == c64_halo_bytes(CS) && c64_lds_bytes(CS, NB, FRONT, WREG) <= 160 * 1024, "LDS budget");
    constexpr int WB = WREG ? C64_WREG_BYTES : C64_WBYTES;
    constexpr int W_OFF = 0, H_OFF = WB, STAGE_OFF = WB + NB * HB, PATCH_OFF = STAGE_OFF + C64_STAGE;
    constexpr unsigned OOB = 0x80000000u;

    const int G = (int)gridDim.x;
    // The n_slices workgroups that walk the SAME tile sequence (one per 64-channel output slice) read the same halos at about the same
    // time.  Workgroups are dealt to the 8 XCDs round-robin by blockIdx, and each XCD has its own L2: with slice = blockIdx % n_slices the
    // partners sat on different XCDs and every halo came from memory once per slice.  So partners share blockIdx % 8.
    int slice = (int)blockIdx.x % p.n_slices, first = (int)blockIdx.x / p.n_slices;
    const int stride = G / p.n_slices;
    if (p.xcd_pairs) {
        const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
        slice = j % p.n_slices;
        first = xcd + 8 * (j / p.n_slices);
    }
    if (first >= p.tiles) return;
    const int co0 = slice * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int r31 = lane & 31, khalf = lane >> 5;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // activations: descriptor base one image row + one pixel BEFORE x, so the halo origin of any tile is a non-negative scalar
    // offset; gfx950 range-checks voffset + soffset, hence the widened num_records (valid lanes still only touch bytes of x)
    const int xneg = (p.W + 1) * 128;
    const i32x4 rx = c64_rsrc(reinterpret_cast<const unsigned char*>(p.x) - xneg, p.x_bytes + 2 * xneg);
    const i32x4 rw = c64_rsrc(p.w, p.w_bytes);

    // ---- resident weights: tap t -> [64 co rows][128 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) ----------
    if (!WREG && wave < 4) {
        const int pos = lane & 7;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int piece = i * 4 + wave;
                const int row = piece * 8 + (lane >> 3);
                const int j = pos ^ ((row >> 1) & 7);
                c64_bload((u32)((co0 + row) * 1152 + t * 128 + j * 16), rw, lds0 + W_OFF + t * 8192 + piece * 1024);
            }
    }

    auto tile_origin = [&](int tile, int& b, int& h0, int& w0) {
        const int wt = tile % p.WT, r = tile / p.WT;
        b = r / p.HT;
        h0 = (r - b * p.HT) * (2 * RP);
        w0 = wt * CC;
    };

    // ---- FRONT: the producer wave.  The halo of a tile is conv1_1 of the (HR + 2) x (HC + 2) x 3 input patch around it: K = 27
    //      (padded to 32, k = kh * 9 + kw * 3 + ci as in conv3x3_cin3_kernel -- same operands, same MFMA, hence the same bf16 values
    //      the separate kernel would have written to memory), 64 channels x 180 halo pixels = 24 v_mfma_f32_32x32x16_bf16 per tile
    //      against the 288 of the four multiplying waves.  Per tile: 12 two-byte global loads per lane (requested a tile ahead),
    //      the patch into LDS, per 32 halo pixels 16 two-byte gathers -> B operand, 4 MFMAs, bias + ReLU + rounding (halo pixels
    //      outside the image become ZERO: they are conv1_2's padding, not conv1_1 of padding), 8-byte stores into the 144-byte halo
    //      rows.  Two halo buffers: tile i + 1's halo is produced while tile i is multiplied. ----------------------------------
    if constexpr (FRONT) {
      if (wave >= 4) {
        const int prod = wave - 4;                           // producer p: halo pixel blocks p, p + NPROD, ...; each producer keeps its own patch copy
        constexpr int PR = HR + 2, PC = HC + 2, PROW = PC * 3, PATCH = PR * PROW;      // patch rows / columns / elements per row
        constexpr int NRAW = (PATCH + 63) / 64, NBLK = (HPX + 31) / 32, MYB = NBLK / C64_NPROD;
        static_assert(NB == 2 && (PR + 1) * PROW * 2 <= C64_PATCH1 && NRAW * 64 * 2 <= C64_PATCH1 && NBLK % C64_NPROD == 0,
                      "two halo buffers; a patch copy holds the patch, a spare row and every raw slot");
        unsigned char* patch = lds + PATCH_OFF + prod * C64_PATCH1;
        for (int i = lane; i < C64_PATCH1 / 4; i += 64) reinterpret_cast<u32*>(patch)[i] = 0u;   // the spare row is read (x zero weights)
        // first-layer filters as the MFMA 'A' operand: row = channel, k = (2 st + khalf) * 8 + j, zero beyond k = 26
        bf16x8 aw[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                union { bf16x8 v; unsigned short u[8]; } t;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = (2 * st + khalf) * 8 + j;
                    t.u[j] = k < 27 ? p.w1[(cb * 32 + r31) * 27 + k] : (unsigned short)0;
                }
                aw[cb][st] = t.v;
            }
        float b1v[2][16];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    b1v[cb][4 * g + e] = p.b1 ? __uint_as_float((u32)p.b1[cb * 32 + 8 * g + 4 * khalf + e] << 16) : 0.f;
        // the lane's halo pixel in each block of 32: patch offset of its top-left tap, halo-row offset, (hr, hc)
        // gather offsets (bytes into the patch) of the lane's 16 k values relative to its pixel: k -> (kh, r = kw * 3 + ci)
        u32 kofs[2][8];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = (2 * st + khalf) * 8 + j;
                kofs[st][j] = (u32)(((k / 9) * PROW + (k % 9)) * 2);          // k = 27 .. 31 read row kh = 3 (x zero weights)
            }
        // the lane's halo pixel in each block of 32: patch offset of its top-left tap, halo-row offset, (hr, hc)
        u32 pbase[MYB], hdst[MYB], hcoord[MYB];
#pragma unroll
        for (int blk = 0; blk < MYB; ++blk) {
            const int hp = (C64_NPROD * blk + prod) * 32 + r31, hpc = hp < HPX ? hp : HPX - 1;
            const int hr = hpc / HC, hc = hpc - hr * HC;
            pbase[blk] = (u32)((hr * PROW + hc * 3) * 2);
            hdst[blk] = (u32)(hpc * 144 + 8 * khalf);
            hcoord[blk] = (u32)((hr << 8) | hc | (hp < HPX ? 0 : 0x10000));
        }
        // raw-load slots: element n = 64 i + lane of the patch = (pr, pc, ci); two-byte buffer loads, an out-of-range offset reads 0
        const i32x4 r3 = c64_rsrc(p.x3, p.B * p.H * p.W * 6);
        u32 rrel[NRAW], rco[NRAW];
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int n = 64 * i + lane, nn = n < PATCH ? n : 0;
            const int pr = nn / PROW, rem = nn - pr * PROW, pc = rem / 3;
            rrel[i] = n < PATCH ? (u32)(((pr * p.W) * 3 + rem) * 2) : OOB;
            rco[i] = (u32)((pr << 8) | pc);
        }
        // (asm loads + a hand-counted wait in front of the patch stores a tile later: with the builtin hipcc converted / paired the
        //  16-bit values right behind the loads and waited for them there -- one exposed memory round trip per tile, the largest
        //  single item of this wave's 6 500 cycles per tile in the in-kernel timers of profiles/r04p1_*)
        auto request = [&](int tile, u32 (&raw)[NRAW]) {
            int b, h0, w0;
            tile_origin(tile, b, h0, w0);
            const int ph = h0 - 2, pw = w0 - 2;                               // image position of patch element (0, 0)
            const bool inside = ph >= 0 && pw >= 0 && ph + PR <= p.H && pw + PC <= p.W;
            if (inside) {                                                     // the tile's position is a scalar offset
                const u32 soff = (u32)((((b * p.H + ph) * p.W + pw) * 3) * 2);
#pragma unroll
                for (int i = 0; i < NRAW; ++i) raw[i] = c64_load_u16(rrel[i], r3, soff);
            } else {
                const int base = ((b * p.H + ph) * p.W + pw) * 6;             // may be negative: only added to offsets of in-image elements
#pragma unroll
                for (int i = 0; i < NRAW; ++i) {
                    const int pr = (int)(rco[i] >> 8) & 0xff, pc = (int)rco[i] & 0xff;
                    const bool ok = rrel[i] != OOB && (unsigned)(ph + pr) < (unsigned)p.H && (unsigned)(pw + pc) < (unsigned)p.W;
                    raw[i] = c64_load_u16(ok ? (u32)(base + (int)rrel[i]) : OOB, r3, 0u);
                }
            }
        };
        C64_PROF_DECL
        // every request is exactly NRAW loads and this wave issues no other VMEM operation: `newer` = a younger request is in flight
        auto landed = [&](u32 (&raw)[NRAW], bool newer) {
            if (newer) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NRAW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NRAW; ++i) asm volatile("" : "+v"(raw[i]));
        };
        // Gathers of ALL the wave's blocks first, then all their MFMAs, then the epilogues: the LDS and matrix-pipe latencies of one block
        // are covered by the other's work (the pipe is shared with the SIMD's multiplying wave, whose MFMAs run back to back).
        auto produce = [&](int tile, const u32 (&raw)[NRAW], int buf) {
            int b, h0, w0;
            tile_origin(tile, b, h0, w0);
#pragma unroll
            for (int i = 0; i < NRAW; ++i) reinterpret_cast<unsigned short*>(patch)[64 * i + lane] = (unsigned short)raw[i];   // slots beyond the patch hold 0
            unsigned char* hb = lds + H_OFF + buf * HB;
            bf16x8 bfr[MYB][2];
#pragma unroll
            for (int blk = 0; blk < MYB; ++blk)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    union { bf16x8 v; unsigned short u[8]; } t;
#pragma unroll
                    for (int j = 0; j < 8; ++j) t.u[j] = *reinterpret_cast<const unsigned short*>(patch + pbase[blk] + kofs[st][j]);
                    bfr[blk][st] = t.v;
                }
            // The matrix pipe of this SIMD is shared with its multiplying wave, whose MFMAs are always ready and which wins the issue
            // arbitration (older wave; raised priority): this wave's eight MFMAs used to sit out the multiplier's whole K loop (in-kernel
            // timers, r04p5: 3 200 cycles in this section, the multipliers then waiting 1 000 at the barrier for the epilogue below).
            // For these eight instructions the producer goes first; they cost the multiplier 256 cycles of pipe time per tile.
            if (p.prio & 2) __builtin_amdgcn_s_setprio(3);
            f32x16 a1[MYB][2];
#pragma unroll
            for (int blk = 0; blk < MYB; ++blk)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) a1[blk][cb][v] = 0.f;
                    a1[blk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[cb][0], bfr[blk][0], a1[blk][cb], 0, 0, 0);
                }
#pragma unroll
            for (int blk = 0; blk < MYB; ++blk)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) a1[blk][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[cb][1], bfr[blk][1], a1[blk][cb], 0, 0, 0);
            if (p.prio & 2) __builtin_amdgcn_s_setprio(0);
            C64_PROF_MARK(2)
            // bias, one rounding (as the separate kernel), conv1_1's ReLU on the rounded pair (v_pk_max_i16 with 0: the compare + select
            // pairs and their hazard states were a third of this wave's instructions).  Halo pixels outside the image are conv1_2's
            // zero padding; only border tiles have any (16 % of a 300 x 300 map), the others skip the masks.
            auto finish = [&](auto edge) {
                constexpr bool EDGE = decltype(edge)::value;
#pragma unroll
                for (int blk = 0; blk < MYB; ++blk) {
                    const int hr = (int)(hcoord[blk] >> 8) & 0xff, hc = (int)hcoord[blk] & 0xff;
                    // only the very last block of 32 has lanes beyond the halo's pixels
                    const bool live = (blk + 1 < MYB) || !(hcoord[blk] & 0x10000);
                    const bool in_img = !EDGE || ((unsigned)(h0 - 1 + hr) < (unsigned)p.H && (unsigned)(w0 - 1 + hc) < (unsigned)p.W);
                    const u32 keep = in_img ? 0xffffffffu : 0u;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = a1[blk][cb][4 * g + e] + b1v[cb][4 * g + e];
                            u32 lo = c64_pkmax_i16(c64_pack2(o[0], o[1]), 0u), hi = c64_pkmax_i16(c64_pack2(o[2], o[3]), 0u);
                            if constexpr (EDGE) { lo &= keep; hi &= keep; }
                            if (live) *reinterpret_cast<uint2*>(hb + hdst[blk] + (cb * 32 + 8 * g) * 2) = make_uint2(lo, hi);
                        }
                }
            };
            if (h0 >= 1 && w0 >= 1 && h0 - 1 + HR <= p.H && w0 - 1 + HC <= p.W) finish(std::false_type{});
            else finish(std::true_type{});
            C64_PROF_MARK(3)
        };
        // Two register sets used alternately (the tile loop is unrolled by two by hand): the set requested in step i is stored into the
        // patch in step i + 1 and never copied, so the only wait for it sits in front of those stores, a whole tile after the request.
        u32 raw_a[NRAW], raw_b[NRAW];
        request(first, raw_a);
        landed(raw_a, false);
        produce(first, raw_a, 0);
        if (first + stride < p.tiles) request(first + stride, raw_a);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // halo of the first tile (+ the multipliers' weights) in place
        int buf = 0;
        auto step = [&](int tile, u32 (&cur)[NRAW], u32 (&nxt)[NRAW]) {
            const bool req = !(SSDHIP_C64_ABLATE & 8) && tile + 2 * stride < p.tiles;                       // 8: idle producers
            if (req) request(tile + 2 * stride, nxt);
            C64_PROF_MARK(0)
            if (!(SSDHIP_C64_ABLATE & 8) && tile + stride < p.tiles) {
                landed(cur, req);
                C64_PROF_MARK(1)
                produce(tile + stride, cur, buf ^ 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            C64_PROF_MARK(4)
            __builtin_amdgcn_s_barrier();                // multipliers done with `buf`; halo of the next tile complete in the other one
            C64_PROF_MARK(5)
            C64_PROF_TILE
            buf ^= 1;
        };
        for (int tile = first; tile < p.tiles; tile += 2 * stride) {
            step(tile, raw_a, raw_b);
            if (tile + stride >= p.tiles) break;
            step(tile + stride, raw_b, raw_a);
        }
        if (wave == 4) { C64_PROF_FLUSH(16) }
        return;
      }
    }

    // ---- the loader wave.  VMEM operations complete in issue order and a global store takes microseconds to be acknowledged:
    //      a wave that both stores outputs and waits for halo loads ends up waiting for its own older stores every tile (the
    //      first version of this kernel: 4 us per tile whatever the prefetch depth).  So the halos are fetched by a fifth wave
    //      that never stores -- its vmcnt counts loads only -- and the four multiplying waves never wait on vmcnt at all.
    //      Per tile i:  loader: halo i+1 landed (halo i+2 may fly) | barrier | issue halo i+3 into the buffer tile i released. ----
    if (!FRONT && wave >= 4) {
        // loader wave lw issues pieces lw, lw + NLOAD, ... of every halo (NPW each).
        // slot n = 64 * piece + lane of a halo buffer -> halo pixel n / 9 (row hr, column hc), 16-byte chunk n % 9 (8 = padding)
        const int lw = wave - 4;
        constexpr int NPW = NP / C64_NLOAD;
        static_assert(NP % C64_NLOAD == 0, "every loader wave issues the same number of pieces (compile-time wait counts)");
        int hrc[NPW];                                    // hr << 16 | hc << 4 | chunk, or -1 (padding slot / beyond the halo)
        u32 rel[NPW];                                    // byte offset of the slot's 16 bytes from the halo origin pixel (or OOB)
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int n = 64 * (k * C64_NLOAD + lw) + lane;
            const int px = n / 9, c = n - 9 * px;
            const int hr = px / HC, hc = px - hr * HC;
            const bool data = c < 8 && px < HPX;
            hrc[k] = data ? ((hr << 16) | (hc << 4) | c) : -1;
            rel[k] = data ? (u32)((hr * p.W + hc) * 128 + c * 16) : OOB;
        }
        // One LDS-DMA instruction costs the issuing wave 60-185 cycles; 26 per tile plus ~10 VALU of address arithmetic each made
        // this wave the bottleneck (the multipliers waited at the barrier).  For a tile whose whole halo lies inside the image --
        // all but the border tiles -- the per-lane offset is the constant rel[k] and the tile only contributes a scalar soffset.
        auto issue_halo = [&](int tile, int buf) {
            int b, h0, w0;
            tile_origin(tile, b, h0, w0);
            const u32 dst = lds0 + H_OFF + buf * HB + lw * 1024;
            if (h0 >= 1 && w0 >= 1 && h0 + 2 * RP + 1 <= p.H && w0 + CC + 1 <= p.W) {
                const u32 soff = (u32)(((b * p.H + h0 - 1) * p.W + (w0 - 1)) * 128 + xneg);
#pragma unroll
                for (int k = 0; k < NPW; ++k) c64_bload(rel[k], rx, dst + k * C64_NLOAD * 1024, soff);
            } else {
#pragma unroll
                for (int k = 0; k < NPW; ++k) {
                    const int h = h0 - 1 + (hrc[k] >> 16), w = w0 - 1 + ((hrc[k] >> 4) & 0xfff);
                    const bool ok = hrc[k] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                    const u32 voff = ok ? (u32)(((b * p.H + h) * p.W + w) * 128 + (hrc[k] & 15) * 16 + xneg) : OOB;
                    c64_bload(voff, rx, dst + k * C64_NLOAD * 1024);
                }
            }
        };
        static_assert(FRONT || (NB == 3 && 2 * NPW <= 63), "wait accounting: two halos of NPW loads in flight must fit vmcnt");
        issue_halo(first, 0);
        if (first + stride < p.tiles) issue_halo(first + stride, 1);
        if (first + 2 * stride < p.tiles) issue_halo(first + 2 * stride, 2);
        if (first + 2 * stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPW) : "memory");
        else if (first + stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // halo of the first tile (+ the multipliers' weights) in place
        int buf = 0;
        C64_PROF_DECL
        for (int tile = first; tile < p.tiles; tile += stride) {
            if (tile + 2 * stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");  // halo i+1 landed (this wave's pieces)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C64_PROF_MARK(0)
            __builtin_amdgcn_s_barrier();                // the multipliers are done with buffer `buf`
            C64_PROF_MARK(1)
            if (tile + 3 * stride < p.tiles) issue_halo(tile + 3 * stride, buf);
            C64_PROF_MARK(2)
            C64_PROF_TILE
            buf = buf + 1 == NB ? 0 : buf + 1;
        }
        if (wave == 4) { C64_PROF_FLUSH(16) }
        return;
    }

    if (p.prio & 1) __builtin_amdgcn_s_setprio(2);           // the multipliers go first whenever they and a producer / loader wave of their SIMD can issue
    // ---- fragment addresses: everything per-lane is fixed for the whole kernel -------------------------------------------
    const int q = wp * 32 + r31;                         // pixel slot of the lane: row pair q >> CS, column q & (CC-1)
    const int rp = q >> CS, col = q & (CC - 1);
    u32 bbase[2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) bbase[pi] = (u32)(H_OFF + ((2 * rp + pi) * HC + col) * 144 + khalf * 16);
    // output: byte offset of the lane's pixel (pooled form: its pooled pixel) from the tile origin, + its 16-byte column of a 32-byte pair (c64_store_runs)
    u32 ylane[2];
    [[maybe_unused]] u32 yfull[2] = {0u, 0u};
    if constexpr (KEEP) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) yfull[pi] = (u32)((((2 * rp + pi) * p.W + col) * p.Cout) * 2 + khalf * 16);
    }
    if constexpr (POOL) {
        ylane[0] = ylane[1] = (u32)(((rp * p.Wo + (col >> 1)) * p.Cout) * 2 + khalf * 16);
    } else {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) ylane[pi] = (u32)((((2 * rp + pi) * p.W + col) * p.Cout) * 2 + khalf * 16);
    }
    u32 abase[4];
    {
        const int row = wc * 32 + r31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) abase[kk] = (u32)(W_OFF + row * 128 + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4));
    }
    // WREG: fragment (tap t, slice kk) of the lane's channel row = 16 bytes of weight[co][t][kk * 16 + khalf * 8 ..], straight from memory
    bf16x8 wreg[WREG ? 36 : 1];
    if constexpr (WREG) {
        const bf16_t* wrow = p.w + (size_t)(co0 + wc * 32 + r31) * 576 + khalf * 8;
#pragma unroll
        for (int s = 0; s < 36; ++s) wreg[s] = *reinterpret_cast<const bf16x8*>(wrow + (s >> 2) * 64 + (s & 3) * 16);
        if (wave == 0) reinterpret_cast<float*>(lds + W_OFF)[lane] = p.bias ? __uint_as_float((u32)p.bias[co0 + lane] << 16) : 0.f;
    }
    // bias of the lane's 16 channels (D row = channel (v & 3) + 8 * (v >> 2) + 4 * khalf of the wave's 32)
    float bv[16];
    if constexpr (!WREG) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = co0 + wc * 32 + 8 * g + 4 * khalf + e;
                bv[4 * g + e] = p.bias ? __uint_as_float((u32)p.bias[ch] << 16) : 0.f;
            }
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's share of the weights (the only loads it ever waits for)
    if constexpr (WREG) {
#pragma unroll
        for (int s = 0; s < 36; ++s) asm volatile("" : "+v"(wreg[s]));       // loaded once: never re-fetched, never re-derived
    }
    __builtin_amdgcn_s_barrier();

    int buf = 0;
    C64_PROF_DECL
    for (int tile = first; tile < p.tiles; tile += stride) {
        const unsigned char* hb = lds + buf * HB;

        f32x16 acc[2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[pi][v] = 0.f;

        // 36 steps (tap t = s / 4, k16 slice kk = s % 4).  One wave per SIMD: nothing but this wave's own MFMAs covers the LDS
        // latency, so fragments are read RD steps ahead through an (RD + 1)-slot register ring (hipcc left to itself sinks
        // every read next to its use; the sched_barriers pin the order).
        constexpr int RD = WREG ? 4 : 5, RS = RD + 1;
        bf16x8 fa[WREG ? 1 : RS], fb0[RS], fb1[RS];
        auto rd = [&](const int s, const int slot) {
            const int t = s >> 2, kk = s & 3;
            const int timm = ((t / 3) * HC + (t % 3)) * 144 + kk * 32;
            fb0[slot] = *reinterpret_cast<const bf16x8*>(hb + bbase[0] + timm);   // the step's first MFMA consumes the LAST two reads,
            if constexpr (!WREG) fa[slot] = *reinterpret_cast<const bf16x8*>(lds + abase[kk] + t * 8192);  // so one counted wait per step covers all three
            fb1[slot] = *reinterpret_cast<const bf16x8*>(hb + bbase[1] + timm);
        };
#pragma unroll
        for (int s = 0; s < RD; ++s) rd(s, s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if ((s + RD < 36) && (!(SSDHIP_C64_ABLATE & 4) || s + RD < RS)) rd(s + RD, (s + RD) % RS);
            __builtin_amdgcn_sched_barrier(0);             // pin the order: hipcc otherwise sinks the reads next to their use
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WREG ? wreg[WREG ? s : 0] : fa[WREG ? 0 : s % RS], fb1[s % RS], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WREG ? wreg[WREG ? s : 0] : fa[WREG ? 0 : s % RS], fb0[s % RS], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (WREG) {                              // the lane's 16 bias values from the LDS table (registers hold the filters),
#pragma unroll                                             // requested ahead of the barrier so that they land while the wave waits there
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = *reinterpret_cast<const float4*>(lds + W_OFF + (wc * 32 + 8 * g + 4 * khalf) * 4);
                bv[4 * g] = t4.x; bv[4 * g + 1] = t4.y; bv[4 * g + 2] = t4.z; bv[4 * g + 3] = t4.w;
            }
        }
        C64_PROF_MARK(0)
        __builtin_amdgcn_s_barrier();                      // the loader wave arrives here once halo i+1 has landed; all four
        buf = buf + 1 == NB ? 0 : buf + 1;                 // multipliers are done reading buffer `buf`, which it may now refill
        C64_PROF_MARK(1)

        // ---- epilogue, straight from the accumulator layout (round 4).  A lane holds, for its pixel, channels 8 g + 4 khalf + (0 .. 3)
        //      of the wave's 32 for g = 0 .. 3: four 8-byte runs, and the (khalf = 0, 1) lane pair of a pixel makes each a 16-byte run.
        //      So the values go out as four buffer_store_dwordx2 per accumulator block with NO transpose through LDS: the first
        //      version staged every block through LDS for 16-byte stores -- write, wait, read, wait, 64-bit address arithmetic per
        //      store -- and with one wave per SIMD all of it was MFMA idle time (in-kernel timers, profiles/r04p1_*: 1 700 - 2 400 of
        //      a tile's 4 600 - 5 200 cycles).  L2 merges the partial lines; a tile writes 16 KB per ~3 000 cycles, nowhere near a
        //      store-path limit.  Per 4 values: 2 packed bias adds, 2 packed conversions, 2 v_pk_max_i16 (ReLU on the rounded pair:
        //      for bf16 bit patterns integer order is numeric order and -0 / negative values become +0, as c64_relu; +NaN stays NaN).
        //      The store address is image base (descriptor) + tile origin (scalar) + a per-lane constant; pixels outside the map get
        //      an out-of-range offset, which the buffer unit drops -- no branches. ------------------------------------------------
        int b, h0, w0;
        tile_origin(tile, b, h0, w0);
        const u32 floor16 = p.relu ? 0u : 0x80008000u;
        if constexpr ((SSDHIP_C64_ABLATE & 2) != 0) {
            asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
        } else
        if constexpr (POOL) {
            if constexpr (KEEP) {                         // the un-pooled form's stores (below), into y2
                const size_t img2 = (size_t)p.H * p.W * p.Cout * 2;
                const __amdgpu_buffer_rsrc_t ry2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.y2) + (size_t)b * img2, 0, (int)img2, 0x00020000);
                const u32 sbase2 = (u32)(((h0 * p.W + w0) * p.Cout + co0 + wc * 32) * 2);
#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    const bool ok = ((h0 + 2 * rp + pi) < p.H) & ((w0 + col) < p.W);
                    const u32 voff2 = yfull[pi] | (ok ? 0u : OOB);
                    u32 lo[4], hi[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        lo[g] = c64_pkmax_i16(c64_pack2(acc[pi][4 * g] + bv[4 * g], acc[pi][4 * g + 1] + bv[4 * g + 1]), floor16);
                        hi[g] = c64_pkmax_i16(c64_pack2(acc[pi][4 * g + 2] + bv[4 * g + 2], acc[pi][4 * g + 3] + bv[4 * g + 3]), floor16);
                    }
                    c64_store_runs(lo, hi, ry2, voff2, sbase2);
                }
            }
            const size_t img = (size_t)p.Ho * p.Wo * p.Cout * 2;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.y) + (size_t)b * img, 0, (int)img, 0x00020000);
            const u32 sbase = (u32)((((h0 >> 1) * p.Wo + (w0 >> 1)) * p.Cout + co0 + wc * 32) * 2);
            const int hq = h0 + 2 * rp, wq = w0 + col;
            const bool okp = (!(r31 & 1)) & (((h0 >> 1) + rp) < p.Ho) & ((wq >> 1) < p.Wo);
            const u32 voff = ylane[0] | (okp ? 0u : OOB);
            // 2x2 maximum on the float32 accumulators (vertical in-lane, horizontal by DPP), then bias, rounding, ReLU (monotonic:
            // equals pooling the rounded activations).  Interior tiles (95 % of a 300x300 map) take the mask-free path.
            auto pooled = [&](auto edge) {
                constexpr bool EDGE = decltype(edge)::value;
                const bool has_below = !EDGE || hq + 1 < p.H, has_right = !EDGE || wq + 1 < p.W;
                u32 lo[4], hi[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[0][4 * g + e];
                        if constexpr (EDGE) {
                            const float below = acc[1][4 * g + e];
                            if (has_below) v = below > v ? below : v;
                            const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
                            if (has_right) v = right > v ? right : v;
                        } else {
                            const float below = acc[1][4 * g + e];
                            v = below > v ? below : v;
                            v = c64_max_with_lane_xor1(v);
                        }
                        o[e] = v + bv[4 * g + e];
                    }
                    lo[g] = c64_pkmax_i16(c64_pack2(o[0], o[1]), floor16);
                    hi[g] = c64_pkmax_i16(c64_pack2(o[2], o[3]), floor16);
                }
                c64_store_runs(lo, hi, ry, voff, sbase);
            };
            if (h0 + 2 * RP <= p.H && w0 + CC <= p.W) pooled(std::false_type{});
            else pooled(std::true_type{});
        } else {
            const size_t img = (size_t)p.H * p.W * p.Cout * 2;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.y) + (size_t)b * img, 0, (int)img, 0x00020000);
            const u32 sbase = (u32)(((h0 * p.W + w0) * p.Cout + co0 + wc * 32) * 2);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const bool ok = ((h0 + 2 * rp + pi) < p.H) & ((w0 + col) < p.W);
                const u32 voff = ylane[pi] | (ok ? 0u : OOB);
                u32 lo[4], hi[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    lo[g] = c64_pkmax_i16(c64_pack2(acc[pi][4 * g] + bv[4 * g], acc[pi][4 * g + 1] + bv[4 * g + 1]), floor16);
                    hi[g] = c64_pkmax_i16(c64_pack2(acc[pi][4 * g + 2] + bv[4 * g + 2], acc[pi][4 * g + 3] + bv[4 * g + 3]), floor16);
                }
                c64_store_runs(lo, hi, ry, voff, sbase);
            }
        }
        C64_PROF_MARK(2)
        C64_PROF_TILE
    }
    if (wave == 0) { C64_PROF_FLUSH(0) }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int CS, bool POOL, int NB, bool FRONT = false, bool WREG = false, bool KEEP = false>
__global__ __launch_bounds__(FRONT ? C64_FRONT_THREADS : C64_THREADS, 1) void conv64_kernel(C64Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[c64_lds_bytes(CS, NB, FRONT, WREG)];
    conv64_body<CS, POOL, NB, FRONT, WREG, KEEP>(p, lds);
#endif
}

}  // namespace ssdhip

using namespace ssdhip;

// pool != 0: MaxPooling2D(2, 2, 'same') fused; y is [B, ceil(H/2), ceil(W/2), Cout].  n_workgroups: persistent workgroups to
// launch (the caller passes the CU count; 0 = 256).
static int c64_launch(const void* x, const void* weight, const void* bias, void* y, void* y2, int B, int H, int W, int Cin, int Cout, int relu,
                      int pool, int n_workgroups, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || Cin != 64 || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)y2) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    if (y2 && !pool) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * 128, wb = (long long)Cout * 1152;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout > 0x7fffffff0LL) return SSDHIP_E_BADARG;
    C64Params p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb;
    p.x3 = nullptr; p.w1 = nullptr; p.b1 = nullptr;
    p.y2 = static_cast<bf16_t*>(y2);
    { const char* e = getenv("SSDHIP_C64_PRIO"); p.prio = e ? atoi(e) : 1; }   // multipliers first; equal within the spread since WREG (profiles/r03zd_*)
    p.n_slices = Cout / 64;
    int cs_best = 4;
    long long best = -1;
    for (int cs = 4; cs >= 3; --cs) {                     // tile shape (16 x 8 or 8 x 16 pixels) with the fewer padded tiles
        const long long wt = (W + (1 << cs) - 1) >> cs, ht = (p.Ho + (64 >> cs) - 1) / (64 >> cs);
        if (best < 0 || wt * ht < best) { best = wt * ht; cs_best = cs; p.WT = (int)wt; p.HT = (int)ht; }
    }
    const long long tiles = (long long)B * p.HT * p.WT;
    if (tiles > 0x3fffffffLL) return SSDHIP_E_BADARG;
    p.tiles = (int)tiles;
    int G = n_workgroups > 0 ? n_workgroups : 256;
    if (G > 4096) G = 4096;
    G = (G / p.n_slices) * p.n_slices;
    if (G < p.n_slices) G = p.n_slices;
    { const char* e = getenv("SSDHIP_C64_XCD"); p.xcd_pairs = (p.n_slices > 1 && G % (8 * p.n_slices) == 0 && !(e && atoi(e) == 0)) ? 1 : 0; }
    // (Round 4 tried these layers on EIGHT multiplying waves without a loader wave -- 16 x 16 tiles, every wave requesting a share of
    //  the halo between its MFMAs, stores and requests in one counted vmcnt queue: 8 % slower (the two waves of a SIMD meet at the same
    //  barrier every tile, so one's epilogue does not overlap the other's MFMAs; profiles/r04v_c64_eight_waves_negative.txt), and on a
    //  second visit a few hundred stale outputs per run, cause not established in the GPU time that was left.  Removed.)
    static const bool wreg = []() { const char* e = getenv("SSDHIP_C64_WREG"); return e ? atoi(e) != 0 : true; }();
#define C64_LAUNCH(CS_, POOL_) do { if (wreg) hipLaunchKernelGGL((conv64_kernel<CS_, POOL_, 3, false, true>), dim3(G), dim3(C64_THREADS), 0, stream, p); \
                                    else hipLaunchKernelGGL((conv64_kernel<CS_, POOL_, 3>), dim3(G), dim3(C64_THREADS), 0, stream, p); } while (0)
    if (pool && y2) {                                    // KEEP: filters in registers only
        if (cs_best == 3) hipLaunchKernelGGL((conv64_kernel<3, true, 3, false, true, true>), dim3(G), dim3(C64_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((conv64_kernel<4, true, 3, false, true, true>), dim3(G), dim3(C64_THREADS), 0, stream, p);
    } else if (pool) {
        if (cs_best == 3) C64_LAUNCH(3, true); else C64_LAUNCH(4, true);
    } else {
        if (cs_best == 3) C64_LAUNCH(3, false); else C64_LAUNCH(4, false);
    }
#undef C64_LAUNCH
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_conv3x3_c64_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                            int Cin, int Cout, int relu, int pool, int n_workgroups, void* stream) {
    return c64_launch(x, weight, bias, y, nullptr, B, H, W, Cin, Cout, relu, pool, n_workgroups, stream);
}

// Conv2D(relu) -> MaxPooling2D(2, 2, 'same') of the TRAINING step in one launch (round 6): y_pooled [B, ceil(H/2), ceil(W/2), Cout] as with
// pool != 0 above AND y_full [B, H, W, Cout], the activation the backward pass needs -- bit-identical to the un-pooled launch followed by
// ssdhip_bias_act_maxpool, without reading the full map back.
extern "C" int ssdhip_conv3x3_c64_pool_keep_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y_full, void* y_pooled, int B,
                                                      int H, int W, int Cin, int Cout, int relu, int n_workgroups, void* stream) {
    if (!y_full) return SSDHIP_E_BADARG;
    return c64_launch(x, weight, bias, y_pooled, y_full, B, H, W, Cin, Cout, relu, 1, n_workgroups, stream);
}

// conv1_1 -> conv1_2 [-> pool1] as ONE kernel (models/keras_ssd300.py:274-276, keras_ssd512.py twin): x3 [B, H, W, 3] bf16 (the
// in-graph preprocessing's output), w1 [64, 3, 3, 3] + b1 the first layer (always followed by ReLU), weight [Cout, 3, 3, 64] + bias
// the second one.  The 64-channel map between them (368 MB at batch 32, 300 x 300) never exists: every tile's halo of it is
// recomputed from the image by a producer wave.  Results are bit-identical to ssdhip_conv3x3_cin3_nhwc_bf16 followed by
// ssdhip_conv3x3_c64_nhwc_bf16.
extern "C" int ssdhip_conv1_block_nhwc_bf16(const void* x3, const void* w1, const void* b1, const void* weight, const void* bias, void* y,
                                            int B, int H, int W, int Cout, int relu, int pool, int n_workgroups, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x3 || !w1 || !weight || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)weight | (uintptr_t)y) & 15 || (((uintptr_t)x3 | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)bias) & 1)) return SSDHIP_E_BADARG;
    const long long wb = (long long)Cout * 1152;
    if ((long long)B * H * W * 6 >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout > 0x7fffffff0LL) return SSDHIP_E_BADARG;   // 31-bit byte offsets
    C64Params p;
    p.x = nullptr; p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.x3 = static_cast<const bf16_t*>(x3); p.w1 = static_cast<const bf16_t*>(w1); p.b1 = static_cast<const bf16_t*>(b1);
    p.y2 = nullptr;
    { const char* e = getenv("SSDHIP_C64_PRIO"); p.prio = e ? atoi(e) : 3; }   // bit 0: multipliers above the producers (equal since WREG, profiles/r03zd_*); bit 1: the producers' own MFMAs above everything
    p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.x_bytes = 0; p.w_bytes = (int)wb;
    p.n_slices = Cout / 64;
    int cs_best = 4;
    long long best = -1;
    for (int cs = 4; cs >= 3; --cs) {                     // tile shape (16 x 8 or 8 x 16 pixels) with the fewer padded tiles
        const long long wt = (W + (1 << cs) - 1) >> cs, ht = (p.Ho + (64 >> cs) - 1) / (64 >> cs);
        if (best < 0 || wt * ht < best) { best = wt * ht; cs_best = cs; p.WT = (int)wt; p.HT = (int)ht; }
    }
    const long long tiles = (long long)B * p.HT * p.WT;
    if (tiles > 0x3fffffffLL) return SSDHIP_E_BADARG;
    p.tiles = (int)tiles;
    int G = n_workgroups > 0 ? n_workgroups : 256;
    if (G > 4096) G = 4096;
    G = (G / p.n_slices) * p.n_slices;
    if (G < p.n_slices) G = p.n_slices;
    { const char* e = getenv("SSDHIP_C64_XCD"); p.xcd_pairs = (p.n_slices > 1 && G % (8 * p.n_slices) == 0 && !(e && atoi(e) == 0)) ? 1 : 0; }
    static const bool wreg = []() { const char* e = getenv("SSDHIP_C64_WREG"); return e ? atoi(e) != 0 : true; }();
#define C64F_LAUNCH(CS_, POOL_) do { if (wreg) hipLaunchKernelGGL((conv64_kernel<CS_, POOL_, 2, true, true>), dim3(G), dim3(C64_FRONT_THREADS), 0, stream, p); \
                                     else hipLaunchKernelGGL((conv64_kernel<CS_, POOL_, 2, true>), dim3(G), dim3(C64_FRONT_THREADS), 0, stream, p); } while (0)
    if (pool) {
        if (cs_best == 3) C64F_LAUNCH(3, true); else C64F_LAUNCH(4, true);
    } else {
        if (cs_best == 3) C64F_LAUNCH(3, false); else C64F_LAUNCH(4, false);
    }
#undef C64F_LAUNCH
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

#ifdef SSDHIP_PROFILE
// profiling build only (32 words): out[0..2] = multiplier wave 0's cycles in (K loop, barrier, epilogue), out[8] its tiles; out[16..] = wave 4's
// (loader: wait, barrier, halo issue; producer: request, wait for the older request, block 0, block 1, LDS drain, barrier), out[24] its
// tiles -- summed over workgroups since the last reset
extern "C" int ssdhip_profile_read_c64(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_prof64), sizeof(g_prof64)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof64), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
