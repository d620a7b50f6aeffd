// ssdhip_wgrad.hip -- weight gradient of the 3x3 'same' stride-1 convolutions of the VGG trunk (models/keras_ssd300.py:274-296, the layers
// model.fit_generator trains: ssd300_training.ipynb:171-173), gfx950, bf16 NHWC operands, float32 accumulation and output.
//
//     dW[co][kh][kw][ci] = sum_{b,h,w} dY[b,h,w,co] * X[b, h + kh - 1, w + kw - 1, ci]
//
// A GEMM whose K dimension is the PIXELS -- and with NHWC tensors both operands have K as their slow index: an MFMA lane wants 8
// consecutive K values of one row (8 pixels of one channel), which sit a whole pixel row apart in memory.  How this kernel deals with it:
//   * positions, not pixels (the forward slab kernel's grid, ssdhip_convh.hip): image b, row h, column w sits at position
//     q = (b (H + 1) + h)(W + 1) + w; the dummy column of every row and the dummy row of every image are out-of-range buffer offsets
//     (the LDS-DMA writes zeros), so tap (kh, kw) is the SAME displacement (kh - 1)(W + 1) + (kw - 1) for every position, with no border
//     masks -- every out-of-image tap lands on zeros, and dY is zero on the dummies;
//   * transposed fragments: both operands are staged in LDS as [position][32 channels] images with 64-byte rows (what the lane-linear
//     LDS-DMA writes when lane L fetches 16 bytes of position L / 4), and ds_read_b64_tr_b16 hands each lane 4 consecutive positions of
//     ITS channel from four rows: a 16-lane group reads one 4 x 16 block, the two groups of a half-wave 256 contiguous bytes (all 64
//     banks once);
//   * every tap's operand is read straight from LDS (two transposed reads each): the row of a lane is free in a transposed read, so the
//     nine taps are nine row displacements of ONE per-lane base address per filter row, the kw and the second-read displacements ride in
//     the instruction's immediate offset.  (First version: one 12-position window per filter row and the kw = 1 / kw = 2 operands built
//     in registers -- 9 + 2 reads but 24 v_perm / v_mov and 12 address operations per 9 MFMAs; the profiling build's ablations,
//     profiles/r04g_wgrad_ablation.json, showed the kernel bound by instruction issue: everything but the MFMAs took as long as the MFMAs.)
//   * tile: 128 (or 64) output channels x 64 input channels x all nine taps per workgroup.  A wave owns 64 output channels x 32 input
//     channels x FIVE or FOUR of the nine taps (ten / eight 32 x 32 accumulators, 160 registers): 8 waves = 2 (output channels) x 2
//     (input channels) x 2 (tap groups), the two tap groups of a channel pair on the SAME SIMD (18 MFMAs per K-step and SIMD, as with
//     nine taps per wave) -- but a wave reads 2 + 5 fragments for 10 MFMAs instead of 1 + 9 for 9 (r04h: the first layout's LDS read
//     stream alone took longer than its MFMAs).  COS = 2 (Cout = 64, conv1_2): 1 x 2 x 2 tap groups x 2 K halves of every block;
//   * the positions are streamed in blocks of 64: dY ring of four blocks, X ring of RB blocks (the halo of (W + 2) positions either side
//     rides along: a block is loaded ONCE per workgroup), requests three blocks ahead with exact vmcnt counts, one barrier per block,
//     fragments of the next K-step read while the current one multiplies (also across the block boundary);
//   * split over positions: S workgroups per (co, ci) tile so that ~256 exist, each writes a float32 partial [co][tap][ci] tile; a
//     second kernel adds the partials in index order (fixed summation order: results are bit-reproducible run to run).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wg_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

struct WgParams {
    const bf16_t* x;             // [B, H, W, Cin]
    const bf16_t* dy;            // [B, H, W, Cout]
    float* part;                 // [slots][Cout][9][Cin] partial sums (slots = splits x (COS == 2 ? 2 : 1))
    int H, W, Cin, Cout;
    int Q;                       // padded positions B (H + 1)(W + 1)
    int n_blocks;                // blocks of 64 positions
    int n_ci_tiles, n_tiles;     // Cin / 64; (Cout / (32 COS)) n_ci_tiles
    int blocks_per_split;
    int HB;                      // halo blocks: ceil((W + 2) / 64)
    int step_h, step_w;          // 64 = step_h (W + 1) + step_w: how a position tracker advances per block
    int x_bytes, dy_bytes;
};

constexpr int WG_THREADS = 512;
[[maybe_unused]] constexpr unsigned WG_OOB = 0x80000000u;
constexpr int wg_lds_bytes(int cos, int rb) { return 2 * (64 * rb + 32) * 64 + 4 * 1024 + cos * 4 * 64 * 64; }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wg_bload(u32 voff, wg_i32x4 rsrc, u32 lds_dst) {     // one 1 KiB LDS-DMA piece (see ch_bload)
    u32 keep;
    lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ wg_i32x4 wg_rsrc(const void* base, int num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    wg_i32x4 r;
    r.x = (int)(u32)a;
    r.y = (int)((u32)(a >> 32) & 0xffffu);
    r.z = num_records;
    r.w = 0x00020000;
    return r;
}
typedef __attribute__((address_space(3))) unsigned char wg_lds_byte;
__device__ __forceinline__ wg_u32x2 wg_tr_read(wg_lds_byte* lds, u32 off) {   // ds_read_b64_tr_b16 at byte `off` of the workgroup's LDS array
    const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(lds + off));
    return __builtin_bit_cast(wg_u32x2, v);
}

// position tracker of one request piece: the lane's position advances by 64 per block
struct WgTrack {
    int p, w, h, bh;             // position, its column and row on the padded grid, b * H
};
#endif

// ABL (profiling build only, wrong results by construction -- each bit removes one cost): 1 no requests inside the block loop, 2 no
// fragment reads, 8 no MFMAs, 16 no wait / barrier per block.
// DIL (round 6: fc6, dilation 6): the same grid with DIL dummy columns per row and DIL dummy rows per image, tap (kh, kw) at displacement
// ((kh - 1) (W + DIL) + (kw - 1)) DIL -- 58 % of fc6's positions are real (19 x 19 of 25 x 25), which still beats gathering the taps.
template <int COS, int RB, int ABL = 0, int DIL = 1>
__global__ __launch_bounds__(WG_THREADS) void conv_wgrad_kernel(WgParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int R = 64 * RB;                           // rows of the X ring (+ 32 guard rows mirroring rows 0..31)
    constexpr int XSUB = (R + 32) * 64;                  // bytes of one 32-channel X sub-image
    constexpr int XR = 0, DUMP = 2 * XSUB, DYB = DUMP + 4 * 1024, DSUB = 4 * 64 * 64;
    constexpr int KS = COS;                              // K-steps (16 positions) a wave multiplies per block: 4, or 2 of the 4 (COS = 2)
    constexpr int NDY = COS / 2;                         // dY pieces a wave requests per block
    __shared__ __attribute__((aligned(1024))) unsigned char lds[wg_lds_bytes(COS, RB)];
    wg_lds_byte* const ldsp = (wg_lds_byte*)lds;
    const u32 lds0 = (u32)(uintptr_t)ldsp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;                             // 32-channel half of the 64 input channels
    const int wm = COS == 4 ? (wave >> 1) & 1 : 0;       // 64-channel half of the output channels
    const int kg = COS == 4 ? 0 : (wave >> 1) & 1;       // COS = 2: which two K-steps of a block
    const int tg = wave >> 2;                            // tap group: 0 = taps 0..4, 1 = taps 5..8 (waves w and w + 4 share a SIMD)
    const int H = p.H, W = p.W, W1 = W + DIL, H1 = H + DIL;

    // workgroup -> (tile, split): the tiles of one split (the same positions) share an XCD's L2 (ids are dealt round robin)
    const int id = (int)blockIdx.x, xcd = id & 7, slot_id = id >> 3;
    const int split = xcd + 8 * (slot_id / p.n_tiles), tile = slot_id % p.n_tiles;
    const int co0 = (tile / p.n_ci_tiles) * (32 * COS), ci0 = (tile % p.n_ci_tiles) * 64;
    const int sa = split * p.blocks_per_split;
    int nst = p.n_blocks - sa;
    nst = nst > p.blocks_per_split ? p.blocks_per_split : nst;          // blocks of this workgroup (may be <= 0: it then writes zeros)

    const wg_i32x4 rx = wg_rsrc(p.x, p.x_bytes), rdy = wg_rsrc(p.dy, p.dy_bytes);

    // ---- request pieces: lane L of a piece fetches 16 bytes of position (block 64 + rg 16 + L / 4), channels sub 32 + (L & 3) 8 ----------
    const int xsub = wave & 1, xrg = wave >> 1;          // this wave's X piece of a block
    const bool guard = xrg < 2;                           // ... and the mirror of its rows (0..31) when the block lands on ring slot 0
    int dsub[NDY], drg[NDY];
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
        dsub[u] = COS == 4 ? wave & 3 : wave & 1;
        drg[u] = COS == 4 ? (wave >> 2) * 2 + u : wave >> 1;
    }
    auto track_init = [&](WgTrack& t, const int pos) {
        const int hw1 = H1 * W1;
        int b = pos / hw1;
        if (pos < 0 && b * hw1 != pos) --b;               // floor: positions before the first image sit in "image -1" (never valid)
        const int r = pos - b * hw1;
        t.p = pos;
        t.h = r / W1;
        t.w = r - t.h * W1;
        t.bh = b * H;
    };
    const bool tiny = p.step_h + 1 >= H1;                 // maps of a handful of rows: a block spans more than one image
    auto track_step = [&](WgTrack& t) {                   // + 64 positions, branch-free (64 = step_h W1 + step_w)
        t.p += 64;
        t.w += p.step_w;
        t.h += p.step_h;
        const bool cw = t.w >= W1;
        t.w -= cw ? W1 : 0;
        t.h += cw ? 1 : 0;
        if (!tiny) {
            const bool ch = t.h >= H1;
            t.h -= ch ? H1 : 0;
            t.bh += ch ? H : 0;
        } else {
            while (t.h >= H1) { t.h -= H1; t.bh += H; }
        }
    };
    auto track_off = [&](const WgTrack& t, const int C, const int c) {       // byte offset of the lane's 16 bytes, or out of range
        const bool ok = (t.p >= 0) & (t.w < W) & (t.h < H) & (t.p < p.Q);
        return ((u32)(((t.bh + t.h) * W + t.w) * C + c) * 2u) | (ok ? 0u : WG_OOB);      // tensors are below 2 GB: bit 31 = out of range
    };
    WgTrack tx, tdy[NDY];
    track_init(tx, (sa - p.HB) * 64 + xrg * 16 + (lane >> 2));
#pragma unroll
    for (int u = 0; u < NDY; ++u) track_init(tdy[u], sa * 64 + drg[u] * 16 + (lane >> 2));
    const int xc = ci0 + xsub * 32 + (lane & 3) * 8;
    int xslot = 0, dslot = 0;                            // ring slots of the NEXT X block / dY block to request
    auto req_x = [&]() {
        const u32 off = track_off(tx, p.Cin, xc);
        wg_bload(off, rx, lds0 + XR + xsub * XSUB + (xslot * 64 + xrg * 16) * 64);
        if (guard) wg_bload(xslot == 0 ? off : WG_OOB, rx, lds0 + (xslot == 0 ? XR + xsub * XSUB + (R + xrg * 16) * 64 : DUMP + wave * 1024));
        track_step(tx);
        xslot = xslot + 1 == RB ? 0 : xslot + 1;
    };
    auto req_dy = [&]() {
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            wg_bload(track_off(tdy[u], p.Cout, co0 + dsub[u] * 32 + (lane & 3) * 8), rdy, lds0 + DYB + dsub[u] * DSUB + (dslot * 64 + drg[u] * 16) * 64);
            track_step(tdy[u]);
        }
        dslot = (dslot + 1) & 3;
    };

    // ---- fragment addressing ----------------------------------------------------------------------------------------------------------------
    const int i16 = lane & 15, g16 = (lane >> 4) & 1, khalf = lane >> 5;
    // K order of the operands: element j of lane (khalf) is position (j >> 2) 8 + khalf 4 + (j & 3) of the K-step -- NOT khalf 8 + j.  Both
    // operands use it (any permutation of K is fine as long as A and B agree), and it makes the 64 lanes of ONE transposed read cover 8
    // consecutive 64-byte rows = 512 contiguous bytes: with the natural order the two half-waves read rows r .. r + 3 and r + 8 .. r + 11,
    // 512 bytes apart = the same banks, and the reads ran at half rate (r04h ablation: 4.5 cycles per read instead of 2).
    const u32 lp = (u32)(khalf * 4 + (i16 >> 2));                             // the lane's row inside a 16-position K-step (first read; second + 8)
    const u32 cbyte = (u32)(g16 * 32 + (i16 & 3) * 8);
    const u32 a_lane = (u32)(DYB + wm * 2 * DSUB) + lp * 64 + cbyte;          // + cb DSUB + (slot 64 + kk 16) 64 (+ 512 for the second read)
    const u32 x_lane = (u32)(XR + wn * XSUB) + lp * 64 + cbyte;               // + ring row 64 + kw 64 (+ 512)

    wg_f32x16 acc[2][5];                                 // [32-channel block of the wave's 64 output channels][tap of the group]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[cb][t][v] = 0.f;

    wg_u32x4 fa[2][2];                                   // [set][cb]: dY^T fragments (32 channels x 16 positions) of this K-step and the next
    wg_u32x4 fx[5];                                      // X fragments of the group's taps: ONE set -- a tap's registers are refilled for
                                                         // the next K-step as soon as its two MFMAs have issued
    // ring row of position (64 (block) - W1 - 1) of the CURRENT block, i.e. of tap (0, 0) of the block's first position
    int ubase = 64 * p.HB - DIL * (W1 + 1);              // >= 0 because 64 HB >= DIL (W1 + 1)
    int rslot = 0;                                       // dY ring slot of the current block
    auto read_a = [&](auto setc, const int ds, const int kk) {
        constexpr int S = decltype(setc)::value;
        const u32 a0 = a_lane + (u32)((ds * 64 + kk * 16) * 64);
        if constexpr (ABL & 2) { asm volatile("" : "+v"(fa[S][0]), "+v"(fa[S][1]) : "v"(a0)); return; }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const wg_u32x2 lo = wg_tr_read(ldsp, a0 + cb * DSUB), hi = wg_tr_read(ldsp, a0 + cb * DSUB + 512);
            fa[S][cb] = wg_u32x4{lo.x, lo.y, hi.x, hi.y};
        }
    };
    // the address of filter row kh of K-step (ub, kk): the row's first ring row is wrapped in SCALAR arithmetic; the lanes' rows (+ lp <= 7,
    // + kw <= 2, + 8 for the second read) run at most 17 rows past it, into the guard rows that mirror the ring's first 32
    auto row_addr = [&](const int ub, const int kk, const int kh) {
        int rs = ub + kk * 16 + kh * DIL * W1;
        rs = rs >= R ? rs - R : rs;
        return x_lane + (u32)(rs * 64);
    };
    auto read_tap = [&](const int slot, const u32 a, const int kw) {
        if constexpr (ABL & 2) { asm volatile("" : "+v"(fx[slot]) : "v"(a)); return; }
        const wg_u32x2 lo = wg_tr_read(ldsp, a + kw * DIL * 64), hi = wg_tr_read(ldsp, a + kw * DIL * 64 + 512);
        fx[slot] = wg_u32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto mfma_tap = [&](auto setc, const int slot) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if constexpr (ABL & 8) {                      // (keeps the operands alive: a VALU operation instead of the MFMA)
                acc[cb][slot][0] = __uint_as_float(__float_as_uint(acc[cb][slot][0]) ^ (fx[slot].x & fx[slot].w & fa[S][cb].x & fa[S][cb].w & 1u));
            } else {
                acc[cb][slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, fa[S][cb]), __builtin_bit_cast(wg_bf16x8, fx[slot]),
                                                                        acc[cb][slot], 0, 0, 0);
            }
        }
    };
    // the fragments of K-step (ub, ds, kk) into A set S and fx (prologue)
    auto read_all = [&](auto tgc, auto setc, const int ub, const int ds, const int kk) {
        constexpr int TG = decltype(tgc)::value, T0 = TG ? 5 : 0, NT = TG ? 4 : 5;
        read_a(setc, ds, kk);
#pragma unroll
        for (int j = 0; j < NT; ++j) read_tap(j, row_addr(ub, kk, (T0 + j) / 3), (T0 + j) % 3);
    };
    // One K-step of tap group TG: per tap the two MFMAs of (A set S, fx[tap]), then the reads that refill fx[tap] for K-step (ub, ds, kk);
    // the next A fragments go to the other A set.
    auto kstep = [&](auto tgc, auto setc, auto nextc, const int ub, const int ds, const int kk) {
        constexpr int TG = decltype(tgc)::value, T0 = TG ? 5 : 0, NT = TG ? 4 : 5;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(setc, j);
            __builtin_amdgcn_sched_barrier(0);
            read_tap(j, row_addr(ub, kk, (T0 + j) / 3), (T0 + j) % 3);
            if (j == 1) read_a(nextc, ds, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    if (nst > 0) {
        // ---- prologue: X blocks sa - HB .. sa + HB + 2 and dY blocks sa .. sa + 2 ----------------------------------------------------------
        for (int j = 0; j < 2 * p.HB + 3; ++j) req_x();
        for (int j = 0; j < 3; ++j) req_dy();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto blocks = [&](auto tgc) {
            read_all(tgc, I0{}, ubase, 0, kg * 2);
            for (int t = 0; t < nst; ++t) {
                // requests of block t + 3 (X: t + HB + 3): their ring slots were last read before block t - 1's barrier
                if constexpr (!(ABL & 1)) {
                    req_x();
                    req_dy();
                }
                int ubn = ubase + 64;
                ubn = ubn >= R ? ubn - R : ubn;
                const int dsn = (rslot + 1) & 3;
                // The block's ONE barrier sits in front of its last K-step: that step prefetches block t + 1's first fragments, so block
                // t + 1's data (requested during block t - 2) must have landed and be visible -- while the requests of blocks t - 1 and t
                // may stay in flight: two blocks (~4 us) of latency cover instead of one.  After the barrier nobody reads block t's
                // LDS data any more (the last K-step's fragments were read before it), so the next block's requests may overwrite it.
                auto sync = [&]() {
                    if constexpr (!(ABL & 16)) {
                        if (guard) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((ABL & 1) ? 0 : 2 * (NDY + 2)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((ABL & 1) ? 0 : 2 * (NDY + 1)) : "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                };
                if constexpr (KS == 4) {
                    kstep(tgc, I0{}, I1{}, ubase, rslot, 1);
                    kstep(tgc, I1{}, I0{}, ubase, rslot, 2);
                    kstep(tgc, I0{}, I1{}, ubase, rslot, 3);
                    sync();
                    kstep(tgc, I1{}, I0{}, ubn, dsn, 0);
                } else {
                    kstep(tgc, I0{}, I1{}, ubase, rslot, kg * 2 + 1);
                    sync();
                    kstep(tgc, I1{}, I0{}, ubn, dsn, kg * 2);
                }
                ubase = ubn;
                rslot = dsn;
            }
        };
        if (tg == 0) blocks(I0{}); else blocks(I1{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // requests still in flight write into LDS: they must not outlive the workgroup

    // ---- partial tile: lane (r31 = input channel, khalf), register v: output channel 8 (v / 4) + 4 khalf + v % 4 ------------------------------
    const int r31 = lane & 31;
    const int slot = COS == 4 ? split : split * 2 + kg;
    float* out = p.part + (size_t)slot * p.Cout * 9 * p.Cin;
    const int t0 = tg ? 5 : 0, nt = tg ? 4 : 5;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j < nt) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int co = co0 + wm * 64 + cb * 32 + 8 * (v >> 2) + 4 * khalf + (v & 3);
                    out[((size_t)co * 9 + t0 + j) * p.Cin + ci0 + wn * 32 + r31] = acc[cb][j][v];
                }
            }
#endif
}

// out[i] = sum over the slots, in index order (float4 per thread) -- blocks 0 .. rb1 - 1.  The blocks behind them (round 5) finish the
// layer's BIAS gradient on the side: brows rows of per-workgroup channel sums [brows][4 bC4] (what the ReLU / pooling backward passes of
// csrc/ssdhip_train.hip leave behind) -> bout [4 bC4].  A block owns 32 channels (8 float4 columns) x 32 row lanes; a lane adds rows
// lane, lane + 32, ... in that order, the 32 lane sums are added in lane order: a fixed summation order, bit-reproducible, and no launch
// of its own (the framework's `partial.sum(0)`: 30 launches, 0.31 ms of the step, profiles/r05i_train_step_timeline.json).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* __restrict__ part, float4* __restrict__ out, int n4, int slots,
                                                           int rb1, const float4* __restrict__ bpart, float4* __restrict__ bout, int bC4,
                                                           int brows) {
    if ((int)blockIdx.x >= rb1) {
        __shared__ float4 red[32][8];
        const int col = (int)threadIdx.x & 7, rl = (int)threadIdx.x >> 3;
        const int c4 = ((int)blockIdx.x - rb1) * 8 + col;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < bC4) {
            for (int r0 = rl; r0 < brows; r0 += 32 * 4) {                  // four rows in flight per trip
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + 32 * u;
                    v[u] = r < brows ? bpart[(size_t)r * bC4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
        }
        red[rl][col] = s;
        __syncthreads();
        if (rl == 0 && c4 < bC4) {
            for (int j = 1; j < 32; ++j) { const float4 v = red[j][col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            bout[c4] = s;
        }
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += rb1 * blockDim.x) {
        float4 s = part[i];
        for (int k = 1; k < slots; k += 8) {                            // eight slots' loads in flight together; added in slot order all the same
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + u < slots ? part[(size_t)(k + u) * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < slots) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        out[i] = s;
    }
}

// ======================================================================================
// Weight gradient of the 1 x 1 layers (fc7, conv6_1 ... conv9_1; round 5): dW[co][ci] = sum over pixels of dY[p][co] X[p][ci], a GEMM
// whose K dimension -- the pixels -- is the slow index of BOTH NHWC operands.  A workgroup owns a 128 x 128 tile of dW over a range of
// pixels: per step 64 pixels of dY and of X are loaded as 16-byte rows (eight loads in flight per thread, requested a step ahead), laid
// into LDS TRANSPOSED ([channel][pixel]: a thread holds two neighbouring pixels of 8 channels and stores them side by side, so that an
// MFMA lane's eight consecutive K values are one 16-byte read), and four waves (2 x 2 halves of the tile, four 32 x 32 accumulators
// each) run 16 v_mfma_f32_32x32x16_bf16 per step.  Split over the pixels so that ~256-512 workgroups exist; the float32 partial tiles
// are added in slot order by wgrad_reduce_kernel (bit-reproducible), which also adds the layer's bias partials.
// ======================================================================================
struct Wg1Params {
    const bf16_t* x;             // [P][Cin]
    const bf16_t* dy;            // [P][Cout]
    float* part;                 // [slots][Cout][Cin]
    int P, Cin, Cout, n_ci_tiles, n_tiles, steps_per_split, n_steps;
};
constexpr int WG1_PITCH = 144;   // bytes per LDS row: 64 pixels + padding, 16-byte aligned

__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(Wg1Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char aT[128 * WG1_PITCH];   // dY tile [co][pixel]
    __shared__ __attribute__((aligned(16))) unsigned char bT[128 * WG1_PITCH];   // X tile  [ci][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x % p.n_tiles, split = blockIdx.x / p.n_tiles;
    const int co0 = (tile / p.n_ci_tiles) * 128, ci0 = (tile % p.n_ci_tiles) * 128;
    const int s0 = split * p.steps_per_split, s1 = min(p.n_steps, s0 + p.steps_per_split);
    const int cg = tid & 15, pp0 = tid >> 4;                 // 8 channels x the pixel pairs pp0 and pp0 + 16
    const int cohalf = wave & 1, cihalf = wave >> 1;
    wg_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    wg_u32x4 ra[4], rb[4];                                   // [unit u][pixel e]: unit u = pair pp0 + 16 u
    auto request = [&](int step) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long long px = (long long)step * 64 + 2 * (pp0 + 16 * u) + e;
                const bool ok = px < p.P;
                ra[2 * u + e] = ok ? *reinterpret_cast<const wg_u32x4*>(p.dy + px * p.Cout + co0 + cg * 8) : wg_u32x4{0u, 0u, 0u, 0u};
                rb[2 * u + e] = ok ? *reinterpret_cast<const wg_u32x4*>(p.x + px * p.Cin + ci0 + cg * 8) : wg_u32x4{0u, 0u, 0u, 0u};
            }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pp = pp0 + 16 * u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // channels cg 8 + 2 q and + 1: the pair's two pixels side by side in one word
                const u32 a0 = ra[2 * u][q], a1 = ra[2 * u + 1][q], b0 = rb[2 * u][q], b1 = rb[2 * u + 1][q];
                // the 16-byte chunk (8 pixels) of a row sits at chunk ^ (row / 8 & 7): rows 8 apart -- the sixteen channel groups of a
                // store instruction -- would otherwise all hit one bank (row pitch 36 words: 8 rows = 288 words = 0 mod 32)
                const int at = ((((pp >> 2) ^ (cg & 7)) << 2) | (pp & 3)) * 4;
                *reinterpret_cast<u32*>(aT + (cg * 8 + 2 * q) * WG1_PITCH + at) = (a0 & 0xffffu) | (a1 << 16);
                *reinterpret_cast<u32*>(aT + (cg * 8 + 2 * q + 1) * WG1_PITCH + at) = (a0 >> 16) | (a1 & 0xffff0000u);
                *reinterpret_cast<u32*>(bT + (cg * 8 + 2 * q) * WG1_PITCH + at) = (b0 & 0xffffu) | (b1 << 16);
                *reinterpret_cast<u32*>(bT + (cg * 8 + 2 * q + 1) * WG1_PITCH + at) = (b0 >> 16) | (b1 & 0xffff0000u);
            }
        }
    };
    if (s0 < s1) request(s0);
    for (int step = s0; step < s1; ++step) {
        stage();
        __syncthreads();
        if (step + 1 < s1) request(step + 1);                // in flight under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = 2 * ks + (lane >> 5);           // pixels 8 chunk .. + 7
            wg_bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra_ = cohalf * 64 + i * 32 + (lane & 31), rb_ = cihalf * 64 + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const wg_bf16x8*>(aT + ra_ * WG1_PITCH + ((chunk ^ ((ra_ >> 3) & 7)) << 4));
                fb[i] = *reinterpret_cast<const wg_bf16x8*>(bT + rb_ * WG1_PITCH + ((chunk ^ ((rb_ >> 3) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                      // the next step's stage() overwrites the tiles
    }
    // D row (co) = (v & 3) + 8 (v >> 2) + 4 (lane >> 5), column (ci) = lane & 31
    float* out = p.part + (size_t)split * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = co0 + cohalf * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                const int ci = ci0 + cihalf * 64 + j * 32 + (lane & 31);
                out[(size_t)co * p.Cin + ci] = acc[i][j][v];
            }
}

// ======================================================================================
// Weight gradient of the 3 x 3 layers the position-grid kernel above does not cover (round 6: fc6 with its dilation 6, the stride-2
// conv6_2 / conv7_2 behind their ZeroPadding2D, the 'valid' conv8_2 / conv9_2 -- models/keras_ssd300.py:294, 299-313):
//     dW[co][kh][kw][ci] = sum over output pixels (b, ho, wo) of dY[b, ho, wo, co] X[b, ho s + kh d - pad, wo s + kw d - pad, ci]
// as the pixel-contraction GEMM of the 1 x 1 form with the X rows GATHERED per tap: a workgroup owns 128 output channels x 128 input
// channels x the three taps of ONE filter row over a range of output pixels.  Per step of 64 pixels the dY rows come in once and the X
// rows of the three taps beside them (16-byte rows, sixteen loads in flight per thread, requested a step ahead; a tap that falls outside
// the image is a row of zeros), all four tiles go into LDS transposed as above, and four waves run 3 x 16 MFMAs -- the dY fragments are
// read once for the three taps.  Any stride / dilation / padding: the address arithmetic is per pixel and per tap, nothing else knows.
// Split over the pixels, partial tiles added in slot order by wgrad_reduce_kernel with the bias partials (bit-reproducible).
// ======================================================================================
struct WgTParams {
    const bf16_t* x;             // [B, H, W, Cin]
    const bf16_t* dy;            // [B, Ho, Wo, Cout]
    float* part;                 // [slots][Cout][9][Cin]
    int P;                       // output pixels B Ho Wo
    int H, W, Ho, Wo, Cin, Cout, stride, pad, dil;
    int n_ci_tiles, n_tiles;     // Cin / 128; (Cout / 128) n_ci_tiles 3
    int steps_per_split, n_steps;
};

__global__ __launch_bounds__(256) void conv_taps_wgrad_kernel(WgTParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char aT[128 * WG1_PITCH];       // dY tile [co][pixel]
    __shared__ __attribute__((aligned(16))) unsigned char bT[3][128 * WG1_PITCH];    // X tiles [kw][ci][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x % p.n_tiles, split = blockIdx.x / p.n_tiles;
    const int kh = tile % 3, t2 = tile / 3;                 // the three filter rows of a (co, ci) tile are neighbours: they share dY and most X rows in L2
    const int co0 = (t2 / p.n_ci_tiles) * 128, ci0 = (t2 % p.n_ci_tiles) * 128;
    const int s0 = split * p.steps_per_split, s1 = min(p.n_steps, s0 + p.steps_per_split);
    const int cg = tid & 15, pp0 = tid >> 4;
    const int cohalf = wave & 1, cihalf = wave >> 1;
    wg_f32x16 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][i][j][v] = 0.f;
    wg_u32x4 ra[4], rb[3][4];
    const u32 howo = (u32)(p.Ho * p.Wo);
    auto request = [&](int step) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const u32 px = (u32)step * 64u + 2u * (u32)(pp0 + 16 * u) + (u32)e;
                const bool ok = px < (u32)p.P;
                const u32 b = px / howo, r = px - b * howo, ho = r / (u32)p.Wo, wo = r - ho * (u32)p.Wo;
                ra[2 * u + e] = ok ? *reinterpret_cast<const wg_u32x4*>(p.dy + (size_t)px * p.Cout + co0 + cg * 8) : wg_u32x4{0u, 0u, 0u, 0u};
                const int hi = (int)ho * p.stride + kh * p.dil - p.pad;
                const bool okh = ok && hi >= 0 && hi < p.H;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int wi = (int)wo * p.stride + t * p.dil - p.pad;
                    const bool okx = okh && wi >= 0 && wi < p.W;
                    rb[t][2 * u + e] = okx ? *reinterpret_cast<const wg_u32x4*>(p.x + (((size_t)b * p.H + hi) * p.W + wi) * p.Cin + ci0 + cg * 8)
                                           : wg_u32x4{0u, 0u, 0u, 0u};
                }
            }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pp = pp0 + 16 * u;
            const int at = ((((pp >> 2) ^ (cg & 7)) << 2) | (pp & 3)) * 4;     // (the chunk swizzle of conv1x1_wgrad_kernel)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 a0 = ra[2 * u][q], a1 = ra[2 * u + 1][q];
                *reinterpret_cast<u32*>(aT + (cg * 8 + 2 * q) * WG1_PITCH + at) = (a0 & 0xffffu) | (a1 << 16);
                *reinterpret_cast<u32*>(aT + (cg * 8 + 2 * q + 1) * WG1_PITCH + at) = (a0 >> 16) | (a1 & 0xffff0000u);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const u32 b0 = rb[t][2 * u][q], b1 = rb[t][2 * u + 1][q];
                    *reinterpret_cast<u32*>(bT[t] + (cg * 8 + 2 * q) * WG1_PITCH + at) = (b0 & 0xffffu) | (b1 << 16);
                    *reinterpret_cast<u32*>(bT[t] + (cg * 8 + 2 * q + 1) * WG1_PITCH + at) = (b0 >> 16) | (b1 & 0xffff0000u);
                }
            }
        }
    };
    if (s0 < s1) request(s0);
    for (int step = s0; step < s1; ++step) {
        stage();
        __syncthreads();
        if (step + 1 < s1) request(step + 1);                // in flight under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = 2 * ks + (lane >> 5);
            wg_bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra_ = cohalf * 64 + i * 32 + (lane & 31);
                fa[i] = *reinterpret_cast<const wg_bf16x8*>(aT + ra_ * WG1_PITCH + ((chunk ^ ((ra_ >> 3) & 7)) << 4));
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                wg_bf16x8 fb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int rb_ = cihalf * 64 + j * 32 + (lane & 31);
                    fb[j] = *reinterpret_cast<const wg_bf16x8*>(bT[t] + rb_ * WG1_PITCH + ((chunk ^ ((rb_ >> 3) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[t][i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                      // the next step's stage() overwrites the tiles
    }
    float* out = p.part + (size_t)split * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int co = co0 + cohalf * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                    const int ci = ci0 + cihalf * 64 + j * 32 + (lane & 31);
                    out[((size_t)co * 9 + kh * 3 + t) * p.Cin + ci] = acc[t][i][j][v];
                }
}

struct WgPlan {
    int cos, splits, slots, n_tiles, n_ci_tiles, n_blocks, blocks_per_split, HB;
    long long Q;
    size_t ws_bytes;
};

static bool wg_plan(int B, int H, int W, int Cin, int Cout, WgPlan& pl, int dil = 1) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 64) || (Cout % 64)) return false;
    if (dil != 1 && !(dil == 6 && Cout % 128 == 0)) return false;            // the dilated form is instantiated for fc6 only
    pl.cos = (Cout % 128) ? 2 : 4;
    pl.HB = (dil * (W + dil + 1) + 63) / 64;                                  // the taps reach dil rows + dil columns either way
    if (pl.HB > (pl.cos == 4 ? 3 : 5)) return false;                          // ring blocks: 2 HB + 4 <= 10 | 14
    pl.Q = (long long)B * (H + dil) * (W + dil);
    if (pl.Q > 0x3fffff00LL) return false;
    if ((long long)B * H * W * Cin * 2 >= 0x7ffff000LL || (long long)B * H * W * Cout * 2 >= 0x7ffff000LL) return false;   // 31-bit byte offsets
    pl.n_blocks = (int)((pl.Q + 63) / 64);
    pl.n_ci_tiles = Cin / 64;
    pl.n_tiles = (Cout / (32 * pl.cos)) * pl.n_ci_tiles;
    // ~256 workgroups (one per CU; the LDS rings take a whole CU), a multiple of 8 splits (the id -> XCD map), at least 12 blocks each
    int s = (256 / pl.n_tiles) / 8 * 8;
    if (s < 8) s = 8;
    while (s > 8 && (pl.n_blocks + s - 1) / s < 12) s -= 8;
    pl.splits = s;
    pl.blocks_per_split = (pl.n_blocks + s - 1) / s;
    pl.slots = pl.cos == 4 ? s : 2 * s;
    pl.ws_bytes = (size_t)pl.slots * Cout * 9 * Cin * sizeof(float);
    return true;
}

}  // namespace ssdhip

using namespace ssdhip;

// Bytes of scratch ssdhip_conv3x3_wgrad_nhwc_bf16 needs for this geometry (0: geometry not supported).
extern "C" size_t ssdhip_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    WgPlan pl;
    return wg_plan(B, H, W, Cin, Cout, pl) ? pl.ws_bytes : 0;
}

// dw[co][kh][kw][ci] (float32, [Cout, 3, 3, Cin] = the channels_last layout of a [Cout, Cin, 3, 3] filter gradient) of the 3x3 'same'
// stride-1 dilation-1 convolution y = conv(x, w): x [B, H, W, Cin] bf16, dy [B, H, W, Cout] bf16 (the gradient w.r.t. the convolution's
// output, ReLU mask applied by the caller).  Cin % 64 == 0; Cout % 128 == 0 with W <= 190, or Cout % 64 == 0 with W <= 318.
// Bit-reproducible run to run (fixed summation order).  SSDHIP_E_BADARG for other geometries (callers fall back to the framework).
// ... and the layer's bias gradient with it: bias_partial [bias_rows][Cout] float32 per-workgroup channel sums of dy (as
// ssdhip_relu_bwd_bias_nhwc_bf16 / ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16 / ssdhip_channel_sums_nhwc_bf16 write them) -> db [Cout]
// float32, rows added in a fixed order by extra workgroups of the reduction launch.  bias_partial == NULL: weight gradient only.
static int wg_grid_launch(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows, float* db, int B, int H, int W,
                          int Cin, int Cout, int dil, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    WgPlan pl;
    if (!x || !dy || !dw || !wg_plan(B, H, W, Cin, Cout, pl, dil)) return SSDHIP_E_BADARG;
    if (bias_partial && (!db || bias_rows <= 0 || (((uintptr_t)bias_partial | (uintptr_t)db) & 15))) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) return SSDHIP_E_BADARG;
    if (!ws || ws_bytes < pl.ws_bytes || ((uintptr_t)ws & 15)) return SSDHIP_E_WORKSPACE;
    WgParams p;
    p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.part = static_cast<float*>(ws);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.Q = (int)pl.Q; p.n_blocks = pl.n_blocks; p.n_ci_tiles = pl.n_ci_tiles; p.n_tiles = pl.n_tiles;
    p.blocks_per_split = pl.blocks_per_split; p.HB = pl.HB;
    p.step_h = 64 / (W + dil); p.step_w = 64 % (W + dil);
    p.x_bytes = (int)((long long)B * H * W * Cin * 2); p.dy_bytes = (int)((long long)B * H * W * Cout * 2);
    const dim3 grid(pl.splits * pl.n_tiles), block(WG_THREADS);
    int abl = 0;
#if defined(SSDHIP_PROFILE)
    if (const char* e = getenv("SSDHIP_WGRAD_ABL")) abl = atoi(e);             // tools/ablate_wgrad.py
    if (pl.cos == 4 && abl) {
        switch (abl) {
            case 1: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 1>), grid, block, 0, stream, p); break;
            case 2: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 2>), grid, block, 0, stream, p); break;
            case 3: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 3>), grid, block, 0, stream, p); break;
            case 8: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 8>), grid, block, 0, stream, p); break;
            case 16: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 16>), grid, block, 0, stream, p); break;
            case 23: hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 23>), grid, block, 0, stream, p); break;
            default: abl = 0; break;
        }
    }
#endif
    if (abl) {}
    else if (dil == 6) hipLaunchKernelGGL((conv_wgrad_kernel<4, 10, 0, 6>), grid, block, 0, stream, p);
    else if (pl.cos == 4) hipLaunchKernelGGL((conv_wgrad_kernel<4, 10>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<2, 14>), grid, block, 0, stream, p);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const int n4 = Cout * 9 * Cin / 4;
    int rb = (n4 + 255) / 256;
    if (rb > 2048) rb = 2048;
    const int bC4 = bias_partial ? Cout / 4 : 0, rb2 = (bC4 + 7) / 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb + rb2), dim3(256), 0, stream, reinterpret_cast<const float4*>(ws), reinterpret_cast<float4*>(dw),
                       n4, pl.slots, rb, reinterpret_cast<const float4*>(bias_partial), reinterpret_cast<float4*>(db), bC4, bias_rows);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_conv3x3_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows,
                                                   float* db, int B, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                                                   void* stream) {
    return wg_grid_launch(x, dy, dw, bias_partial, bias_rows, db, B, H, W, Cin, Cout, 1, ws, ws_bytes, stream);
}

extern "C" int ssdhip_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, void* ws,
                                              size_t ws_bytes, void* stream) {
    return ssdhip_conv3x3_wgrad_bias_nhwc_bf16(x, dy, dw, nullptr, 0, nullptr, B, H, W, Cin, Cout, ws, ws_bytes, stream);
}

// Weight gradient (and, with bias_partial, the bias gradient) of a 1 x 1 stride-1 convolution (fc7, conv6_1 ... conv9_1 of
// models/keras_ssd300.py:296-313): dw [Cout][Cin] float32 from x [P][Cin], dy [P][Cout] bf16 (P = B H W pixels).  Cin % 128 == 0,
// Cout % 128 == 0.  Bit-reproducible (fixed summation order).  bias_partial / bias_rows / db as ssdhip_conv3x3_wgrad_bias_nhwc_bf16.
static bool wg1_plan(long long P, int Cin, int Cout, int& splits, int& steps_per_split, int& n_steps) {
    if (P <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 128) || (Cout % 128)) return false;
    if (P * Cin * 2 >= 0x7ffff000LL || P * Cout * 2 >= 0x7ffff000LL) return false;
    n_steps = (int)((P + 63) / 64);
    const int tiles = (Cout / 128) * (Cin / 128);
    int s = 512 / tiles;
    if (s < 1) s = 1;
    if (s > n_steps) s = n_steps;
    while (s > 1 && (n_steps + s - 1) / s < 4) --s;                           // at least four steps per workgroup
    steps_per_split = (n_steps + s - 1) / s;
    splits = (n_steps + steps_per_split - 1) / steps_per_split;
    return true;
}

extern "C" size_t ssdhip_conv1x1_wgrad_workspace_bytes(long long n_pixels, int Cin, int Cout) {
    int s, sp, ns;
    return wg1_plan(n_pixels, Cin, Cout, s, sp, ns) ? (size_t)s * Cout * Cin * sizeof(float) : 0;
}

extern "C" int ssdhip_conv1x1_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows, float* db,
                                                   long long n_pixels, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int splits, sps, n_steps;
    if (!x || !dy || !dw || !wg1_plan(n_pixels, Cin, Cout, splits, sps, n_steps)) return SSDHIP_E_BADARG;
    if (bias_partial && (!db || bias_rows <= 0 || (((uintptr_t)bias_partial | (uintptr_t)db) & 15))) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) return SSDHIP_E_BADARG;
    if (!ws || ws_bytes < (size_t)splits * Cout * Cin * sizeof(float) || ((uintptr_t)ws & 15)) return SSDHIP_E_WORKSPACE;
    Wg1Params p;
    p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.part = static_cast<float*>(ws);
    p.P = (int)n_pixels; p.Cin = Cin; p.Cout = Cout;
    p.n_ci_tiles = Cin / 128; p.n_tiles = (Cout / 128) * p.n_ci_tiles; p.steps_per_split = sps; p.n_steps = n_steps;
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(splits * p.n_tiles), dim3(256), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const int n4 = Cout * Cin / 4;
    int rb = (n4 + 255) / 256;
    if (rb > 2048) rb = 2048;
    const int bC4 = bias_partial ? Cout / 4 : 0, rb2 = (bC4 + 7) / 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb + rb2), dim3(256), 0, stream, reinterpret_cast<const float4*>(ws), reinterpret_cast<float4*>(dw),
                       n4, splits, rb, reinterpret_cast<const float4*>(bias_partial), reinterpret_cast<float4*>(db), bC4, bias_rows);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Weight gradient (and, with bias_partial, the bias gradient) of a 3 x 3 convolution of ANY stride / padding / dilation (fc6: dilation 6;
// conv6_2, conv7_2: stride 2 behind ZeroPadding2D; conv8_2, conv9_2: 'valid' -- models/keras_ssd300.py:294, 299-313):
// dw [Cout][3][3][Cin] float32 (the channels_last layout of a [Cout, Cin, 3, 3] gradient) from x [B, H, W, Cin] and dy
// [B, Ho, Wo, Cout] bf16, Ho = (H + 2 pad - 2 dilation - 1) / stride + 1 (likewise Wo; anything else is SSDHIP_E_BADARG).
// Cin % 128 == 0, Cout % 128 == 0.  Bit-reproducible (fixed summation order).  bias_partial / bias_rows / db as
// ssdhip_conv3x3_wgrad_bias_nhwc_bf16.
static bool wgt_plan(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int pad, int dil, int& splits, int& steps_per_split,
                     int& n_steps) {
    if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 128) || (Cout % 128)) return false;
    if (stride < 1 || pad < 0 || dil < 1) return false;
    if (H + 2 * pad - 2 * dil - 1 < 0 || W + 2 * pad - 2 * dil - 1 < 0) return false;
    if (Ho != (H + 2 * pad - 2 * dil - 1) / stride + 1 || Wo != (W + 2 * pad - 2 * dil - 1) / stride + 1) return false;
    const long long P = (long long)B * Ho * Wo;
    if ((long long)B * H * W * Cin * 2 >= 0x7ffff000LL || P * Cout * 2 >= 0x7ffff000LL || P >= 0x7fffff00LL) return false;
    n_steps = (int)((P + 63) / 64);
    const int tiles = (Cout / 128) * (Cin / 128) * 3;
    int s = 512 / tiles;
    if (s < 1) s = 1;
    if (s > n_steps) s = n_steps;
    while (s > 1 && (n_steps + s - 1) / s < 4) --s;                           // at least four steps per workgroup
    steps_per_split = (n_steps + s - 1) / s;
    splits = (n_steps + steps_per_split - 1) / steps_per_split;
    return true;
}

// (a dilated 'same' layer the position-grid kernel is instantiated for -- fc6 -- goes there: 175 us against 250 us with gathered taps)
static bool wgt_on_grid(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int pad, int dil, WgPlan& pl) {
    return stride == 1 && dil > 1 && pad == dil && Ho == H && Wo == W && wg_plan(B, H, W, Cin, Cout, pl, dil);
}

extern "C" size_t ssdhip_conv3x3_taps_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int padding,
                                                            int dilation) {
    int s, sp, ns;
    if (!wgt_plan(B, H, W, Cin, Ho, Wo, Cout, stride, padding, dilation, s, sp, ns)) return 0;
    WgPlan pl;
    const size_t gather = (size_t)s * Cout * 9 * Cin * sizeof(float);
    if (wgt_on_grid(B, H, W, Cin, Ho, Wo, Cout, stride, padding, dilation, pl)) return pl.ws_bytes > gather ? pl.ws_bytes : gather;
    return gather;
}

extern "C" int ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16(const void* x, const void* dy, float* dw, const float* bias_partial, int bias_rows,
                                                        float* db, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride,
                                                        int padding, int dilation, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int splits, sps, n_steps;
    if (!x || !dy || !dw || !wgt_plan(B, H, W, Cin, Ho, Wo, Cout, stride, padding, dilation, splits, sps, n_steps)) return SSDHIP_E_BADARG;
    WgPlan pl;
    if (wgt_on_grid(B, H, W, Cin, Ho, Wo, Cout, stride, padding, dilation, pl) && !getenv("SSDHIP_WGRAD_GATHER_ONLY"))
        return wg_grid_launch(x, dy, dw, bias_partial, bias_rows, db, B, H, W, Cin, Cout, dilation, ws, ws_bytes, stream_);
    if (bias_partial && (!db || bias_rows <= 0 || (((uintptr_t)bias_partial | (uintptr_t)db) & 15))) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) return SSDHIP_E_BADARG;
    if (!ws || ws_bytes < (size_t)splits * Cout * 9 * Cin * sizeof(float) || ((uintptr_t)ws & 15)) return SSDHIP_E_WORKSPACE;
    WgTParams p;
    p.x = static_cast<const bf16_t*>(x); p.dy = static_cast<const bf16_t*>(dy); p.part = static_cast<float*>(ws);
    p.P = B * Ho * Wo; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Cin = Cin; p.Cout = Cout; p.stride = stride; p.pad = padding; p.dil = dilation;
    p.n_ci_tiles = Cin / 128; p.n_tiles = (Cout / 128) * p.n_ci_tiles * 3; p.steps_per_split = sps; p.n_steps = n_steps;
    hipLaunchKernelGGL(conv_taps_wgrad_kernel, dim3(splits * p.n_tiles), dim3(256), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const int n4 = Cout * 9 * Cin / 4;
    int rb = (n4 + 255) / 256;
    if (rb > 2048) rb = 2048;
    const int bC4 = bias_partial ? Cout / 4 : 0, rb2 = (bC4 + 7) / 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb + rb2), dim3(256), 0, stream, reinterpret_cast<const float4*>(ws), reinterpret_cast<float4*>(dw),
                       n4, splits, rb, reinterpret_cast<const float4*>(bias_partial), reinterpret_cast<float4*>(db), bC4, bias_rows);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// out[c] = sum over the rows of partial [rows][C] float32, rows added in a fixed order (the bias half of wgrad_reduce_kernel on its own):
// the per-workgroup partial sums of conv1_1_bwd_kernel and l2norm_bwd_kernel.  Round 6: these were `partial.sum(0)` in the framework,
// whose reduction zeroes its semaphores with a MEMSET node -- the one thing that kept the training step from replaying as a HIP graph
// under the runtime's defaults (profiles/r06zd_train_graph_conv1_1_gradient.txt).  C % 4 == 0.
extern "C" int ssdhip_row_sums_f32(const float* partial, int rows, int C, float* out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!partial || !out || rows <= 0 || C <= 0 || (C % 4) || (((uintptr_t)partial | (uintptr_t)out) & 15)) return SSDHIP_E_BADARG;
    const int bC4 = C / 4, rb2 = (bC4 + 7) / 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb2), dim3(256), 0, stream, static_cast<const float4*>(nullptr), static_cast<float4*>(nullptr), 0, 0, 0,
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(out), bC4, rows);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
