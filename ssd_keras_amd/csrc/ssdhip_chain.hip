// ssdhip_chain.hip -- a CHAIN of small convolutions (+ bias + ReLU) in ONE launch, one workgroup per image, every intermediate map in LDS:
// the tail of the SSD extra layers, conv7_1 -> conv7_2 -> conv8_1 -> conv8_2 -> conv9_1 -> conv9_2 (models/keras_ssd300.py:304-313:
// 1x1 reduce + 3x3 stride-2 / 'valid' pairs on maps of 10 x 10 pixels and below), gfx950, bf16 NHWC, float32 accumulation.
//
// Why.  At batch 32 these six layers are 25 / 9 / 1 pixels per image: as separate launches each is a handful of tiles walking a K loop
// of 16-72 steps -- pure latency (split-K + reduce: twelve launches, ~85 us of a 2.3 ms step, the matrix pipe busy 6 % of it).  But a
// whole image's chain fits one CU: the 10 x 10 x 512 input is 102 KB, every later map is below 26 KB.  So one workgroup per image loads
// the input map once, runs the six layers back to back out of LDS (ping-pong buffers, a barrier between layers) and writes only the three
// maps the predictor heads read.  What remains is streaming 2 MB of filters from L2 per workgroup:
//   * filters are PRE-PACKED in MFMA fragment order (ssdhip_conv_chain_pack_weight, once per set of weights): fragment (32-channel
//     block, tap, 16-channel block) is 1 KiB contiguous, lane L's 16 bytes at offset 16 L -- a wave reads it with ONE fully coalesced
//     global_load_dwordx4 straight into the MFMA operand registers, no LDS round trip;
//   * every wave keeps eight fragments in flight (a register ring) -- the K loops are latency chains, this is what hides L2;
//   * the pixel operand comes from the LDS map by per-lane row addresses (tap displacement, stride, zero row for padding): im2col on
//     the fly, rows padded by 16 bytes so that 32 pixels do not hit one bank group.
// GEMM view as everywhere in libssdhip: MFMA 'A' = filters (row = output channel), 'B' = pixels; K order = taps outer, channels inner.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 cc_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cc_f32x2 __attribute__((ext_vector_type(2)));

constexpr int CC_MAX_LAYERS = 8;
constexpr int CC_THREADS = 512;
constexpr int CC_LDS = 156 * 1024;
[[maybe_unused]] constexpr int CC_PF = 8;                                 // K-steps of a block (one tap x 128 channels: the pixel operands' unit)
// Filter fragments a wave keeps in flight (the kernel's CC_RING: 8, 16 or 32; 1 KiB each).  Round 6, fourth session, measured and NOT
// adopted (profiles/r06zz5_chain_ring_depth_negative.txt): if the K loops were chains of memory round trips, twice the fragments in
// flight would halve them -- 16 is 7 % SLOWER than 8 back to back (45.3 against 42.1 us at batch 32; 181 against 149 VGPRs) and equal
// inside the step, 32 spills 84 bytes per lane (74.7 us).  The loads are not what a step waits for.  SSDHIP_CHAIN_RING=8|16|32 selects at
// launch (the forms stay instantiated: the test checks that the depth changes no bit).
constexpr int CC_RING_DEFAULT = 8;

struct ChainLayerDev {
    const uint4* wp;             // packed filters: [Cout / 32][k k][Cin / 16][64 lanes] x 16 bytes
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [B, Hout, Wout, Cout] or null (the map stays in LDS only)
    int k, stride, pad, Cin, Cout, relu;
    int Hin, Win, Hout, Wout;
    int in_off, out_off;         // LDS byte offsets of the input / output map ([pixel][C] rows of C 2 + 16 bytes)
};
struct ChainParams {
    const bf16_t* x;             // [B, H, W, C0]
    int n_layers, zero_off;      // zero_off: LDS offset of a row of zeros (padding taps)
    ChainLayerDev L[CC_MAX_LAYERS];
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32 cc_pack2(float a, float b) {
    const cc_f32x2 v = {a, b};
    return __builtin_bit_cast(u32, __builtin_convertvector(v, cc_bf16x2));
}
#endif

template <int CC_RING>
__global__ __launch_bounds__(CC_THREADS) void conv_chain_kernel(ChainParams p) {
    static_assert(CC_RING == 8 || CC_RING == 16 || CC_RING == 32, "ring depth: one, two or four blocks (the blocks alternate between the halves of `bq`)");
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned char lds[CC_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, khalf = lane >> 5;
    const int b = (int)blockIdx.x;

    // ---- the image's input map -> LDS (16-byte chunks, coalesced), the zero row ------------------------------------------------------------
    {
        const ChainLayerDev& l0 = p.L[0];
        const int cpr = l0.Cin / 8, n = l0.Hin * l0.Win * cpr, stride = l0.Cin * 2 + 16;
        const uint4* src = reinterpret_cast<const uint4*>(p.x + (size_t)b * l0.Hin * l0.Win * l0.Cin);
        for (int i = tid; i < n; i += CC_THREADS) {
            const int pix = i / cpr, c = i - pix * cpr;
            *reinterpret_cast<uint4*>(lds + l0.in_off + pix * stride + c * 16) = src[i];
        }
        for (int i = tid; i < 66; i += CC_THREADS) *reinterpret_cast<uint4*>(lds + p.zero_off + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();

    for (int li = 0; li < p.n_layers; ++li) {
        const ChainLayerDev& l = p.L[li];
        const int c16n = l.Cin >> 4;                     // 16-channel blocks per tap (a multiple of 8)
        const int taps = l.k * l.k, kt = taps * c16n;    // K-steps of 16
        const int bpt = c16n >> 3;                       // blocks of eight K-steps per tap
        const int nblk = l.Cout >> 5, npix = l.Hout * l.Wout, mblk = (npix + 31) >> 5;
        const int in_stride = l.Cin * 2 + 16, out_stride = l.Cout * 2 + 16;
        for (int t = wave; t < nblk * mblk; t += CC_THREADS / 64) {      // wave-uniform: a tile = 32 output channels x 32 pixels
            const int nb = t % nblk, mb = t / nblk;
            const int pix = mb * 32 + r31;
            const bool live = pix < npix;
            const int ho = live ? pix / l.Wout : 0, wo = live ? pix - ho * l.Wout : 0;
            const int hi0 = ho * l.stride - l.pad, wi0 = wo * l.stride - l.pad;
            const uint4* wsrc = l.wp + (size_t)nb * kt * 64 + lane;
            cc_f32x16 acc, acc1;                          // even / odd K-steps: two dependency chains through the matrix pipe instead of one
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[v] = 0.f; acc1[v] = 0.f; }
            uint4 ring[CC_RING];
#pragma unroll
            for (int j = 0; j < CC_RING; ++j) ring[j] = wsrc[(size_t)(j < kt ? j : kt - 1) * 64];
            // the pixel operands of a block of eight K-steps (one tap, 128 channels) are read from LDS a whole block AHEAD, into the
            // other half of `bq`: read right before their MFMA, every step waited out an LDS round trip (r04o: 417 cycles per step)
            uint4 bq[2][CC_PF];
            auto brow_of = [&](const int kb) {
                const int tap = (kb >> 3) / bpt, cb = (kb >> 3) - tap * bpt;
                const int kh = tap / l.k, kw = tap - kh * l.k;
                const int hi = hi0 + kh, wi = wi0 + kw;
                const bool ok = live & ((unsigned)hi < (unsigned)l.Hin) & ((unsigned)wi < (unsigned)l.Win) & (kb < kt);
                // a padding tap (or a lane without a pixel) reads the row of zeros, eight times the same 16 bytes
                return ok ? l.in_off + (hi * l.Win + wi) * in_stride + cb * 256 + khalf * 16 : -(p.zero_off + khalf * 16);
            };
            auto read_block = [&](auto hc, const int kb) {
                constexpr int HB = decltype(hc)::value;
                const int a = brow_of(kb);
                const unsigned char* brow = lds + (a < 0 ? -a : a);
                const int step = a < 0 ? 0 : 32;
#pragma unroll
                for (int j = 0; j < CC_PF; ++j) bq[HB][j] = *reinterpret_cast<const uint4*>(brow + j * step);
            };
            auto mul_block = [&](auto hc, auto sc, const int kb) {
                constexpr int HB = decltype(hc)::value, SB = decltype(sc)::value;     // half of `bq`; first ring slot of the block
#pragma unroll
                for (int j = 0; j < CC_PF; ++j) {
                    const uint4 a = ring[SB + j];
                    // (unconditional, the index clamped: a conditional load makes the compiler wait for ALL loads in flight at every
                    // step -- vmcnt(0) -- and the ring hides nothing)
                    const int kn = kb + CC_RING + j;
                    ring[SB + j] = wsrc[(size_t)(kn < kt ? kn : kt - 1) * 64];
                    __builtin_amdgcn_sched_barrier(0);    // the refill is issued HERE, eight steps ahead of its use, not batched at the block's end
                    if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cc_bf16x8, a), __builtin_bit_cast(cc_bf16x8, bq[HB][j]), acc1, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cc_bf16x8, a), __builtin_bit_cast(cc_bf16x8, bq[HB][j]), acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
            read_block(H0{}, 0);
            // blocks of eight K-steps walk the ring's CC_RING / 8 block slots in turn and the two halves of `bq` alternately: the loop body
            // is unrolled over the ring so that both are compile-time indices (kt / 8 need not be a multiple: a block past the end multiplies
            // nothing)
            for (int kb = 0; kb < kt; kb += CC_RING < 2 * CC_PF ? 2 * CC_PF : CC_RING) {
                read_block(H1{}, kb + CC_PF);
                mul_block(H0{}, std::integral_constant<int, 0>{}, kb);
                if (kb + CC_PF < kt) {
                    read_block(H0{}, kb + 2 * CC_PF);
                    mul_block(H1{}, std::integral_constant<int, CC_RING >= 16 ? 8 : 0>{}, kb + CC_PF);
                }
                if constexpr (CC_RING == 32) {
                    if (kb + 2 * CC_PF < kt) {
                        read_block(H1{}, kb + 3 * CC_PF);
                        mul_block(H0{}, std::integral_constant<int, 16>{}, kb + 2 * CC_PF);
                    }
                }
                if constexpr (CC_RING == 32) {
                    if (kb + 3 * CC_PF < kt) {
                        read_block(H0{}, kb + 4 * CC_PF);
                        mul_block(H1{}, std::integral_constant<int, 24>{}, kb + 3 * CC_PF);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] += acc1[v];
            // bias + ReLU + one rounding; the lane holds 16 channels (four runs of four) of ONE pixel
            if (live) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nb * 32 + 8 * g + 4 * khalf;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * g + e];
                        if (l.bias) v += __uint_as_float((u32)l.bias[c + e] << 16);
                        o[e] = l.relu ? (v <= 0.f ? 0.f : v) : v;
                    }
                    const uint2 pk = make_uint2(cc_pack2(o[0], o[1]), cc_pack2(o[2], o[3]));
                    *reinterpret_cast<uint2*>(lds + l.out_off + pix * out_stride + c * 2) = pk;
                    if (l.y) *reinterpret_cast<uint2*>(l.y + ((size_t)b * npix + pix) * l.Cout + c) = pk;
                }
            }
        }
        __syncthreads();
    }
#endif
}

// ======================================================================================
// The same chain at the reference's precision (round 6; models/precise.py): activations are float16 (hi, lo) PAIR maps, filters float16
// pairs of the float32 filters / oscale, three float16 MFMAs per K-step -- w_hi x_hi, w_hi x_lo, w_lo x_hi -- into three float32
// accumulators (three independent chains through the matrix pipe), the epilogue mul * sum + bias in float32, ReLU, re-split.  One
// workgroup per image as above.  Differences: the FIRST layer (a 1 x 1 layer: conv7_1) reads its pixels straight from global memory --
// conv6_2's 10 x 10 x 512 map is 205 KB as pairs, more than a CU's LDS -- and every later map (53 KB and below) lives in LDS as
// [pixel][hi C | lo C] rows; filter fragments stream as (hi, lo) twins, four K-steps in flight per wave (the register budget of 2
// waves per SIMD: 32 ring + 64 pixel-operand + 48 accumulator registers).  Replaces six launches of 22-102 us each on the
// reference-precision step's critical path.
// ======================================================================================
typedef _Float16 cc_f16x8 __attribute__((ext_vector_type(8)));
[[maybe_unused]] constexpr int CX_PF = 4;                                 // K-steps of a block (one tap x 64 channels)
// K-steps of filter (hi, lo) twins a wave keeps in flight (CX_RING: 4 or 8 -- one or two blocks); 8 measured 3 % slower than 4
// (104.4 against 101.2 us, same file): SSDHIP_CHAIN_X3_RING=4|8 selects at launch
constexpr int CX_RING_DEFAULT = 4;

struct ChainX3LayerDev {
    const uint4* wp;             // packed filters: [Cout / 32][k k][Cin / 16][2 = hi, lo][64 lanes] x 16 bytes
    const float* bias;           // [Cout] float32, already divided by the layer's output divisor, or null
    unsigned short* y;           // [B, Hout, Wout, 2 Cout] float16 pairs or null
    float mul;                   // oscale * s_in / s_out
    int k, stride, pad, Cin, Cout, relu;
    int Hin, Win, Hout, Wout;
    int in_off, out_off;         // LDS byte offsets of the input / output map (rows of 4 C + 16 bytes); layer 0 reads global memory
};
struct ChainX3Params {
    const unsigned short* x;     // [B, H, W, 2 C0] float16 pairs
    int n_layers, zero_off;
    ChainX3LayerDev L[CC_MAX_LAYERS];
};

template <int CX_RING>
__global__ __launch_bounds__(CC_THREADS) void conv_chain_x3_kernel(ChainX3Params p) {
    static_assert(CX_RING == 4 || CX_RING == 8, "ring depth: one or two blocks");
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) unsigned char lds[CC_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, khalf = lane >> 5;
    const int b = (int)blockIdx.x;
    for (int i = tid; i < 66; i += CC_THREADS) *reinterpret_cast<uint4*>(lds + p.zero_off + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    for (int li = 0; li < p.n_layers; ++li) {
        const ChainX3LayerDev& l = p.L[li];
        const int c16n = l.Cin >> 4;                     // 16-channel blocks per tap (a multiple of 4)
        const int taps = l.k * l.k, kt = taps * c16n;
        const int bpt = c16n >> 2;                       // blocks of four K-steps per tap
        const int nblk = l.Cout >> 5, npix = l.Hout * l.Wout, mblk = (npix + 31) >> 5;
        const int out_stride = l.Cout * 4 + 16;
        // (two instantiations of the tile loop: layer 0's pixel rows are GLOBAL memory, the others' LDS -- one generic pointer would make
        //  every read a flat load)
        auto tiles = [&](auto from_global) {
        constexpr bool GLOBAL = decltype(from_global)::value;
        const int in_stride = GLOBAL ? l.Cin * 4 : l.Cin * 4 + 16;
        const unsigned char* const gbase = reinterpret_cast<const unsigned char*>(p.x) + (size_t)b * l.Hin * l.Win * (l.Cin * 4);
        for (int t = wave; t < nblk * mblk; t += CC_THREADS / 64) {
            const int nb = t % nblk, mb = t / nblk;
            const int pix = mb * 32 + r31;
            const bool live = pix < npix;
            const int ho = live ? pix / l.Wout : 0, wo = live ? pix - ho * l.Wout : 0;
            const int hi0 = ho * l.stride - l.pad, wi0 = wo * l.stride - l.pad;
            const uint4* wsrc = l.wp + (size_t)nb * kt * 128 + lane;
            cc_f32x16 a_hh, a_hl, a_lh;
#pragma unroll
            for (int v = 0; v < 16; ++v) { a_hh[v] = 0.f; a_hl[v] = 0.f; a_lh[v] = 0.f; }
            uint4 rh[CX_RING], rl[CX_RING];
#pragma unroll
            for (int j = 0; j < CX_RING; ++j) {
                const size_t o = (size_t)(j < kt ? j : kt - 1) * 128;
                rh[j] = wsrc[o];
                rl[j] = wsrc[o + 64];
            }
            uint4 bh[2][CX_PF], bl[2][CX_PF];
            // the lane's pixel row of K-block kb (one tap, 64 channels), or the row of zeros (padding tap / lane without a pixel / behind the end)
            auto read_block = [&](auto hc, const int kb) {
                constexpr int HB = decltype(hc)::value;
                const int tap = (kb >> 2) / bpt, cb = (kb >> 2) - tap * bpt;
                const int kh = tap / l.k, kw = tap - kh * l.k;
                const int hi = hi0 + kh, wi = wi0 + kw;
                const bool ok = live & ((unsigned)hi < (unsigned)l.Hin) & ((unsigned)wi < (unsigned)l.Win) & (kb < kt);
                if constexpr (GLOBAL) {
                    // (a 1 x 1 layer without padding: only lanes without a pixel and reads behind the end are not `ok` -- they re-read the
                    //  image's first row, their products are never stored / the step multiplies the clamped last filter fragment into
                    //  nothing that is kept: kb >= kt only happens in the skipped half of the last double block)
                    const unsigned char* row = gbase + (size_t)(ok ? hi * l.Win + wi : 0) * in_stride + (kb < kt ? cb * 128 : 0) + khalf * 16;
#pragma unroll
                    for (int j = 0; j < CX_PF; ++j) {
                        bh[HB][j] = *reinterpret_cast<const uint4*>(row + j * 32);
                        bl[HB][j] = *reinterpret_cast<const uint4*>(row + l.Cin * 2 + j * 32);
                    }
                } else {
                    const unsigned char* row = lds + (ok ? l.in_off + (hi * l.Win + wi) * in_stride + cb * 128 + khalf * 16 : p.zero_off + khalf * 16);
                    const int step = ok ? 32 : 0, lo = ok ? l.Cin * 2 : 0;
#pragma unroll
                    for (int j = 0; j < CX_PF; ++j) {
                        bh[HB][j] = *reinterpret_cast<const uint4*>(row + j * step);
                        bl[HB][j] = *reinterpret_cast<const uint4*>(row + lo + j * step);
                    }
                }
            };
            auto mul_block = [&](auto hc, const int kb) {
                constexpr int HB = decltype(hc)::value, SB = CX_RING == 8 ? HB * CX_PF : 0;     // half of bh / bl; first ring slot of the block
#pragma unroll
                for (int j = 0; j < CX_PF; ++j) {
                    const uint4 wh = rh[SB + j], wl = rl[SB + j];
                    const int kn = kb + CX_RING + j;
                    const size_t o = (size_t)(kn < kt ? kn : kt - 1) * 128;
                    rh[SB + j] = wsrc[o];
                    rl[SB + j] = wsrc[o + 64];
                    __builtin_amdgcn_sched_barrier(0);
                    a_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cc_f16x8, wh), __builtin_bit_cast(cc_f16x8, bh[HB][j]), a_hh, 0, 0, 0);
                    a_hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cc_f16x8, wh), __builtin_bit_cast(cc_f16x8, bl[HB][j]), a_hl, 0, 0, 0);
                    a_lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cc_f16x8, wl), __builtin_bit_cast(cc_f16x8, bh[HB][j]), a_lh, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
            read_block(H0{}, 0);
            for (int kb = 0; kb < kt; kb += 2 * CX_PF) {
                read_block(H1{}, kb + CX_PF);
                mul_block(H0{}, kb);
                if (kb + CX_PF < kt) {
                    read_block(H0{}, kb + 2 * CX_PF);
                    mul_block(H1{}, kb + CX_PF);
                }
            }
            if (live) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nb * 32 + 8 * g + 4 * khalf;
                    unsigned short hs[4], ls[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = (a_hh[4 * g + e] + (a_hl[4 * g + e] + a_lh[4 * g + e])) * l.mul;
                        if (l.bias) v += l.bias[c + e];
                        v = l.relu ? (v > 0.f ? v : (v != v ? v : 0.f)) : v;
                        const _Float16 h = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)h);
                        hs[e] = __builtin_bit_cast(unsigned short, h);
                        ls[e] = __builtin_bit_cast(unsigned short, lo);
                    }
                    const uint2 ph = make_uint2((u32)hs[0] | ((u32)hs[1] << 16), (u32)hs[2] | ((u32)hs[3] << 16));
                    const uint2 pl = make_uint2((u32)ls[0] | ((u32)ls[1] << 16), (u32)ls[2] | ((u32)ls[3] << 16));
                    unsigned char* orow = lds + l.out_off + pix * out_stride;
                    *reinterpret_cast<uint2*>(orow + c * 2) = ph;
                    *reinterpret_cast<uint2*>(orow + l.Cout * 2 + c * 2) = pl;
                    if (l.y) {
                        unsigned short* yrow = l.y + ((size_t)b * npix + pix) * (2 * l.Cout);
                        *reinterpret_cast<uint2*>(yrow + c) = ph;
                        *reinterpret_cast<uint2*>(yrow + l.Cout + c) = pl;
                    }
                }
            }
        }
        };
        if (li == 0) tiles(std::true_type{}); else tiles(std::false_type{});
        __syncthreads();
    }
#endif
}

// x3 [Cout][taps][3 Cin] float16 = [w hi | w lo | w hi] (what x3_pack_weight builds) -> fragment twins:
// packed[(((nb T + tap) (Cin / 16) + c16) 2 + part) 64 + lane] (16 bytes) = part(w)[nb 32 + (lane & 31)][tap][c16 16 + (lane >> 5) 8 .. + 7]
__global__ __launch_bounds__(256) void chain_x3_pack_kernel(const uint4* __restrict__ w, uint4* __restrict__ packed, int taps, int Cin, int Cout) {
    const int c16n = Cin >> 4;
    const size_t n = (size_t)(Cout >> 5) * taps * c16n * 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), part = (int)((i >> 6) & 1);
        size_t r = i >> 7;
        const int c16 = (int)(r % c16n); r /= c16n;
        const int tap = (int)(r % taps);
        const int nb = (int)(r / taps);
        const int co = nb * 32 + (lane & 31);
        packed[i] = w[((size_t)co * taps + tap) * (3 * Cin >> 3) + (size_t)part * (Cin >> 3) + c16 * 2 + (lane >> 5)];
    }
}

// packed[((nb T + tap) (Cin / 16) + c16) 64 + lane] (16 bytes) = w[nb 32 + (lane & 31)][tap][c16 16 + (lane >> 5) 8 .. + 7]
__global__ __launch_bounds__(256) void chain_pack_kernel(const uint4* __restrict__ w, uint4* __restrict__ packed, int taps, int Cin, int Cout) {
    const int c16n = Cin >> 4;
    const size_t n = (size_t)(Cout >> 5) * taps * c16n * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        size_t r = i >> 6;
        const int c16 = (int)(r % c16n); r /= c16n;
        const int tap = (int)(r % taps);
        const int nb = (int)(r / taps);
        const int co = nb * 32 + (lane & 31);
        packed[i] = w[((size_t)co * taps + tap) * (Cin >> 3) + c16 * 2 + (lane >> 5)];
    }
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" size_t ssdhip_conv_chain_packed_bytes(int k, int Cin, int Cout) {
    if (k <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 128) || (Cout % 32)) return 0;
    return (size_t)Cout * k * k * Cin * 2;
}

// weight [Cout, k, k, Cin] bf16 -> the fragment order conv_chain_kernel streams (same byte count)
extern "C" int ssdhip_conv_chain_pack_weight(const void* weight, void* packed, int k, int Cin, int Cout, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!weight || !packed || !ssdhip_conv_chain_packed_bytes(k, Cin, Cout)) return SSDHIP_E_BADARG;
    if (((uintptr_t)weight | (uintptr_t)packed) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(chain_pack_kernel, dim3(256), dim3(256), 0, stream, static_cast<const uint4*>(weight), static_cast<uint4*>(packed), k * k, Cin,
                       Cout);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// n_layers (<= 8) convolutions applied one after the other to x [B, H, W, C0] (bf16 NHWC), layer i: k_i x k_i, stride_i, zero padding
// pad_i, bias, ReLU if relu_i; y_i [B, H_i, W_i, Cout_i] is written for the layers whose y_h[i] is not null (the others exist in LDS only).
// Arrays are HOST arrays.  packed_h[i] from ssdhip_conv_chain_pack_weight.  Cin_i % 128 == 0, Cout_i % 32 == 0, and the maps of one
// image must fit the CU's LDS (SSDHIP_E_BADARG otherwise; callers fall back to one launch per layer).
extern "C" int ssdhip_conv_chain_nhwc_bf16(const void* x, int B, int H, int W, int C0, int n_layers, const void* const* packed_h,
                                           const void* const* bias_h, void* const* y_h, const int* k_h, const int* stride_h, const int* pad_h,
                                           const int* cout_h, const int* relu_h, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || B <= 0 || H <= 0 || W <= 0 || n_layers < 1 || n_layers > CC_MAX_LAYERS || !packed_h || !y_h || !k_h || !stride_h || !pad_h || !cout_h)
        return SSDHIP_E_BADARG;
    ChainParams p;
    p.x = static_cast<const bf16_t*>(x);
    p.n_layers = n_layers;
    int h = H, w = W, c = C0;
    // ping-pong: even layers read buffer 0 and write buffer 1, odd layers the other way round; buffer sizes = the largest map placed there
    size_t need[2] = {0, 0};
    int hs[CC_MAX_LAYERS + 1], wsz[CC_MAX_LAYERS + 1], cs[CC_MAX_LAYERS + 1];
    hs[0] = h; wsz[0] = w; cs[0] = c;
    for (int i = 0; i < n_layers; ++i) {
        const int k = k_h[i], s = stride_h[i], pd = pad_h[i], co = cout_h[i];
        if (k < 1 || k > 7 || s < 1 || pd < 0 || !ssdhip_conv_chain_packed_bytes(k, c, co) || !packed_h[i]) return SSDHIP_E_BADARG;
        if (h + 2 * pd < k || w + 2 * pd < k) return SSDHIP_E_BADARG;
        if (((uintptr_t)packed_h[i] & 15) || (bias_h && bias_h[i] && ((uintptr_t)bias_h[i] & 1)) || (y_h[i] && ((uintptr_t)y_h[i] & 7))) return SSDHIP_E_BADARG;
        h = (h + 2 * pd - k) / s + 1; w = (w + 2 * pd - k) / s + 1; c = co;
        hs[i + 1] = h; wsz[i + 1] = w; cs[i + 1] = c;
    }
    for (int i = 0; i <= n_layers; ++i) {
        const size_t bytes = (size_t)hs[i] * wsz[i] * (cs[i] * 2 + 16);
        if (bytes > need[i & 1]) need[i & 1] = bytes;
    }
    const size_t off1 = (need[0] + 15) / 16 * 16, zoff = off1 + (need[1] + 15) / 16 * 16;
    if (zoff + 66 * 16 > (size_t)CC_LDS) return SSDHIP_E_BADARG;
    if (((uintptr_t)x & 15) || (C0 % 8)) return SSDHIP_E_BADARG;
    p.zero_off = (int)zoff;
    for (int i = 0; i < n_layers; ++i) {
        ChainLayerDev& l = p.L[i];
        l.wp = static_cast<const uint4*>(packed_h[i]);
        l.bias = bias_h ? static_cast<const bf16_t*>(bias_h[i]) : nullptr;
        l.y = static_cast<bf16_t*>(y_h[i]);
        l.k = k_h[i]; l.stride = stride_h[i]; l.pad = pad_h[i]; l.Cin = cs[i]; l.Cout = cs[i + 1]; l.relu = relu_h ? relu_h[i] : 1;
        l.Hin = hs[i]; l.Win = wsz[i]; l.Hout = hs[i + 1]; l.Wout = wsz[i + 1];
        l.in_off = (i & 1) ? (int)off1 : 0;
        l.out_off = (i & 1) ? 0 : (int)off1;
    }
    for (int i = n_layers; i < CC_MAX_LAYERS; ++i) p.L[i] = p.L[0];
    int ring = CC_RING_DEFAULT;
    if (const char* e = getenv("SSDHIP_CHAIN_RING")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32) ring = v; }
    if (ring == 8) hipLaunchKernelGGL(conv_chain_kernel<8>, dim3(B), dim3(CC_THREADS), 0, stream, p);
    else if (ring == 32) hipLaunchKernelGGL(conv_chain_kernel<32>, dim3(B), dim3(CC_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(conv_chain_kernel<16>, dim3(B), dim3(CC_THREADS), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// ---- the chain at the reference's precision (float16 pairs, three products; see conv_chain_x3_kernel) -------------------------------------
extern "C" size_t ssdhip_conv_chain_x3_packed_bytes(int k, int Cin, int Cout) {
    if (k <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 64) || (Cout % 32)) return 0;
    return (size_t)Cout * k * k * Cin * 4;
}

// weight_x3 [Cout, k, k, 3 Cin] float16 = [w hi | w lo | w hi] (ssdhip_conv2d_x3_nhwc_f16's filter layout) -> the fragment twins the chain streams
extern "C" int ssdhip_conv_chain_x3_pack_weight(const void* weight_x3, void* packed, int k, int Cin, int Cout, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!weight_x3 || !packed || !ssdhip_conv_chain_x3_packed_bytes(k, Cin, Cout)) return SSDHIP_E_BADARG;
    if (((uintptr_t)weight_x3 | (uintptr_t)packed) & 15) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(chain_x3_pack_kernel, dim3(256), dim3(256), 0, stream, static_cast<const uint4*>(weight_x3), static_cast<uint4*>(packed), k * k,
                       Cin, Cout);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// n_layers (<= 8) convolutions applied one after the other to the pair map x [B, H, W, 2 C0] float16 = [hi | lo]; layer i computes
// act(mul_i * (w_i * x) + bias_i) in float32 -- mul_i = oscale_i * (divisor of its input) / (divisor of its output), bias_i float32 already
// divided by the output divisor -- and re-splits; y_i [B, H_i, W_i, 2 Cout_i] pairs are written where y_h[i] is not null.  The first layer
// must be 1 x 1, stride 1, no padding (it reads x from global memory; the later maps must fit the CU's LDS as pairs).  Cin_i % 64 == 0,
// Cout_i % 32 == 0.  SSDHIP_E_BADARG: the chain does not fit (callers run the layers one by one).
extern "C" int ssdhip_conv_chain_x3_nhwc_f16(const void* x, int B, int H, int W, int C0, int n_layers, const void* const* packed_h,
                                             const float* const* bias_h, void* const* y_h, const int* k_h, const int* stride_h, const int* pad_h,
                                             const int* cout_h, const int* relu_h, const float* mul_h, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || B <= 0 || H <= 0 || W <= 0 || n_layers < 1 || n_layers > CC_MAX_LAYERS || !packed_h || !y_h || !k_h || !stride_h || !pad_h || !cout_h ||
        !mul_h)
        return SSDHIP_E_BADARG;
    if (k_h[0] != 1 || stride_h[0] != 1 || pad_h[0] != 0 || ((uintptr_t)x & 15)) return SSDHIP_E_BADARG;
    ChainX3Params p;
    p.x = static_cast<const unsigned short*>(x);
    p.n_layers = n_layers;
    int h = H, w = W, c = C0;
    size_t need[2] = {0, 0};
    int hs[CC_MAX_LAYERS + 1], wsz[CC_MAX_LAYERS + 1], cs[CC_MAX_LAYERS + 1];
    hs[0] = h; wsz[0] = w; cs[0] = c;
    for (int i = 0; i < n_layers; ++i) {
        const int k = k_h[i], s = stride_h[i], pd = pad_h[i], co = cout_h[i];
        if (k < 1 || k > 7 || s < 1 || pd < 0 || !ssdhip_conv_chain_x3_packed_bytes(k, c, co) || !packed_h[i] || !(mul_h[i] > 0.f)) return SSDHIP_E_BADARG;
        if (h + 2 * pd < k || w + 2 * pd < k) return SSDHIP_E_BADARG;
        if (((uintptr_t)packed_h[i] & 15) || (bias_h && bias_h[i] && ((uintptr_t)bias_h[i] & 3)) || (y_h[i] && ((uintptr_t)y_h[i] & 7))) return SSDHIP_E_BADARG;
        h = (h + 2 * pd - k) / s + 1; w = (w + 2 * pd - k) / s + 1; c = co;
        hs[i + 1] = h; wsz[i + 1] = w; cs[i + 1] = c;
    }
    for (int i = 1; i <= n_layers; ++i) {                 // map 0 stays in global memory
        const size_t bytes = (size_t)hs[i] * wsz[i] * (cs[i] * 4 + 16);
        if (bytes > need[i & 1]) need[i & 1] = bytes;
    }
    const size_t off1 = (need[0] + 15) / 16 * 16, zoff = off1 + (need[1] + 15) / 16 * 16;
    if (zoff + 66 * 16 > (size_t)CC_LDS) return SSDHIP_E_BADARG;
    p.zero_off = (int)zoff;
    for (int i = 0; i < n_layers; ++i) {
        ChainX3LayerDev& l = p.L[i];
        l.wp = static_cast<const uint4*>(packed_h[i]);
        l.bias = bias_h ? bias_h[i] : nullptr;
        l.y = static_cast<unsigned short*>(y_h[i]);
        l.mul = mul_h[i];
        l.k = k_h[i]; l.stride = stride_h[i]; l.pad = pad_h[i]; l.Cin = cs[i]; l.Cout = cs[i + 1]; l.relu = relu_h ? relu_h[i] : 1;
        l.Hin = hs[i]; l.Win = wsz[i]; l.Hout = hs[i + 1]; l.Wout = wsz[i + 1];
        l.in_off = (i & 1) ? (int)off1 : 0;
        l.out_off = (i & 1) ? 0 : (int)off1;
    }
    for (int i = n_layers; i < CC_MAX_LAYERS; ++i) p.L[i] = p.L[0];
    int ring = CX_RING_DEFAULT;
    if (const char* e = getenv("SSDHIP_CHAIN_X3_RING")) { const int v = atoi(e); if (v == 4 || v == 8) ring = v; }
    if (ring == 4) hipLaunchKernelGGL(conv_chain_x3_kernel<4>, dim3(B), dim3(CC_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(conv_chain_x3_kernel<8>, dim3(B), dim3(CC_THREADS), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
