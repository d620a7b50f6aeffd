// ssdhip_conv64.hip -- 3x3 'same' convolution for the Cin = 64 layers (conv1_2, conv2_1 of models/keras_ssd300.py:275-279
// and twins) + bias + ReLU [+ 2x2/2 max-pool], gfx950, bf16 NHWC, float32 accumulation.
//
// Why its own kernel.  On these layers the implicit-GEMM kernel of ssdhip_conv.hip moves 216 KB from L2 into LDS per
// 128-pixel x 64-channel tile (nine tap tiles of activations + nine weight tiles) for 72 MFMAs per wave, and pays nine
// barriers: it runs at ~19 % of the MFMA peak there.  With Cin = 64 the WHOLE filter bank of a 64-channel output slice is
// 9 x 64 x 64 x 2 B = 72 KB -- it fits in LDS next to the activations.  So:
//   * persistent workgroups (one per CU): the 72 KB weight slice is loaded ONCE and stays resident;
//   * per tile only the activation HALO (the (2 RP + 2) x (CC + 2) pixels a 2 RP x CC tile reads through its nine taps,
//     ~23 KB) comes in -- by LDS-DMA into the other half of a double buffer while the current tile is multiplied; the nine
//     taps are nine constant byte displacements into that halo (rows are padded to 144 B so a displacement does not change
//     the bank pattern: no swizzle term, every fragment address is base register + immediate);
//   * image borders need no per-tap masks: out-of-image halo pixels are out-of-range buffer offsets, which the buffer unit
//     turns into zeros in LDS;
//   * ONE barrier per tile (9x fewer), 8 LDS-DMA loads per wave per tile (7x fewer), L2 -> LDS traffic 9x lower.
// One wave per SIMD (the workgroup owns the CU's LDS), so the K loop is software pipelined by hand: the 36 (tap, k16) steps of
// a tile are fully unrolled with the fragment reads running two steps ahead of the MFMAs through a three-slot register ring.
//
// Tile geometry, epilogues and numerics are those of conv_igemm4_pool_kernel (ssdhip_conv.hip): 2-D tiles of RP row pairs x
// CC columns; pooled epilogue = 2x2 max on the float32 accumulators (vertical in-lane, horizontal by DPP), then bias, ReLU, one
// bf16 rounding.  Results are bit-identical to the implicit-GEMM kernel (same K order per output: taps outer, channels inner).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int C64_THREADS = 320;                         // four multiplying waves (one per SIMD) + one loader wave
constexpr int C64_WBYTES = 9 * 64 * 128;                 // resident weight slice: [tap][co][64 ci] bf16, 128-byte rows
constexpr int C64_STAGE = 4 * 2048;                      // per-wave output transpose: 32 px x 64 B (non-pooled: twice per tile)
constexpr int c64_halo_bytes(int cs) { return ((9 * (2 * (64 >> cs) + 2) * ((1 << cs) + 2) + 63) / 64) * 1024; }
constexpr int c64_lds_bytes(int cs, int nb) { return C64_WBYTES + nb * c64_halo_bytes(cs) + C64_STAGE; }

struct C64Params {
    const bf16_t* x;             // [B, H, W, 64]
    const bf16_t* w;             // [Cout, 3, 3, 64]
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [B, H, W, Cout]  or pooled [B, Ho, Wo, Cout]
    int B, H, W, Cout, relu;
    int HT, WT, n_slices, tiles; // tiles per image grid: HT x WT, tiles = B * HT * WT
    int Ho, Wo;                  // pooled map (POOL)
    int x_bytes, w_bytes;
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
__device__ __forceinline__ float c64_relu(float v) { return v <= 0.f ? 0.f : v; }     // NaN stays NaN, -0 -> +0 (as ssdhip_conv.hip)

// one wave-wide 1 KiB LDS-DMA load (see ssdhip_conv.hip: inline asm so hipcc does not drain vmcnt before aliasing ds_reads)
__device__ __forceinline__ void c64_bload(u32 voff, i32x4 rsrc, u32 lds_dst, u32 soff = 0) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
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

template <int CS, bool POOL, int NB>
__device__ __forceinline__ void conv64_body(const C64Params& p, unsigned char* lds) {
    constexpr int CC = 1 << CS, RP = 64 >> CS;           // tile: RP row pairs x CC columns = 128 pixels
    constexpr int HC = CC + 2, HR = 2 * RP + 2;          // halo columns / rows
    constexpr int HPX = HR * HC;                         // halo pixels, one 144-byte LDS row each (128 data + 16 pad)
    constexpr int SLOTS = 9 * HPX;                       // 16-byte slots of a halo buffer
    constexpr int NP = (SLOTS + 63) / 64;                // 1 KiB LDS-DMA pieces per halo
    constexpr int HB = NP * 1024;                        // NB halo buffers: loads run NB - 1 tiles ahead of the MFMAs
    static_assert(HB == c64_halo_bytes(CS) && c64_lds_bytes(CS, NB) <= 160 * 1024, "LDS budget");
    constexpr int W_OFF = 0, H_OFF = C64_WBYTES, STAGE_OFF = C64_WBYTES + NB * HB;
    constexpr unsigned OOB = 0x80000000u;

    const int G = (int)gridDim.x;
    const int slice = (int)blockIdx.x % p.n_slices;
    const int first = (int)blockIdx.x / p.n_slices, stride = G / p.n_slices;
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
    if (wave < 4) {
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

    // ---- the loader wave.  VMEM operations complete in issue order and a global store takes microseconds to be acknowledged:
    //      a wave that both stores outputs and waits for halo loads ends up waiting for its own older stores every tile (the
    //      first version of this kernel: 4 us per tile whatever the prefetch depth).  So the halos are fetched by a fifth wave
    //      that never stores -- its vmcnt counts loads only -- and the four multiplying waves never wait on vmcnt at all.
    //      Per tile i:  loader: halo i+1 landed (halo i+2 may fly) | barrier | issue halo i+3 into the buffer tile i released. ----
    if (wave == 4) {
        // slot n = 64 * piece + lane of a halo buffer -> halo pixel n / 9 (row hr, column hc), 16-byte chunk n % 9 (8 = padding)
        int hrc[NP];                                     // hr << 16 | hc << 4 | chunk, or -1 (padding slot / beyond the halo)
        u32 rel[NP];                                     // byte offset of the slot's 16 bytes from the halo origin pixel (or OOB)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int n = 64 * k + lane;
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
            const u32 dst = lds0 + H_OFF + buf * HB;
            if (h0 >= 1 && w0 >= 1 && h0 + 2 * RP + 1 <= p.H && w0 + CC + 1 <= p.W) {
                const u32 soff = (u32)(((b * p.H + h0 - 1) * p.W + (w0 - 1)) * 128 + xneg);
#pragma unroll
                for (int k = 0; k < NP; ++k) c64_bload(rel[k], rx, dst + k * 1024, soff);
            } else {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int h = h0 - 1 + (hrc[k] >> 16), w = w0 - 1 + ((hrc[k] >> 4) & 0xfff);
                    const bool ok = hrc[k] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                    const u32 voff = ok ? (u32)(((b * p.H + h) * p.W + w) * 128 + (hrc[k] & 15) * 16 + xneg) : OOB;
                    c64_bload(voff, rx, dst + k * 1024);
                }
            }
        };
        static_assert(NB == 3 && 2 * NP <= 63, "wait accounting: two halos of NP loads in flight must fit vmcnt");
        issue_halo(first, 0);
        if (first + stride < p.tiles) issue_halo(first + stride, 1);
        if (first + 2 * stride < p.tiles) issue_halo(first + 2 * stride, 2);
        if (first + 2 * stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP) : "memory");
        else if (first + stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // halo of the first tile (+ the multipliers' weights) in place
        int buf = 0;
        for (int tile = first; tile < p.tiles; tile += stride) {
            if (tile + 2 * stride < p.tiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");   // halo i+1 landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // the multipliers are done with buffer `buf`
            if (tile + 3 * stride < p.tiles) issue_halo(tile + 3 * stride, buf);
            buf = buf + 1 == NB ? 0 : buf + 1;
        }
        return;
    }

    // ---- fragment addresses: everything per-lane is fixed for the whole kernel -------------------------------------------
    const int q = wp * 32 + r31;                         // pixel slot of the lane: row pair q >> CS, column q & (CC-1)
    const int rp = q >> CS, col = q & (CC - 1);
    u32 bbase[2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) bbase[pi] = (u32)(H_OFF + ((2 * rp + pi) * HC + col) * 144 + khalf * 16);
    u32 abase[4];
    {
        const int row = wc * 32 + r31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) abase[kk] = (u32)(W_OFF + row * 128 + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4));
    }
    // bias of the lane's 16 channels (D row = channel (v & 3) + 8 * (v >> 2) + 4 * khalf of the wave's 32)
    float bv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = co0 + wc * 32 + 8 * g + 4 * khalf + e;
            bv[4 * g + e] = p.bias ? __uint_as_float((u32)p.bias[ch] << 16) : 0.f;
        }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the weights (the only loads it ever waits for)
    __builtin_amdgcn_s_barrier();

    int buf = 0;
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
        constexpr int RD = 5, RS = RD + 1;
        bf16x8 fa[RS], fb0[RS], fb1[RS];
        auto rd = [&](const int s, const int slot) {
            const int t = s >> 2, kk = s & 3;
            const int timm = ((t / 3) * HC + (t % 3)) * 144 + kk * 32;
            fb0[slot] = *reinterpret_cast<const bf16x8*>(hb + bbase[0] + timm);   // the step's first MFMA consumes the LAST two reads,
            fa[slot] = *reinterpret_cast<const bf16x8*>(lds + abase[kk] + t * 8192);  // so one counted wait per step covers all three
            fb1[slot] = *reinterpret_cast<const bf16x8*>(hb + bbase[1] + timm);
        };
#pragma unroll
        for (int s = 0; s < RD; ++s) rd(s, s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + RD < 36) rd(s + RD, (s + RD) % RS);
            __builtin_amdgcn_sched_barrier(0);             // pin the order: hipcc otherwise sinks the reads next to their use
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % RS], fb1[s % RS], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % RS], fb0[s % RS], acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        __builtin_amdgcn_s_barrier();                      // the loader wave arrives here once halo i+1 has landed; all four
        buf = buf + 1 == NB ? 0 : buf + 1;                 // multipliers are done reading buffer `buf`, which it may now refill

        // ---- epilogue (wave-private LDS stage: DS operations of one wave execute in order) ------------------------------
        int b, h0, w0;
        tile_origin(tile, b, h0, w0);
        if constexpr (POOL) {
            unsigned char* stage = lds + STAGE_OFF + wave * 2048;          // [16 pooled px][64 B]
            const int hq = h0 + 2 * rp, wq = w0 + col;
            // 2x2 maximum in registers, then bias + ReLU + rounding (monotonic: equals pooling the rounded activations).
            // Interior tiles (all 128 pixels inside the image: 95 % of a 300x300 map) take the mask-free path.
            auto pooled = [&](auto edge) {
                constexpr bool EDGE = decltype(edge)::value;
                const bool has_below = !EDGE || hq + 1 < p.H, has_right = !EDGE || wq + 1 < p.W;
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
                        v += bv[4 * g + e];
                        o[e] = p.relu ? c64_relu(v) : v;
                    }
                    if (!(r31 & 1)) {
                        const int px = r31 >> 1;
                        *reinterpret_cast<uint2*>(stage + px * 64 + ((g ^ (px & 3)) << 4) + khalf * 8) =
                            make_uint2(c64_pack2(o[0], o[1]), c64_pack2(o[2], o[3]));
                    }
                }
            };
            if (h0 + 2 * RP <= p.H && w0 + CC <= p.W) pooled(std::false_type{});
            else pooled(std::true_type{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const int px = lane >> 2, c = lane & 3;                    // 16 px x 4 chunks = 64 lanes
                const int qe = wp * 32 + 2 * px;
                const int ho = (h0 >> 1) + (qe >> CS), wo = (w0 + (qe & (CC - 1))) >> 1;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 64 + ((c ^ (px & 3)) << 4));
                if (ho < p.Ho && wo < p.Wo)
                    *reinterpret_cast<uint4*>(p.y + ((size_t)(b * p.Ho + ho) * p.Wo + wo) * p.Cout + co0 + wc * 32 + c * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            unsigned char* stage = lds + STAGE_OFF + wave * 2048;          // [32 px][64 B], used once per accumulator block
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[pi][4 * g + e] + bv[4 * g + e];
                        o[e] = p.relu ? c64_relu(v) : v;
                    }
                    *reinterpret_cast<uint2*>(stage + r31 * 64 + ((g ^ (r31 & 3)) << 4) + khalf * 8) =
                        make_uint2(c64_pack2(o[0], o[1]), c64_pack2(o[2], o[3]));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int idx = j * 64 + lane, px = idx >> 2, c = idx & 3;          // 32 px x 4 chunks
                    const int qq = wp * 32 + px;
                    const int h = h0 + 2 * (qq >> CS) + pi, w = w0 + (qq & (CC - 1));
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 64 + ((c ^ (px & 3)) << 4));
                    if (h < p.H && w < p.W)
                        *reinterpret_cast<uint4*>(p.y + ((size_t)(b * p.H + h) * p.W + w) * p.Cout + co0 + wc * 32 + c * 8) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int CS, bool POOL, int NB>
__global__ __launch_bounds__(C64_THREADS, 1) void conv64_kernel(C64Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[c64_lds_bytes(CS, NB)];
    conv64_body<CS, POOL, NB>(p, lds);
#endif
}

}  // namespace ssdhip

using namespace ssdhip;

// pool != 0: MaxPooling2D(2, 2, 'same') fused; y is [B, ceil(H/2), ceil(W/2), Cout].  n_workgroups: persistent workgroups to
// launch (the caller passes the CU count; 0 = 256).
extern "C" int ssdhip_conv3x3_c64_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                            int Cin, int Cout, int relu, int pool, int n_workgroups, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || Cin != 64 || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * 128, wb = (long long)Cout * 1152;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout > 0x7fffffff0LL) return SSDHIP_E_BADARG;
    C64Params p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.B = B; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb;
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
#define C64_LAUNCH(CS_, POOL_) hipLaunchKernelGGL((conv64_kernel<CS_, POOL_, 3>), dim3(G), dim3(C64_THREADS), 0, stream, p)
    if (pool) {
        if (cs_best == 3) C64_LAUNCH(3, true); else C64_LAUNCH(4, true);
    } else {
        if (cs_best == 3) C64_LAUNCH(3, false); else C64_LAUNCH(4, false);
    }
#undef C64_LAUNCH
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
