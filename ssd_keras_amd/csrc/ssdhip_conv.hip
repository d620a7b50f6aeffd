// ssdhip_conv.hip -- 'same' convolution (3x3 or 1x1, stride 1, any dilation) + bias + ReLU as ONE implicit-GEMM MFMA kernel
// for gfx950 (MI355X), bf16 NHWC activations, float32 accumulation.
//
// Reference: every Conv2D(..., padding='same', activation='relu') of the VGG-16 trunk and fc6/fc7,
// models/keras_ssd300.py:274-300 (keras_ssd512.py twin).  The framework path runs conv (MIOpen) + bias add + clamp as
// three kernels; here the bias/activation lives in the epilogue of the GEMM, so the activation tensor is written once.
//
// GEMM view:  D[co][m] = sum_{tap, ci} Wt[co][tap][ci] * X[m + off(tap)][ci]     m = flattened (b, h, w) pixel index
//   * MFMA 'A' operand = weights (rows = output channels), 'B' operand = pixels: each lane then owns 4 consecutive
//     channels of one pixel per accumulator quad -> 8-byte NHWC stores straight from registers.
//   * workgroup tile: BC (64|128) channels x 128 pixels x 64 input channels of one tap per K-step; 4 waves as 2x2,
//     v_mfma_f32_32x32x16_bf16, 16 (BC=128) MFMAs per wave per K-step.
//   * global -> LDS with global_load_lds_dwordx4 (no VGPR staging): LDS image is lane-linear, so the 16-byte chunks of
//     a 128-byte row are permuted on the SOURCE side (chunk j of row r sits at position j ^ ((r >> 1) & 7)) and the
//     same involution is applied when reading fragments -> ds_read_b128 free of bank conflicts.
//   * padding: a pixel whose tap falls outside the image reads from a 64-byte zero block instead (per-lane source
//     pointer select; validity of the 9 taps is a bit mask computed once per thread).
//   * two LDS buffers, one barrier per K-step: loads of step s+1 are in flight while step s is multiplied.
//   * blockIdx -> tile map keeps the channel tiles of one pixel tile on the same XCD (shared L2 for X).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(64))) const unsigned int g_zero_block[16] = {0};

constexpr int CONV_BP = 128;     // pixels per workgroup tile
constexpr int CONV_BK = 64;      // input channels per K-step (128 bytes per row)
constexpr int CONV_THREADS = 256;

struct ConvParams {
    const bf16_t* x;             // [M, Cin]
    const bf16_t* w;             // [Cout, KS*KS, Cin]  (torch OIHW weight in channels_last memory)
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [M, Cout]
    int H, W, Cin, Cout, KS, dil, relu;
    int M, m_tiles, n_tiles;
    int Ho, Wo, WT, HT, cshift;  // fused 2x2/2 max-pool: pooled map size; a tile is (64 >> cshift) row pairs x (1 << cshift)
                                 // columns, WT x HT tiles per image
    int stride, coff, Min;       // v4 only: output pixel (ho, wo) is centred on input pixel (ho*stride + coff, wo*stride + coff),
                                 // coff = (KS/2)*dil - pad >= 0; then Ho x Wo is the output map, M = B*Ho*Wo and Min = B*H*W.
                                 // stride 1, coff 0 is the 'same' convolution (Min == M)
    int ksplit;                  // split-K form only: K ranges per tile; range ks writes its float32 partial tile to slab[ks]
    float* slab;                 // [ksplit, M, Cout] float32 partial sums (caller's workspace)
    // reference-precision form (X3) only: x rows hold xC = 2 C float16 channels [hi | lo], the K loop walks Cin = 3 C channels
    // [x hi . w hi | x hi . w lo | x lo . w hi]; bias32 is float32; the accumulator is multiplied by oscale (the weights' power-of-two
    // scale undone) before the bias; out_f32: y is float32 [M, Cout], otherwise float16 [M, 2 Cout] = [hi | lo]
    int xC, out_f32;
    const float* bias32;
    float oscale;
};

__device__ __forceinline__ u32 f2bf_rn(float f) {
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BC>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm_kernel(ConvParams p) {
    constexpr int CI = BC / 64;                       // 32-channel MFMA tiles per wave
    constexpr int WROWS = BC;                         // weight rows per tile
    constexpr int XBYTES = CONV_BP * 128, WBYTES = WROWS * 128, BUF = XBYTES + WBYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];

    // XCD-aware tile map: workgroup id -> (pixel tile, channel tile); all channel tiles of a pixel tile on one XCD
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int mt = (slot / p.n_tiles) * 8 + xcd, nt = slot % p.n_tiles;
    if (mt >= p.m_tiles) return;
    const int m0 = mt * CONV_BP, co0 = nt * BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;
    const int Cin = p.Cin, KK = p.KS * p.KS, half = p.KS >> 1;
    const int csteps = Cin / CONV_BK, T = KK * csteps;

    // ---- per-thread load descriptors: 4 X rows (+ tap validity) and CI*2 weight rows --------------------------
    const bf16_t* xsrc[4];
    u32 xok[4];
    const int pos = lane & 7;
    {
        // rows of this thread: m0 + wave*8 + (lane>>3) + 32*i; (h, w) of the first by division, the others by stepping
        int m = m0 + wave * 8 + (lane >> 3);
        int wq = m % p.W, hq = (m / p.W) % p.H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            const int j = pos ^ ((row >> 1) & 7);
            u32 rmask = 0, cmask = 0;                  // taps valid along h / along w; tap (kh, kw) valid iff both
            for (int k = 0; k < p.KS; ++k) {
                const int d = (k - half) * p.dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (m < p.M)
                for (int kh = 0; kh < p.KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * p.KS);
            xok[i] = ok;
            xsrc[i] = p.x + (size_t)m * Cin + j * 8;
            m += 32;
            wq += 32;
            while (wq >= p.W) { wq -= p.W; if (++hq == p.H) hq = 0; }
        }
    }
    const bf16_t* wsrc[CI * 2];
#pragma unroll
    for (int i = 0; i < CI * 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int j = pos ^ ((row >> 1) & 7);
        wsrc[i] = p.w + (size_t)(co0 + row) * KK * Cin + j * 8;
    }
    const bf16_t* zsrc = reinterpret_cast<const bf16_t*>(g_zero_block);               // 16 readable zero bytes

    auto issue = [&](int s, int buf) {
        const int t = s / csteps, c0 = (s - t * csteps) * CONV_BK;
        const int dh = (t / p.KS - half) * p.dil, dw = (t % p.KS - half) * p.dil;
        const long xoff = ((long)dh * p.W + dw) * Cin + c0;
        const long woff = (long)t * Cin + c0;
        unsigned char* xb = lds + buf * BUF;
        unsigned char* wb = xb + XBYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = ((xok[i] >> t) & 1u) ? xsrc[i] + xoff : zsrc;
            glds16(src, xb + (i * 4 + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < CI * 2; ++i) glds16(wsrc[i] + woff, wb + (i * 4 + wave) * 1024);
    };

    f32x16 acc[CI][2];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;

    const int r31 = lane & 31, khalf = lane >> 5;
    const int swz = (r31 >> 1) & 7;                   // tile-row bases are multiples of 32, so (row >> 1) & 7 == this
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        if (s + 1 < T) issue(s + 1, (s + 1) & 1);
        const unsigned char* xb = lds + (s & 1) * BUF;
        const unsigned char* wb = xb + XBYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int choff = ((2 * kk + khalf) ^ swz) << 4;
            bf16x8 a[CI], b[2];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
                a[ci] = *reinterpret_cast<const bf16x8*>(wb + (wc * (BC / 2) + ci * 32 + r31) * 128 + choff);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
                b[pi] = *reinterpret_cast<const bf16x8*>(xb + (wp * 64 + pi * 32 + r31) * 128 + choff);
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int pi = 0; pi < 2; ++pi) acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ci], b[pi], acc[ci][pi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next step's tile has landed (this wave's part) ...
        __syncthreads();                                         // ... and everybody's; nobody still reads this step's buffer
    }

    // ---- epilogue: D row = channel (v&3) + 8*(v>>2) + 4*(lane>>5), column = pixel lane&31.  A lane owns 4-channel
    //      slivers of 2 pixels; written straight out they would be 8-byte stores scattered over 32 cache lines per
    //      instruction.  The wave transposes its 64-pixel x (32*CI)-channel tile through LDS (idle by now: the loop
    //      ended on a barrier) and stores whole 16-byte chunks, 8 (CI = 2) lanes per contiguous 128-byte pixel row. ----
    constexpr int ROWB = 64 * CI;                      // bytes per pixel row of the wave tile
    unsigned char* stage = lds + wave * (64 * ROWB);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int px = pi * 32 + r31;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ch);
                    bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
                    bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
                }
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[ci][pi][4 * g + q] + bv[q];
                    if (p.relu) v = v > 0.f ? v : (v != v ? v : 0.f);
                    o[q] = f2bf_rn(v);
                }
                const int chunk = ci * 4 + g;          // 16-byte chunk of the pixel row; this lane fills half of it
                *reinterpret_cast<uint2*>(stage + px * ROWB + ((chunk ^ (px & (4 * CI - 1))) << 4) + khalf * 8) =
                    make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
    }
    __syncthreads();
    constexpr int CPR = 4 * CI;                        // chunks per pixel row
#pragma unroll
    for (int j = 0; j < CPR; ++j) {
        const int idx = j * 64 + lane, px = idx / CPR, c = idx % CPR;
        const int m = m0 + wp * 64 + px;
        if (m < p.M)
            *reinterpret_cast<uint4*>(p.y + (size_t)m * p.Cout + co0 + wc * (BC / 2) + c * 8) =
                *reinterpret_cast<const uint4*>(stage + px * ROWB + ((c ^ (px & (CPR - 1))) << 4));
    }
}



// =================================================================================================================
// v4: the v1 tile (BC channels x 128 pixels x 64 input channels per K-step, 4 waves as 2x2, two LDS buffers) with the
// per-step instruction overhead taken out -- rocprofv3 PMC of v1 on conv4_2 showed each wave ISSUING for ~840 cycles per
// K-step against 512 cycles of MFMA work (SQ_ACTIVE_INST_ANY), and the ISA showed why:
//   * two scalar integer divisions per step to turn the step number into (tap, channel slice)  -> the (kh, kw, slice)
//     counters are advanced incrementally;
//   * per LDS-DMA load two 64-bit pointer adds, a mask test and a two-register pointer select against the zero block,
//     and a v_readfirstlane for M0                                                               -> buffer addressing:
//     the row's byte offset is ONE 32-bit VGPR, the step's (tap, slice) displacement rides in the scalar soffset (not
//     range checked), a tap that falls outside the image becomes an out-of-range voffset (0x80000000 >= num_records:
//     the buffer unit returns zeros into LDS), and the wave number is made scalar once so the LDS destination is SALU;
//   * the compiler read one kk-slice of fragments, waited lgkmcnt(0), issued 4 MFMAs, four times per step (LDS latency
//     exposed four times)                                 -> all 16 fragment reads of the step are issued back to back
//     into four register sets and the MFMAs start as each set lands (counted lgkmcnt).
// The descriptor base is x minus the largest negative tap displacement `neg`, so every soffset is non-negative, and
// num_records = bytes(x) + 2 * neg because on gfx950 the range check covers voffset + soffset (measured: with
// num_records = bytes(x) the last 2W+2 pixels of the batch lost their positive-displacement taps); valid lanes still
// address only bytes inside x.  Needs x and w below 2 GiB each (checked on the host; v1 otherwise).
// =================================================================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The bodies below use amdgcn-only types and builtins (buffer resources, DPP, inline ISA): they exist in the device pass only;
// the host pass sees just the kernel stubs.
#if defined(__HIP_DEVICE_COMPILE__)
// (the body is a __device__ function: the host pass never sees the buffer-resource type, which only exists for amdgcn)
// SK: the split-K form for layers whose K loop is one workgroup deep (the SSD extra layers: a handful of tiles, each walking 4-36
// K-steps at one L2 round trip per step): work item = (tile, K range), the partial tile leaves as float32 to slab[range] and
// splitk_reduce_kernel adds the ranges in order, then bias + activation + one rounding.
// X3: the reference-precision form.  The reference's convolutions are float32 (models/keras_ssd300.py:274-335); a float32 value is
// the sum of two float16 numbers to 2^-22 (hi = fl16(v), lo = fl16(v - hi)), and hi.hi + hi.lo + lo.hi reproduces a product to
// 2^-22 as well -- three float16 MFMA passes with float32 accumulation instead of the 1/16-rate float32 MFMA.  The three passes are
// ONE K loop over 3 C channels: slices [0, n) multiply x hi by w hi, [n, 2n) x hi by w lo, [2n, 3n) x lo by w hi (the filters are
// packed [w hi | w lo | w hi] on the host, x is read through the slice map j -> j < n ? j : j - n).  The epilogue scales, adds the
// float32 bias, applies the activation on float32 values and splits the result again (or leaves it float32 for the graph glue).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32 x3_split2(float a, float b, u32& lo_out) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    lo_out = (u32)__builtin_bit_cast(unsigned short, la) | ((u32)__builtin_bit_cast(unsigned short, lb) << 16);
    return (u32)__builtin_bit_cast(unsigned short, ha) | ((u32)__builtin_bit_cast(unsigned short, hb) << 16);
}

template <int BC, bool POOL, bool SK = false, bool X3 = false>
__device__ __forceinline__ void conv_igemm4_body(const ConvParams& p, unsigned char* lds, const int id_) {
    constexpr int CI = BC / 64;
    constexpr int XBYTES = CONV_BP * 128, WBYTES = BC * 128, BUF = XBYTES + WBYTES;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(!(SK && POOL), "the split-K form has no pooled epilogue");

    const int ks = SK ? id_ % p.ksplit : 0;
    const int id = SK ? id_ / p.ksplit : id_;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / p.n_tiles) * 8 + xcd, nt = slot % p.n_tiles;
    if (mt >= p.m_tiles) return;
    const int m0 = mt * CONV_BP, co0 = nt * BC;       // POOL: mt = ((b * HT + ht) * WT + wt), see below
    int pb = 0, php = 0, pwt = 0;                      // POOL: image, first row pair, column tile of this workgroup
    if constexpr (POOL) { pwt = mt % p.WT; const int r = mt / p.WT; php = (r % p.HT) * (64 >> p.cshift); pb = r / p.HT; }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int Cin = p.Cin, KS = p.KS, KK = KS * KS, half = KS >> 1, dil = p.dil;
    const int csteps = Cin / CONV_BK, T = KK * csteps;
    const int XC = X3 ? p.xC : Cin;                                       // channels of an x row (X3: 2 C of the loop's 3 C)
    const int xslices = X3 ? csteps / 3 : csteps;                         // X3: slice j of the loop reads x slice j < n ? j : j - n

    const int neg = (half * dil * p.W + half * dil) * XC * 2;             // bytes; largest negative tap displacement
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.x)) - neg, 0, p.Min * XC * 2 + 2 * neg, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.w)), 0, p.Cout * KK * Cin * 2, 0x00020000);

    // ---- per-thread load descriptors: byte offsets of 4 X rows (+ tap validity) and CI*2 weight rows ---------------
    u32 xoff[4], xok[4], woff[CI * 2];
    const int pos = lane & 7;
    if (!POOL && (p.stride != 1 || p.coff != 0)) {
        // strided and / or partially padded: the tile's 128 OUTPUT pixels, each centred on its own input pixel; the taps
        // are the same displacements around that centre, so everything after this prologue is unchanged
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            const int j = pos ^ ((row >> 1) & 7);
            const int m = m0 + row;
            const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, b = t / p.Ho;
            const int hq = ho * p.stride + p.coff, wq = wo * p.stride + p.coff;
            u32 rmask = 0, cmask = 0;
            for (int k = 0; k < KS; ++k) {
                const int d = (k - half) * dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (m < p.M)
                for (int kh = 0; kh < KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * KS);
            xok[i] = ok;
            xoff[i] = (u32)((b * p.H + hq) * p.W + wq) * (u32)(XC * 2) + (u32)(j * 16);
        }
    } else if constexpr (!POOL) {
        int m = m0 + wave * 8 + (lane >> 3);
        int wq = m % p.W, hq = (m / p.W) % p.H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            const int j = pos ^ ((row >> 1) & 7);
            u32 rmask = 0, cmask = 0;
            for (int k = 0; k < KS; ++k) {
                const int d = (k - half) * dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (m < p.M)
                for (int kh = 0; kh < KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * KS);
            xok[i] = ok;
            xoff[i] = (u32)m * (u32)(XC * 2) + (u32)(j * 16);
            m += 32;
            wq += 32;
            while (wq >= p.W) { wq -= p.W; if (++hq == p.H) hq = 0; }
        }
    } else {
        // tile = RP = 64 >> cshift row pairs x CC = 1 << cshift columns (shape picked on the host to waste the fewest
        // padded pixels).  Tile-local pixel i (= LDS row = MFMA column wp*64 + pi*32 + r31) has q = (i >> 6) * 32 + (i & 31):
        // image row 2 * (php + q / CC) + ((i >> 5) & 1), column CC * pwt + q % CC -- a lane's two accumulator blocks (pi = 0, 1)
        // are the SAME column of the two rows of a pair, and lane ^ 1 is the neighbouring column of the same pair.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 4 + wave) * 8 + (lane >> 3);
            const int j = pos ^ ((row >> 1) & 7);
            const int q = (row >> 6) * 32 + (row & 31);
            const int hq = 2 * (php + (q >> p.cshift)) + ((row >> 5) & 1), wq = (pwt << p.cshift) + (q & ((1 << p.cshift) - 1));
            u32 rmask = 0, cmask = 0;
            for (int k = 0; k < KS; ++k) {
                const int d = (k - half) * dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (hq < p.H && wq < p.W)
                for (int kh = 0; kh < KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * KS);
            xok[i] = ok;
            xoff[i] = (u32)((pb * p.H + hq) * p.W + wq) * (u32)(XC * 2) + (u32)(j * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < CI * 2; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int j = pos ^ ((row >> 1) & 7);
        woff[i] = (u32)(co0 + row) * (u32)(KK * Cin * 2) + (u32)(j * 16);
    }

    int n_kh = 0, n_kw = 0, n_cs = 0;                  // the step the next issue() loads
    int s_first = 0, s_end = T;
    if constexpr (SK) {                                // steps [T ks / ksplit, T (ks + 1) / ksplit); step s = (slice, kh, kw), taps innermost
        s_first = (int)((long long)T * ks / p.ksplit);
        s_end = (int)((long long)T * (ks + 1) / p.ksplit);
        n_cs = s_first / KK;
        const int t0 = s_first - n_cs * KK;
        n_kh = t0 / KS;
        n_kw = t0 - n_kh * KS;
    }
    auto issue = [&](int buf) {
        const int t = n_kh * KS + n_kw;
        const int xs = (X3 && n_cs >= xslices) ? n_cs - xslices : n_cs;
        const int soff_x = neg + (((n_kh - half) * dil * p.W + (n_kw - half) * dil) * XC + xs * CONV_BK) * 2;
        const int soff_w = (t * Cin + n_cs * CONV_BK) * 2;
        const u32 tapbit = 1u << t;
        unsigned char* xb = lds + buf * BUF + wave * 1024;
        unsigned char* wb = xb + XBYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xb + i * 4096), 16, (xok[i] & tapbit) ? xoff[i] : OOB,
                                                     soff_x, 0, 0);
#pragma unroll
        for (int i = 0; i < CI * 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wb + i * 4096), 16, woff[i], soff_w, 0, 0);
        // taps innermost: the nine taps of one 64-channel slice touch the SAME cache lines (shifted by one pixel / one image
        // row), so consecutive steps re-read lines that are still in L2; with the slice innermost (v1) a line's reuse distance is
        // Cin/64 steps x every workgroup of the XCD -- beyond the 4 MB L2, and each tap's tile came back from MALL/HBM again
        if (++n_kw == KS) { n_kw = 0; if (++n_kh == KS) { n_kh = 0; ++n_cs; } }
    };

    f32x16 acc[CI][2];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;

    const int r31 = lane & 31, khalf = lane >> 5;
    const int swz = (r31 >> 1) & 7;
    const int arow = (wc * (BC / 2) + r31) * 128, brow = (wp * 64 + r31) * 128;
    int choff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) choff[kk] = ((2 * kk + khalf) ^ swz) << 4;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int TS = s_end - s_first;                    // the steps this work item walks (all of them unless split-K)
    for (int s = 0; s < TS; ++s) {
        const unsigned char* xb = lds + (s & 1) * BUF;
        const unsigned char* wb = xb + XBYTES;
        bf16x8 a[4][CI], b[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) a[kk][ci] = *reinterpret_cast<const bf16x8*>(wb + arow + ci * 4096 + choff[kk]);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) b[kk][pi] = *reinterpret_cast<const bf16x8*>(xb + brow + pi * 4096 + choff[kk]);
        }
        __builtin_amdgcn_sched_barrier(0);             // the step's 16 fragment reads go out first ...
        if (s + 1 < TS) issue((s + 1) & 1);             // ... their latency hides behind issuing the next step's LDS-DMA loads
        __builtin_amdgcn_sched_barrier(0);             // (measured: loads before the reads, or s_setprio around the MFMAs, change nothing)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
                    if constexpr (X3)
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[kk][ci]), __builtin_bit_cast(f16x8, b[kk][pi]),
                                                                             acc[ci][pi], 0, 0, 0);
                    else
                        acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][ci], b[kk][pi], acc[ci][pi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);             // all MFMAs are queued before the wave parks on the loads / the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (X3) {
        // ---- reference-precision epilogue: float32 values straight from the registers (a lane holds 4 consecutive channels of a
        //      pixel per accumulator quad); POOL: the 2 x 2 maximum first, on the raw accumulators (scale > 0, the bias is per
        //      channel and the activation monotonic: the same as pooling the finished values) ----------------------------------------
        const int q = wp * 32 + r31;
        int pix[2];
        bool ok[2];
        if constexpr (POOL) {
            const int hq = 2 * (php + (q >> p.cshift)), wq = (pwt << p.cshift) + (q & ((1 << p.cshift) - 1));
            const bool has_below = hq + 1 < p.H, has_right = wq + 1 < p.W;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float m = acc[ci][0][v];
                    const float below = acc[ci][1][v];
                    if (has_below) m = below > m ? below : m;
                    const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xf, 0xf, false));
                    if (has_right) m = right > m ? right : m;
                    acc[ci][0][v] = m;
                }
            const int ho = php + (q >> p.cshift), wo = wq >> 1;
            pix[0] = (pb * p.Ho + ho) * p.Wo + wo;
            ok[0] = !(r31 & 1) && ho < p.Ho && wo < p.Wo && hq < p.H && wq < p.W;
            pix[1] = 0; ok[1] = false;
        } else {
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) { pix[pi] = m0 + wp * 64 + pi * 32 + r31; ok[pi] = pix[pi] < p.M; }
        }
#pragma unroll
        for (int pi = 0; pi < (POOL ? 1 : 2); ++pi)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[ci][pi][4 * g + e] * p.oscale + (p.bias32 ? p.bias32[ch + e] : 0.f);
                        if (p.relu) t = t > 0.f ? t : (t != t ? t : 0.f);
                        v[e] = t;
                    }
                    if (!ok[pi]) continue;
                    if (p.out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)pix[pi] * p.Cout + ch) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        u32 l0, l1;
                        const u32 h0 = x3_split2(v[0], v[1], l0), h1 = x3_split2(v[2], v[3], l1);
                        bf16_t* row = p.y + (size_t)pix[pi] * (2 * p.Cout) + ch;
                        *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(row + p.Cout) = make_uint2(l0, l1);
                    }
                }
        return;
    }
    if constexpr (POOL) {
        // ---- pooled epilogue: max over the 2x2 window in registers (vertical: the lane's two accumulator blocks; horizontal:
        //      lane ^ 1 by DPP quad_perm), THEN bias + ReLU + one bf16 rounding -- all three are monotonic, so this equals
        //      pooling the rounded activations (what Conv2D(activation='relu') -> MaxPooling2D computes).  Even lanes hold the
        //      16 pooled columns of the wave; they go through the LDS transpose and leave as 16-byte stores. ----------------
        constexpr int ROWB = 64 * CI;
        unsigned char* stage = lds + wave * (16 * ROWB);
        const int q = wp * 32 + r31;
        const int hq = 2 * (php + (q >> p.cshift)), wq = (pwt << p.cshift) + (q & ((1 << p.cshift) - 1));
        const bool has_below = hq + 1 < p.H, has_right = wq + 1 < p.W;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ch);
                    bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
                    bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
                }
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[ci][0][4 * g + q];
                    const float below = acc[ci][1][4 * g + q];
                    if (has_below) v = below > v ? below : v;
                    const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
                    if (has_right) v = right > v ? right : v;
                    v += bv[q];
                    if (p.relu) v = v > 0.f ? v : (v != v ? v : 0.f);
                    o[q] = f2bf_rn(v);
                }
                if (!(r31 & 1)) {
                    const int px = r31 >> 1, chunk = ci * 4 + g;
                    *reinterpret_cast<uint2*>(stage + px * ROWB + ((chunk ^ (px & (4 * CI - 1))) << 4) + khalf * 8) =
                        make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
                }
            }
        __syncthreads();
        constexpr int CPR = 4 * CI;
#pragma unroll
        for (int j = 0; j < CPR / 4; ++j) {
            const int idx = j * 64 + lane, px = idx / CPR, c = idx % CPR;
            const int qe = wp * 32 + 2 * px;
            const int ho = php + (qe >> p.cshift), wo = ((pwt << p.cshift) + (qe & ((1 << p.cshift) - 1))) >> 1;
            if (ho < p.Ho && wo < p.Wo)
                *reinterpret_cast<uint4*>(p.y + ((size_t)(pb * p.Ho + ho) * p.Wo + wo) * p.Cout + co0 + wc * (BC / 2) + c * 8) =
                    *reinterpret_cast<const uint4*>(stage + px * ROWB + ((c ^ (px & (CPR - 1))) << 4));
        }
        return;
    }
    if constexpr (SK) {
        // a lane holds 4 consecutive channels of a pixel per accumulator quad: 16-byte float32 stores straight from registers
        float* out = p.slab + (size_t)ks * (size_t)p.M * (size_t)p.Cout;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int m = m0 + wp * 64 + pi * 32 + r31;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                    if (m < p.M)
                        *reinterpret_cast<float4*>(out + (size_t)m * p.Cout + ch) =
                            make_float4(acc[ci][pi][4 * g], acc[ci][pi][4 * g + 1], acc[ci][pi][4 * g + 2], acc[ci][pi][4 * g + 3]);
                }
        }
        return;
    }
    // ---- epilogue: identical to v1 (LDS transpose, 16-byte stores) --------------------------------------------------
    constexpr int ROWB = 64 * CI;
    unsigned char* stage = lds + wave * (64 * ROWB);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int px = pi * 32 + r31;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ch);
                    bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
                    bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
                }
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[ci][pi][4 * g + q] + bv[q];
                    if (p.relu) v = v > 0.f ? v : (v != v ? v : 0.f);
                    o[q] = f2bf_rn(v);
                }
                const int chunk = ci * 4 + g;
                *reinterpret_cast<uint2*>(stage + px * ROWB + ((chunk ^ (px & (4 * CI - 1))) << 4) + khalf * 8) =
                    make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
    }
    __syncthreads();
    constexpr int CPR = 4 * CI;
#pragma unroll
    for (int j = 0; j < CPR; ++j) {
        const int idx = j * 64 + lane, px = idx / CPR, c = idx % CPR;
        const int m = m0 + wp * 64 + px;
        if (m < p.M)
            *reinterpret_cast<uint4*>(p.y + (size_t)m * p.Cout + co0 + wc * (BC / 2) + c * 8) =
                *reinterpret_cast<const uint4*>(stage + px * ROWB + ((c ^ (px & (CPR - 1))) << 4));
    }
}

#endif  // __HIP_DEVICE_COMPILE__

template <int BC>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm4_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (CONV_BP * 128 + BC * 128)];
    conv_igemm4_body<BC, false>(p, lds, (int)blockIdx.x);
#endif
}

template <int BC, bool POOL>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm4_x3_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (CONV_BP * 128 + BC * 128)];
    conv_igemm4_body<BC, POOL, false, true>(p, lds, (int)blockIdx.x);
#endif
}

template <int BC>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm4_splitk_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (CONV_BP * 128 + BC * 128)];
    conv_igemm4_body<BC, false, true>(p, lds, (int)blockIdx.x);
#endif
}

// y[m, co .. co + 7] = act(bias + slab[0][m, co ..] + slab[1][m, co ..] + ...): the ranges are added in order (one fixed float32
// summation order whatever the grid), one rounding to bf16.  One thread per 8 channels of a pixel.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const bf16_t* __restrict__ bias,
                                                            bf16_t* __restrict__ y, int M, int Cout, int ksplit, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n8 = (long long)M * Cout / 8;
    if (i >= n8) return;
    const size_t plane = (size_t)M * Cout;
    float a[8];
    {
        const float4 lo = *reinterpret_cast<const float4*>(slab + i * 8), hi = *reinterpret_cast<const float4*>(slab + i * 8 + 4);
        a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w; a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
    }
    for (int k0 = 1; k0 < ksplit; k0 += 4) {             // four ranges' loads in flight per trip (a plain accumulate loop pays one
        float4 lo[4], hi[4];                              // memory round trip per range: 8 us for the 18 ranges of conv9_2, r03n)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* src = slab + (size_t)(k0 + u < ksplit ? k0 + u : 0) * plane + i * 8;
            lo[u] = *reinterpret_cast<const float4*>(src);
            hi[u] = *reinterpret_cast<const float4*>(src + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < ksplit) {
                a[0] += lo[u].x; a[1] += lo[u].y; a[2] += lo[u].z; a[3] += lo[u].w;
                a[4] += hi[u].x; a[5] += hi[u].y; a[6] += hi[u].z; a[7] += hi[u].w;
            }
    }
    const int co = (int)((i * 8) % Cout);
    u32 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = a[e] + (bias ? __uint_as_float((u32)bias[co + e] << 16) : 0.f);
        if (relu) v = v > 0.f ? v : (v != v ? v : 0.f);
        o[e] = f2bf_rn(v);
    }
    *reinterpret_cast<uint4*>(y + i * 8) = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
}

template <int BC>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm4_pool_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (CONV_BP * 128 + BC * 128)];
    conv_igemm4_body<BC, true>(p, lds, (int)blockIdx.x);
#endif
}

// Several independent convolutions in ONE launch (the six predictor heads of an SSD: a 1x1-pixel head is a single workgroup
// walking 36 K-steps, ~40 us of pure latency when launched alone; side by side the small problems hide behind the big ones).
constexpr int CONV_MAX_GROUP = 8;
struct ConvGroup {
    ConvParams p[CONV_MAX_GROUP];
    int first_block[CONV_MAX_GROUP + 1];     // workgroups of problem k: [first_block[k], first_block[k+1]), each a multiple of 8
    int n;
};

__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm4_group_kernel(ConvGroup g) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (CONV_BP * 128 + 128 * 128)];
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.first_block[k + 1]) ++k;
    // problems whose channel count is a multiple of 128 take the 128-channel tile (same pixels, half the activation re-reads);
    // the host sized n_tiles accordingly
    if (g.p[k].Cout % 128 == 0) conv_igemm4_body<128, false>(g.p[k], lds, (int)blockIdx.x - g.first_block[k]);
    else conv_igemm4_body<64, false>(g.p[k], lds, (int)blockIdx.x - g.first_block[k]);
#endif
}

// =================================================================================================================
// v5: multi-stage LDS ring.  The trace of v4 showed ~1 us per K-step on EVERY layer, big or small, busy CU or idle: a step
// cannot finish before the loads issued at its start have come back from L2 / MALL (prefetch distance = one step), so the
// kernel is bound by load latency x bytes in flight (64 KB per CU), not by MFMA or L2 bandwidth.  v5 keeps v4's tile
// (BC channels x 128 pixels, 4 waves as 2x2) but steps through K in 32-channel slices (16 KB per stage at BC = 128) and
// keeps NS stages in LDS: loads run NS-1 steps ahead of the MFMAs.
//   * NS = 4: 64 KB per workgroup, 2 workgroups per CU, 96 KB in flight per CU; NS = 3: 48 KB, 3 workgroups per CU.
//   * One s_barrier per step, none of them draining the loads: the LDS-DMA loads are issued from inline asm (a
//     __builtin load makes hipcc put s_waitcnt vmcnt(0) in front of every ds_read that might alias an in-flight
//     LDS-DMA -- that is what serialised v3 -- and __syncthreads() waits vmcnt(0) as well), completion is counted by
//     hand: before step s a wave waits until all but its newest (NS-2) x LPW loads have landed, then the barrier makes
//     that true for every wave and also says everybody is done reading the stage that the next loads overwrite.
//   * LDS rows are 64 bytes (32 channels); 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3), which keeps
//     both the lane-linear LDS-DMA image and the ds_read_b128 fragment reads free of bank conflicts.
// =================================================================================================================
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

#if defined(__HIP_DEVICE_COMPILE__)
// one wave-wide 1 KiB LDS-DMA load: lane L writes 16 bytes at lds_dst + 16 L from base(rsrc) + soff + voff (zeros if voff is
// out of range).  M0 is saved and restored inside the statement (hipcc does not model it around asm).
__device__ __forceinline__ void bload_lds16(u32 voff, i32x4 rsrc, u32 lds_dst, u32 soff) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, long offset_bytes, int num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base + (unsigned long long)offset_bytes;
    i32x4 r;
    r.x = (int)(u32)a;
    r.y = (int)((u32)(a >> 32) & 0xffffu);             // stride 0, no swizzle
    r.z = num_records;
    r.w = 0x00020000;
    return r;
}

template <int BC, int NS>
__device__ __forceinline__ void conv_igemm5_body(const ConvParams& p, unsigned char* lds) {
    constexpr int CI = BC / 64;
    constexpr int BK = 32, ROW = 64;                   // bytes per LDS row
    constexpr int XBYTES = CONV_BP * ROW, WBYTES = BC * ROW, STG = XBYTES + WBYTES;
    constexpr int XP = 2, WP = BC / 64, LPW = XP + WP; // 1 KiB pieces per wave per step
    constexpr int D = NS - 1;                          // prefetch distance in steps
    constexpr unsigned OOB = 0x80000000u;

    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int mt = (slot / p.n_tiles) * 8 + xcd, nt = slot % p.n_tiles;
    if (mt >= p.m_tiles) return;
    const int m0 = mt * CONV_BP, co0 = nt * BC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int Cin = p.Cin, KS = p.KS, KK = KS * KS, half = KS >> 1, dil = p.dil;
    const int csteps = Cin / BK, T = KK * csteps;

    const int neg = (half * dil * p.W + half * dil) * Cin * 2;
    const i32x4 rx = make_rsrc(p.x, -(long)neg, p.Min * Cin * 2 + 2 * neg);
    const i32x4 rw = make_rsrc(p.w, 0, p.Cout * KK * Cin * 2);
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---- per-thread load descriptors: a 1 KiB piece = 16 rows x 64 bytes; lane L -> row L >> 2, position L & 3 ------
    u32 xoff[XP], xok[XP], woff[WP];
    const int pos = lane & 3;
    if (p.stride != 1 || p.coff != 0) {
        // strided and / or partially padded (see v4): the tile's OUTPUT pixels, each centred on its own input pixel
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int row = (i * 4 + wave) * 16 + (lane >> 2);
            const int j = pos ^ ((row >> 2) & 3);
            const int m = m0 + row;
            const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, b = t / p.Ho;
            const int hq = ho * p.stride + p.coff, wq = wo * p.stride + p.coff;
            u32 rmask = 0, cmask = 0;
            for (int k = 0; k < KS; ++k) {
                const int d = (k - half) * dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (m < p.M)
                for (int kh = 0; kh < KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * KS);
            xok[i] = ok;
            xoff[i] = (u32)((b * p.H + hq) * p.W + wq) * (u32)(Cin * 2) + (u32)(j * 16);
        }
    } else {
        int m = m0 + wave * 16 + (lane >> 2);
        int wq = m % p.W, hq = (m / p.W) % p.H;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int row = (i * 4 + wave) * 16 + (lane >> 2);
            const int j = pos ^ ((row >> 2) & 3);
            u32 rmask = 0, cmask = 0;
            for (int k = 0; k < KS; ++k) {
                const int d = (k - half) * dil;
                if ((unsigned)(hq + d) < (unsigned)p.H) rmask |= 1u << k;
                if ((unsigned)(wq + d) < (unsigned)p.W) cmask |= 1u << k;
            }
            u32 ok = 0;
            if (m < p.M)
                for (int kh = 0; kh < KS; ++kh)
                    if ((rmask >> kh) & 1u) ok |= cmask << (kh * KS);
            xok[i] = ok;
            xoff[i] = (u32)m * (u32)(Cin * 2) + (u32)(j * 16);
            m += 64;
            wq += 64;
            while (wq >= p.W) { wq -= p.W; if (++hq == p.H) hq = 0; }
        }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = (i * 4 + wave) * 16 + (lane >> 2);
        const int j = pos ^ ((row >> 2) & 3);
        woff[i] = (u32)(co0 + row) * (u32)(KK * Cin * 2) + (u32)(j * 16);
    }

    int n_kh = 0, n_kw = 0, n_cs = 0;                  // the step the next issue() loads (taps innermost, see v4)
    auto issue = [&](int stage) {
        const int t = n_kh * KS + n_kw;
        const u32 soff_x = (u32)(neg + (((n_kh - half) * dil * p.W + (n_kw - half) * dil) * Cin + n_cs * BK) * 2);
        const u32 soff_w = (u32)((t * Cin + n_cs * BK) * 2);
        const u32 tapbit = 1u << t;
        const u32 xb = lds0 + stage * STG + wave * 1024, wb = xb + XBYTES;
#pragma unroll
        for (int i = 0; i < XP; ++i) bload_lds16((xok[i] & tapbit) ? xoff[i] : OOB, rx, xb + i * 4096, soff_x);
#pragma unroll
        for (int i = 0; i < WP; ++i) bload_lds16(woff[i], rw, wb + i * 4096, soff_w);
        if (++n_kw == KS) { n_kw = 0; if (++n_kh == KS) { n_kh = 0; ++n_cs; } }
    };

    f32x16 acc[CI][2];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;

    const int r31 = lane & 31, khalf = lane >> 5;
    const int swz = (r31 >> 2) & 3;                    // tile-row bases are multiples of 32
    const int arow = XBYTES + (wc * (BC / 2) + r31) * ROW, brow = (wp * 64 + r31) * ROW;
    int choff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) choff[kk] = ((2 * kk + khalf) ^ swz) << 4;

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < T) issue(d);

    for (int s = 0; s < T; ++s) {
        // step s's tile has landed (this wave's pieces) once at most min(D-1, T-1-s) newer steps are still in flight
        const int ahead = (T - 1 - s) < (D - 1) ? (T - 1 - s) : (D - 1);
        if (ahead >= 2) wait_vmcnt<2 * LPW>();
        else if (ahead == 1) wait_vmcnt<LPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // ... everybody's pieces; and nobody still reads stage (s-1) % NS
        const unsigned char* sb = lds + (s % NS) * STG;
        bf16x8 a[2][CI], b[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) a[kk][ci] = *reinterpret_cast<const bf16x8*>(sb + arow + ci * 32 * ROW + choff[kk]);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) b[kk][pi] = *reinterpret_cast<const bf16x8*>(sb + brow + pi * 32 * ROW + choff[kk]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + D < T) issue((s + D) % NS);            // overwrites the stage read during step s-1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
                    acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][ci], b[kk][pi], acc[ci][pi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // all fragment reads done: the ring becomes the store stage

    // ---- epilogue: identical to v1 / v4 (LDS transpose, 16-byte stores) ------------------------------------------------
    constexpr int ROWB = 64 * CI;
    unsigned char* stage = lds + wave * (64 * ROWB);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int px = pi * 32 + r31;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = co0 + wc * (BC / 2) + ci * 32 + 8 * g + 4 * khalf;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + ch);
                    bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
                    bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
                }
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[ci][pi][4 * g + q] + bv[q];
                    if (p.relu) v = v > 0.f ? v : (v != v ? v : 0.f);
                    o[q] = f2bf_rn(v);
                }
                const int chunk = ci * 4 + g;
                *reinterpret_cast<uint2*>(stage + px * ROWB + ((chunk ^ (px & (4 * CI - 1))) << 4) + khalf * 8) =
                    make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
    }
    __syncthreads();
    constexpr int CPR = 4 * CI;
#pragma unroll
    for (int j = 0; j < CPR; ++j) {
        const int idx = j * 64 + lane, px = idx / CPR, c = idx % CPR;
        const int m = m0 + wp * 64 + px;
        if (m < p.M)
            *reinterpret_cast<uint4*>(p.y + (size_t)m * p.Cout + co0 + wc * (BC / 2) + c * 8) =
                *reinterpret_cast<const uint4*>(stage + px * ROWB + ((c ^ (px & (CPR - 1))) << 4));
    }
}

#endif  // __HIP_DEVICE_COMPILE__

template <int BC, int NS>
__global__ __launch_bounds__(CONV_THREADS, (NS >= 4 ? 2 : 3)) void conv_igemm5_kernel(ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * (CONV_BP * 64 + BC * 64)];
    conv_igemm5_body<BC, NS>(p, lds);
#endif
}


// =================================================================================================================
// First layer: 3x3 'same' convolution of a 3-channel image into 64 channels (+ bias + ReLU), conv1_1 of
// models/keras_ssd300.py:274.  K = 27 is far too shallow for the implicit-GEMM kernel (MIOpen's generic kernel takes
// 155 us + 148 us of bias/ReLU passes at batch 32); the op is bound by WRITING the 64-channel map (368 MB at batch 32).
// Per workgroup: 256 flattened pixels.  Each thread im2col's its own pixel into one 64-byte LDS row (27 taps, zero padded
// to K = 32, border taps zeroed), then every wave multiplies its 64 pixels by the [64][32] weight image with 8
// v_mfma_f32_32x32x16_bf16 and stores through the same LDS transpose as the implicit-GEMM epilogue.
// =================================================================================================================
constexpr int C1_BP = 256;

__global__ __launch_bounds__(C1_BP) void conv3x3_cin3_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ y,
                                                             int H, int W, int M, int relu) {
    constexpr int CIN = 3, COUT = 64, K = 27, SLEN = (C1_BP + 2) * CIN;
    constexpr int SCHUNKS = (SLEN * 2 + 14 + 15) / 16;      // 16-byte chunks covering a strip at any 2-byte phase
    constexpr int X2_OFF = 0, W2_OFF = 16384, STRIP_OFF = W2_OFF + COUT * 64, STRIP_STRIDE = SCHUNKS * 16;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[STRIP_OFF + 3 * STRIP_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t* w2 = reinterpret_cast<bf16_t*>(lds + W2_OFF);                     // [64][32] bf16, chunk c of row r at c ^ ((r>>2)&3)
    for (int i = tid; i < COUT * 32; i += C1_BP) {
        const int co = i >> 5, k = i & 31;
        const bf16_t v = k < K ? w[co * K + k] : (bf16_t)0;                   // w[co][kh][kw][ci] is already k-major
        w2[co * 32 + ((((k >> 3) ^ ((co >> 2) & 3)) << 3) | (k & 7))] = v;
    }
    // persistent workgroups: the filter bank is repacked into LDS once, then the workgroup walks its tiles (the per-tile
    // barriers below order every reuse of the LDS regions; the store stage of a wave overlays only that wave's im2col rows)
    const int n_tiles = (M + C1_BP - 1) / C1_BP;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int m0 = tile * C1_BP;
    // the three input strips (image rows h-1, h, h+1 of the tile's pixels, one pixel of halo each side) are contiguous
    // byte ranges of x: copied as aligned 16-byte chunks, the 2-byte phase is kept and added back when indexing
    const long total_bytes = (long)M * CIN * 2;
    int sphase[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const long sb = ((long)m0 + (long)(r - 1) * W - 1) * CIN * 2;         // byte offset of the strip in x (may be < 0)
        const long ab = sb & ~15L;
        sphase[r] = (int)(sb - ab);
        for (int j = tid; j < SCHUNKS; j += C1_BP) {
            const long pos = ab + 16L * j;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pos >= 0 && pos + 16 <= total_bytes) {
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(x) + pos);
            } else if (pos + 16 > 0 && pos < total_bytes) {                     // straddles an end of the tensor
                u32 e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const long bp = pos + 2 * q;
                    e[q] = (bp >= 0 && bp < total_bytes) ? (u32)x[bp >> 1] : 0u;
                }
                v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
            *reinterpret_cast<uint4*>(lds + STRIP_OFF + r * STRIP_STRIDE + 16 * j) = v;
        }
    }
    __syncthreads();
    // ---- im2col of this thread's pixel: k = (kh*3 + kw)*3 + ci = kh*9 + r, value strip[kh][tid*3 + r] ---------------
    {
        const int m = m0 + tid;
        const int wq = m % W, hq = (m / W) % H;
        u32 okm = 0;                                    // bit kh*3+kw: tap inside the image
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                if ((unsigned)(hq + kh - 1) < (unsigned)H && (unsigned)(wq + kw - 1) < (unsigned)W) okm |= 1u << (kh * 3 + kw);
        u32 pk[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            u32 v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = 2 * k2 + h;
                if (k < K) {
                    const int kh = k / 9, r = k % 9;
                    const bf16_t* st = reinterpret_cast<const bf16_t*>(lds + STRIP_OFF + kh * STRIP_STRIDE + sphase[kh]);
                    const u32 raw = st[tid * CIN + r];
                    v[h] = ((okm >> (kh * 3 + r / 3)) & 1u) ? raw : 0u;
                } else {
                    v[h] = 0u;
                }
            }
            pk[k2] = v[0] | (v[1] << 16);
        }
        unsigned char* row = lds + X2_OFF + tid * 64;
        const int sw = (tid >> 2) & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<uint4*>(row + ((c ^ sw) << 4)) = make_uint4(pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]);
    }
    __syncthreads();
    // ---- 64 channels x 64 pixels per wave: D[channel][pixel], K = 32 in two MFMA steps --------------------------------
    const int r31 = lane & 31, khalf = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int ch = st * 2 + khalf;
        bf16x8 a[2], b[2];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const int row = ci * 32 + r31;
            a[ci] = *reinterpret_cast<const bf16x8*>(lds + W2_OFF + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const int row = wave * 64 + pi * 32 + r31;
            b[pi] = *reinterpret_cast<const bf16x8*>(lds + X2_OFF + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ci], b[pi], acc[ci][pi], 0, 0, 0);
    }
    __syncthreads();                                    // the im2col rows are dead: their space becomes the store stage
    // per wave and per 32-pixel half: transpose through 4 KB of LDS (wave-private: DS operations of one wave execute in
    // order, so only the compiler has to be held back) and store 16-byte chunks, 8 lanes per contiguous 128-byte pixel row
    unsigned char* stage = lds + wave * 4096;           // [32 pixels][128 bytes]
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = ci * 32 + 8 * g + 4 * khalf;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (bias) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(bias + ch);
                    bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
                    bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
                }
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[ci][pi][4 * g + q] + bv[q];
                    if (relu) v = v > 0.f ? v : (v != v ? v : 0.f);
                    o[q] = f2bf_rn(v);
                }
                *reinterpret_cast<uint2*>(stage + r31 * 128 + (((ci * 4 + g) ^ (r31 & 7)) << 4) + khalf * 8) =
                    make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 64 + lane, px = idx >> 3, c = idx & 7;
            const int m = m0 + wave * 64 + pi * 32 + px;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 128 + ((c ^ (px & 7)) << 4));
            if (m < M) *reinterpret_cast<uint4*>(y + (size_t)m * COUT + c * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    }   // tiles
}

}  // namespace ssdhip

using namespace ssdhip;

static int conv_run(int variant, const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                    int Cin, int Cout, int kernel, int dilation, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || dilation <= 0 || dilation > 8) return SSDHIP_E_BADARG;
    if (kernel != 1 && kernel != 3) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % CONV_BK) || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 7)) return SSDHIP_E_BADARG;
    const long long M = (long long)B * H * W;
    if (M * (long long)(Cin > Cout ? Cin : Cout) > 0x7fffffff0LL || M > 0x7fffff00LL) return SSDHIP_E_BADARG;
    ConvParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = kernel; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.M = (int)M;
    p.Ho = p.Wo = p.WT = p.HT = p.cshift = 0;
    p.stride = 1; p.coff = 0; p.Min = (int)M;
    const bool wide = (Cout % 128) == 0;
    p.n_tiles = Cout / (wide ? 128 : 64);
    const bool small = M * Cin * 2 + 4LL * (dilation * W + dilation) * Cin < 0x7ffff000LL && (long long)Cout * kernel * kernel * Cin * 2 < 0x7ffff000LL;
    if (variant == 4 && !small) variant = 1;  // buffer addressing needs 31-bit byte offsets
    if ((variant == 5 || variant == 6) && !small) variant = 1;
    if (variant == 5 || variant == 6) {       // multi-stage LDS ring: 5 = four 16 KB stages (2 WG/CU), 6 = three (3 WG/CU)
        p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
        const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (variant == 5) {
            if (wide) hipLaunchKernelGGL((conv_igemm5_kernel<128, 4>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm5_kernel<64, 4>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        } else {
            if (wide) hipLaunchKernelGGL((conv_igemm5_kernel<128, 3>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
            else hipLaunchKernelGGL((conv_igemm5_kernel<64, 3>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        }
    } else if (variant == 4) {                // v1's tile with buffer-addressed LDS-DMA and batched fragment reads
        p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
        const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (wide) hipLaunchKernelGGL(conv_igemm4_kernel<128>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL(conv_igemm4_kernel<64>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    } else {                                  // variant 1: per-lane 64-bit addressing, the fallback for tensors beyond 2 GiB
        p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
        const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (wide) hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// y[b,h,w,co] = act(bias[co] + sum_{kh,kw,ci} x[b, h + (kh-k/2)*dil, w + (kw-k/2)*dil, ci] * w[co,kh,kw,ci]), zero padding.
extern "C" int ssdhip_conv2d_same_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                            int Cin, int Cout, int kernel, int dilation, int relu, void* stream) {
    return conv_run(4, x, weight, bias, y, B, H, W, Cin, Cout, kernel, dilation, relu, stream);
}

// General form: stride >= 1 and zero padding 0 <= pad <= (kernel/2)*dilation on every side (torch.nn.Conv2d semantics):
//   y[b,ho,wo,co] = act(bias[co] + sum x[b, ho*stride - pad + kh*dil, wo*stride - pad + kw*dil, ci] * w[co,kh,kw,ci]),
//   Ho = (H + 2*pad - dil*(kernel-1) - 1) / stride + 1 (same for Wo).
// The SSD extra layers: conv6_2 / conv7_2 (ZeroPadding2D(1) + 3x3 stride 2, models/keras_ssd300.py:302-307) and conv8_2 /
// conv9_2 (3x3 'valid', :310-313).  Same kernel as the 'same' convolution: only the tile prologue's pixel -> address map differs.
static int conv_general_run(int variant, const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin,
                            int Cout, int kernel, int stride, int pad, int dilation, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || dilation <= 0 || dilation > 8) return SSDHIP_E_BADARG;
    if ((kernel != 1 && kernel != 3) || stride < 1 || stride > 4) return SSDHIP_E_BADARG;
    const int reach = (kernel / 2) * dilation;
    if (pad < 0 || pad > reach) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % CONV_BK) || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 7)) return SSDHIP_E_BADARG;
    const int Ho = (H + 2 * pad - 2 * reach - 1) / stride + 1, Wo = (W + 2 * pad - 2 * reach - 1) / stride + 1;
    if (H + 2 * pad - 2 * reach < 1 || W + 2 * pad - 2 * reach < 1) return SSDHIP_E_BADARG;
    const long long Min = (long long)B * H * W, M = (long long)B * Ho * Wo;
    if (Min * Cin * 2 + 4LL * (dilation * W + dilation) * Cin >= 0x7ffff000LL || (long long)Cout * kernel * kernel * Cin * 2 >= 0x7ffff000LL ||
        M * Cout > 0x7fffffff0LL)
        return SSDHIP_E_BADARG;                // buffer addressing: 31-bit byte offsets
    ConvParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = kernel; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.M = (int)M; p.Min = (int)Min;
    p.Ho = Ho; p.Wo = Wo; p.WT = p.HT = p.cshift = 0;
    p.stride = stride; p.coff = reach - pad;
    const bool wide = (Cout % 128) == 0;
    p.n_tiles = Cout / (wide ? 128 : 64);
    p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
    const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
    if (variant == 5) {                        // four-stage ring of 32-channel slices: loads three steps ahead
        if (wide) hipLaunchKernelGGL((conv_igemm5_kernel<128, 4>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm5_kernel<64, 4>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    } else if (variant == 6) {
        if (wide) hipLaunchKernelGGL((conv_igemm5_kernel<128, 3>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm5_kernel<64, 3>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    } else {
        if (wide) hipLaunchKernelGGL(conv_igemm4_kernel<128>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL(conv_igemm4_kernel<64>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Split-K form of ssdhip_conv2d_nhwc_bf16 for layers with a handful of tiles (the SSD extra layers conv6_1 ... conv9_2,
// models/keras_ssd300.py:301-313): `ksplit` K ranges per tile (0: chosen so that the launch has about one workgroup per CU)
// write float32 partial tiles to the caller's workspace, a second launch adds them in order and applies bias / activation.
static int splitk_choose(long long tiles, int T) {       // about 200 workgroups, at most 12 ranges (every range is a float32 copy of the output)
    long long ks = (200 + tiles - 1) / tiles;
    if (ks > T) ks = T;
    if (ks > 12) ks = 12;
    return ks < 1 ? 1 : (int)ks;
}
static bool splitk_geometry(int B, int H, int W, int Cin, int Cout, int kernel, int stride, int pad, int dilation, long long* M_out, int* T_out,
                            long long* tiles_out) {
    if (B <= 0 || H <= 0 || W <= 0 || dilation <= 0 || dilation > 8 || (kernel != 1 && kernel != 3) || stride < 1 || stride > 4) return false;
    const int reach = (kernel / 2) * dilation;
    if (pad < 0 || pad > reach || Cin <= 0 || (Cin % CONV_BK) || Cout <= 0 || (Cout % 64)) return false;
    if (H + 2 * pad - 2 * reach < 1 || W + 2 * pad - 2 * reach < 1) return false;
    const int Ho = (H + 2 * pad - 2 * reach - 1) / stride + 1, Wo = (W + 2 * pad - 2 * reach - 1) / stride + 1;
    *M_out = (long long)B * Ho * Wo;
    *T_out = kernel * kernel * (Cin / CONV_BK);
    *tiles_out = ((*M_out + CONV_BP - 1) / CONV_BP) * (Cout / ((Cout % 128) == 0 ? 128 : 64));
    return true;
}

extern "C" size_t ssdhip_conv2d_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kernel, int stride, int pad, int dilation,
                                                       int ksplit) {
    long long M, tiles;
    int T;
    if (!splitk_geometry(B, H, W, Cin, Cout, kernel, stride, pad, dilation, &M, &T, &tiles)) return 0;
    if (ksplit <= 0) ksplit = splitk_choose(tiles, T);
    if (ksplit > T) ksplit = T;
    return (size_t)ksplit * (size_t)M * (size_t)Cout * sizeof(float);
}

extern "C" int ssdhip_conv2d_splitk_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin,
                                              int Cout, int kernel, int stride, int pad, int dilation, int relu, int ksplit, void* ws,
                                              size_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    long long M, tiles;
    int T;
    if (!x || !weight || !y || !ws || !splitk_geometry(B, H, W, Cin, Cout, kernel, stride, pad, dilation, &M, &T, &tiles)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)ws) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    const int reach = (kernel / 2) * dilation;
    const long long Min = (long long)B * H * W;
    if (Min * Cin * 2 + 4LL * (dilation * W + dilation) * Cin >= 0x7ffff000LL || (long long)Cout * kernel * kernel * Cin * 2 >= 0x7ffff000LL ||
        M * Cout > 0x7fffffff0LL)
        return SSDHIP_E_BADARG;                // buffer addressing: 31-bit byte offsets
    if (ksplit <= 0) ksplit = splitk_choose(tiles, T);
    if (ksplit > T) ksplit = T;
    if (ws_bytes < (size_t)ksplit * (size_t)M * (size_t)Cout * sizeof(float)) return SSDHIP_E_WORKSPACE;
    ConvParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = nullptr;
    p.y = nullptr;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = kernel; p.dil = dilation; p.relu = 0;
    p.M = (int)M; p.Min = (int)Min;
    p.Ho = (H + 2 * pad - 2 * reach - 1) / stride + 1; p.Wo = (W + 2 * pad - 2 * reach - 1) / stride + 1; p.WT = p.HT = p.cshift = 0;
    p.stride = stride; p.coff = reach - pad;
    p.ksplit = ksplit; p.slab = static_cast<float*>(ws);
    const bool wide = (Cout % 128) == 0;
    p.n_tiles = Cout / (wide ? 128 : 64);
    p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
    const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8 * ksplit;
    if (wide) hipLaunchKernelGGL(conv_igemm4_splitk_kernel<128>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(conv_igemm4_splitk_kernel<64>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    const long long n8 = M * Cout / 8;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, p.slab, static_cast<const bf16_t*>(bias),
                       static_cast<bf16_t*>(y), (int)M, Cout, ksplit, relu ? 1 : 0);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Reference-precision convolution (see conv_igemm4_body, X3): x [B,H,W,2C] float16 = [hi | lo], weight [Cout,k,k,3C] float16 =
// [w hi | w lo | w hi] of the float32 filters times a power of two 1/oscale, bias float32 [Cout] or NULL; y float32 [B,Ho,Wo,Cout]
// (out_f32) or float16 [B,Ho,Wo,2 Cout] = [hi | lo].  pool != 0: MaxPooling2D(2, 2, 'same') fused (stride 1, pad = reach only).
extern "C" int ssdhip_conv2d_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C, int Cout,
                                         int kernel, int stride, int pad, int dilation, int relu, int pool, int out_f32, float oscale,
                                         void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || dilation <= 0 || dilation > 8) return SSDHIP_E_BADARG;
    if ((kernel != 1 && kernel != 3) || stride < 1 || stride > 4 || !(oscale > 0.f)) return SSDHIP_E_BADARG;
    const int reach = (kernel / 2) * dilation;
    if (pad < 0 || pad > reach || C <= 0 || (C % CONV_BK) || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (pool && (stride != 1 || pad != reach)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 3)) return SSDHIP_E_BADARG;
    if (H + 2 * pad - 2 * reach < 1 || W + 2 * pad - 2 * reach < 1) return SSDHIP_E_BADARG;
    const int Ho = (H + 2 * pad - 2 * reach - 1) / stride + 1, Wo = (W + 2 * pad - 2 * reach - 1) / stride + 1;
    const long long Min = (long long)B * H * W, M = (long long)B * Ho * Wo;
    if (Min * 2 * C * 2 + 4LL * (dilation * W + dilation) * 2 * C >= 0x7ffff000LL || (long long)Cout * kernel * kernel * 3 * C * 2 >= 0x7ffff000LL ||
        M * Cout > 0x3fffffff0LL)
        return SSDHIP_E_BADARG;                // buffer addressing: 31-bit byte offsets
    ConvParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = nullptr;
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = 3 * C; p.Cout = Cout; p.KS = kernel; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.M = (int)M; p.Min = (int)Min;
    p.Ho = Ho; p.Wo = Wo; p.WT = p.HT = p.cshift = 0;
    p.stride = stride; p.coff = reach - pad;
    p.ksplit = 1; p.slab = nullptr;
    p.xC = 2 * C; p.out_f32 = out_f32 ? 1 : 0; p.bias32 = bias; p.oscale = oscale;
    const bool wide = (Cout % 128) == 0;
    p.n_tiles = Cout / (wide ? 128 : 64);
    if (pool) {
        p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
        long long best = -1;                  // tile shape (64 >> cs row pairs x 1 << cs columns) with the smallest padded area
        for (int cs = 6; cs >= 1; --cs) {
            const long long wt = (W + (1 << cs) - 1) >> cs, ht = (p.Ho + (64 >> cs) - 1) / (64 >> cs);
            if (best < 0 || wt * ht < best) { best = wt * ht; p.cshift = cs; p.WT = (int)wt; p.HT = (int)ht; }
        }
        const long long mt = (long long)B * p.HT * p.WT;
        if (mt > 0x3fffffffLL / p.n_tiles) return SSDHIP_E_BADARG;
        p.m_tiles = (int)mt;
        const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (wide) hipLaunchKernelGGL((conv_igemm4_x3_kernel<128, true>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm4_x3_kernel<64, true>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    } else {
        p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
        const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (wide) hipLaunchKernelGGL((conv_igemm4_x3_kernel<128, false>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
        else hipLaunchKernelGGL((conv_igemm4_x3_kernel<64, false>), dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_conv2d_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W, int Cin,
                                       int Cout, int kernel, int stride, int pad, int dilation, int relu, void* stream) {
    return conv_general_run(4, x, weight, bias, y, B, H, W, Cin, Cout, kernel, stride, pad, dilation, relu, stream);
}

// The same with an explicit kernel variant: 4 the default (two LDS stages), 5 / 6 the multi-stage ring (loads three / two steps
// ahead of the MFMAs) -- small maps leave one workgroup per CU, where nothing but the prefetch depth hides the L2 latency.
extern "C" int ssdhip_conv2d_nhwc_bf16_variant(int variant, const void* x, const void* weight, const void* bias, void* y, int B, int H,
                                               int W, int Cin, int Cout, int kernel, int stride, int pad, int dilation, int relu,
                                               void* stream) {
    if (variant != 4 && variant != 5 && variant != 6) return SSDHIP_E_BADARG;
    return conv_general_run(variant, x, weight, bias, y, B, H, W, Cin, Cout, kernel, stride, pad, dilation, relu, stream);
}

// Profiling aid: the same with an explicit kernel variant (4: the shipped kernel; 1: its predecessor with per-lane
// pointers, per-step divisions and per-kk fragment waits -- 10-27 % slower on the VGG shapes; 5 / 6: the multi-stage LDS ring;
// 7: ssdhip_conv3x3_halo_nhwc_bf16).
extern "C" int ssdhip_conv2d_same_nhwc_bf16_variant(int variant, const void* x, const void* weight, const void* bias, void* y,
                                                    int B, int H, int W, int Cin, int Cout, int kernel, int dilation, int relu,
                                                    void* stream) {
    if (variant == 7) {                       // the slab kernel of ssdhip_convh.hip (3x3, dilation 1, Cin % 128 == 0, Cout % 128 == 0)
        if (kernel != 3 || dilation != 1) return SSDHIP_E_BADARG;
        return ssdhip_conv3x3_halo_nhwc_bf16(x, weight, bias, y, B, H, W, Cin, Cout, relu, 0, stream);
    }
    if (variant != 1 && variant != 4 && variant != 5 && variant != 6) return SSDHIP_E_BADARG;
    return conv_run(variant, x, weight, bias, y, B, H, W, Cin, Cout, kernel, dilation, relu, stream);
}

// Conv2D(k x k, padding='same', activation='relu') followed by MaxPooling2D(2, 2, padding='same') as ONE kernel
// (models/keras_ssd300.py:274-283 conv1_2 -> pool1 and twins): y [B, ceil(H/2), ceil(W/2), Cout].
extern "C" int ssdhip_conv2d_same_pool2_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                                  int Cin, int Cout, int kernel, int dilation, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || dilation <= 0 || dilation > 8) return SSDHIP_E_BADARG;
    if (kernel != 1 && kernel != 3) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % CONV_BK) || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 7)) return SSDHIP_E_BADARG;
    const long long M = (long long)B * H * W;
    if (M * Cin * 2 + 4LL * (dilation * W + dilation) * Cin >= 0x7ffff000LL || (long long)Cout * kernel * kernel * Cin * 2 >= 0x7ffff000LL)
        return SSDHIP_E_BADARG;
    ConvParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = kernel; p.dil = dilation; p.relu = relu ? 1 : 0;
    p.M = (int)M;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.stride = 1; p.coff = 0; p.Min = (int)M;
    long long best = -1;                      // tile shape (64 >> cs row pairs x 1 << cs columns) with the smallest padded area
    for (int cs = 6; cs >= 1; --cs) {
        const long long wt = (W + (1 << cs) - 1) >> cs, ht = (p.Ho + (64 >> cs) - 1) / (64 >> cs);
        if (best < 0 || wt * ht < best) { best = wt * ht; p.cshift = cs; p.WT = (int)wt; p.HT = (int)ht; }
    }
    const bool wide = (Cout % 128) == 0;
    p.n_tiles = Cout / (wide ? 128 : 64);
    const long long mt = (long long)B * p.HT * p.WT;
    if (mt > 0x3fffffffLL / p.n_tiles) return SSDHIP_E_BADARG;
    p.m_tiles = (int)mt;
    const int grid = ((p.m_tiles + 7) / 8) * p.n_tiles * 8;
    if (wide) hipLaunchKernelGGL(conv_igemm4_pool_kernel<128>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(conv_igemm4_pool_kernel<64>, dim3(grid), dim3(CONV_THREADS), 0, stream, p);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// n_problems independent 'same' convolutions (no pooling) in one launch; arrays are HOST arrays of per-problem arguments.
extern "C" int ssdhip_conv2d_same_group_nhwc_bf16(int n_problems, const void* const* x_h, const void* const* weight_h,
                                                  const void* const* bias_h, void* const* y_h, const int* B_h, const int* H_h,
                                                  const int* W_h, const int* Cin_h, const int* Cout_h, const int* kernel_h,
                                                  const int* dilation_h, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_problems <= 0 || n_problems > CONV_MAX_GROUP || !x_h || !weight_h || !y_h || !B_h || !H_h || !W_h || !Cin_h || !Cout_h ||
        !kernel_h || !dilation_h)
        return SSDHIP_E_BADARG;
    ConvGroup g;
    g.n = n_problems;
    long long blocks = 0;
    for (int k = 0; k < CONV_MAX_GROUP; ++k) {
        g.first_block[k] = (int)blocks;
        if (k >= n_problems) { g.p[k] = g.p[0]; continue; }
        const int B = B_h[k], H = H_h[k], W = W_h[k], Cin = Cin_h[k], Cout = Cout_h[k], ks = kernel_h[k], dil = dilation_h[k];
        const void* bias = bias_h ? bias_h[k] : nullptr;
        if (!x_h[k] || !weight_h[k] || !y_h[k] || B <= 0 || H <= 0 || W <= 0 || dil <= 0 || dil > 8 || (ks != 1 && ks != 3))
            return SSDHIP_E_BADARG;
        if (Cin <= 0 || (Cin % CONV_BK) || Cout <= 0 || (Cout % 64)) return SSDHIP_E_BADARG;
        if (((uintptr_t)x_h[k] | (uintptr_t)weight_h[k] | (uintptr_t)y_h[k]) & 15 || ((uintptr_t)bias & 7)) return SSDHIP_E_BADARG;
        const long long M = (long long)B * H * W;
        if (M * Cin * 2 + 4LL * (dil * W + dil) * Cin >= 0x7ffff000LL || (long long)Cout * ks * ks * Cin * 2 >= 0x7ffff000LL ||
            M * Cout > 0x7fffffff0LL)
            return SSDHIP_E_BADARG;
        ConvParams& p = g.p[k];
        p.x = static_cast<const bf16_t*>(x_h[k]); p.w = static_cast<const bf16_t*>(weight_h[k]); p.bias = static_cast<const bf16_t*>(bias);
        p.y = static_cast<bf16_t*>(y_h[k]);
        p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KS = ks; p.dil = dil; p.relu = relu ? 1 : 0;
        p.M = (int)M;
        p.Ho = p.Wo = p.WT = p.HT = p.cshift = 0;
        p.stride = 1; p.coff = 0; p.Min = (int)M;
        p.n_tiles = (Cout % 128 == 0) ? Cout / 128 : Cout / 64;
        p.m_tiles = (int)((M + CONV_BP - 1) / CONV_BP);
        blocks += (long long)((p.m_tiles + 7) / 8) * p.n_tiles * 8;
        if (blocks > 0x3fffffffLL) return SSDHIP_E_BADARG;
    }
    g.first_block[CONV_MAX_GROUP] = (int)blocks;
    for (int k = n_problems; k < CONV_MAX_GROUP; ++k) g.first_block[k] = (int)blocks;
    hipLaunchKernelGGL(conv_igemm4_group_kernel, dim3((unsigned)blocks), dim3(CONV_THREADS), 0, stream, g);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// First layer: 3x3 'same' convolution with Cin = 3, Cout = 64 (+ bias + ReLU).
extern "C" int ssdhip_conv3x3_cin3_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                             int Cin, int Cout, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || Cin != 3 || Cout != 64) return SSDHIP_E_BADARG;
    if ((((uintptr_t)y) & 15) || (((uintptr_t)bias) & 7)) return SSDHIP_E_BADARG;
    const long long M = (long long)B * H * W;
    if (M > 0x7fffff00LL) return SSDHIP_E_BADARG;
    hipLaunchKernelGGL(conv3x3_cin3_kernel, dim3((unsigned)(((M + C1_BP - 1) / C1_BP) < 1536 ? ((M + C1_BP - 1) / C1_BP) : 1536)), dim3(C1_BP), 0, stream,
                       static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(weight), static_cast<const bf16_t*>(bias),
                       static_cast<bf16_t*>(y), H, W, (int)M, relu ? 1 : 0);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
