// ssdhip_convh.hip -- 3x3 'same' convolution (stride 1, dilation 1) + bias + ReLU for the deep VGG layers (Cin a multiple of
// 128: conv3_x, conv4_x, conv5_x of models/keras_ssd300.py:284-296 and twins), gfx950, bf16 NHWC, float32 accumulation.
//
// Why its own kernel.  The implicit-GEMM kernel of ssdhip_conv.hip moves 32 KB from L2 into LDS per K-step (one tap of one
// 64-channel slice: a 128-pixel activation tile + a 128-channel weight tile) for 512 cycles of MFMA work; two workgroups per
// CU make that 64 KB per 1024 cycles -- exactly what the CU's vector-memory path delivers (64 B/clk).  Loads and MFMAs are
// balanced, so the kernel tops out at ~41 % of the MFMA peak whatever the prefetch depth (measured).  The cure
// is fewer bytes per FLOP:
//   * the nine taps of a 64-channel slice read the SAME activations shifted by a pixel / an image row, so the tile's
//     activations come in ONCE per slice as a 'slab' (the tile's positions plus a halo of W + 2 positions on either side)
//     and the taps are nine row displacements into it: activation traffic / 9 x (1 + halo);
//   * positions, not pixels: the batch is laid out on a padded grid -- image b, row h, column w sits at position
//     q = (b (H + 1) + h)(W + 1) + w, with one dummy column (w = W) per row and one dummy row (h = H) per image.  Dummy
//     positions are out-of-range buffer offsets (the buffer unit writes zeros into LDS), and every tap of every pixel that
//     falls outside its image lands on one: tap (kh, kw) is the SAME displacement (kh - 1)(W + 1) + (kw - 1) for all positions,
//     with no per-tap validity masks.  A tile is 256 consecutive positions; dummy positions cost (H + 1)(W + 1) / HW - 1 of the
//     MFMA work (5 % on a 38 x 38 map) and are skipped by the epilogue;
//   * tile = 128 channels x 256 positions, 8 waves as 2 x 4 (64 x 64 each, 16 v_mfma_f32_32x32x16_bf16 per K-step), ONE
//     workgroup per CU: per K-step 16 KB of weights + 1/9 slab (~6 KB) for 1024 cycles of MFMA work per SIMD -- a third of the
//     implicit-GEMM kernel's bytes per FLOP.
// Pipeline (per K-step = one tap of one slice, one s_barrier):
//   * weights: ring of NW 16 KB stages, loads NW steps ahead; slab: two buffers, slice cs + 1 arrives during the first taps
//     of slice cs; all by LDS-DMA from inline asm with hand-counted vmcnt (see ssdhip_conv.hip v5 for why not the builtin) --
//     the taps are unrolled, so every s_waitcnt immediate is an exact compile-time count;
//   * the MFMA operands of step s + 1 are read from LDS into a second register set WHILE the 16 MFMAs of step s issue, in 16
//     slots of one MFMA + 0..2 reads (reads in bursts fill the LDS queue and drain the MFMA pipe), so no LDS latency is exposed
//     and the barrier only orders LDS reuse;
//   * persistent workgroups (one per CU) walk over tiles and request the next tile's first slab and weights during the last
//     slice of the current one; the epilogue stages through the slab buffer the last slice has just vacated.
// Geometries: the padded position grid above (maps up to 94 wide), or 2-D tiles of 16 x 16 / 8 x 32 pixels with a one-pixel ring
// (any map size; MaxPooling2D(2, 2, 'same') fused on the float32 accumulators).  Several problems can share one launch (the
// packed predictor heads): work items sorted by depth, dealt to the workgroups in snake order.
// LDS rows are 128 bytes (64 channels); 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7) (source-side permutation
// of the lane-linear DMA image, undone by the fragment reads): conflict-free ds_read_b128 for ANY 32 consecutive rows, so a
// tap displacement only changes the swizzle term, which the readers recompute per step (a handful of VALU operations).
// Accumulation order per output = ssdhip_conv.hip's (slices outer, taps, 16-channel blocks): results are bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

// In-kernel phase timers, profiling build only (tools/prof_build.sh): shader cycles of waves 0 and 4 (the two waves of SIMD 0) in
// (K loop, epilogue, end-of-tile barrier), summed over tiles and workgroups; read back with ssdhip_profile_read_convh.
#ifdef SSDHIP_PROFILE
__device__ unsigned long long g_profh[16];
#define CH_PROF_DECL long long _pt = clock64(); long long _pa[4] = {0, 0, 0, 0}; int _pn = 0;
#define CH_PROF_MARK(i) { const long long _t = clock64(); _pa[i] += _t - _pt; _pt = _t; }
#define CH_PROF_TILE ++_pn;
#define CH_PROF_FLUSH(base) if ((threadIdx.x & 63) == 0) { for (int _i = 0; _i < 4; ++_i) atomicAdd(&g_profh[(base) + _i], (unsigned long long)_pa[_i]); atomicAdd(&g_profh[(base) + 4], (unsigned long long)_pn); }
#else
#define CH_PROF_DECL
#define CH_PROF_MARK(i)
#define CH_PROF_TILE
#define CH_PROF_FLUSH(base)
#endif

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_THREADS = 512;
constexpr int CH_BN = 256;                               // positions per tile
constexpr int CH_BM = 128;                               // output channels per tile
constexpr int CH_WST = CH_BM * 128;                      // bytes of one weight stage
constexpr int ch_lds_bytes(int nw, int spw) { return nw * CH_WST + 2 * spw * 8192; }

struct ConvHParams {
    const bf16_t* x;             // [B, H, W, Cin]
    const bf16_t* w;             // [Cout, 3, 3, Cin]
    const bf16_t* bias;          // [Cout] or null
    bf16_t* y;                   // [B, H, W, Cout]
    int H, W, Cin, Cout, relu;
    int Q, q_tiles, n_tiles;     // padded positions B (H+1)(W+1); tiles of 256 positions; tiles of 128 channels
    int total_ids;               // workgroup ids that map to tiles: ceil(q_tiles / 8) * n_tiles * 8 (some of the last ones to none)
    int HT, WT, Ho, Wo;          // 2-D tiles: tiles per image (q_tiles = B HT WT); pooled map size
    int os, ooff, Hs, Ws;        // position grid only: output pixel (ho, wo) = the 'same' result at (ho os + ooff, wo os + ooff), map Hs x Ws
                                 // (os = 1, ooff = 0, Hs = H, Ws = W: the plain 'same' convolution)
    int x_bytes, w_bytes, y_bytes;
    // reference-precision form (X3, see ssdhip_conv.hip conv_igemm4_body): x rows hold xC = 2 C float16 channels [hi | lo], the K loop
    // walks Cin = 3 C channels (slice j reads x slice j < nx ? j : j - nx), y rows hold 2 Cout float16 channels [hi | lo]; float32 bias
    int xC, nx;
    const float* bias32;
    float oscale;
    // 2-D tiles over the STACKED batch (round 6, fourth session; Hp > 0, pooled forms): the images are laid on top of each other at a
    // pitch of Hp rows (even, >= H + 1: at least one row of zeros between two images, row pairs aligned with every image's first row)
    // and the tile rows walk the stack -- HT = ceil(nB Hp / TR) rows of tiles for the whole batch instead of ceil(H / TR) per image.
    // Row R of the stack is row R - b Hp of image b = R / Hp = umulhi(R, hp_magic); rows H .. Hp - 1 of an image read zeros and store
    // nothing.  SSD300's conv3_3 + pool3 (75 x 75, 16 x 16 tiles): 152 x 5 = 760 position tiles instead of 32 x 25 = 800, i.e. 1 520
    // tile-units = six rounds of 256 CUs where 1 600 needed a seventh.  Same accumulation order per output: bit-identical results.
    int Hp = 0, nB = 0;
    unsigned hp_magic = 0;
    // MSK (training, round 6 fourth session): the data gradient of a layer whose input x is the ReLU output of the layer below leaves
    // the kernel already masked by x > 0 -- mask = that activation, [B, H, W, Cout] like y; what threshold_backward(dL/dx, x, 0) would
    // do in a pass of its own over both maps (csrc/ssdhip_train.hip, relu_bwd_bias_kernel)
    const bf16_t* mask = nullptr;
    // ... and the channel sums of the masked result (the bias gradient of the layer below) on the side: bsum [4 grid / n_tiles][Cout]
    // float32 or null.  A workgroup keeps ONE channel tile over all its position tiles (the launch makes grid / 8 a multiple of n_tiles),
    // so wave (wm, wn) of workgroup w owns row 4 ((w / 8 / n_tiles) 8 + w % 8) + wn, columns co0 + 64 wm ..: written by its first tile,
    // added to by the later ones in program order -- no atomics, a fixed summation order; the weight gradient's reduction launch adds the rows
    float* bsum = nullptr;
    // KEEP (training, fourth session of round 6; pooled forms): the full-resolution activation [B, H, W, Cout] the backward pass needs is
    // stored too -- the accumulators survive the pooled epilogue -- instead of an un-pooled launch followed by a pooling pass that reads
    // the map back (conv2_2 -> pool2, conv3_3 -> pool3; conv1_2 -> pool1 is ssdhip_conv64.hip's KEEP)
    bf16_t* y_full = nullptr;
    int yf_bytes = 0;
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 ch_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ch_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 ch_pack2(float a, float b) {           // v_cvt_pk_bf16_f32: round to nearest even
    const ch_f32x2 v = {a, b};
    return __builtin_bit_cast(u32, __builtin_convertvector(v, ch_bf16x2));
}
__device__ __forceinline__ float ch_relu(float v) { return v <= 0.f ? 0.f : v; }       // NaN stays NaN, -0 -> +0
// Two bf16 values at once as signed 16-bit integers (v_pk_max_i16).  On ROUNDED activations this is the whole activation step:
// max(x, 0) sends every value with the sign bit set (negative numbers, -0) to +0 and leaves the others (+NaN included) alone -- the
// same bits as "v <= 0 ? 0 : v" before the rounding, because rounding to bf16 is monotonic and keeps the sign; max(x, 0x8000) is the
// identity (no activation).  And on NON-NEGATIVE bf16 values integer order is numeric order, so it is also the pooling maximum.
typedef short ch_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 ch_pkmax_i16(u32 a, u32 b) {
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(ch_s16x2, a), __builtin_bit_cast(ch_s16x2, b)));
}

// one wave-wide 1 KiB LDS-DMA load: lane L writes 16 bytes at lds_dst + 16 L from base(rsrc) + soff + voff (zeros if the offset
// is out of range).  M0 is saved and restored inside the statement (hipcc does not model it around asm).
__device__ __forceinline__ void ch_bload(u32 voff, i32x4 rsrc, u32 lds_dst, u32 soff) {
    u32 keep;
    // wave-uniform by construction; the explicit readfirstlane keeps them in SGPRs when hipcc has folded a common factor of the
    // expression into a vector register (an "s" constraint does not insert one by itself)
    lds_dst = (u32)__builtin_amdgcn_readfirstlane((int)lds_dst);
    soff = (u32)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 ch_rsrc(const void* base, int num_records) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    i32x4 r;
    r.x = (int)(u32)a;
    r.y = (int)((u32)(a >> 32) & 0xffffu);
    r.z = num_records;
    r.w = 0x00020000;
    return r;
}
// MODE bits -- 64: the second wave of every SIMD (waves 4..7) reads its fragments two slots later; 8: ... issues its requests two
// slots later; 128: persistent workgroups (one per CU) that walk over tiles and request the next tile's first slab and weights during
// the last slice of the current one.  Profiling build only (wrong results, they isolate one cost each): 1 no loads in the K loop, 2 no
// fragment reads, 4 no waits / barrier, 32 no MFMAs.
// CSH = 0: tiles of 256 consecutive positions of the padded grid (maps up to 94 wide).  CSH = 4 | 5: 2-D tiles of 16 x 16 | 8 x 32
// pixels of ONE image with a one-pixel halo ring in the slab ((TR + 2)(TC + 2) <= 340 rows whatever the map size): the same pipeline,
// only the slab <-> pixel map, the lanes' slab rows and the epilogue differ.  A lane's two position blocks (pi = 0, 1) are then the
// same column of the two rows of a row pair and lane ^ 1 is the neighbouring column, so POOL (MaxPooling2D(2, 2, 'same') fused:
// models/keras_ssd300.py:279-283) takes the 2 x 2 maximum on the float32 accumulators in registers, as conv_igemm4_pool_kernel does.
// SMALL: the map may be narrower than 7 pixels (the epilogue then steps its positions with a loop instead of one select).
// threshold_backward's keep rule on two bf16 activations at once: all ones where the value is NOT <= 0 (positive, or NaN), zero elsewhere
__device__ __forceinline__ u32 ch_keep2(u32 m) {
    const ch_s16x2 h = __builtin_bit_cast(ch_s16x2, m);
    const ch_s16x2 a = h & (short)0x7fff;
    const ch_s16x2 k = (h > (short)0) | (a > (short)0x7f80);
    return __builtin_bit_cast(u32, k);
}
typedef _Float16 ch_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32 ch_split2(float a, float b, u32& lo_out) {        // two float32 -> packed float16 hi parts, lo parts
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    lo_out = (u32)__builtin_bit_cast(unsigned short, la) | ((u32)__builtin_bit_cast(unsigned short, lb) << 16);
    return (u32)__builtin_bit_cast(unsigned short, ha) | ((u32)__builtin_bit_cast(unsigned short, hb) << 16);
}

// NWV = waves per workgroup.  8 (2 x 4, each 64 channels x 64 positions, two waves per SIMD) is the round-2 layout.  4 (round 4: 2 x 2,
// each 64 channels x 128 positions, ONE wave per SIMD with the 512-register budget) reads 24 fragments for 32 MFMAs per K-step
// instead of 2 x 16 for 2 x 16: a quarter less LDS read traffic per FLOP, and no second wave competing for the SIMD's matrix pipe.
// Same tile, same LDS image, same requests (each wave issues twice the pieces), same accumulation order: bit-identical results.
template <int NW, int SPW, int MODE, int CSH, bool POOL, bool SMALL, bool X3 = false, int NWV = 8, bool STK = false, bool MSK = false,
          bool KEEP = false>
__device__ __forceinline__ void convh_body(const ConvHParams& p, unsigned char* lds, const int first_id) {
    constexpr bool G2 = CSH != 0;
    static_assert(!KEEP || (POOL && !X3 && NWV == 8 && !(MODE & (16384 | 1024))), "KEEP: the staged pooled bf16 epilogue, plain waits");
    static_assert(!MSK || (!POOL && !X3 && NWV == 8 && !(MODE & 16384)), "masked outputs: the staged bf16 epilogue without pooling");
    static_assert(!STK || (POOL && !((MODE & 16384) != 0 && !X3)), "stacked-batch tiles: the staged pooled epilogues");
    static_assert(NWV == 8 || (NWV == 4 && !X3), "8 waves, or 4 (bf16 forms only)");
    constexpr int NPI = 16 / NWV;                        // 32-position blocks per wave: 2 | 4
    constexpr int WPK = 8 / NWV;                         // 1 KiB request pieces a wave issues per 64-row chunk: 1 | 2
    constexpr int TC = G2 ? (1 << CSH) : 1, TR = G2 ? (CH_BN >> CSH) : 1, SC2 = TC + 2;   // tile columns, rows; slab columns
    static_assert(!POOL || G2, "the pooled epilogue needs 2-D tiles");
    static_assert(!G2 || (TR + 2) * (TC + 2) <= 64 * SPW, "the 2-D slab must fit the slab buffer");
    // MODE bit 8192 (four weight stages only): requests run NW - 1 steps ahead instead of NW, i.e. one ring stage of slack -- the stage
    // a step's requests overwrite was last READ two steps earlier, and those reads were consumed (waited for, per register, by the MFMAs
    // that use them) before the previous barrier.  The step's closing wait then covers the requests only: fragment reads stay in
    // flight across the barrier instead of all eight waves draining the LDS queue in lockstep before every barrier.
    constexpr bool LZ = (MODE & 8192) != 0 && NW == 4;
    constexpr int D = LZ ? NW - 1 : NW;                  // weights of step s + D are requested during step s
    constexpr int SLAB0 = NW * CH_WST, SLB = SPW * 8192;
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool PERSIST = (MODE & 128) != 0;
    static_assert(SPW + D <= 10, "slice cs+1's slab must have landed two steps before tap 8 of slice cs reads it");
    static_assert(SLB >= 32768, "the epilogue stages half a tile (8 waves x 32 positions x 128 B) in the idle slab buffer");

    // workgroup id -> tile: the channel tiles of a position tile share an XCD (ids are dealt to the XCDs round robin)
    auto tile_of = [&](const int id, int& q0, int& co0) {
        const int xcd = id & 7, slot = id >> 3;
        const int qt = (slot / p.n_tiles) * 8 + xcd;
        q0 = G2 ? qt : qt * CH_BN;                       // 1-D: first position of the tile; 2-D: the tile's index (image, tile row, tile column)
        co0 = (slot % p.n_tiles) * CH_BM;
        return id < p.total_ids && qt < p.q_tiles;
    };
    int id = first_id, q0, co0;
    if (!tile_of(id, q0, co0)) {
        if constexpr (MSK) {                             // no tile at all (the ragged last group of eight ids): its rows of the channel sums are zeros
            if (p.bsum) {
                const int w_ = (int)threadIdx.x >> 6, l_ = (int)threadIdx.x & 63;
                const int slot_ = first_id >> 3, cz = (slot_ % p.n_tiles) * CH_BM + (w_ >> 2) * 64 + l_;
                p.bsum[(size_t)(4 * ((slot_ / p.n_tiles) * 8 + (first_id & 7)) + (w_ & 3)) * p.Cout + cz] = 0.f;
            }
        }
        return;
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NWV == 8 ? wave >> 2 : wave >> 1;     // 64-channel half of the tile
    const int wn = NWV == 8 ? wave & 3 : wave & 1;       // 64-position quarter | 128-position half
    const int r31 = lane & 31, khalf = lane >> 5;
    const int H = p.H, W = p.W, Cin = p.Cin, W1 = W + 1, H1 = H + 1;
    const int csteps = Cin >> 6;
    const int XC = X3 ? p.xC : Cin;                       // channels of an x row
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const i32x4 rx = ch_rsrc(p.x, p.x_bytes);
    const i32x4 rw = ch_rsrc(p.w, p.w_bytes);
    // Outputs leave through a buffer descriptor: a lane that has nothing to store gets an out-of-range offset (dropped by the
    // buffer unit), so every wave issues EXACTLY NST store instructions per tile -- the K loop's first waits after an epilogue
    // count on that (see `post` in step()).
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));
    auto store16 = [&](const uint4 v, const bool ok, const u32 elem) {     // elem: index of the first of the 8 channels (y is below 2 GB)
        if constexpr (!(MODE & 256)) {
            const ch_u32x4 d = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_raw_buffer_store_b128(d, ry, ok ? elem * 2u : OOB, 0, 0);
        }
    };
    auto store16_at = [&](const uint4 v, const u32 byte_off) {            // byte_off: OOB for a lane that has nothing to store
        if constexpr (!(MODE & 256)) {
            const ch_u32x4 d = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_raw_buffer_store_b128(d, ry, byte_off, 0, 0);
        }
    };
    constexpr bool STR = (MODE & 2048) != 0;             // the strided / cropped forms (os, ooff, Hs, Ws); otherwise the plain 'same' result
    constexpr bool DIRECT = (MODE & 16384) != 0 && !X3;  // the epilogue stores from the accumulator layout (v_permlane32_swap), no LDS transpose
    constexpr int NST = (MODE & (256 | 512)) ? 0 : (POOL ? (DIRECT ? 4 : 2) : 8) * (X3 ? 2 : 1) * (NPI / 2);   // global stores a wave issues per epilogue
    const u32 YC = X3 ? 2u * (u32)p.Cout : (u32)p.Cout;  // channels of a y row (X3: [hi | lo])

    // ---- per-lane load descriptors --------------------------------------------------------------------------------------
    // slab row r <-> position q0 - (W + 2) + r; piece (k, wave) = rows 64 k + 8 wave .. + 7, lane -> row (lane >> 3), chunk slot
    // (lane & 7) holding source chunk (lane & 7) ^ ((row >> 1) & 7)
    const int SP = G2 ? (TR + 2) * SC2 : CH_BN + 2 * W + 4;   // slab rows a tile reads
    // STK: the tile's first row of the stack is row h0 of image b; a row h0 + d (d <= TR + 1 <= Hp) of the tile belongs to image b
    // (+ 1 once it reaches Hp: stack_row).  Row -1 of an image is the last row of the gap above it: zeros, like the top of image 0.
    auto tile_origin = [&](const int qt, int& b, int& h0, int& w0) {       // 2-D
        const int wt = qt % p.WT, r = qt / p.WT;
        if constexpr (STK) {
            b = (int)__umulhi((u32)(r * TR), p.hp_magic);
            h0 = r * TR - b * p.Hp;
        } else {
            b = r / p.HT;
            h0 = (r - b * p.HT) * TR;
        }
        w0 = wt * TC;
    };
    auto stack_row = [&](int& b, int& hh) {
        if constexpr (STK) {
            const bool wrap = hh >= p.Hp;
            hh -= wrap ? p.Hp : 0;
            b += wrap ? 1 : 0;
        }
    };
    [[maybe_unused]] const int stackB = STK ? p.nB : 0x7fffffff;
    // request piece idx = k WPK + u of a wave: slab rows (64 / WPK) idx + 8 wave .. + 7
    constexpr int NXO = SPW * WPK, XSTEP = 64 / WPK;
    u32 xoff[NXO];
    auto make_xoff = [&](const int tile_q0) {
        const int row0 = wave * 8 + (lane >> 3);
        if constexpr (G2) {
            // slab row s = (sr, sc) of the (TR + 2) x (TC + 2) halo window: pixel (h0 - 1 + sr, w0 - 1 + sc), zeros outside the image
            int b, h0, w0;
            tile_origin(tile_q0, b, h0, w0);
#pragma unroll
            for (int k = 0; k < NXO; ++k) {
                const int row = row0 + XSTEP * k;
                const int j = (lane & 7) ^ ((row >> 1) & 7);
                const int sr = row / SC2, sc = row - sr * SC2;
                int bb = b, hh = h0 - 1 + sr;
                const int ww = w0 - 1 + sc;
                stack_row(bb, hh);
                const bool ok = row < SP && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W && (!STK || bb < stackB);
                xoff[k] = ok ? (u32)(((bb * H + hh) * W + ww) * (XC * 2) + j * 16) : OOB;
            }
            return;
        }
        int q = tile_q0 - (W + 2) + row0;
        int b = 0, h = 0, w = 0;
        if (q >= 0) { b = q / (H1 * W1); const int r = q - b * (H1 * W1); h = r / W1; w = r - h * W1; }
        else { w = q; }                                  // negative positions: before the first image (zeros)
#pragma unroll
        for (int k = 0; k < NXO; ++k) {
            const int row = row0 + XSTEP * k;
            const int j = (lane & 7) ^ ((row >> 1) & 7);
            const bool ok = w >= 0 && w < W && h < H && q < p.Q && row < SP;
            xoff[k] = ok ? (u32)(((b * H + h) * W + w) * (XC * 2) + j * 16) : OOB;
            q += XSTEP;
            w += XSTEP;
            while (w >= W1) { w -= W1; if (++h == H1) { h = 0; ++b; } }
        }
    };
    make_xoff(q0);
    auto xdst = [&](const int k) { return (u32)((XSTEP * k + 8 * wave) * 128); };   // LDS byte offset of piece k inside a slab buffer
    constexpr int NWP = 2 * WPK;                         // weight pieces a wave issues per stage (16 pieces of 8 channels)
    u32 woff[NWP];                                       // relative to the tile's first channel (which rides in the scalar offset)
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int row = (i * NWV + wave) * 8 + (lane >> 3);
        const int j = (lane & 7) ^ ((row >> 1) & 7);
        woff[i] = (u32)(row * (9 * Cin * 2) + j * 16);
    }

    // the weights of tap `tap` of slice `cs` of channel tile `co` into ring stage `stage` (all wave-uniform): this wave's piece i
    auto issue_w_piece = [&](const int co, const int cs, const int tap, const int stage, const int i) {
        const u32 soff = (u32)(((co * 9 + tap) * Cin + cs * 64) * 2);
        ch_bload(woff[i], rw, lds0 + stage * CH_WST + (i * NWV + wave) * 1024, soff);
    };

    // ---- fragment addressing -----------------------------------------------------------------------------------------------
    u32 abase[4];                                        // weight stage: row = channel, chunk (2 kk + khalf) ^ ((row >> 1) & 7)
    {
        const int row = wm * 64 + r31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) abase[kk] = (u32)(row * 128 + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4));
    }
    // slab row of the lane's first position (pi = 0) at tap (0, 0), and the distance to its second one.  2-D: slot = wn * 32 + r31 is
    // (row pair, column) = (slot / TC, slot % TC) of the tile, pi the row of the pair
    // (NWV = 4: position block pi = 2 ph + pr is row pr of the row pairs of slots wn 64 + ph 32 + r31)
    const int slot2 = wn * (16 * NPI) + r31;
    const int prow = G2 ? (2 * (slot2 >> CSH)) * SC2 + (slot2 & (TC - 1)) : wn * (32 * NPI) + r31;
    auto pirow = [&](const int pi) {                     // slab row distance of position block pi from block 0
        return G2 ? (pi >> 1) * (2 * (32 >> CSH)) * SC2 + (pi & 1) * SC2 : pi * 32;
    };

    f32x16 acc[2][NPI];
    bf16x8 fa[2][4][2], fb[2][4][NPI];                   // [register set][k16 block][ci | pi]

    u32 ra[4], rb[NPI], re[NPI];                         // addresses of the pending fragment reads
    auto read_addr = [&](const int cs, const int tap, const int stage) {     // the fragments of tap `tap` of slice `cs`
        u32 wst = (u32)(stage * CH_WST);
        u32 toff = (u32)((tap / 3) * (G2 ? SC2 : W1) + tap % 3);
        u32 sl = (u32)(SLAB0 + (cs & 1) * SLB);
        // opaque to the optimiser: with the taps unrolled it otherwise computes the addresses of all nine steps up front (72 VGPRs)
        asm volatile("" : "+s"(wst), "+s"(toff), "+s"(sl));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ra[kk] = abase[kk] + wst;
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) {
            const u32 row = (u32)(prow + pirow(pi)) + toff;
            rb[pi] = sl + (row << 7);
            re[pi] = (((row >> 1) & 7u) ^ (u32)khalf) << 4;     // chunk (2 kk + khalf) ^ swz = (2 kk) ^ (khalf ^ swz)
        }
    };
    constexpr int RPK = 2 + NPI;                          // fragment reads per k16 block: weights ci 0, 1 then positions pi 0 ..
    auto read_one = [&](auto setc, auto jc) {             // fragment read j of a step: k16 block j / RPK
        constexpr int S = decltype(setc)::value, j = decltype(jc)::value, kk = j / RPK, w = j % RPK;
        if constexpr (w < 2) fa[S][kk][w] = *reinterpret_cast<const bf16x8*>(lds + ra[kk] + w * 4096);
        else fb[S][kk][w - 2] = *reinterpret_cast<const bf16x8*>(lds + rb[w - 2] + (re[w - 2] ^ (u32)(kk << 5)));
    };
    auto mfma_one = [&](auto setc, auto ic) {             // MFMA i of a step: k16 block i / (2 NPI), accumulator (ci, pi) = ((i / NPI) & 1, i % NPI)
        constexpr int S = decltype(setc)::value, i = decltype(ic)::value, kk = i / (2 * NPI), ci = (i / NPI) & 1, pi = i % NPI;
        if constexpr (X3)
            acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ch_f16x8, fa[S][kk][ci]), __builtin_bit_cast(ch_f16x8, fb[S][kk][pi]),
                                                                 acc[ci][pi], 0, 0, 0);
        else
            acc[ci][pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][kk][ci], fb[S][kk][pi], acc[ci][pi], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    // ---- prologue of the workgroup's first tile: slab of slice 0, weights of steps 0 .. D-1 ------------------------------------
#pragma unroll
    for (int k = 0; k < NXO; ++k) ch_bload(xoff[k], rx, lds0 + SLAB0 + xdst(k), 0u);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int i = 0; i < NWP; ++i) issue_w_piece(co0, 0, d, d, i);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWP * (D - 2)) : "memory");   // slab 0 and the weights of steps 0 and 1 (this wave's share)
    __builtin_amdgcn_s_barrier();                        // ... everybody's
    read_addr(0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) fa[0][kk][ci] = *reinterpret_cast<const bf16x8*>(lds + ra[kk] + ci * 4096);
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) fb[0][kk][pi] = *reinterpret_cast<const bf16x8*>(lds + rb[pi] + (re[pi] ^ (u32)(kk << 5)));
    }

    int vbase = 0;                                       // slices this workgroup has finished (the weight ring runs on across tiles)
    bool has_next = false;
    int q0n = 0, co0n = 0;                               // the workgroup's next tile

    // One K-step; its tap is known at compile time, so the vmcnt immediates below are exact counts.  Wave-uniform flags: `nomore` =
    // the last slice of the workgroup's last tile (requests nothing beyond the tile), `nxt` = the last slice of any other tile (what
    // it requests beyond the slice belongs to the NEXT tile: its slice 0 slab through the refreshed xoff[], its weights at co0n).
    auto step = [&](auto grpc, auto setc, auto tapc, const int cs, const bool nomore, const bool nxt) {
        constexpr int GRP = decltype(grpc)::value, S = decltype(setc)::value, TAP = decltype(tapc)::value;
        using Sn = std::integral_constant<int, 1 - S>;
        // loads this wave issues during tap t of a slice: 2 weight pieces (+ 1 slab piece)
        constexpr auto issued = [](int t, bool lst) { return (lst ? (t < 9 - D ? 2 : 0) : 2 + (t < SPW ? 1 : 0)) * WPK; };
        constexpr int n_mid = (D == 3 ? 0 : (TAP > 0 ? issued(TAP - 1, false) : 2 * WPK)) + issued(TAP, false);
        constexpr int n_last = (D == 3 ? 0 : (TAP > 0 ? issued(TAP - 1, true) : 2 * WPK)) + issued(TAP, true);
        // Ring stage of a step is (global step number) mod NW: 9 slices-so-far + TAP -> TAP mod 3 for three stages, (slices + TAP) mod 4
        const int vs = vbase + cs;
        const bool post = cs == 0 && vbase > 0;
        const int st = NW == 3 ? (TAP + D) % 3 : ((vs + TAP + D) & 3);   // ring stage of step s + D
        // 16 slots, slot i = MFMA i, then 0..2 fragment reads of step s + 1, then at three slots one LDS-DMA request.  Nothing comes
        // in bursts: with 4 MFMAs, then 8 reads from all eight waves at once, the LDS queue fills up, the waves stall on issuing reads
        // and the MFMA pipe drains.  The step opens with an MFMA: hipcc puts an s_waitcnt lgkmcnt in front of the first use of a
        // register set (it cannot see that the previous step already waited), and that must not catch reads issued in this step.
        // After the very last step the reads fetch a stage nobody uses (in-bounds LDS addresses): cheaper than branches.
        constexpr int RSH = ((MODE & 64) && GRP) ? 2 : 0, QSH = ((MODE & 8) && GRP) ? 2 : 0;
        // (the addresses of this step's fragment reads are computed AFTER the step's first MFMA has issued -- in slot 0 below: eight
        // VALU operations in front of it left the matrix pipe idle at the head of every K-step, where VALU issue after a barrier
        // release is at its slowest (MI355X_MICROARCH.md, "start-of-segment VALU penalty"): 0.5-1.2 % per layer, r04w)
        auto addr_now = [&]() { read_addr(cs + (TAP == 8 ? 1 : 0), (TAP + 1) % 9, NW == 3 ? (TAP + 1) % 3 : ((vs + TAP + 1) & 3)); };
        // the weight piece `wi` of step s + D: the same slice, the next one, or slice 0 of the next tile
        auto req_w = [&](const int wi) {
            constexpr bool CARRY = TAP + D >= 9;
            if (!nomore || !CARRY) {
                const bool over = CARRY && nxt;
                issue_w_piece(over ? co0n : co0, over ? 0 : cs + (CARRY ? 1 : 0), (TAP + D) % 9, st, wi);
            }
        };
        auto req_x = [&](const int u) {                    // piece TAP WPK + u of the next slice's slab (TAP < SPW)
            if (!nomore)
                ch_bload(xoff[TAP * WPK + u], rx, lds0 + SLAB0 + ((cs + 1) & 1) * SLB + xdst(TAP * WPK + u),
                         nxt ? 0u : (u32)(((X3 && cs + 1 >= p.nx) ? cs + 1 - p.nx : cs + 1) * 128));
        };
        auto slot = [&](auto ic) {
            constexpr int i = decltype(ic)::value, j = i - RSH;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(MODE & 32)) mfma_one(setc, ic);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i == 0) {
                addr_now();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NWV == 8) {
                if constexpr (!(MODE & 2)) {
                    if constexpr (j >= 0 && j < 4) {
                        read_one(Sn{}, std::integral_constant<int, 2 * (j < 0 ? 0 : j)>{});
                        read_one(Sn{}, std::integral_constant<int, 2 * (j < 0 ? 0 : j) + 1>{});
                    } else if constexpr (j >= 4 && j < 12) {
                        read_one(Sn{}, std::integral_constant<int, (j < 4 ? 4 : j) + 4>{});
                    }
                }
                if constexpr (!(MODE & 1)) {
                    if constexpr (i == 1 + QSH) req_w(0);
                    else if constexpr (i == 5 + QSH) req_w(1);
                    else if constexpr (i == 9 + QSH && TAP < SPW) req_x(0);
                }
            } else {
                // 32 MFMAs, 24 fragment reads (one per slot), six requests (slots 2, 6, 10, 14: weights; 18, 22: slab)
                if constexpr (!(MODE & 2) && i < 24) read_one(Sn{}, std::integral_constant<int, (i < 24 ? i : 0)>{});
                if constexpr (!(MODE & 1)) {
                    if constexpr (i == 2) req_w(0);
                    else if constexpr (i == 6) req_w(1);
                    else if constexpr (i == 10) req_w(2);
                    else if constexpr (i == 14) req_w(3);
                    else if constexpr ((i == 18 || i == 22) && TAP < SPW) req_x(i == 18 ? 0 : 1);
                }
            }
        };
        slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{}); slot(std::integral_constant<int, 2>{});
        slot(std::integral_constant<int, 3>{}); slot(std::integral_constant<int, 4>{}); slot(std::integral_constant<int, 5>{});
        slot(std::integral_constant<int, 6>{}); slot(std::integral_constant<int, 7>{}); slot(std::integral_constant<int, 8>{});
        slot(std::integral_constant<int, 9>{}); slot(std::integral_constant<int, 10>{}); slot(std::integral_constant<int, 11>{});
        slot(std::integral_constant<int, 12>{}); slot(std::integral_constant<int, 13>{}); slot(std::integral_constant<int, 14>{});
        slot(std::integral_constant<int, 15>{});
        if constexpr (NWV == 4) {
            slot(std::integral_constant<int, 16>{}); slot(std::integral_constant<int, 17>{}); slot(std::integral_constant<int, 18>{});
            slot(std::integral_constant<int, 19>{}); slot(std::integral_constant<int, 20>{}); slot(std::integral_constant<int, 21>{});
            slot(std::integral_constant<int, 22>{}); slot(std::integral_constant<int, 23>{}); slot(std::integral_constant<int, 24>{});
            slot(std::integral_constant<int, 25>{}); slot(std::integral_constant<int, 26>{}); slot(std::integral_constant<int, 27>{});
            slot(std::integral_constant<int, 28>{}); slot(std::integral_constant<int, 29>{}); slot(std::integral_constant<int, 30>{});
            slot(std::integral_constant<int, 31>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        // everything step s + 2 needs has landed (in-order completion: only the newest D - 2 steps' requests may be in flight), this
        // wave's fragment reads are done (their stage is overwritten next step), then the barrier
        if constexpr (!(MODE & 4)) {
            // Right after an epilogue (`post`: the first slice of a tile that is not the workgroup's first) the wave's VMEM queue
            // reads [next-tile requests of the last slice | NST stores | the bias load | this tile's requests].  vmcnt counts loads
            // AND stores in issue order and a store is acknowledged microseconds after it was issued, while what steps 0 .. D-3 need
            // (the weights of steps 2 .. D-1) was requested BEFORE the stores: these steps let the stores and the bias load stay in
            // flight too.  From step D-2 on the awaited requests are younger than the stores and the plain counts apply.
            constexpr bool TOL = (MODE & 1024) != 0 && TAP < D - 2;
            if constexpr (LZ) {
                if (TOL && post) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((MODE & 1) ? 0 : n_mid + NST + 1) : "memory");
                else if (!nomore) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((MODE & 1) ? 0 : n_mid) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((MODE & 1) ? 0 : n_last) : "memory");
            } else {
            if (TOL && post) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((MODE & 1) ? 0 : n_mid + NST + 1) : "memory");
            else if (!nomore) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((MODE & 1) ? 0 : n_mid) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((MODE & 1) ? 0 : n_last) : "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    };
    auto slice = [&](auto grpc, auto parc, const int cs, const bool nomore, const bool nxt) {   // nine taps; register set of tap t: (par + t) & 1
        constexpr int P = decltype(parc)::value;
        using A = std::integral_constant<int, P>; using B = std::integral_constant<int, 1 - P>;
        step(grpc, A{}, std::integral_constant<int, 0>{}, cs, nomore, nxt); step(grpc, B{}, std::integral_constant<int, 1>{}, cs, nomore, nxt);
        step(grpc, A{}, std::integral_constant<int, 2>{}, cs, nomore, nxt); step(grpc, B{}, std::integral_constant<int, 3>{}, cs, nomore, nxt);
        step(grpc, A{}, std::integral_constant<int, 4>{}, cs, nomore, nxt); step(grpc, B{}, std::integral_constant<int, 5>{}, cs, nomore, nxt);
        step(grpc, A{}, std::integral_constant<int, 6>{}, cs, nomore, nxt); step(grpc, B{}, std::integral_constant<int, 7>{}, cs, nomore, nxt);
        step(grpc, A{}, std::integral_constant<int, 8>{}, cs, nomore, nxt);
    };
    auto k_loop = [&](auto grpc) {
        for (int cs = 0; cs < csteps; cs += 2) {         // csteps is even (Cin % 128 == 0): the last slice is an odd one
            const bool fin = cs + 2 >= csteps;
            slice(grpc, I0{}, cs, false, false);
            if (PERSIST && fin && has_next) make_xoff(q0n);   // the current tile's xoff[] was used for the last time in the slice above
            slice(grpc, I1{}, cs + 1, fin && !has_next, fin && has_next);
        }
    };

    CH_PROF_DECL
    for (;;) {
        if constexpr (PERSIST) has_next = tile_of(id + (int)gridDim.x, q0n, co0n);
        // The tile's bias: ONE dword per lane (lane l: channels 2 (l & 31), + 1 of the wave's 64), requested here -- ahead of the
        // tile's K loop in the VMEM queue, so the loop's counted waits retire it -- and handed round by ds_bpermute in the
        // epilogue.  (32 two-byte loads per lane IN the epilogue made every tile wait for a full memory round trip, and with it
        // for every request of the next tile already in flight: r03h, 10-47 us per layer.)  An asm load so that hipcc does not
        // drain vmcnt before its first use; without a bias the lanes read the filters (any valid address) and ignore the value.
        u32 bias_dw;
        {
            const bf16_t* bsrc = (p.bias ? p.bias + co0 + wm * 64 : p.w) + 2 * r31;
            asm volatile("global_load_dword %0, %1, off" : "=v"(bias_dw) : "v"(bsrc) : "memory");
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[ci][pi][v] = 0.f;
        if constexpr ((MODE & (8 | 64)) != 0) {          // the two waves of a SIMD run differently ordered code (same barriers)
            if (wave < 4) k_loop(I0{}); else k_loop(I1{});
        } else {
            k_loop(I0{});
        }

        // ---- epilogue: bias + ReLU + one rounding, transpose through LDS, 16-byte stores.  The stage is the slab buffer of the
        //      last (odd) slice: everything else in LDS may already hold the next tile's first slab and weights.  Two passes of 32
        //      positions per wave (4 KB, wave private: DS operations of one wave execute in order). ----------------------------------
        asm volatile("" : "+v"(bias_dw));                 // landed: the K loop's last counted wait is younger than the request
        CH_PROF_MARK(0)
        if constexpr ((MODE & 512) != 0) {                 // ablation: no epilogue (the accumulators only stay alive)
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
            if constexpr (NPI == 4) asm volatile("" :: "v"(acc[0][2]), "v"(acc[0][3]), "v"(acc[1][2]), "v"(acc[1][3]));
        } else {
        unsigned char* stage = lds + SLAB0 + SLB + wave * 4096;
        float bv[2][16];
        if (!p.bias) bias_dw = 0u;
        // ReLU as "v <= floor ? floor : v" with floor = +0 (NaN stays NaN, -0 -> +0), no activation as floor = -inf: one compare and one
        // select per value either way (a runtime `relu ? ... : v` costs a third, scalar, instruction per value)
        const float rfloor = p.relu ? 0.f : -__builtin_inff();
        if constexpr (X3) {
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bv[ci][4 * g + e] = p.bias32 ? p.bias32[co0 + wm * 64 + ci * 32 + 8 * g + 4 * khalf + e] : 0.f;
        } else
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // channels ci 32 + 8 g + 4 khalf + e, e = 0 .. 3: dwords ci 16 + 4 g + 2 khalf, + 1 (held by the lanes of that number)
                const int dw = ci * 16 + 4 * g + 2 * khalf;
                const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(dw * 4, (int)bias_dw);
                const u32 hi = (u32)__builtin_amdgcn_ds_bpermute(dw * 4 + 4, (int)bias_dw);
                bv[ci][4 * g + 0] = __uint_as_float(lo << 16);
                bv[ci][4 * g + 1] = __uint_as_float(lo & 0xffff0000u);
                bv[ci][4 * g + 2] = __uint_as_float(hi << 16);
                bv[ci][4 * g + 3] = __uint_as_float(hi & 0xffff0000u);
            }
        if constexpr (!X3) {
        // ---- bf16 epilogue (round 4: the instruction diet of DESIGN 8 item 2).  Per 4 values: 2 v_pk_add_f32 (bias), 2
        //      v_cvt_pk_bf16_f32, 2 v_pk_max_i16 (the activation on the rounded pair, see ch_pkmax_i16) -- instead of four compare +
        //      select pairs with their VCC hazard states; the store offsets of the 1-D form are computed once per tile, branch-free.
        const u32 floor16 = p.relu ? 0u : 0x80008000u;
        auto pack4 = [&](const int ci, const int pi, const int g, u32& p0, u32& p1) {
            const float s0 = acc[ci][pi][4 * g + 0] + bv[ci][4 * g + 0], s1 = acc[ci][pi][4 * g + 1] + bv[ci][4 * g + 1];
            const float s2 = acc[ci][pi][4 * g + 2] + bv[ci][4 * g + 2], s3 = acc[ci][pi][4 * g + 3] + bv[ci][4 * g + 3];
            p0 = ch_pkmax_i16(ch_pack2(s0, s1), floor16);
            p1 = ch_pkmax_i16(ch_pack2(s2, s3), floor16);
        };
        // DIRECT: a lane holds channels ci 32 + 8 g + 4 khalf + (0 .. 3) of its position as (lo[g], hi[g]); v_permlane32_swap between the
        // two lanes of a position (lane, lane + 32) leaves the khalf = 0 lane with the 16 bytes of g = 0 / 2 and the khalf = 1 lane with
        // those of g = 1 / 3: two 16-byte stores per (ci, position block), at off = the position's byte offset + 16 khalf (or OOB).
        auto store_runs = [&](const u32 (&lo)[4], const u32 (&hi)[4], const u32 off) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const auto sl = __builtin_amdgcn_permlane32_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
                const auto sh = __builtin_amdgcn_permlane32_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
                store16_at(make_uint4(sl[0], sh[0], sl[1], sh[1]), off + pr * 32);
            }
        };
        if constexpr (POOL && DIRECT) {
            // as the staged form below (rounded and activated first, pooled as 16-bit integers; float path without ReLU), the even lanes
            // store their pooled pixel straight from the registers
            static_assert(NPI == 2, "eight-wave form");
            int b, h0, w0;
            tile_origin(q0, b, h0, w0);
            const int pair = slot2 >> CSH, col = slot2 & (TC - 1);
            int hr = h0 + 2 * pair;                       // the pair's first row (stacked batch: of its image b)
            stack_row(b, hr);
            const int ho = hr >> 1, wo = (w0 + col) >> 1;
            const bool okp = (!(r31 & 1)) & (ho < p.Ho) & (wo < p.Wo);
            const u32 off = (((u32)((b * p.Ho + ho) * p.Wo + wo) * YC + (u32)(co0 + wm * 64)) * 2u + (u32)khalf * 16u) | (okp ? 0u : OOB);
            const u32 mrow1 = hr + 1 < H ? 0xffffffffu : 0u, mcol = w0 + col < W ? 0xffffffffu : 0u;
            const bool has_below = hr + 1 < H, has_right = w0 + col + 1 < W;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                u32 lo[4], hi[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (p.relu) {
                        u32 a0, a1, b0, b1;
                        pack4(ci, 0, g, a0, a1);
                        pack4(ci, 1, g, b0, b1);
                        u32 v0 = ch_pkmax_i16(a0, b0 & mrow1) & mcol, v1 = ch_pkmax_i16(a1, b1 & mrow1) & mcol;
                        lo[g] = ch_pkmax_i16(v0, (u32)__builtin_amdgcn_update_dpp(0, (int)v0, 0xB1, 0xf, 0xf, false));
                        hi[g] = ch_pkmax_i16(v1, (u32)__builtin_amdgcn_update_dpp(0, (int)v1, 0xB1, 0xf, 0xf, false));
                    } else {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[ci][0][4 * g + e];
                            const float below = acc[ci][1][4 * g + e];
                            if (has_below) v = below > v ? below : v;
                            const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
                            if (has_right) v = right > v ? right : v;
                            o[e] = v + bv[ci][4 * g + e];
                        }
                        lo[g] = ch_pack2(o[0], o[1]);
                        hi[g] = ch_pack2(o[2], o[3]);
                    }
                }
                store_runs(lo, hi, off + ci * 64);
            }
        } else
        if constexpr (POOL) {
            // Rounded and activated FIRST, pooled as 16-bit integers: after ReLU every value is non-negative, where integer order is
            // numeric order, so the 2 x 2 maximum is two v_pk_max_i16 per pair (vertical: the lane's two position blocks; horizontal:
            // lane ^ 1 by DPP) and exactly MaxPooling2D of the rounded activations.  Pixels outside the image (odd maps: the last row
            // pair / column pair) are masked to +0, the neutral element.  Without ReLU the float path below (rare: tests only).
            // (NWV = 4: two row-pair blocks ph per wave, position blocks 2 ph and 2 ph + 1, staged side by side: 2 KB each)
            int b, h0, w0;
            tile_origin(q0, b, h0, w0);
#pragma unroll
            for (int ph = 0; ph < NPI / 2; ++ph) {
            const int slot2p = slot2 + ph * 32;
            const int pair = slot2p >> CSH, col = slot2p & (TC - 1);
            unsigned char* pstage = stage + ph * 2048;
            int hr = h0 + 2 * pair, bp = b;               // the pair's first row (stacked batch: of its image bp)
            stack_row(bp, hr);
            if (p.relu) {
                const u32 mrow1 = hr + 1 < H ? 0xffffffffu : 0u, mcol = w0 + col < W ? 0xffffffffu : 0u;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32 a0, a1, b0, b1;
                        pack4(ci, 2 * ph, g, a0, a1);
                        pack4(ci, 2 * ph + 1, g, b0, b1);
                        u32 v0 = ch_pkmax_i16(a0, b0 & mrow1) & mcol, v1 = ch_pkmax_i16(a1, b1 & mrow1) & mcol;
                        v0 = ch_pkmax_i16(v0, (u32)__builtin_amdgcn_update_dpp(0, (int)v0, 0xB1, 0xf, 0xf, false));
                        v1 = ch_pkmax_i16(v1, (u32)__builtin_amdgcn_update_dpp(0, (int)v1, 0xB1, 0xf, 0xf, false));
                        if (!(r31 & 1)) {
                            const int px = r31 >> 1, chunk = ci * 4 + g;
                            *reinterpret_cast<uint2*>(pstage + px * 128 + ((chunk ^ (px & 7)) << 4) + khalf * 8) = make_uint2(v0, v1);
                        }
                    }
            } else {
                const bool has_below = hr + 1 < H, has_right = w0 + col + 1 < W;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[ci][2 * ph][4 * g + e];
                            const float below = acc[ci][2 * ph + 1][4 * g + e];
                            if (has_below) v = below > v ? below : v;
                            const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
                            if (has_right) v = right > v ? right : v;
                            o[e] = v + bv[ci][4 * g + e];
                        }
                        if (!(r31 & 1)) {
                            const int px = r31 >> 1, chunk = ci * 4 + g;
                            *reinterpret_cast<uint2*>(pstage + px * 128 + ((chunk ^ (px & 7)) << 4) + khalf * 8) =
                                make_uint2(ch_pack2(o[0], o[1]), ch_pack2(o[2], o[3]));
                        }
                    }
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ph = 0; ph < NPI / 2; ++ph)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = j * 64 + lane, px = idx >> 3, c = idx & 7;     // 16 pooled pixels x 8 chunks
                const int se = wn * (16 * NPI) + ph * 32 + 2 * px;              // the even lane's slot
                int hs = h0 + 2 * (se >> CSH), bs = b;
                stack_row(bs, hs);
                const int ho = hs >> 1, wo = (w0 + (se & (TC - 1))) >> 1;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + ph * 2048 + px * 128 + ((c ^ (px & 7)) << 4));
                store16(v, ho < p.Ho && wo < p.Wo && (!STK || bs < stackB), (u32)((bs * p.Ho + ho) * p.Wo + wo) * YC + (u32)(co0 + wm * 64 + c * 8));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KEEP) {
                // ---- the same accumulators once more, un-pooled: bias + one rounding + activation, the LDS transpose, 16-byte stores into
                //      y_full (the plain 2-D epilogue below; a row of the stacked batch finds its image as everywhere else) ----
                const __amdgpu_buffer_rsrc_t ryf = __builtin_amdgcn_make_buffer_rsrc(p.y_full, 0, p.yf_bytes, 0x00020000);
                const int c = lane & 7;
                const u32 cb = (u32)(co0 + wm * 64 + c * 8) * 2u;
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) {
                    u32 so[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int sl = wn * (16 * NPI) + (pi >> 1) * 32 + j * 8 + (lane >> 3);
                        int hh = h0 + 2 * (sl >> CSH) + (pi & 1), bb = b;
                        const int ww = w0 + (sl & (TC - 1));
                        stack_row(bb, hh);
                        so[j] = ((u32)((bb * H + hh) * W + ww) * (YC * 2u) + cb) | (((hh < H) & (ww < W) & (!STK || bb < stackB)) ? 0u : OOB);
                    }
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            u32 p0, p1;
                            pack4(ci, pi, g, p0, p1);
                            const int chunk = ci * 4 + g;
                            *reinterpret_cast<uint2*>(stage + r31 * 128 + ((chunk ^ (r31 & 7)) << 4) + khalf * 8) = make_uint2(p0, p1);
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int px = j * 8 + (lane >> 3);
                        const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 128 + ((c ^ (px & 7)) << 4));
                        const ch_u32x4 d = {v.x, v.y, v.z, v.w};
                        __builtin_amdgcn_raw_buffer_store_b128(d, ryf, so[j], 0, 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        } else if constexpr (DIRECT) {
            static_assert(NPI == 2, "eight-wave form");
            u32 soff1[NPI];                              // byte offset of the lane's position in block pi (+ its 16-byte column), or OOB
            {
                const u32 cb = (u32)(co0 + wm * 64) * 2u + (u32)khalf * 16u;
                if constexpr (G2) {
                    int b, h0, w0;
                    tile_origin(q0, b, h0, w0);
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi) {
                        const int sl = wn * (16 * NPI) + (pi >> 1) * 32 + r31;
                        const int hh = h0 + 2 * (sl >> CSH) + (pi & 1), ww = w0 + (sl & (TC - 1));
                        soff1[pi] = ((u32)((b * H + hh) * W + ww) * (YC * 2u) + cb) | (((hh < H) & (ww < W)) ? 0u : OOB);
                    }
                } else {
                    int q = q0 + wn * (32 * NPI) + r31;
                    asm volatile("" : "+v"(q));          // after the K loop: more live registers inside it would spill
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi) {
                        const int b = q / (H1 * W1);
                        const int r = q - b * (H1 * W1);
                        const int h = r / W1, w = r - h * W1;
                        bool ok = (w < W) & (h < H) & (q < p.Q);
                        u32 pix;
                        if constexpr (STR) {
                            const int sh = p.os - 1;
                            const int hh = h - p.ooff, ww = w - p.ooff;
                            const int ho = hh >> sh, wo = ww >> sh;
                            ok = ok & ((hh | ww) >= 0) & !((hh | ww) & sh) & (ho < p.Hs) & (wo < p.Ws);
                            pix = (u32)((b * p.Hs + ho) * p.Ws + wo);
                        } else {
                            pix = (u32)((b * H + h) * W + w);
                        }
                        soff1[pi] = (pix * (YC * 2u) + cb) | (ok ? 0u : OOB);
                        q += 32;
                    }
                }
            }
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    u32 lo[4], hi[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) pack4(ci, pi, g, lo[g], hi[g]);
                    store_runs(lo, hi, soff1[pi] + ci * 64);
                }
        } else {
            const int c = lane & 7;
            u32 soff[NPI][4];                            // byte offsets of the lane's 16-byte stores (OOB: nothing to store)
            {
                const u32 cb = (u32)(co0 + wm * 64 + c * 8) * 2u;
                if constexpr (G2) {
                    int b, h0, w0;
                    tile_origin(q0, b, h0, w0);
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int sl = wn * (16 * NPI) + (pi >> 1) * 32 + j * 8 + (lane >> 3);
                            const int hh = h0 + 2 * (sl >> CSH) + (pi & 1), ww = w0 + (sl & (TC - 1));
                            soff[pi][j] = ((u32)((b * H + hh) * W + ww) * (YC * 2u) + cb) | (((hh < H) & (ww < W)) ? 0u : OOB);
                        }
                } else {
                    int q = q0 + wn * (32 * NPI) + (lane >> 3);
                    asm volatile("" : "+v"(q));          // after the K loop: eight more live registers inside it would spill
                    int b = q / (H1 * W1);
                    const int r = q - b * (H1 * W1);
                    int h = r / W1, w = r - h * W1;
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bool ok = (w < W) & (h < H) & (q < p.Q);           // '&': no short-circuit branches
                            u32 pix;
                            if constexpr (STR) {
                                // strided / 'valid' forms keep the positions (ho os + ooff, wo os + ooff) of the 'same' result; os is 1 or 2
                                const int sh = p.os - 1;
                                const int hh = h - p.ooff, ww = w - p.ooff;
                                const int ho = hh >> sh, wo = ww >> sh;
                                ok = ok & ((hh | ww) >= 0) & !((hh | ww) & sh) & (ho < p.Hs) & (wo < p.Ws);
                                pix = (u32)((b * p.Hs + ho) * p.Ws + wo);
                            } else {
                                pix = (u32)((b * H + h) * W + w);
                            }
                            soff[pi][j] = (pix * (YC * 2u) + cb) | (ok ? 0u : OOB);   // y is below 2 GB: bit 31 puts the offset out of range
                            q += 8;
                            w += 8;
                            if (!SMALL || W1 >= 8) {       // at most one row wrap per step: branch-free selects, no divergent loop
                                const bool wr = w >= W1;
                                w -= wr ? W1 : 0;
                                h += wr ? 1 : 0;
                                const bool hr = h == H1;
                                h = hr ? 0 : h;
                                b += hr ? 1 : 0;
                            } else {
                                while (w >= W1) { w -= W1; if (++h == H1) { h = 0; ++b; } }
                            }
                        }
                }
            }
            // MSK: the activation at the lane's eight store addresses, requested before the packing and the LDS round trip below
            // (an address that stores nothing reads zeros); the loads are the compiler's, it waits for them where they are used
            [[maybe_unused]] ch_u32x4 mk[NPI][4];
            [[maybe_unused]] float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // MSK: sums of the lane's eight channels over its eight positions
            [[maybe_unused]] float4 bold0 = make_float4(0.f, 0.f, 0.f, 0.f), bold1 = bold0;  // what the wave's earlier tiles left in its row
            [[maybe_unused]] float* brow = nullptr;
            if constexpr (MSK) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.mask), 0, p.y_bytes, 0x00020000);
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[pi][j] = __builtin_amdgcn_raw_buffer_load_b128(rm, soff[pi][j], 0, 0);
                if (p.bsum && lane < 8) {
                    const int slot0 = first_id >> 3;
                    brow = p.bsum + (size_t)(4 * ((slot0 / p.n_tiles) * 8 + (first_id & 7)) + wn) * p.Cout + co0 + wm * 64 + lane * 8;
                    if (id != first_id) { bold0 = reinterpret_cast<const float4*>(brow)[0]; bold1 = reinterpret_cast<const float4*>(brow)[1]; }
                }
            }
#pragma unroll
            for (int pi = 0; pi < NPI; ++pi) {
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32 p0, p1;
                        pack4(ci, pi, g, p0, p1);
                        const int chunk = ci * 4 + g;
                        *reinterpret_cast<uint2*>(stage + r31 * 128 + ((chunk ^ (r31 & 7)) << 4) + khalf * 8) = make_uint2(p0, p1);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int px = j * 8 + (lane >> 3);                  // of this pass's 32 positions
                    uint4 v = *reinterpret_cast<const uint4*>(stage + px * 128 + ((c ^ (px & 7)) << 4));
                    if constexpr (MSK) {
                        v.x &= ch_keep2(mk[pi][j][0]); v.y &= ch_keep2(mk[pi][j][1]);
                        v.z &= ch_keep2(mk[pi][j][2]); v.w &= ch_keep2(mk[pi][j][3]);
                        // (a position that stores nothing read a mask of zeros: it adds zeros)
                        bs[0] += __uint_as_float(v.x << 16); bs[1] += __uint_as_float(v.x & 0xffff0000u);
                        bs[2] += __uint_as_float(v.y << 16); bs[3] += __uint_as_float(v.y & 0xffff0000u);
                        bs[4] += __uint_as_float(v.z << 16); bs[5] += __uint_as_float(v.z & 0xffff0000u);
                        bs[6] += __uint_as_float(v.w << 16); bs[7] += __uint_as_float(v.w & 0xffff0000u);
                    }
                    store16_at(v, soff[pi][j]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is rewritten by the next pass
            }
            if constexpr (MSK) {
                if (p.bsum) {
                    // the eight lanes that share a channel group (lane & 7) hold eight positions each: xor-tree over lane bits 3 .. 5
                    // (ds_bpermute; a fixed order), then lanes 0 .. 7 add the wave's 64 positions to their row
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        bs[q] += __shfl_xor(bs[q], 8);
                        bs[q] += __shfl_xor(bs[q], 16);
                        bs[q] += __shfl_xor(bs[q], 32);
                    }
                    if (lane < 8) {
                        reinterpret_cast<float4*>(brow)[0] = make_float4(bold0.x + bs[0], bold0.y + bs[1], bold0.z + bs[2], bold0.w + bs[3]);
                        reinterpret_cast<float4*>(brow)[1] = make_float4(bold1.x + bs[4], bold1.y + bs[5], bold1.z + bs[6], bold1.w + bs[7]);
                    }
                }
            }
        }
        } else
        if constexpr (POOL) {
            // 2 x 2 maximum in registers (vertical: the lane's two position blocks; horizontal: lane ^ 1 by DPP), THEN bias + ReLU +
            // one rounding -- all monotonic, so this equals pooling the rounded activations.  Even lanes hold the wave's 16 pooled
            // pixels; they go through the LDS transpose and leave as 16-byte stores.
            int b, h0, w0;
            tile_origin(q0, b, h0, w0);
            const int pair = slot2 >> CSH, col = slot2 & (TC - 1);
            int hr = h0 + 2 * pair, bp = b;               // the pair's first row (stacked batch: of its image bp)
            stack_row(bp, hr);
            const bool has_below = hr + 1 < H, has_right = w0 + col + 1 < W;
#pragma unroll
            for (int part = 0; part < (X3 ? 2 : 1); ++part) {             // X3: the hi parts, then the lo parts
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[ci][0][4 * g + e];
                        const float below = acc[ci][1][4 * g + e];
                        if (has_below) v = below > v ? below : v;
                        const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
                        if (has_right) v = right > v ? right : v;
                        v = X3 ? v * p.oscale + bv[ci][4 * g + e] : v + bv[ci][4 * g + e];
                        o[e] = v <= rfloor ? rfloor : v;
                    }
                    if (!(r31 & 1)) {
                        const int px = r31 >> 1, chunk = ci * 4 + g;
                        u32 p0, p1;
                        if constexpr (X3) {
                            u32 l0, l1;
                            const u32 h0_ = ch_split2(o[0], o[1], l0), h1_ = ch_split2(o[2], o[3], l1);
                            p0 = part ? l0 : h0_;
                            p1 = part ? l1 : h1_;
                        } else {
                            p0 = ch_pack2(o[0], o[1]);
                            p1 = ch_pack2(o[2], o[3]);
                        }
                        *reinterpret_cast<uint2*>(stage + px * 128 + ((chunk ^ (px & 7)) << 4) + khalf * 8) = make_uint2(p0, p1);
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = j * 64 + lane, px = idx >> 3, c = idx & 7;     // 16 pooled pixels x 8 chunks
                const int se = wn * 32 + 2 * px;                                // the even lane's slot
                int hs = h0 + 2 * (se >> CSH), bs = b;
                stack_row(bs, hs);
                const int ho = hs >> 1, wo = (w0 + (se & (TC - 1))) >> 1;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 128 + ((c ^ (px & 7)) << 4));
                store16(v, ho < p.Ho && wo < p.Wo && (!STK || bs < stackB), (u32)((bs * p.Ho + ho) * p.Wo + wo) * YC + (u32)(part * p.Cout + co0 + wm * 64 + c * 8));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            int q = 0, b = 0, h = 0, w = 0, h0 = 0, w0 = 0;
            if constexpr (G2) {
                tile_origin(q0, b, h0, w0);
            } else {
                q = q0 + wn * 64 + (lane >> 3);
                b = q / (H1 * W1);
                const int r = q - b * (H1 * W1);
                h = r / W1;
                w = r - h * W1;
            }
            const int c = lane & 7;
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
              const int q_s = q, b_s = b, h_s = h, w_s = w;             // X3 walks the pass's positions twice (hi parts, lo parts)
#pragma unroll
              for (int part = 0; part < (X3 ? 2 : 1); ++part) {
                q = q_s; b = b_s; h = h_s; w = w_s;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = X3 ? acc[ci][pi][4 * g + e] * p.oscale + bv[ci][4 * g + e] : acc[ci][pi][4 * g + e] + bv[ci][4 * g + e];
                            o[e] = v <= rfloor ? rfloor : v;
                        }
                        const int chunk = ci * 4 + g;
                        u32 p0, p1;
                        if constexpr (X3) {
                            u32 l0, l1;
                            const u32 h0_ = ch_split2(o[0], o[1], l0), h1_ = ch_split2(o[2], o[3], l1);
                            p0 = part ? l0 : h0_;
                            p1 = part ? l1 : h1_;
                        } else {
                            p0 = ch_pack2(o[0], o[1]);
                            p1 = ch_pack2(o[2], o[3]);
                        }
                        *reinterpret_cast<uint2*>(stage + r31 * 128 + ((chunk ^ (r31 & 7)) << 4) + khalf * 8) = make_uint2(p0, p1);
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int px = j * 8 + (lane >> 3);                  // of this pass's 32 positions
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + px * 128 + ((c ^ (px & 7)) << 4));
                    if constexpr (G2) {
                        const int sl = wn * 32 + px;
                        const int hh = h0 + 2 * (sl >> CSH) + pi, ww = w0 + (sl & (TC - 1));
                        store16(v, hh < H && ww < W, (u32)((b * H + hh) * W + ww) * YC + (u32)(part * p.Cout + co0 + wm * 64 + c * 8));
                    } else {
                        {
                            // strided / 'valid' forms keep the positions (ho os + ooff, wo os + ooff) of the 'same' result; os is 1 or 2:
                            // a shift and a parity test (an integer division by a runtime value is ~35 instructions, and there were
                            // sixteen of them per tile in this loop)
                            const int sh = p.os - 1;
                            const int hh = h - p.ooff, ww = w - p.ooff;
                            const int ho = hh >> sh, wo = ww >> sh;
                            const bool ok = w < W && h < H && q < p.Q && (hh | ww) >= 0 && !((hh | ww) & sh) && ho < p.Hs && wo < p.Ws;
                            store16(v, ok, (u32)((b * p.Hs + ho) * p.Ws + wo) * YC + (u32)(part * p.Cout + co0 + wm * 64 + c * 8));
                        }
                        q += 8;
                        w += 8;
                        if (!SMALL || W1 >= 8) {           // at most one row wrap per step: branch-free selects, no divergent loop
                            const bool wr = w >= W1;
                            w -= wr ? W1 : 0;
                            h += wr ? 1 : 0;
                            const bool hr = h == H1;
                            h = hr ? 0 : h;
                            b += hr ? 1 : 0;
                        } else {
                            while (w >= W1) { w -= W1; if (++h == H1) { h = 0; ++b; } }
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is rewritten by the next pass
              }
            }
        }
        }
        CH_PROF_MARK(1)
        CH_PROF_TILE
        if (!has_next) break;
        __builtin_amdgcn_s_barrier();                    // the stage is a slab buffer again: the next tile's slice 1 lands there
        CH_PROF_MARK(2)
        id += (int)gridDim.x;
        q0 = q0n;
        co0 = co0n;
        vbase += csteps;
    }
    if (wave == 0) { CH_PROF_FLUSH(0) }
    if (wave == 4) { CH_PROF_FLUSH(8) }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int NW, int SPW, int MODE, int CSH, bool POOL, bool X3 = false, int NWV = 8, bool STK = false, bool MSK = false, bool KEEP = false>
__global__ __launch_bounds__(64 * NWV) void convh_kernel(ConvHParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[ch_lds_bytes(NW, SPW)];
    convh_body<NW, SPW, MODE, CSH, POOL, SPW == 5, X3, NWV, STK, MSK, KEEP>(p, lds, (int)blockIdx.x);   // SPW == 5: maps up to 30 wide
#endif
}

// Several independent convolutions in ONE launch -- the packed predictor heads of all source maps (models/keras_ssd300.py:322-335):
// persistent workgroups take work items off one list, problems with the longest K loop first (the fc7 head walks 144 K-steps; started
// last it would be the tail of the launch).  Every problem runs on the padded position grid with the <4 stages, 6 pieces> layout
// (maps up to 62 wide); a tile is processed as in the one-workgroup-per-tile schedule (no prefetch across tiles of different problems).
constexpr int CH_MAX_GROUP = 8;
struct ConvHGroup {
    ConvHParams p[CH_MAX_GROUP];
    int first_id[CH_MAX_GROUP + 1];                      // work items of problem k: [first_id[k], first_id[k+1]), each count a multiple of 8
    int n;
};

template <int MODE>
__global__ __launch_bounds__(CH_THREADS) void convh_group_kernel(ConvHGroup g) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[ch_lds_bytes(4, 6)];
    // Work items are sorted by depth and dealt in SNAKE order (workgroup w takes items w, 2G-1-w, 2G+w, ...): the workgroups that
    // started with the deepest items (the fc7 head's 144 K-steps) get their second item last, if at all -- with the plain stride the
    // same workgroups took a long AND a medium item and the launch ended ~70 us after most CUs had gone idle.
    const int G = (int)gridDim.x, total = g.first_id[g.n];
    for (int round = 0;; ++round) {
        const int id = (round & 1) ? (round + 1) * G - 1 - (int)blockIdx.x : round * G + (int)blockIdx.x;
        if (round * G >= total) break;
        if (id >= total) continue;
        int k = 0;
        while (k + 1 < g.n && id >= g.first_id[k + 1]) ++k;
        convh_body<4, 6, MODE, 0, false, true>(g.p[k], lds, id - g.first_id[k]);
    }
#endif
}

// the schedules that exist with stacked-batch tiles (ConvHParams::Hp): the product's pooled forms -- bf16 with the tolerant waits, X3
constexpr bool convh_has_stk(int mode, bool x3) { return x3 ? mode == 128 : mode == 1152; }
// geom: 0 = padded position grid; 4 | 5 = 2-D tiles of 16 x 16 | 8 x 32 pixels
template <int MODE, bool X3 = false>
static void convh_launch(const ConvHParams& p, int geom, int pool, int n_cu, hipStream_t stream) {
    int grid = p.total_ids;
    if ((MODE & 128) && grid > n_cu) grid = n_cu;        // persistent: one workgroup per CU (a multiple of 8: the id -> XCD map)
    if constexpr ((MODE & 4096) != 0 && !X3) {           // four waves per workgroup (64 channels x 128 positions each)
        constexpr int M4 = MODE & ~4096;
        const dim3 g(grid), t(256);
        if (geom == 4) {
            if (pool) hipLaunchKernelGGL((convh_kernel<4, 6, M4, 4, true, false, 4>), g, t, 0, stream, p);
            else hipLaunchKernelGGL((convh_kernel<4, 6, M4, 4, false, false, 4>), g, t, 0, stream, p);
        } else if (geom == 5) {
            if (pool) hipLaunchKernelGGL((convh_kernel<4, 6, M4, 5, true, false, 4>), g, t, 0, stream, p);
            else hipLaunchKernelGGL((convh_kernel<4, 6, M4, 5, false, false, 4>), g, t, 0, stream, p);
        }
        else if (p.W <= 30) hipLaunchKernelGGL((convh_kernel<4, 5, M4, 0, false, false, 4>), g, t, 0, stream, p);
        else if (p.W <= 62) hipLaunchKernelGGL((convh_kernel<4, 6, M4, 0, false, false, 4>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<3, 7, M4, 0, false, false, 4>), g, t, 0, stream, p);
    } else {
    const dim3 g(grid), t(CH_THREADS);
    if constexpr ((MODE & 2048) != 0) {                  // the strided / cropped forms exist on the padded position grid only
        if (p.W <= 30) hipLaunchKernelGGL((convh_kernel<4, 5, MODE, 0, false, X3>), g, t, 0, stream, p);
        else if (p.W <= 62) hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 0, false, X3>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<3, 7, MODE, 0, false, X3>), g, t, 0, stream, p);
    } else
    if (geom == 4) {
        if constexpr (convh_has_stk(MODE, X3)) {
            if (pool && p.Hp) { hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 4, true, X3, 8, true>), g, t, 0, stream, p); return; }
        }
        if (pool) hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 4, true, X3>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 4, false, X3>), g, t, 0, stream, p);
    } else if (geom == 5) {
        if constexpr (convh_has_stk(MODE, X3)) {
            if (pool && p.Hp) { hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 5, true, X3, 8, true>), g, t, 0, stream, p); return; }
        }
        if (pool) hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 5, true, X3>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 5, false, X3>), g, t, 0, stream, p);
    }
    // slab rows = 256 + 2 W + 4 <= 64 SPW: three weight stages + 7 pieces per wave (W <= 94), four + 6 (W <= 62), four + 5 (W <= 30)
    else if (p.W <= 30) hipLaunchKernelGGL((convh_kernel<4, 5, MODE, 0, false, X3>), g, t, 0, stream, p);
    else if (p.W <= 62) hipLaunchKernelGGL((convh_kernel<4, 6, MODE, 0, false, X3>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((convh_kernel<3, 7, MODE, 0, false, X3>), g, t, 0, stream, p);
    }
}

// the pooled forms that also store the full-resolution map (ConvHParams::y_full): persistent workgroups, 2-D tiles per image or stacked
static void convh_launch_keep(const ConvHParams& p, int geom, int n_cu, hipStream_t stream) {
    int grid = p.total_ids;
    if (grid > n_cu) grid = n_cu;
    const dim3 g(grid), t(CH_THREADS);
    if (geom == 4) {
        if (p.Hp) hipLaunchKernelGGL((convh_kernel<4, 6, 128, 4, true, false, 8, true, false, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<4, 6, 128, 4, true, false, 8, false, false, true>), g, t, 0, stream, p);
    } else {
        if (p.Hp) hipLaunchKernelGGL((convh_kernel<4, 6, 128, 5, true, false, 8, true, false, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((convh_kernel<4, 6, 128, 5, true, false, 8, false, false, true>), g, t, 0, stream, p);
    }
}

// the masked-output forms (ConvHParams::mask): persistent workgroups, no pooling
// workgroups of a masked launch: one per CU, and grid / 8 a multiple of the channel tiles so that a workgroup keeps its channel tile
static int convh_masked_grid(int total_ids, int n_tiles, int n_cu) {
    int grid = total_ids < n_cu ? total_ids : n_cu;
    return (grid / (8 * n_tiles)) * (8 * n_tiles);       // (total_ids is a multiple of 8 n_tiles)
}
static void convh_launch_masked(const ConvHParams& p, int geom, int grid, hipStream_t stream) {
    const dim3 g(grid), t(CH_THREADS);
    if (geom == 4) hipLaunchKernelGGL((convh_kernel<4, 6, 128, 4, false, false, 8, false, true>), g, t, 0, stream, p);
    else if (geom == 5) hipLaunchKernelGGL((convh_kernel<4, 6, 128, 5, false, false, 8, false, true>), g, t, 0, stream, p);
    else if (p.W <= 30) hipLaunchKernelGGL((convh_kernel<4, 5, 128, 0, false, false, 8, false, true>), g, t, 0, stream, p);
    else if (p.W <= 62) hipLaunchKernelGGL((convh_kernel<4, 6, 128, 0, false, false, 8, false, true>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((convh_kernel<3, 7, 128, 0, false, false, 8, false, true>), g, t, 0, stream, p);
}

}  // namespace ssdhip

using namespace ssdhip;

static int convh_cu_count() {
    static int cu_count = 0;                              // persistent variants launch one workgroup per CU
    if (cu_count == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cu_count = (n / 8) * 8 > 0 ? (n / 8) * 8 : 8;
    }
    return cu_count;
}

// 2-D tiles: 16 x 16 or 8 x 32 pixels, per image or -- pooled forms only -- over the stacked batch (ConvHParams::Hp), whichever covers
// the batch with the fewest tiles (ties: per image, 16 x 16 first).  SSDHIP_CONVH_STACK=0 keeps the per-image tiles (A/B runs).
static bool convh_pick_2d(ConvHParams& p, int B, int H, int W, int pool, int& geom, bool has_stk) {
    long long best = -1;
    p.Hp = 0; p.nB = B; p.hp_magic = 0;
    for (int csh = 4; csh <= 5; ++csh) {
        const long long wt = (W + (1 << csh) - 1) >> csh, ht = (H + (256 >> csh) - 1) / (256 >> csh);
        if (best < 0 || (long long)B * wt * ht < best) { best = (long long)B * wt * ht; geom = csh; p.WT = (int)wt; p.HT = (int)ht; }
    }
    const char* e = getenv("SSDHIP_CONVH_STACK");
    if (pool && has_stk && !(e && atoi(e) == 0)) {
        const long long hp = H + 1 + ((H + 1) & 1);      // even, >= H + 1
        for (int csh = 4; csh <= 5; ++csh) {
            const long long tr = 256 >> csh, wt = (W + (1 << csh) - 1) >> csh, ht = ((long long)B * hp + tr - 1) / tr;
            // umulhi(R, floor(2^32 / hp) + 1) == R / hp for every row R the kernel asks about (R < (ht + 1) tr) while R hp < 2^32
            if (wt * ht < best && (ht + 1) * tr * hp < 0x7fffffffLL && tr + 2 <= hp) {      // (a tile's rows span at most two images)
                best = wt * ht; geom = csh; p.WT = (int)wt; p.HT = (int)ht;
                p.Hp = (int)hp; p.hp_magic = (unsigned)(0x100000000ULL / (unsigned long long)hp) + 1u;
            }
        }
    }
    if (best > 0x3fffff00LL) return false;
    p.Q = 0;
    p.q_tiles = (int)best;
    return true;
}

// An un-pooled map up to 94 wide runs on the padded position grid -- unless 2-D tiles finish in FEWER ROUNDS of one workgroup per CU
// (fourth session of round 6): SSD512's conv4_x at batch 16 is 265 position tiles x 4 channel tiles = 1 060 units = five rounds of 256 CUs
// on the grid and 256 x 4 = 1 024 = exactly four on 16 x 16 tiles, its conv5_x 276 units (two rounds) against 256 (one).  SSD300's maps
// (75, 38 wide) stay on the grid.  Fills the tile fields either way; false: sizes beyond the index range.  SSDHIP_CONVH_GRID=1 keeps the grid.
static bool convh_plan_unpooled(ConvHParams& p, int B, int H, int W, int n_tiles, int& geom) {
    geom = 0;
    p.HT = p.WT = 0;
    p.Hp = 0;
    if (W > 94) return convh_pick_2d(p, B, H, W, 0, geom, false);
    const long long Q = (long long)B * (H + 1) * (W + 1);
    if (Q > 0x3fffff00LL) return false;
    const long long grid_tiles = (Q + CH_BN - 1) / CH_BN;
    const char* e = getenv("SSDHIP_CONVH_GRID");
    if (!(e && atoi(e) == 1)) {
        int g2 = 0;
        if (convh_pick_2d(p, B, H, W, 0, g2, false)) {
            const long long n_cu = convh_cu_count();
            const long long r_grid = (((grid_tiles + 7) / 8) * 8 * n_tiles + n_cu - 1) / n_cu;
            const long long r_2d = ((((long long)p.q_tiles + 7) / 8) * 8 * n_tiles + n_cu - 1) / n_cu;
            if (r_2d < r_grid) { geom = g2; return true; }
        }
    }
    p.HT = p.WT = 0;
    p.Hp = 0;
    p.Q = (int)Q;
    p.q_tiles = (int)grid_tiles;
    return true;
}

// The tiling ssdhip_conv3x3_halo_nhwc_bf16 / ssdhip_conv3x3_halo_x3_nhwc_f16 pick for a (batch, map, pool) -- host arithmetic only, no
// launch: plan[0] = 0 (padded position grid) | 4 (16 x 16 pixel tiles) | 5 (8 x 32), plan[1] = position tiles (x Cout / 128 = tile
// units), plan[2] = the row pitch of the stacked batch (0: tiles per image), plan[3] = rows of tiles (per image, or of the stack).
// (Cout: an un-pooled call's choice between grid and 2-D tiles counts rounds of tile units.)
extern "C" int ssdhip_conv3x3_halo_plan(int B, int H, int W, int Cout, int pool, int* plan) {
    if (!plan || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
    ConvHParams p;
    int geom = 0;
    if (pool) {
        if (!convh_pick_2d(p, B, H, W, pool, geom, true)) return SSDHIP_E_BADARG;
    } else if (!convh_plan_unpooled(p, B, H, W, Cout / CH_BM, geom)) return SSDHIP_E_BADARG;
    plan[0] = geom; plan[1] = p.q_tiles; plan[2] = p.Hp; plan[3] = p.HT;
    return SSDHIP_OK;
}

// 3x3 convolution with stride 1 | 2 and zero padding 0 | 1 (torch.nn.Conv2d semantics) on a map up to 94 wide: the SSD extra layers
// conv6_2 / conv7_2 (ZeroPadding2D(1) + stride 2, models/keras_ssd300.py:302-307) and conv8_2 / conv9_2 ('valid', :310-313).  The
// slab kernel computes the stride-1 'same' result on the position grid and keeps the positions the strided / cropped form asks for:
// up to 4x redundant MFMA work, on layers whose cost is the latency of a K loop one workgroup deep, not FLOPs.
extern "C" int ssdhip_conv3x3_halo_strided_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                                     int Cin, int Cout, int stride, int pad, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || W > 94) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % 128) || Cout <= 0 || (Cout % CH_BM) || (stride != 1 && stride != 2) || (pad != 0 && pad != 1)) return SSDHIP_E_BADARG;
    if (H + 2 * pad < 3 || W + 2 * pad < 3) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * Cin * 2, wb = (long long)Cout * 9 * Cin * 2, Q = (long long)B * (H + 1) * (W + 1);
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout * 2 >= 0x7ffff000LL || Q > 0x3fffff00LL) return SSDHIP_E_BADARG;
    ConvHParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2; p.HT = p.WT = 0;
    p.os = stride; p.ooff = 1 - pad;
    p.Hs = (H + 2 * pad - 3) / stride + 1; p.Ws = (W + 2 * pad - 3) / stride + 1;
    p.Q = (int)Q;
    p.q_tiles = (int)((Q + CH_BN - 1) / CH_BN);
    p.n_tiles = Cout / CH_BM;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)((long long)B * p.Hs * p.Ws * Cout * 2);
    p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
    convh_launch<128 | 2048>(p, 0, 0, convh_cu_count(), stream);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// n_problems (<= 8) independent 3x3 'same' convolutions (no pooling; Cin % 128 == 0, Cout % 128 == 0, maps up to 62 wide) in one
// launch; arrays are HOST arrays of per-problem arguments; max_workgroups > 0 caps the persistent workgroups (one per CU) so that a
// concurrent stream finds free CUs.  Results are bit-identical to the single-problem entry.
extern "C" int ssdhip_conv3x3_halo_group_nhwc_bf16(int n_problems, const void* const* x_h, const void* const* weight_h,
                                                   const void* const* bias_h, void* const* y_h, const int* B_h, const int* H_h,
                                                   const int* W_h, const int* Cin_h, const int* Cout_h, int relu, int max_workgroups,
                                                   void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_problems < 1 || n_problems > CH_MAX_GROUP || !x_h || !weight_h || !y_h || !B_h || !H_h || !W_h || !Cin_h || !Cout_h) return SSDHIP_E_BADARG;
    int order[CH_MAX_GROUP];
    for (int k = 0; k < n_problems; ++k) order[k] = k;
    for (int i = 1; i < n_problems; ++i)                  // deepest K loop first (stable insertion sort)
        for (int j = i; j > 0 && Cin_h[order[j]] > Cin_h[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    ConvHGroup g;
    long long ids = 0;
    for (int s = 0; s < n_problems; ++s) {
        const int k = order[s];
        const int B = B_h[k], H = H_h[k], W = W_h[k], Cin = Cin_h[k], Cout = Cout_h[k];
        const void* x = x_h[k]; const void* weight = weight_h[k]; const void* bias = bias_h ? bias_h[k] : nullptr; void* y = y_h[k];
        if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || W > 62) return SSDHIP_E_BADARG;
        if (Cin <= 0 || (Cin % 128) || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
        if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
        const long long xb = (long long)B * H * W * Cin * 2, wb = (long long)Cout * 9 * Cin * 2, Q = (long long)B * (H + 1) * (W + 1);
        if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout * 2 >= 0x7ffff000LL || Q > 0x3fffff00LL) return SSDHIP_E_BADARG;
        ConvHParams& p = g.p[s];
        p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
        p.y = static_cast<bf16_t*>(y);
        p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu ? 1 : 0;
        p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2; p.HT = p.WT = 0;
        p.os = 1; p.ooff = 0; p.Hs = H; p.Ws = W;
        p.Q = (int)Q;
        p.q_tiles = (int)((Q + CH_BN - 1) / CH_BN);
        p.n_tiles = Cout / CH_BM;
        p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)((long long)B * H * W * Cout * 2);
        p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
        g.first_id[s] = (int)ids;
        ids += p.total_ids;
        if (ids > 0x3fffffffLL) return SSDHIP_E_BADARG;
    }
    for (int s = n_problems; s <= CH_MAX_GROUP; ++s) g.first_id[s] = (int)ids;
    for (int s = n_problems; s < CH_MAX_GROUP; ++s) g.p[s] = g.p[0];
    g.n = n_problems;
    int grid = (int)ids;
    if (grid > convh_cu_count()) grid = convh_cu_count();
    if (max_workgroups >= 8 && grid > (max_workgroups / 8) * 8) grid = (max_workgroups / 8) * 8;   // leave CUs to a concurrent stream
    hipLaunchKernelGGL((convh_group_kernel<0>), dim3(grid), dim3(CH_THREADS), 0, stream, g);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// The reference-precision form of ssdhip_conv3x3_halo_nhwc_bf16 (see ssdhip_conv2d_x3_nhwc_f16 for the arithmetic): x [B,H,W,2C]
// float16 = [hi | lo], weight [Cout,3,3,3C] float16 = [w hi | w lo | w hi] of the float32 filters / oscale, bias float32, y
// [B,Ho,Wo,2 Cout] float16 = [hi | lo] of act(oscale * sum + bias).  C % 128 == 0, Cout % 128 == 0.
// C == 64 (conv2_1): the K loop walks 64-channel slices in pairs, and 3 C is three of them -- the filters are then [Cout,3,3,256] =
// [w hi | w hi | w lo | 0] against the activation slices (hi, lo, hi, lo): the same three products plus a slice of zeros.
extern "C" int ssdhip_conv3x3_halo_x3_nhwc_f16(const void* x, const void* weight, const float* bias, void* y, int B, int H, int W, int C,
                                               int Cout, int relu, int pool, float oscale, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0 || !(oscale > 0.f)) return SSDHIP_E_BADARG;
    const bool c64 = C == 64;
    if (C <= 0 || ((C % 128) && !c64) || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 3)) return SSDHIP_E_BADARG;
    const int Kc = c64 ? 256 : 3 * C;                   // channels the K loop walks
    const long long xb = (long long)B * H * W * 2 * C * 2, wb = (long long)Cout * 9 * Kc * 2;
    const int Ho = pool ? (H + 1) / 2 : H, Wo = pool ? (W + 1) / 2 : W;
    const long long yb = (long long)B * Ho * Wo * 2 * Cout * 2;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || yb >= 0x7ffff000LL) return SSDHIP_E_BADARG;   // 31-bit byte offsets
    ConvHParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = nullptr;
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Kc; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.HT = p.WT = 0;
    p.os = 1; p.ooff = 0; p.Hs = H; p.Ws = W;
    p.xC = 2 * C; p.nx = c64 ? 2 : C / 64; p.bias32 = bias; p.oscale = oscale;
    int geom = 0;
    if (pool) {
        if (!convh_pick_2d(p, B, H, W, pool, geom, true)) return SSDHIP_E_BADARG;
    } else if (!convh_plan_unpooled(p, B, H, W, Cout / CH_BM, geom)) return SSDHIP_E_BADARG;
    p.n_tiles = Cout / CH_BM;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)yb;
    p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
    convh_launch<128, true>(p, geom, pool, convh_cu_count(), stream);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// y[b,h,w,co] = act(bias[co] + sum_{kh,kw,ci} x[b, h + kh - 1, w + kw - 1, ci] * w[co,kh,kw,ci]), zero padding: the 3x3 'same'
// convolutions of the VGG blocks with Cin % 128 == 0 and Cout % 128 == 0; pool != 0: MaxPooling2D(2, 2, 'same') fused, y is
// [B, ceil(H/2), ceil(W/2), Cout].  Maps up to 94 wide without pooling run on the padded position grid, everything else on 2-D
// tiles.  SSDHIP_E_BADARG for other channel counts (callers fall back to ssdhip_conv2d_same[_pool2]_nhwc_bf16: bit-identical results).
extern "C" int ssdhip_conv3x3_halo_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y, int B, int H, int W,
                                             int Cin, int Cout, int relu, int pool, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y || B <= 0 || H <= 0 || W <= 0) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % 128) || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * Cin * 2, wb = (long long)Cout * 9 * Cin * 2;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || (long long)B * H * W * Cout * 2 >= 0x7ffff000LL) return SSDHIP_E_BADARG;   // 31-bit byte offsets
    ConvHParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.HT = p.WT = 0;
    p.os = 1; p.ooff = 0; p.Hs = H; p.Ws = W;
    int mode = pool ? 1152 : 128;                         // the schedule: see the switch below
    if (const char* e = getenv("SSDHIP_CONVH_MODE")) mode = atoi(e);
    int geom = 0;
    if (pool) {
        if (!convh_pick_2d(p, B, H, W, pool, geom, mode == 1152)) return SSDHIP_E_BADARG;
    } else if (!convh_plan_unpooled(p, B, H, W, Cout / CH_BM, geom)) return SSDHIP_E_BADARG;
    p.n_tiles = Cout / CH_BM;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)(pool ? (long long)B * p.Ho * p.Wo * Cout * 2 : (long long)B * H * W * Cout * 2);
    p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
    const int cu_count = convh_cu_count();
    // Schedules: 128 = persistent workgroups, one per CU, that request the next tile's first slab and weights during the last slice
    // of the current tile; 1152 = the same with the tolerant waits after an epilogue.  SSDHIP_CONVH_MODE selects.  (The round-2
    // schedule with one workgroup per tile is gone: it bought nothing over 128 and its <3, 7> variant spilled registers.)
    // r03i / r03j / r03k A/Bs: the tolerant waits are worth 5-7 % on the pooled tiles (2 stores per wave and tile: the whole burst fits
    // the tolerance) and nothing measurable on the plain ones
#if defined(SSDHIP_PROFILE)
    if (const char* e = getenv("SSDHIP_CONVH_WAVES")) { if (atoi(e) == 4) mode |= 4096; }
    if (const char* e = getenv("SSDHIP_CONVH_LAZY")) { if (atoi(e) == 1) mode |= 8192; }
#endif
    switch (mode) {
        case 1152: convh_launch<1152>(p, geom, pool, cu_count, stream); break;                             // 128 + 1024: the first waits after an epilogue let its stores stay in flight
#if defined(SSDHIP_PROFILE)
        // Round-4 experiments, bit-identical to the product modes and tested as such (tests/test_conv_gpu.py with SSDHIP_LIB pointing at
        // the profiling build), measured and NOT adopted (profiles/r04bc_slab_four_waves_and_lazy_lds_waits_negative.json):
        // + 4096: four waves per workgroup (a quarter less LDS read traffic per FLOP, one wave per SIMD): 2-7 % SLOWER on every layer;
        // + 8192: requests three steps ahead, fragment reads in flight across the barrier: equal within 0.5 %.
        case 4224: convh_launch<128 | 4096>(p, geom, pool, cu_count, stream); break;
        case 5248: convh_launch<1152 | 4096>(p, geom, pool, cu_count, stream); break;
        case 8320: convh_launch<128 | 8192>(p, geom, pool, cu_count, stream); break;
        case 9344: convh_launch<1152 | 8192>(p, geom, pool, cu_count, stream); break;
        case 12416: convh_launch<128 | 8192 | 4096>(p, geom, pool, cu_count, stream); break;
        // + 16384: the epilogue stores from the accumulator layout (v_permlane32_swap, 32-byte runs per position and instruction) instead
        // of transposing through LDS for 128-byte runs -- what paid on the one-wave-per-SIMD Cin = 64 kernels (ssdhip_conv64.hip) is
        // 1-3 % SLOWER here on every layer (profiles/r04p8_slab_direct_epilogue_stores_negative.txt): with two waves per SIMD the LDS
        // round trips are covered, and the four-times-more cache lines per store instruction are not.
        case 16512: convh_launch<128 | 16384>(p, geom, pool, cu_count, stream); break;
        case 17536: convh_launch<1152 | 16384>(p, geom, pool, cu_count, stream); break;
        // ablations (wrong results by construction): tools/ablate_convh.py, tools/ablate_convh2.py
        case 129: convh_launch<129>(p, geom, pool, cu_count, stream); break;                               // no loads in the K loop
        case 130: convh_launch<130>(p, geom, pool, cu_count, stream); break;                               // no fragment reads
        case 135: convh_launch<135>(p, geom, pool, cu_count, stream); break;                               // MFMAs only
        case 384: convh_launch<384>(p, geom, pool, cu_count, stream); break;                               // the epilogue without its global stores
        case 640: convh_launch<640>(p, geom, pool, cu_count, stream); break;                               // no epilogue at all
#endif
        default: convh_launch<128>(p, geom, pool, cu_count, stream); break;
    }
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

#ifdef SSDHIP_PROFILE
// profiling build only: out[0..2] = wave 0's cycles in (K loop, epilogue, end-of-tile barrier), out[4] its tiles; out[8..12] the same for wave 4
extern "C" int ssdhip_profile_read_convh(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_profh), sizeof(g_profh)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_profh), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

// Tiling of a masked call: fills p.{HT, WT, Q, q_tiles, n_tiles, total_ids} and geom; false: sizes beyond the kernel's index range
static bool convh_masked_plan(ConvHParams& p, int B, int H, int W, int Cout, int& geom) {
    if (!convh_plan_unpooled(p, B, H, W, Cout / CH_BM, geom)) return false;
    p.n_tiles = Cout / CH_BM;
    p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
    return true;
}

// rows of the channel-sum table a masked call fills ([rows][Cout] float32; 0: geometry not supported)
extern "C" int ssdhip_conv3x3_halo_masked_bias_rows(int B, int H, int W, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout % CH_BM)) return 0;
    ConvHParams p;
    int geom;
    if (!convh_masked_plan(p, B, H, W, Cout, geom)) return 0;
    const int grid = convh_masked_grid(p.total_ids, p.n_tiles, convh_cu_count());
    return grid > 0 ? 4 * grid / p.n_tiles : 0;
}

// y = the 3x3 'same' convolution of x (no bias, no activation) where mask > 0 (or NaN), zero elsewhere: the data gradient of a layer
// whose input is the ReLU output `mask` [B, H, W, Cout] of the layer below, with that layer's threshold_backward folded into the
// epilogue (the training step: conv2_2 / conv3_2 / conv3_3 / conv4_2 / conv4_3 of models/keras_ssd300.py:279-291 towards the layer
// under them).  Bit-identical to ssdhip_conv3x3_halo_nhwc_bf16 followed by ssdhip_relu_bwd_bias_nhwc_bf16's mask.
// bias_partial (or NULL): [bias_rows][Cout] float32, bias_rows = ssdhip_conv3x3_halo_masked_bias_rows(B, H, W, Cout) -- every entry is
// written; the column sums are the channel sums of y (the bias gradient of the layer below), added in a fixed order.
extern "C" int ssdhip_conv3x3_halo_masked_nhwc_bf16(const void* x, const void* weight, const void* mask, void* y, float* bias_partial,
                                                    int bias_rows, int B, int H, int W, int Cin, int Cout, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !mask || !y || B <= 0 || H <= 0 || W <= 0) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % 128) || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)mask | (uintptr_t)bias_partial) & 15) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * Cin * 2, wb = (long long)Cout * 9 * Cin * 2, yb = (long long)B * H * W * Cout * 2;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || yb >= 0x7ffff000LL) return SSDHIP_E_BADARG;   // 31-bit byte offsets
    ConvHParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = nullptr;
    p.y = static_cast<bf16_t*>(y); p.mask = static_cast<const bf16_t*>(mask); p.bsum = bias_partial;
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.os = 1; p.ooff = 0; p.Hs = H; p.Ws = W;
    p.xC = 0; p.nx = 0; p.bias32 = nullptr; p.oscale = 1.f;
    int geom = 0;
    if (!convh_masked_plan(p, B, H, W, Cout, geom)) return SSDHIP_E_BADARG;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)yb;
    const int grid = convh_masked_grid(p.total_ids, p.n_tiles, convh_cu_count());
    if (grid <= 0 || (bias_partial && bias_rows != 4 * grid / p.n_tiles)) return SSDHIP_E_BADARG;
    convh_launch_masked(p, geom, grid, stream);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

// Conv2D(relu) -> MaxPooling2D(2, 2, 'same') of the training step in ONE launch that writes the activation the backward pass needs
// (y_full [B, H, W, Cout]) AND the pooled map (y_pooled [B, ceil(H/2), ceil(W/2), Cout]): models/keras_ssd300.py:279-287 (conv2_2 ->
// pool2, conv3_3 -> pool3).  Cin % 128 == 0, Cout % 128 == 0.  Both maps bit-identical to ssdhip_conv3x3_halo_nhwc_bf16 (pool = 0 / 1).
extern "C" int ssdhip_conv3x3_halo_pool_keep_nhwc_bf16(const void* x, const void* weight, const void* bias, void* y_full, void* y_pooled,
                                                       int B, int H, int W, int Cin, int Cout, int relu, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!x || !weight || !y_full || !y_pooled || B <= 0 || H <= 0 || W <= 0) return SSDHIP_E_BADARG;
    if (Cin <= 0 || (Cin % 128) || Cout <= 0 || (Cout % CH_BM)) return SSDHIP_E_BADARG;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y_full | (uintptr_t)y_pooled) & 15 || ((uintptr_t)bias & 1)) return SSDHIP_E_BADARG;
    const long long xb = (long long)B * H * W * Cin * 2, wb = (long long)Cout * 9 * Cin * 2, yfb = (long long)B * H * W * Cout * 2;
    if (xb >= 0x7ffff000LL || wb >= 0x7ffff000LL || yfb >= 0x7ffff000LL) return SSDHIP_E_BADARG;   // 31-bit byte offsets
    ConvHParams p;
    p.x = static_cast<const bf16_t*>(x); p.w = static_cast<const bf16_t*>(weight); p.bias = static_cast<const bf16_t*>(bias);
    p.y = static_cast<bf16_t*>(y_pooled); p.y_full = static_cast<bf16_t*>(y_full);
    p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu ? 1 : 0;
    p.Ho = (H + 1) / 2; p.Wo = (W + 1) / 2;
    p.HT = p.WT = 0;
    p.os = 1; p.ooff = 0; p.Hs = H; p.Ws = W;
    p.xC = 0; p.nx = 0; p.bias32 = nullptr; p.oscale = 1.f;
    int geom = 0;
    if (!convh_pick_2d(p, B, H, W, 1, geom, true)) return SSDHIP_E_BADARG;
    p.n_tiles = Cout / CH_BM;
    p.x_bytes = (int)xb; p.w_bytes = (int)wb; p.y_bytes = (int)((long long)B * p.Ho * p.Wo * Cout * 2); p.yf_bytes = (int)yfb;
    p.total_ids = ((p.q_tiles + 7) / 8) * p.n_tiles * 8;
    convh_launch_keep(p, geom, convh_cu_count(), stream);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}
