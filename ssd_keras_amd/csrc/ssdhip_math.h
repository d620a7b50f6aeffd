// Device-side leaf math shared by the decode / encode / loss kernels (gfx950 only).
// Compiled with -ffp-contract=off: every + - * / below is one IEEE-754 operation, in the
// order written, so results are reproducible against the CPU oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssdhip {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned __int128 u128;

// Zero `bytes` (a multiple of 4) of device memory on `stream` with a plain KERNEL instead of hipMemsetAsync.  Inside a captured HIP
// graph the runtime's memset node is not reliable on this stack (ROCm 7.2, graph packet capture on -- the default): replayed after a
// device-wide synchronize, the second and later replays of SSDLoss saw histograms that had NOT been cleared (loss 754 instead of 18.7,
// the same value on every replay; tools/debug_loss_graph.py, profiles/r04l_loss_graph_memset_node.txt; correct with
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) -- the "diverging" graph-replayed training leg of rounds 2 and 3.  A kernel node has no such
// problem, and is cheaper than the runtime's fill kernel (4.3 us in the step timeline).
static __global__ void zero_words_kernel(u32* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<u32*>(p), n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// float32 exp through a fixed chain of float64 adds/multiplies (the reference decodes box
// sizes with np.exp on float32, whose SIMD implementation is neither correctly rounded nor
// the same on every host; this one is <= 0.5000001 ulp and identical everywhere).
//   k = rint(x*log2 e);  r = x - k*ln2_hi - k*ln2_lo;  p = sum_{n<=13} r^n/n! (Horner);
//   result = (float)(p * 2^k)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float det_expf(float xf) {
    if (xf != xf) return xf;
    double x = (double)xf;
    x = x < -104.0 ? -104.0 : x;
    x = x > 89.0 ? 89.0 : x;
    const double t = x * 0x1.71547652b82fep+0 + 0x1.8p52;
    const double k = t - 0x1.8p52;
    const double r = (x - k * 0x1.62e42fee00000p-1) - k * 0x1.a39ef35793c76p-33;
    double p = 0x1.6124613a86d09p-33;
    p = p * r + 0x1.1eed8eff8d898p-29;
    p = p * r + 0x1.ae64567f544e4p-26;
    p = p * r + 0x1.27e4fb7789f5cp-22;
    p = p * r + 0x1.71de3a556c734p-19;
    p = p * r + 0x1.a01a01a01a01ap-16;
    p = p * r + 0x1.a01a01a01a01ap-13;
    p = p * r + 0x1.6c16c16c16c17p-10;
    p = p * r + 0x1.1111111111111p-7;
    p = p * r + 0x1.5555555555555p-5;
    p = p * r + 0x1.5555555555555p-3;
    p = p * r + 0x1.0000000000000p-1;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const long long ki = (long long)k;
    const double scale = __longlong_as_double((ki + 1023LL) << 52);
    return (float)(p * scale);
}

// float64 exp for the decoders' float64 flow (float64 predictions): the same chain without the final narrowing,
// scaled by ldexp (one rounding, also for subnormal results).  <= 1 ulp; the CPU checker repeats the chain verbatim.
__device__ __forceinline__ double det_exp64(double x) {
    if (x != x) return x;
    x = x < -746.0 ? -746.0 : x;
    x = x > 710.0 ? 710.0 : x;
    const double t = x * 0x1.71547652b82fep+0 + 0x1.8p52;
    const double k = t - 0x1.8p52;
    const double r = (x - k * 0x1.62e42fee00000p-1) - k * 0x1.a39ef35793c76p-33;
    double p = 0x1.6124613a86d09p-33;
    p = p * r + 0x1.1eed8eff8d898p-29;
    p = p * r + 0x1.ae64567f544e4p-26;
    p = p * r + 0x1.27e4fb7789f5cp-22;
    p = p * r + 0x1.71de3a556c734p-19;
    p = p * r + 0x1.a01a01a01a01ap-16;
    p = p * r + 0x1.a01a01a01a01ap-13;
    p = p * r + 0x1.6c16c16c16c17p-10;
    p = p * r + 0x1.1111111111111p-7;
    p = p * r + 0x1.5555555555555p-5;
    p = p * r + 0x1.5555555555555p-3;
    p = p * r + 0x1.0000000000000p-1;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return ldexp(p, (int)k);
}

// ---------------------------------------------------------------------------------------
// IoU of two 'corners' boxes exactly as bounding_box_utils.py:283-383 evaluates it in
// 'element-wise' mode: intersection with d = 0 (the reference forgets to forward
// border_pixels, :345), areas with d, union = (area_a + area_b) - inter, IEEE division.
// ---------------------------------------------------------------------------------------
template <typename F>
struct PxBox {
    F x0, y0, x1, y1, area;
};

template <typename F>
__device__ __forceinline__ F clamp0(F v) { return v < (F)0 ? (F)0 : v; }   // np.maximum(0, v); NaN stays NaN

template <typename F>
__device__ __forceinline__ F box_area(F x0, F y0, F x1, F y1, F d) { return (x1 - x0 + d) * (y1 - y0 + d); }

// np.maximum / np.minimum: a NaN operand wins (unlike fmax/fmin and unlike a bare `a > b ? a : b`)
template <typename F>
__device__ __forceinline__ F np_maximum(F a, F b) { return (a > b || a != a) ? a : b; }
template <typename F>
__device__ __forceinline__ F np_minimum(F a, F b) { return (a < b || a != a) ? a : b; }

template <typename F>
__device__ __forceinline__ F iou_px(const PxBox<F>& a, const PxBox<F>& b) {
    const F ix0 = np_maximum<F>(a.x0, b.x0);
    const F iy0 = np_maximum<F>(a.y0, b.y0);
    const F ix1 = np_minimum<F>(a.x1, b.x1);
    const F iy1 = np_minimum<F>(a.y1, b.y1);
    const F iw = clamp0<F>(ix1 - ix0);
    const F ih = clamp0<F>(iy1 - iy0);
    const F inter = iw * ih;
    const F uni = (a.area + b.area) - inter;
    return inter / uni;
}

// order-preserving map float -> u32 (larger float <=> larger key; -0 < +0, NaNs at the ends)
__device__ __forceinline__ u32 float_key(float f) {
    const u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(u32 k) {
    const u32 u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ u64 lanemask_lt() {
    const u32 lane = threadIdx.x & 63u;
    return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

// Inclusive suffix sum over the 64 lanes of a wave (lane l gets the sum of lanes l..63) as a butterfly: five ds_swizzle
// exchanges (xor 1..16: an immediate pattern, NO address register -- __shfl_down costs one VGPR of lane addresses per distance,
// which the compiler hoists out of every loop around the call and then spills in the 80-register NMS kernel) and one bpermute for xor 32.
__device__ __forceinline__ int wave_suffix_sum(int v, int lane) {
    int suf = v;
#if defined(__HIP_DEVICE_COMPILE__)
    int tot = v;
#define SSDHIP_SWZ_STEP(D)                                                              \
    {                                                                                   \
        const int o = __builtin_amdgcn_ds_swizzle(tot, 0x1F | ((D) << 10));             \
        if (!(lane & (D))) suf += o;                                                    \
        tot += o;                                                                       \
    }
    SSDHIP_SWZ_STEP(1) SSDHIP_SWZ_STEP(2) SSDHIP_SWZ_STEP(4) SSDHIP_SWZ_STEP(8) SSDHIP_SWZ_STEP(16)
#undef SSDHIP_SWZ_STEP
    const int o32 = __shfl_xor(tot, 32);
    if (lane < 32) suf += o32;
#endif
    return suf;
}

// ---------------------------------------------------------------------------------------
// Radix-select helper: hist[] holds per-digit counts (blockDim.x * PER bins); find the digit d with
//   #(digit > d) < want <= #(digit >= d)      (want >= 1, want <= total count)
// All threads call it; result in out[0] = d, out[1] = #(digit > d).  wave_tot: LDS, blockDim.x/64 ints.
// Parallel suffix scan: per-thread bin sums, shuffle scan inside each wave, wave totals through LDS.
// ---------------------------------------------------------------------------------------
template <int PER>
__device__ __forceinline__ void block_find_digit(const u32* hist, int want, int* wave_tot, int* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    u32 local[PER];
    int s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { local[j] = hist[tid * PER + j]; s += (int)local[j]; }
    const int suf = wave_suffix_sum(s, lane);      // inclusive suffix sum over the lanes of this wave
    if (lane == 0) wave_tot[wave] = suf;
    __syncthreads();
    int above = suf - s;
    for (int w = wave + 1; w < nw; ++w) above += wave_tot[w];
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) {
        if (above < want && want <= above + (int)local[j]) { out[0] = tid * PER + j; out[1] = above; }
        above += (int)local[j];
    }
    __syncthreads();
}

// The same scan answering TWO questions from one read of the histogram: out[0..1] as block_find_digit(hist, want); then
// out[2] = the digit d2 with #(digit > d2) < out[1] + want2 <= #(digit >= d2) and out[3] = #(digit > d2).  When fewer than
// out[1] + want2 keys are counted at all, out[2..3] = (none2, above_none2).
template <int PER>
__device__ __forceinline__ void block_find_digit2(const u32* hist, int want, int want2, int* wave_tot, int* out, int none2, int above_none2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    u32 local[PER];
    int s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { local[j] = hist[tid * PER + j]; s += (int)local[j]; }
    const int suf = wave_suffix_sum(s, lane);
    if (lane == 0) wave_tot[wave] = suf;
    if (tid == 0) { out[2] = none2; out[3] = above_none2; }
    __syncthreads();
    int above0 = suf - s;
    for (int w = wave + 1; w < nw; ++w) above0 += wave_tot[w];
    int above = above0;
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) {
        if (above < want && want <= above + (int)local[j]) { out[0] = tid * PER + j; out[1] = above; }
        above += (int)local[j];
    }
    __syncthreads();
    const int w2 = out[1] + want2;
    above = above0;
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) {
        if (above < w2 && w2 <= above + (int)local[j]) { out[2] = tid * PER + j; out[3] = above; }
        above += (int)local[j];
    }
    __syncthreads();
}

}  // namespace ssdhip
