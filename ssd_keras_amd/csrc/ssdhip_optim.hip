// ssdhip_optim.hip -- the parameter side of the TRAINING step on gfx950 (MI355X): one launch over ALL parameters instead of one or
// two framework kernels per layer.
//
// The reference trains with keras.optimizers.SGD(lr=0.001, momentum=0.9) on float32 weights (ssd300_training.ipynb:169-173).  Here the
// float32 master weights stay the optimizer's; what the MFMA kernels read are bf16 copies in the layouts they want:
//   * the layer's filters [Cout][kh][kw][Cin] bf16 (channels_last) for the forward pass and the weight gradient,
//   * the same filters transposed (Cin <-> Cout) with their taps flipped, [Cin][kh][kw][Cout], for the data gradient (a stride-1 'same'
//     convolution of dL/dy with exactly those filters).
// Round 4 built them per layer and per step with framework ops: a cast, a layout copy where a kernel wanted channels_last, torch.flip +
// permute + contiguous in every backward, torch.cat for the packed predictor heads -- 52 copies, 19 flips, 15 concatenations, 24 fills:
// ~0.8 ms and ~110 launches of an 11 ms step (profiles/r05h_train_step_timeline.json).
//
//   shadow_refresh_kernel   grid = tiles of 32 x 32 (Cout x Cin) filter taps over every weight tensor + a tail for the 1-D tensors
//                           (biases, L2Normalization's gamma): float32 master -> bf16 through LDS, both layouts written as 64-byte runs.
//   sgd_momentum_kernel     torch.optim.SGD's update (buf = momentum buf + g [+ wd p]; p -= lr buf; the first step's buf = g), every
//                           parameter in one launch, float32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssdhip.h"
#include "ssdhip_math.h"

namespace ssdhip {

__device__ __forceinline__ unsigned short opt_f2b(float f) {          // round to nearest even, NaN stays NaN (as c10::BFloat16)
    const u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

constexpr int SH_T = 32;                                  // tile edge (output x input channels)
constexpr int SH_MAXKK = 16;                              // taps of a filter: 1 x 1, 3 x 3, 4 x 4 (SSD512's conv10_2)

// One block = one (32 Cout x 32 Cin) tile of one weight tensor, or one 256-element run of a 1-D tensor.
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const ssdhip_shadow_desc* __restrict__ tab, int n_w, int n_tiles, int n_v) {
    __shared__ unsigned short tile[SH_MAXKK * SH_T * (SH_T + 2)];
    __shared__ int sh_which;
    const int tid = threadIdx.x;
    int blk = (int)blockIdx.x;
    if (blk >= n_tiles) {                                 // ---- 1-D tensors: descriptors n_w .. n_w + n_v - 1, tile0 = first run
        blk -= n_tiles;
        if (tid == 0) {
            int k = n_w;
            while (k + 1 < n_w + n_v && tab[k + 1].tile0 <= blk) ++k;
            sh_which = k;
        }
        __syncthreads();
        const ssdhip_shadow_desc d = tab[sh_which];
        const int i = (blk - d.tile0) * 256 + tid;
        if (i < d.O) static_cast<unsigned short*>(d.cl)[i] = opt_f2b(static_cast<const float*>(d.src)[i]);
        return;
    }
    if (tid == 0) {
        int k = 0;
        while (k + 1 < n_w && tab[k + 1].tile0 <= blk) ++k;
        sh_which = k;
    }
    __syncthreads();
    const ssdhip_shadow_desc d = tab[sh_which];
    const int kk = d.KK;
    const int ti = (d.I + SH_T - 1) / SH_T;
    const int t = blk - d.tile0;
    const int o0 = (t / ti) * SH_T, i0 = (t % ti) * SH_T;
    const int no = min(SH_T, d.O - o0), ni = min(SH_T, d.I - i0);
    const float* src = static_cast<const float*>(d.src);
    const int run = ni * kk;
    // (six loads in flight per thread and trip: one at a time, a tile of 9216 values was 36 dependent memory round trips per workgroup
    //  and the launch ran at a third of the HBM rate)
    constexpr int SH_U = 6;
    const int n_el = no * run;
    if (d.src_channels_last) {
        // master [O][kk][I] float32 (a channels_last parameter): runs of ni values per (o, tap)
        for (int e0 = tid; e0 < n_el; e0 += 256 * SH_U) {
            float v[SH_U];
            int at[SH_U];
#pragma unroll
            for (int u = 0; u < SH_U; ++u) {
                const int e = e0 + 256 * u, ec = e < n_el ? e : 0;
                const int o = ec / run, j = ec - o * run;
                const int tap = j / ni, i = j - tap * ni;
                at[u] = e < n_el ? (tap * SH_T + o) * (SH_T + 2) + i : -1;
                v[u] = src[((size_t)(o0 + o) * kk + tap) * d.I + i0 + i];
            }
#pragma unroll
            for (int u = 0; u < SH_U; ++u)
                if (at[u] >= 0) tile[at[u]] = opt_f2b(v[u]);
        }
    } else {
        // master [O][I][kk] float32: the tile's row o is the contiguous run [i0 .. i0 + ni) x kk
        for (int e0 = tid; e0 < n_el; e0 += 256 * SH_U) {
            float v[SH_U];
            int at[SH_U];
#pragma unroll
            for (int u = 0; u < SH_U; ++u) {
                const int e = e0 + 256 * u, ec = e < n_el ? e : 0;
                const int o = ec / run, j = ec - o * run;
                const int i = j / kk, tap = j - i * kk;
                at[u] = e < n_el ? (tap * SH_T + o) * (SH_T + 2) + i : -1;
                v[u] = src[((size_t)(o0 + o) * d.I + i0) * kk + j];
            }
#pragma unroll
            for (int u = 0; u < SH_U; ++u)
                if (at[u] >= 0) tile[at[u]] = opt_f2b(v[u]);
        }
    }
    __syncthreads();
    // channels_last copy [O][kk][I]: runs of ni values
    // (four values = 8 bytes per store where the runs allow it: two-byte stores left the launch at 1.6 TB/s)
    unsigned short* cl = static_cast<unsigned short*>(d.cl);
    if (cl) {
        if (!(d.I & 3) && !((uintptr_t)cl & 7)) {                       // i0 is a multiple of 32, so ni % 4 == 0 as well
            for (int e = tid; e < no * kk * (SH_T / 4); e += 256) {
                const int i = (e & (SH_T / 4 - 1)) * 4, r = e >> 3;
                const int tap = r % kk, o = r / kk;
                if (i < ni) {
                    const u32* t2 = reinterpret_cast<const u32*>(&tile[(tap * SH_T + o) * (SH_T + 2) + i]);   // rows are 68 bytes: 4-byte aligned
                    *reinterpret_cast<uint2*>(&cl[((size_t)(o0 + o) * kk + tap) * d.I + i0 + i]) = make_uint2(t2[0], t2[1]);
                }
            }
        } else {
            for (int e = tid; e < no * kk * SH_T; e += 256) {
                const int i = e & (SH_T - 1), r = e >> 5;
                const int tap = r % kk, o = r / kk;
                if (i < ni) cl[((size_t)(o0 + o) * kk + tap) * d.I + i0 + i] = tile[(tap * SH_T + o) * (SH_T + 2) + i];
            }
        }
    }
    // transposed, taps flipped [I][kk][tr_ostride] at channel offset tr_ooff: runs of no values
    unsigned short* tr = static_cast<unsigned short*>(d.tr);
    if (tr) {
        if (!(d.tr_ostride & 3) && !(d.tr_ooff & 3) && !(no & 3) && !((uintptr_t)tr & 7)) {
            for (int e = tid; e < ni * kk * (SH_T / 4); e += 256) {
                const int o = (e & (SH_T / 4 - 1)) * 4, r = e >> 3;
                const int tap = r % kk, i = r / kk;
                if (o < no) {
                    const unsigned short* tp = &tile[(tap * SH_T + o) * (SH_T + 2) + i];
                    const u32 a = tp[0], b = tp[SH_T + 2], c = tp[2 * (SH_T + 2)], e3 = tp[3 * (SH_T + 2)];
                    *reinterpret_cast<uint2*>(&tr[((size_t)(i0 + i) * kk + (kk - 1 - tap)) * d.tr_ostride + d.tr_ooff + o0 + o]) =
                        make_uint2(a | (b << 16), c | (e3 << 16));
                }
            }
        } else {
            for (int e = tid; e < ni * kk * SH_T; e += 256) {
                const int o = e & (SH_T - 1), r = e >> 5;
                const int tap = r % kk, i = r / kk;
                if (o < no) tr[((size_t)(i0 + i) * kk + (kk - 1 - tap)) * d.tr_ostride + d.tr_ooff + o0 + o] = tile[(tap * SH_T + o) * (SH_T + 2) + i];
            }
        }
    }
}

// torch.optim.SGD (momentum, dampening 0, no Nesterov): four values per thread and step, up to SGD_CHUNK tensors per launch.  The
// tensor table travels in the KERNEL ARGUMENTS: gradients are new tensors every step (zero_grad(set_to_none=True)), and a device-side
// table would have to be re-uploaded -- a pageable host-to-device copy that blocks the host until the whole backward pass has drained
// (measured: the step 0.3-0.5 ms SLOWER than with the framework's optimizer although the kernels took 250 us less).
constexpr int SGD_CHUNK = 80;
struct SgdArgs {
    float* p[SGD_CHUNK];
    const float* g[SGD_CHUNK];
    float* m[SGD_CHUNK];
    long long n[SGD_CHUNK];
    int block0[SGD_CHUNK];                                 // first block of each tensor (4096 values per block)
    int count;
};

__global__ __launch_bounds__(256) void sgd_momentum_kernel(const SgdArgs a, float lr, float momentum, float weight_decay) {
    const int tid = threadIdx.x, blk = (int)blockIdx.x;
    int lo = 0, hi = a.count - 1;                          // last tensor with block0 <= blk (uniform: scalar loads of the arguments)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.block0[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    float* p = a.p[lo];
    const float* g = a.g[lo];
    float* m = a.m[lo];
    const long long n = a.n[lo];
    const long long base = (long long)(blk - a.block0[lo]) * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + (long long)u * 1024 + tid * 4;
        if (i + 3 < n) {
            const float4 pv = *reinterpret_cast<const float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
            const float4 mv = *reinterpret_cast<const float4*>(m + i);
            float gg[4] = {gv.x, gv.y, gv.z, gv.w}, pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float gq = gg[q];
                if (weight_decay != 0.f) gq = gq + weight_decay * pp[q];
                mm[q] = momentum * mm[q] + gq;
                pp[q] = pp[q] - lr * mm[q];
            }
            *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) {
                float gq = g[j];
                if (weight_decay != 0.f) gq = gq + weight_decay * p[j];
                const float mq = momentum * m[j] + gq;
                m[j] = mq;
                p[j] = p[j] - lr * mq;
            }
        }
    }
}

}  // namespace ssdhip

using namespace ssdhip;

extern "C" int ssdhip_shadow_refresh(const ssdhip_shadow_desc* table_dev, int n_weights, int n_tiles, int n_vectors, int n_vector_blocks,
                                     void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!table_dev || n_weights < 0 || n_vectors < 0 || n_tiles < 0 || n_vector_blocks < 0 || (n_weights == 0) != (n_tiles == 0)
        || (n_vectors == 0) != (n_vector_blocks == 0))
        return SSDHIP_E_BADARG;
    if (n_tiles + n_vector_blocks == 0) return SSDHIP_OK;
    hipLaunchKernelGGL(shadow_refresh_kernel, dim3((unsigned)(n_tiles + n_vector_blocks)), dim3(256), 0, stream, table_dev, n_weights,
                       n_tiles, n_vectors);
    return hipGetLastError() == hipSuccess ? SSDHIP_OK : SSDHIP_E_LAUNCH;
}

extern "C" int ssdhip_sgd_momentum_step(int n_tensors, void* const* params_h, const void* const* grads_h, void* const* bufs_h,
                                        const long long* numel_h, double lr, double momentum, double weight_decay, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_tensors <= 0 || !params_h || !grads_h || !bufs_h || !numel_h) return SSDHIP_E_BADARG;
    for (int k = 0; k < n_tensors; ++k) {
        if (!params_h[k] || !grads_h[k] || !bufs_h[k] || numel_h[k] <= 0) return SSDHIP_E_BADARG;
        if (((uintptr_t)params_h[k] | (uintptr_t)grads_h[k] | (uintptr_t)bufs_h[k]) & 15) return SSDHIP_E_BADARG;
        if ((numel_h[k] + 4095) / 4096 > 0x3fffffffLL) return SSDHIP_E_BADARG;
    }
    for (int k0 = 0; k0 < n_tensors; k0 += SGD_CHUNK) {
        SgdArgs a;
        a.count = n_tensors - k0 < SGD_CHUNK ? n_tensors - k0 : SGD_CHUNK;
        long long blocks = 0;
        for (int k = 0; k < SGD_CHUNK; ++k) {
            const int src = k < a.count ? k0 + k : k0;     // (unused slots repeat the first tensor: never selected)
            a.p[k] = static_cast<float*>(params_h[src]);
            a.g[k] = static_cast<const float*>(grads_h[src]);
            a.m[k] = static_cast<float*>(bufs_h[src]);
            a.n[k] = numel_h[src];
            a.block0[k] = (int)blocks;
            if (k < a.count) blocks += (numel_h[src] + 4095) / 4096;
            if (blocks > 0x7fffffffLL) return SSDHIP_E_BADARG;
        }
        hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, (float)lr, (float)momentum, (float)weight_decay);
        if (hipGetLastError() != hipSuccess) return SSDHIP_E_LAUNCH;
    }
    return SSDHIP_OK;
}
