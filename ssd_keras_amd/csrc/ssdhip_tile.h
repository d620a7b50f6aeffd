// Coalesced global -> LDS copy of a contiguous run of floats (a tile of whole [C+12]-rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssdhip {

// Copies src[0..total) to LDS with 16-byte vector loads where global memory is 16-byte aligned.
// `lds_base` must be 16-byte aligned and have room for total + 4 floats; returns the pointer p with
// p[i] <-> src[i] (LDS keeps the 16-byte phase of the global address).  Caller must __syncthreads().
//
// The plain loop below keeps ONE load in flight per thread (the compiler emits load / wait / ds_write per iteration).
// Measured on MI355X (profiles/r01l_loss_copy_variants.txt): batching 8 or 16 loads per thread ahead of the LDS stores
// made the loss kernels SLOWER (L1 27 -> 36 -> 41 us, backward 37 -> 46 -> 52 us); the latency is covered by the other
// workgroups of the CU instead, so the kernels that use this keep their LDS footprint small (>= 4 workgroups per CU).
__device__ __forceinline__ float* tile_copy_f32(float* lds_base, const float* __restrict__ src, int total, int tid, int nthreads) {
    const int phase = (int)(((uintptr_t)src & 15u) >> 2);
    float* tile = lds_base + phase;
    const int head = min(total, (4 - phase) & 3);
    if (tid < head) tile[tid] = src[tid];
    const int nvec = (total - head) >> 2;
    const float4* vsrc = reinterpret_cast<const float4*>(src + head);
    float4* vdst = reinterpret_cast<float4*>(tile + head);
    for (int i = tid; i < nvec; i += nthreads) vdst[i] = vsrc[i];
    const int done = head + (nvec << 2);
    if (tid < total - done) tile[done + tid] = src[done + tid];
    return tile;
}

}  // namespace ssdhip
