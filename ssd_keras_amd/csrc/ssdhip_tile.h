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

// ---- LDS-DMA streaming of row tiles (round 5: the loss kernels) ---------------------------------------------------------------
// One wave-wide buffer_load ... lds: lane l writes 16 (or 4) bytes at lds_dst + 16 l (4 l) from base(rsrc) + voff; a voff at or past
// num_records writes zeros.  No VGPR staging, so a wave keeps a whole tile (two or three dozen 1 KiB loads) in flight while it works on
// the previous one.  Issued from inline asm and counted by hand (tile_wait_vmcnt): a __builtin load makes hipcc drain vmcnt in front of
// every LDS read that might alias the DMA.  M0 is saved and restored inside the statement (hipcc does not model it around asm).
typedef int tile_i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned int TILE_OOB = 0x80000000u;

__device__ __forceinline__ tile_i32x4 tile_rsrc(const void* base, unsigned int num_bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    tile_i32x4 r;
    r.x = (int)(unsigned int)a;
    r.y = (int)((unsigned int)(a >> 32) & 0xffffu);    // stride 0, no swizzle
    r.z = (int)num_bytes;
    r.w = 0x00020000;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void tile_dma16(unsigned int voff, tile_i32x4 rsrc, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void tile_dma4(unsigned int voff, tile_i32x4 rsrc, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 63 (the immediate has to be a constant: a jump over 64 one-instruction cases)
__device__ __forceinline__ void tile_wait_vmcnt(int n) {
#define SSDHIP_W1(k) case k: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(k) : "memory"); break;
#define SSDHIP_W8(k) SSDHIP_W1(k) SSDHIP_W1(k + 1) SSDHIP_W1(k + 2) SSDHIP_W1(k + 3) SSDHIP_W1(k + 4) SSDHIP_W1(k + 5) SSDHIP_W1(k + 6) SSDHIP_W1(k + 7)
    switch (n) {
        SSDHIP_W8(0) SSDHIP_W8(8) SSDHIP_W8(16) SSDHIP_W8(24) SSDHIP_W8(32) SSDHIP_W8(40) SSDHIP_W8(48) SSDHIP_W8(56)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef SSDHIP_W8
#undef SSDHIP_W1
}
#else
__device__ inline void tile_dma16(unsigned int, tile_i32x4, unsigned int) {}
__device__ inline void tile_dma4(unsigned int, tile_i32x4, unsigned int) {}
__device__ inline void tile_wait_vmcnt(int) {}
#endif

}  // namespace ssdhip
