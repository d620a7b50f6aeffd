// Do DS reads of 4 / 8 / 16 bytes work at 2-byte-aligned LDS addresses on gfx950, and what do they cost?  (The conv1 block's producer
// waves gather nine 16-bit values per filter row from a 6-byte-per-pixel patch: 16 ds_read_u16 per 32 pixels today.)
// LDS holds value = index as 16-bit words; lane l reads at byte address 2 * (3 l + c) for c = 0, 1 -- even c: aligned for even l only.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ua tools/micro/lds_unaligned_reads.hip && /tmp/ua
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    for (int c = 0; c < 2; ++c) {
        const unsigned a = base + 2 * (3 * l + c);
        unsigned v1; u32x2 v2; u32x4 v4;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v1) : "v"(a) : "memory");
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v2) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v4) : "v"(a) : "memory");
        unsigned* o = out + (c * 64 + l) * 8;
        o[0] = v1; o[1] = v2.x; o[2] = v2.y; o[3] = v4.x; o[4] = v4.y; o[5] = v4.z; o[6] = v4.w; o[7] = 0;
    }
    // cost: 64 back-to-back reads of each kind, aligned (stride 16 B) vs the 6-byte stride, one wave
    for (int mode = 0; mode < 8; ++mode) {
        const int kind = mode >> 1, un = mode & 1;
        const unsigned a = base + (un ? 6 * l : 16 * l);
        unsigned acc = 0;
        const long long t0 = clock64();
        for (int r = 0; r < 64; ++r) {
            const unsigned ar = a + (r & 7) * 1024;
            if (kind == 0) { unsigned short v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v; }
            else if (kind == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v; }
            else if (kind == 2) { u32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v.x; }
            else { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v.x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (l == 0) cyc[mode] = t1 - t0;
        out[1024 + mode * 64 + l] = acc;
    }
}
int main() {
    unsigned* d; long long* c; static unsigned h[2048]; long long hc[8];
    hipMalloc(&d, sizeof(h)); hipMalloc(&c, sizeof(hc));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    if (hipDeviceSynchronize() != hipSuccess) { printf("lds_unaligned_reads: kernel FAILED (%s)\n", hipGetErrorString(hipGetLastError())); return 2; }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int cc = 0; cc < 2; ++cc)
        for (int l = 0; l < 64; ++l) {
            const unsigned e = 3 * l + cc;   // first 16-bit element
            auto pair = [&](unsigned i) { return (e + 2 * i) | ((e + 2 * i + 1) << 16); };
            const unsigned* o = h + (cc * 64 + l) * 8;
            if (o[0] != pair(0)) ++bad[0];
            if (o[1] != pair(0) || o[2] != pair(1)) ++bad[1];
            if (o[3] != pair(0) || o[4] != pair(1) || o[5] != pair(2) || o[6] != pair(3)) { if (bad[2] < 4) printf("b128 lane %d c %d: %08x %08x %08x %08x want %08x ..\n", l, cc, o[3], o[4], o[5], o[6], pair(0)); ++bad[2]; }
        }
    printf("lds_unaligned_reads: mismatches b32 %d  b64 %d  b128 %d (of 128 lane reads each)\n", bad[0], bad[1], bad[2]);
    const char* names[4] = {"u16", "b32", "b64", "b128"};
    for (int m = 0; m < 8; ++m) printf("  %-4s %s: %.1f cycles per read (one wave, 64 reads)\n", names[m >> 1], (m & 1) ? "6-byte stride (unaligned)" : "16-byte stride (aligned)  ", hc[m] / 64.0);
    return 0;
}
