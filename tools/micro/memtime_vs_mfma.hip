// What does s_memtime (clock64) tick in?  One wave issues N back-to-back v_mfma_f32_32x32x16_bf16 on two alternating accumulators -- 32
// shader cycles each when the matrix pipe is the limit (MI355X_MICROARCH.md) -- between two clock64() reads and two wall_clock64() reads
// (s_memrealtime, a constant 100 MHz).  ticks / MFMA = 32 means clock64 counts shader cycles; the two clocks together give the shader
// clock during the loop.  Run once with ONE workgroup (an idle chip clocks high) and once with 1024 (every SIMD busy: the power-limited clock).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mt tools/micro/memtime_vs_mfma.hip && /tmp/mt
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(long long* out, float* sink, int n) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 c0 = {}, c1 = {};
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; i += 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
int main() {
    long long* d; float* s; long long h[2];
    hipMalloc(&d, 16); hipMalloc(&s, 1024 * 256 * 4);
    const int n = 1 << 16;
    for (int wgs : {1, 1024, 1024}) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, s, n);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double us = h[1] / 100.0;
        printf("memtime_vs_mfma: %4d workgroups x 4 waves, %d MFMAs per wave: %.2f clock64 ticks per MFMA, %.1f us -> %.3f GHz of clock64 ticks; %.1f ns per MFMA\n",
               wgs, n, (double)h[0] / n, us, h[0] / us / 1e3, us * 1e3 / n);
    }
    return 0;
}
