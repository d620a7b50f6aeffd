// Does a chain of v_mfma_f32_32x32x16_bf16 on ONE accumulator (every MFMA reads the previous one's result as C) issue at the matrix
// pipe's rate?  One wave per SIMD, N MFMAs, CH = 1, 2, 4 independent accumulators dealt round-robin; clock64 ticks per MFMA.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mdc tools/micro/mfma_dependent_chain.hip && /tmp/mdc
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(long long* out, float* sink, int n) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x16 c[CH];
    for (int j = 0; j < CH; ++j) for (int v = 0; v < 16; ++v) c[j][v] = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % CH], 0, 0, 0);
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    const long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < CH; ++j) s += c[j][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    long long* d; float* s; long long h;
    hipMalloc(&d, 16); hipMalloc(&s, 1024 * 256 * 4);
    const int n = 1 << 16;
    for (int wgs : {1, 256}) {
        for (int ch : {1, 2, 4, 1, 2}) {
            if (ch == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, s, n);
            else if (ch == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, d, s, n);
            else hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, d, s, n);
            hipDeviceSynchronize();
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("mfma_dependent_chain: %4d workgroups x 4 waves, %d accumulator chain(s): %.2f clock64 ticks per MFMA\n", wgs, ch, (double)h / n);
        }
    }
    return 0;
}
