// ds_read_b64_tr_b16 semantics check (gfx950): the layout csrc/ssdhip_wgrad.hip relies on.  LDS holds a [rows][32] image of 16-bit
// values (64-byte rows), value = row * 32 + column.  Lane l = (khalf, g16, i): address of row khalf 8 + i / 4, columns g16 16 + (i & 3) 4.
// Expected: lane l, element j = value at row khalf 8 + j, column g16 16 + (l & 15).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/tr tools/micro/tr_read_semantics.hip && /tmp/tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g16 = (l >> 4) & 1, kh = l >> 5;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (kh * 8 + (i >> 2)) * 32 + g16 * 16 + (i & 3) * 4));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    int* d; int h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int want = ((l >> 5) * 8 + j) * 32 + ((l >> 4) & 1) * 16 + (l & 15);
            if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h[l * 4 + j], want); ++bad; }
        }
    printf("tr_read_semantics: %d mismatches\n", bad);
    return bad != 0;
}
