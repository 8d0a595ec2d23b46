#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const int l = threadIdx.x;
    unsigned a[4];
    for (int j = 0; j < 4; ++j) a[j] = 100 * j + 10 * (l >> 4) + 0;   // value encodes (register j, lane group g)
    u32x2 t;
    t = __builtin_amdgcn_permlane32_swap(a[0], a[2], false, false); a[0] = t[0]; a[2] = t[1];
    t = __builtin_amdgcn_permlane32_swap(a[1], a[3], false, false); a[1] = t[0]; a[3] = t[1];
    t = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false); a[0] = t[0]; a[1] = t[1];
    t = __builtin_amdgcn_permlane16_swap(a[2], a[3], false, false); a[2] = t[0]; a[3] = t[1];
    for (int j = 0; j < 4; ++j) out[j * 64 + l] = a[j];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { printf("reg %d:", j); for (int g = 0; g < 4; ++g) printf(" %3u", h[j * 64 + 16 * g]); printf("   (want j'=lane group, g'=reg: %d0x)\n", j); }
    return 0;
}
