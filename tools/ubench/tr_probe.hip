#include <hip/hip_runtime.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int q = (l & 15) >> 2, p = l & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4*)(&img[q * 64 + 4 * p + 16 * (l >> 4)]);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    out[l * 4 + 0] = v[0]; out[l * 4 + 1] = v[1]; out[l * 4 + 2] = v[2]; out[l * 4 + 3] = v[3];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
