// Probe: range checking of STRUCTURED buffer loads (stride = one image row) on gfx950: is a 16-byte load that starts before the
// row (offset -4) or runs past its end (offset stride-8) clipped per dword?  (Would give free column masking for 3x3 taps.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ u32x4 sload(u32x4 rsrc, unsigned index, unsigned offset) {
    u32x4 v;
    u32x2 io = {index, offset};
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(io), "s"(rsrc) : "memory");
    return v;
}
__global__ void k(const unsigned* src, unsigned W, unsigned rows, unsigned* out, unsigned flags) {
    const unsigned long long p = (unsigned long long)src;
    u32x4 r = {(unsigned)p, (unsigned)(p >> 32) | ((W * 4u) << 16), rows, flags};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    u32x4 a = sload(r, 2, 0xFFFFFFFCu);      // row 2, x = -1..2
    u32x4 b = sload(r, 2, W * 4 - 8);        // row 2, x = W-2..W+1
    u32x4 c = sload(r, 2, 4);                // row 2, x = 1..4
    u32x4 d = sload(r, rows, 0);             // row out of range
    u32x4 e = sload(r, 0xFFFFFFFFu, 0);      // row -1
    for (int i = 0; i < 4; ++i) { out[i] = a[i]; out[4 + i] = b[i]; out[8 + i] = c[i]; out[12 + i] = d[i]; out[16 + i] = e[i]; }
}
int main() {
    const unsigned W = 16, rows = 8;
    unsigned h[W * rows + 64];
    for (unsigned i = 0; i < W * rows + 64; ++i) h[i] = 1000 + i;
    unsigned *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const unsigned flagsets[3] = {0x00020000u, 0x00020000u | (3u << 28), 0x00027000u};
    for (int f = 0; f < 3; ++f) {
        k<<<1, 1>>>(d, W, rows, o, flagsets[f]);
        unsigned r[20]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("flags %08x  (row 2 starts at value %u, W=%u)\n", flagsets[f], 1000 + 2 * W, W);
        printf("  offset -4      : %u %u %u %u\n", r[0], r[1], r[2], r[3]);
        printf("  offset 4W-8    : %u %u %u %u\n", r[4], r[5], r[6], r[7]);
        printf("  offset 4       : %u %u %u %u\n", r[8], r[9], r[10], r[11]);
        printf("  index = rows   : %u %u %u %u\n", r[12], r[13], r[14], r[15]);
        printf("  index = -1     : %u %u %u %u\n", r[16], r[17], r[18], r[19]);
    }
    return 0;
}
