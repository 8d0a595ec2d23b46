// Probe: is the SGPR offset (soffset) of a raw buffer load part of the range check on gfx950?
// A 1024-byte descriptor over a 4096-byte allocation filled with word index + 1.  GPU box only; not part of libag2v.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const unsigned* buf, unsigned* out, unsigned s1, unsigned s2) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, 1024, 0x00020000);
    // 0: voffset in range, soffset moves the address past num_records (still inside the allocation)
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 1020u, s1, 0);
    // 1: voffset out of range, voffset + soffset wraps around to 16
    out[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 0xFFFFFFF0u, s2, 0);
    // 2: plain in-range with soffset
    out[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 16u, s1, 0);
    // 3: voffset just out of range, soffset 0
    out[3] = __builtin_amdgcn_raw_buffer_load_b32(r, 1024u, 0, 0);
}
int main() {
    unsigned h[1024], *d, *o, ho[4];
    for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    hipMalloc(&d, 4096); hipMalloc(&o, 16);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    probe<<<1, 1>>>(d, o, 8u, 32u);
    hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
    printf("voffset 1020 + soffset 8 (past num_records 1024): %u (258 = read, 0 = range-checked)\n", ho[0]);
    printf("voffset 0xFFFFFFF0 + soffset 32 (wraps to 16):     %u (0 = voffset alone is checked, 5 = wrapped sum passes)\n", ho[1]);
    printf("voffset 16 + soffset 8:                            %u (expect 7)\n", ho[2]);
    printf("voffset 1024, soffset 0:                           %u (expect 0)\n", ho[3]);
    return 0;
}
