// Probe: what does an LDS-DMA load (buffer_load_dwordx4 ... lds) write for a lane whose buffer offset is out of range -- zeros, or
// nothing (the lane's 16 LDS bytes keep their old contents)?  conv_halo.hip is correct under both (its padding positions are cleared
// once and only ever targeted by out-of-range lanes); a persistent-tile variant would need "zeros".  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_vptr;
__global__ void probe(const unsigned* buf, unsigned* out) {
    __shared__ __attribute__((aligned(1024))) unsigned lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, 1024, 0x00020000);
    // lanes 0..31 in range (16 bytes each), lanes 32..47 voffset out of range, lanes 48..63 in range again
    const unsigned voff = (lane >= 32 && lane < 48) ? 0xFFFFFFF0u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    unsigned h[256], *d, *o, ho[256];
    for (int i = 0; i < 256; ++i) h[i] = i + 1;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    printf("lane 0  (in range):     %08x %08x (expect 00000001 00000002)\n", ho[0], ho[1]);
    printf("lane 32 (out of range): %08x %08x (00000000 = zeros written, dead0080 = untouched)\n", ho[128], ho[129]);
    printf("lane 47 (out of range): %08x\n", ho[47 * 4]);
    printf("lane 48 (in range):     %08x (expect %08x)\n", ho[192], 193);
    return 0;
}
