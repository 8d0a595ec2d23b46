// TEST AIDS -- built into libag2v_test.so, NOT into the product library (a 128 KB-LDS busy-wait kernel and its entry point
// do not belong in libag2v.so / include/ag2v.h).  Loaded by tests/_test_aids.py only.
#include "common.h"

// ---- TEST AID (tests/test_graph_program_gpu.py, tests/test_distributed_gpu.py): occupy `n_wgs` CUs -- 1024 threads and 128 KB of LDS
// per workgroup, so nothing that needs a CU's LDS can share them -- for `ms` milliseconds of wall time (at most 500), the way another
// stream's persistent kernel or a communication kernel would.  Bounded by the clock: it cannot hang.
__global__ __launch_bounds__(1024) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
    __shared__ unsigned held[32768];
    held[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    }
    __syncthreads();
    if (sink && held[(threadIdx.x * 7) & 32767] == 0xFFFFFFFFu) *sink = 1;       // (keeps the LDS array alive)
}
extern "C" int ag2v_test_hold_cus(int n_wgs, float ms, void* stream) {
    if (n_wgs <= 0 || n_wgs > 256 || !(ms > 0.f) || ms > 500.f) return AG2V_EINVAL;
    hipLaunchKernelGGL(hold_cus_kernel, dim3(n_wgs), dim3(1024), 0, (hipStream_t)stream, (unsigned long long)(ms * 1e5f), (unsigned*)nullptr);
    AG2V_CHECK_LAUNCH();
    return AG2V_OK;
}

