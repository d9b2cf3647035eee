// A stand-in for a resident collective kernel: `nwg` workgroups of 256 threads that each hold a CU slot (32 KB of LDS,
// so a GEMM block with its ~150 KB operand ring cannot share the CU — as it cannot with a register-heavy RCCL
// workgroup) until `ticks` of the 100 MHz wall clock have passed.  Every wave reaches the exit condition by itself (clock based), so the grid always drains.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) occupy_kernel(int64_t ticks, float* sink) {
    __shared__ float hold[8192];
    hold[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const int64_t t0 = wall_clock64();
    float a = hold[(threadIdx.x * 7) & 8191];
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        a = a * 1.0001f + 1.0f;
    }
    if (a == -1.f) sink[0] = a;
}
extern "C" int probe_occupy(int nwg, int64_t ticks, float* sink, void* stream) {
    hipLaunchKernelGGL(occupy_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, ticks, sink);
    return (int)hipGetLastError();
}
