// tools/cotenant.hip -- a "co-tenant" kernel for tools/cotenant.py: `blocks` workgroups of `threads` lanes that hold their CU
// slots (and `lds` bytes of LDS each) for `micros` microseconds without using memory bandwidth -- a stand-in for the
// collective kernels RCCL runs on its own stream beside the step kernel in an N > 1 rollout.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/cotenant.hip -o tools/libcotenant.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void squat_kernel(unsigned long long ticks, int *sink) {
    extern __shared__ int smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const unsigned long long t0 = wall_clock64();  // 100 MHz constant clock
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && smem[0] == 12345) sink[blockIdx.x] = 1;
}

extern "C" int cotenant_launch(int blocks, int threads, int lds, double micros, void *stream) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)squat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(squat_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (unsigned long long)(micros * 100.0), nullptr);
    return (int)hipGetLastError();
}
