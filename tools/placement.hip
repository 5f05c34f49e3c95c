// Microbenchmark (GPU box): does HBM write bandwidth depend on WHERE a buffer lies?  Allocates `count` buffers of
// `gb` GB each with hipMalloc and streams plain 16-byte stores (flat grid-stride, grid 1536) into each in turn,
// several rounds; prints TB/s per buffer with its device address.
// build: hipcc -O3 --offload-arch=gfx950 tools/placement.hip -o tools/placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_flat(double2 *out, size_t chunks, double v) {
    for (size_t g = blockIdx.x * 256ull + threadIdx.x; g < chunks; g += gridDim.x * 256ull) out[g] = make_double2(v, v);
}

int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 20.0;
    int count = argc > 2 ? atoi(argv[2]) : 12;
    int grid = argc > 3 ? atoi(argv[3]) : 1536;
    size_t bytes = (size_t)(gb * 1e9) / 4096 * 4096;
    std::vector<double2 *> buf(count);
    for (int i = 0; i < count; ++i) CK(hipMalloc(&buf[i], bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int reps = bytes > (size_t)5e9 ? 3 : 50;
    for (int round = 0; round < 3; ++round)
        for (int i = 0; i < count; ++i) {
            hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, buf[i], bytes / 16, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, buf[i], bytes / 16, 1.5);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d buffer %2d at %p (%6.1f GB): %7.3f ms  %5.2f TB/s\n", round, i, (void *)buf[i], bytes / 1e9, ms / reps,
                   bytes / (ms / reps * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
