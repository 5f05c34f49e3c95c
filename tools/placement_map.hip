// Microbenchmark (GPU box): map of HBM write bandwidth over ONE big allocation: streams plain 16-byte stores (flat
// grid-stride, grid 1536) into consecutive windows of `win` GB of a `total` GB hipMalloc buffer.
// build: hipcc -O3 --offload-arch=gfx950 tools/placement_map.hip -o tools/placement_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_flat(double2 *out, size_t chunks, double v) {
    for (size_t g = blockIdx.x * 256ull + threadIdx.x; g < chunks; g += gridDim.x * 256ull) out[g] = make_double2(v, v);
}
int main(int argc, char **argv) {
    double total_gb = argc > 1 ? atof(argv[1]) : 240.0, win_gb = argc > 2 ? atof(argv[2]) : 1.0;
    size_t total = (size_t)(total_gb * 1e9) / (1 << 21) * (1 << 21), win = (size_t)(win_gb * 1e9) / (1 << 21) * (1 << 21);
    char *buf; CK(hipMalloc(&buf, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int reps = win > (size_t)5e9 ? 3 : 20;
    printf("buffer at %p, %zu windows of %.2f GB\n", buf, total / win, win / 1e9);
    for (int round = 0; round < 2; ++round) {
        for (size_t off = 0; off + win <= total; off += win) {
            double2 *p = (double2 *)(buf + off);
            hipLaunchKernelGGL(k_flat, dim3(1536), dim3(256), 0, 0, p, win / 16, 1.0);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_flat, dim3(1536), dim3(256), 0, 0, p, win / 16, 1.5);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5.2f ", win / (ms / reps * 1e-3) / 1e12);
            if (((off / win) + 1) % 16 == 0) printf("\n");
        }
        printf("\n");
    }
    return 0;
}
