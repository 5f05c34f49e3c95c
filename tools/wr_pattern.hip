// Microbenchmark (run on the GPU box): does the ORDER in which workgroups stream plain
// 16-byte stores matter for HBM write bandwidth?  flat grid-stride vs per-tile regions.
// build: hipcc -O3 --offload-arch=gfx950 tools/wr_pattern.hip -o tools/wr_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__device__ __forceinline__ void st(double2 *p, double2 v) {
    if constexpr (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

// A: flat grid-stride over 16-byte chunks
template <bool NT>
__global__ __launch_bounds__(256) void k_flat(double2 *out, size_t chunks, double v) {
    for (size_t g = blockIdx.x * 256ull + threadIdx.x; g < chunks; g += gridDim.x * 256ull) st<NT>(out + g, make_double2(v, v));
}
// B: tiles of `tile_chunks` contiguous chunks, tiles grid-strided, each tile streamed by one block
template <bool NT>
__global__ __launch_bounds__(256) void k_tile(double2 *out, size_t chunks, size_t tile_chunks, double v) {
    size_t ntiles = (chunks + tile_chunks - 1) / tile_chunks;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t base = t * tile_chunks;
        size_t n = chunks - base < tile_chunks ? chunks - base : tile_chunks;
#pragma unroll 4
        for (size_t c = threadIdx.x; c < n; c += 256) st<NT>(out + base + c, make_double2(v, v));
    }
}

int main(int argc, char **argv) {
    size_t sizes[] = {(size_t)167772160, (size_t)20132659200ull, (size_t)100000000000ull};
    int grids[] = {1536, 2048, 4096};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t bytes : sizes) {
        double2 *buf[2];
        int nb = bytes < (size_t)50e9 ? 2 : 1;
        for (int i = 0; i < nb; ++i) CK(hipMalloc(&buf[i], bytes));
        size_t chunks = bytes / 16;
        int reps = bytes < (size_t)1e9 ? 200 : (bytes < (size_t)50e9 ? 20 : 5);
        auto run = [&](const char *name, auto launch) {
            for (int i = 0; i < 2; ++i) launch(buf[i % nb]);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(buf[i % nb]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%8.3f GB  %-34s %7.3f ms  %6.2f TB/s\n", bytes / 1e9, name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
            fflush(stdout);
        };
        char nm[128];
        for (int g : grids) {
            snprintf(nm, sizeof nm, "flat grid=%d", g);
            run(nm, [&](double2 *o) { hipLaunchKernelGGL(k_flat<false>, dim3(g), dim3(256), 0, 0, o, chunks, 1.5); });
        }
        run("flat grid=2048 nontemporal", [&](double2 *o) { hipLaunchKernelGGL(k_flat<true>, dim3(2048), dim3(256), 0, 0, o, chunks, 1.5); });
        size_t tiles[] = {2560 / 16, 81920 / 16, 614400 / 16, 1228800 / 16, 4915200 / 16};
        for (size_t tc : tiles)
            for (int g : {1536, 2048}) {
                snprintf(nm, sizeof nm, "tile=%zuB grid=%d", tc * 16, g);
                run(nm, [&](double2 *o) { hipLaunchKernelGGL(k_tile<false>, dim3(g), dim3(256), 0, 0, o, chunks, tc, 1.5); });
            }
        run("tile=614400B grid=2048 nontemporal", [&](double2 *o) { hipLaunchKernelGGL(k_tile<true>, dim3(2048), dim3(256), 0, 0, o, chunks, (size_t)614400 / 16, 1.5); });
        for (int i = 0; i < nb; ++i) CK(hipFree(buf[i]));
    }
    return 0;
}
