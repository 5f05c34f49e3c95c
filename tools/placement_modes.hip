// Microbenchmark (GPU box): the per-buffer spread of HBM write bandwidth (tools/placement.hip: 5.7 ... 6.5 TB/s for the
// same flat store kernel on different hipMalloc buffers) under different workgroup -> address mappings.
//   mode 0  flat grid-stride, 16 B per thread: the whole grid writes one moving ~6 MB front
//   mode 1  blocked: workgroup b owns the contiguous bytes [b, b+1) * bytes / grid
//   mode 2  XCD-blocked: XCD x (= b % 8) owns the contiguous eighth x of the buffer, its workgroups grid-stride inside
//   mode 3  tiles of TILE bytes, grid-strided (tile t -> workgroup t % grid): what the step kernel does (20 KiB tiles)
//   mode 4  mode 3 with the tile order scrambled (multiplicative hash of the tile index)
//   mode 5  mode 3 with tile t written by XCD (t / 8) % 8 ... i.e. 8 consecutive tiles per XCD
//   mode 6  XCD-blocked tiles: XCD x owns the contiguous eighth x of the TILES, its workgroups stride through them
// build: hipcc -O3 --offload-arch=gfx950 tools/placement_modes.hip -o tools/placement_modes
// usage: placement_modes <GB per buffer> <buffers> <grid> <tile bytes>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_store(double2 *out, size_t chunks, size_t tile_chunks, double v) {
    const size_t G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    const double2 val = make_double2(v, v);
    if (MODE == 0) {
        for (size_t g = b * 256 + t; g < chunks; g += G * 256) out[g] = val;
    } else if (MODE == 1) {
        const size_t per = (chunks + G - 1) / G, lo = b * per, hi = lo + per < chunks ? lo + per : chunks;
        for (size_t g = lo + t; g < hi; g += 256) out[g] = val;
    } else if (MODE == 2) {
        const size_t x = b % 8, j = b / 8, gx = G / 8;
        const size_t per = (chunks + 7) / 8, lo = x * per, hi = lo + per < chunks ? lo + per : chunks;
        for (size_t g = lo + j * 256 + t; g < hi; g += gx * 256) out[g] = val;
    } else {
        const size_t ntiles = (chunks + tile_chunks - 1) / tile_chunks;
        for (size_t i = b; i < ntiles; i += G) {
            size_t tile = i;
            if (MODE == 4) tile = (i * 2654435761ull) % ntiles;  // a permutation only when gcd = 1; good enough for timing
            if (MODE == 5) { const size_t blk = i / 64, r = i % 64; tile = blk * 64 + (r % 8) * 8 + r / 8; }  // XCD x gets tiles 8x..8x+7 of every 64
            if (MODE == 6) {
                const size_t x = b % 8, j = b / 8, gx = G / 8, per = (ntiles + 7) / 8;
                const size_t k = j + ((i - b) / G) * gx;  // this workgroup's k-th tile inside its XCD's eighth
                tile = k < per ? x * per + k : ntiles;
                // the loop variable i only counts iterations here; stop once past the eighth
                if (k >= per) break;
            }
            if (tile >= ntiles) continue;
            const size_t lo = tile * tile_chunks, hi = lo + tile_chunks < chunks ? lo + tile_chunks : chunks;
            for (size_t g = lo + t; g < hi; g += 256) out[g] = val;
        }
    }
}

template <int MODE>
static float run(double2 *buf, size_t bytes, int grid, size_t tile_bytes, int reps, hipEvent_t e0, hipEvent_t e1) {
    hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, tile_bytes / 16, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, tile_bytes / 16, 1.5);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return bytes / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 20.0;
    int count = argc > 2 ? atoi(argv[2]) : 8;
    int grid = argc > 3 ? atoi(argv[3]) : 1024;
    size_t tile_bytes = argc > 4 ? (size_t)atol(argv[4]) : 20480;
    size_t bytes = (size_t)(gb * 1e9) / 4096 * 4096;
    std::vector<double2 *> buf(count);
    for (int i = 0; i < count; ++i) CK(hipMalloc(&buf[i], bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int reps = bytes > (size_t)5e9 ? 3 : (bytes > (size_t)5e8 ? 20 : 200);
    printf("%.3f GB x %d buffers, grid %d, tile %zu B, TB/s per mode (0 flat, 1 blocked, 2 XCD-blocked, 3 tiles, 4 scrambled tiles, 5 tiles 8-per-XCD, 6 XCD-blocked tiles)\n",
           bytes / 1e9, count, grid, tile_bytes);
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < count; ++i) {
            float r0 = run<0>(buf[i], bytes, grid, tile_bytes, reps, e0, e1), r1 = run<1>(buf[i], bytes, grid, tile_bytes, reps, e0, e1);
            float r2 = run<2>(buf[i], bytes, grid, tile_bytes, reps, e0, e1), r3 = run<3>(buf[i], bytes, grid, tile_bytes, reps, e0, e1);
            float r4 = run<4>(buf[i], bytes, grid, tile_bytes, reps, e0, e1), r5 = run<5>(buf[i], bytes, grid, tile_bytes, reps, e0, e1);
            float r6 = run<6>(buf[i], bytes, grid, tile_bytes, reps, e0, e1);
            printf("round %d buffer %2d at %p: %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f\n", round, i, (void *)buf[i], r0, r1, r2, r3, r4, r5, r6);
            fflush(stdout);
        }
    return 0;
}
