// Microbenchmark (GPU box): does the ORDER in which physical memory backs a big buffer decide its write bandwidth?
// tools/placement_study.py found that a 20 GB buffer's write bandwidth (5.4 ... 6.9 TB/s for the same kernel) is a
// property of its physical pages: virtual offsets and launch geometry do not matter, physically contiguous allocations are
// uniformly slow, scattered ones are sometimes fast.  Here a buffer is assembled from fixed-size physical granules with
// the HIP virtual-memory API and the SAME granules are mapped (a) in creation order, (b) in a random permutation,
// (c) in a bit-reversed-like stride order; a tile-striding store kernel (what the step kernel does) is timed on each.
// build: hipcc -O3 --offload-arch=gfx950 tools/vmm_shuffle.hip -o tools/vmm_shuffle
// usage: vmm_shuffle <GB> <granule MiB> <tile KiB> <grid>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_store(dbl2 *out, size_t chunks, size_t tile_chunks, double v) {
    const size_t G = gridDim.x, t = threadIdx.x;
    const size_t ntiles = (chunks + tile_chunks - 1) / tile_chunks;
    const dbl2 val = {v, v};
    for (size_t tile = blockIdx.x; tile < ntiles; tile += G) {
        const size_t lo = tile * tile_chunks, hi = lo + tile_chunks < chunks ? lo + tile_chunks : chunks;
        for (size_t g = lo + t; g < hi; g += 256) __builtin_nontemporal_store(val, &out[g]);
    }
}

static double run(void *buf, size_t bytes, int grid, size_t tile_bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, (dbl2 *)buf, bytes / 16, tile_bytes / 16, 1.0);
    CK(hipDeviceSynchronize());
    double best = 0;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, (dbl2 *)buf, bytes / 16, tile_bytes / 16, 1.5 + r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double tb = bytes / (ms / 3 * 1e-3) / 1e12;
        if (tb > best) best = tb;
    }
    return best;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t gran_req = (size_t)(argc > 2 ? atol(argv[2]) : 2) << 20;
    const size_t tile_bytes = (size_t)(argc > 3 ? atol(argv[3]) : 600) << 10;
    const int grid = argc > 4 ? atoi(argv[4]) : 1536;
    int dev = 0; CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    size_t gran = (gran_req + gmin - 1) / gmin * gmin;
    const size_t n = (size_t)(gb * 1e9) / gran;
    const size_t bytes = n * gran;
    printf("%.2f GB = %zu granules of %zu KiB (driver minimum %zu, recommended %zu KiB), tile %zu KiB, grid %d\n", bytes / 1e9, n, gran >> 10, gmin >> 10, grec >> 10,
           tile_bytes >> 10, grid);
    // reference points: plain hipMalloc buffers
    for (int i = 0; i < 3; ++i) {
        void *p; CK(hipMalloc(&p, bytes));
        printf("hipMalloc buffer %d:            %5.2f TB/s\n", i, run(p, bytes, grid, tile_bytes));
        CK(hipFree(p));
    }
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], gran, &prop, 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::mt19937_64 rng(12345);
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = i;
        const char *name = "creation order";
        if (mode == 1 || mode == 2) { std::shuffle(order.begin(), order.end(), rng); name = mode == 1 ? "random permutation A" : "random permutation B"; }
        if (mode == 3) { for (size_t i = 0; i < n; ++i) order[i] = n - 1 - i; name = "reversed"; }
        if (mode == 4) {  // granule g of the buffer <- granule (g % 8) * (n / 8) + g / 8: eight far-apart regions interleaved
            const size_t per = n / 8;
            for (size_t g = 0; g < n; ++g) order[g] = g < per * 8 ? (g % 8) * per + g / 8 : g;
            name = "8 far-apart regions interleaved";
        }
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
        for (size_t i = 0; i < n; ++i) CK(hipMemMap((char *)va + i * gran, gran, 0, h[order[i]], 0));
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        printf("VMM, %-44s %5.2f TB/s\n", name, run(va, bytes, grid, tile_bytes));
        fflush(stdout);
        CK(hipMemUnmap(va, bytes));
        CK(hipMemAddressFree(va, bytes));
    }
    for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i]));
    return 0;
}
