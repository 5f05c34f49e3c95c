// Microbenchmark (GPU box): how long does the dispatcher take to START the workgroups of one launch?
// Every workgroup stamps s_memrealtime (100 MHz) at entry, spins ~20 us so that nothing retires meanwhile, and
// exits.  Prints the start-time distribution over the grid for several block sizes / grids / LDS sizes.
// build: hipcc -O3 --offload-arch=gfx950 tools/dispatch_ramp.hip -o tools/dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k(unsigned long long *out, int spin_ticks) {
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
    smem[threadIdx.x] = (char)t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
}
int main() {
    unsigned long long *d; CK(hipMalloc(&d, 8192 * 8));
    std::vector<unsigned long long> h(8192);
    struct { int block, grid, lds; } cases[] = {{256, 512, 20608}, {256, 1024, 20608}, {256, 1792, 20736}, {256, 2048, 0}, {256, 2048, 20608},
                                                {512, 512, 41216}, {512, 1024, 41216}, {1024, 256, 65536}, {1024, 512, 65536}, {512, 512, 0}, {64, 4096, 5120}, {128, 2048, 10240}};
    for (auto c : cases) {
        std::vector<double> p10, p50, p90, mx;
        for (int rep = 0; rep < 12; ++rep) {
            hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), c.lds, 0, d, 2000);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d, c.grid * 8, hipMemcpyDeviceToHost));
            std::vector<double> t(c.grid);
            unsigned long long m = *std::min_element(h.begin(), h.begin() + c.grid);
            for (int i = 0; i < c.grid; ++i) t[i] = (h[i] - m) * 0.01;
            std::sort(t.begin(), t.end());
            if (rep >= 2) { p10.push_back(t[c.grid / 10]); p50.push_back(t[c.grid / 2]); p90.push_back(t[c.grid * 9 / 10]); mx.push_back(t.back()); }
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("block %4d grid %5d lds %6d (waves %5d): start p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f us\n", c.block, c.grid, c.lds,
               c.grid * c.block / 64, med(p10), med(p50), med(p90), med(mx));
    }
    return 0;
}
