// Store-pattern microbenchmark (round 6): which way of covering a 20 GB buffer with 16-byte stores reaches what torch's fill
// reaches (6.7 TB/s on MI355X), and which property of the step kernel's pattern costs the difference (it runs 5.4 - 5.9 TB/s on
// the same memory: profiles/r06_microbench/config3_launch_size.md).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_patterns tools/store_patterns.hip      (here; the binary travels with gpurun)
//   tools/store_patterns [GB]
//
// Every kernel writes the whole buffer once per launch, 16 B per lane per store, full 128-B lines.  Reported: TB/s over trains of
// back-to-back launches (HIP events), median of 5 trains.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

using u4 = __attribute__((ext_vector_type(4))) unsigned int;

template <bool NT>
__device__ __forceinline__ void st16(u4 *p, u4 v) {
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// (1) the fill clone: one workgroup of 256 lanes per `per_wg` bytes, every lane 16 B per store, workgroup b owns bytes [b * per_wg, ...)
template <bool NT>
__global__ __launch_bounds__(256) void k_flat(u4 *buf, size_t n16, int per_wg16) {
    const size_t base = (size_t)blockIdx.x * per_wg16;
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    for (int i = threadIdx.x; i < per_wg16; i += 256)
        if (base + i < n16) st16<NT>(buf + base + i, v);
}

// (1b) the same with any workgroup size: ONE 16-byte store per lane, the workgroup writes block x 16 contiguous bytes and ends
__global__ void k_one_store(u4 *buf, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    if (i < n16) buf[i] = v;
}

// (1c) one store per lane, but the workgroup first does what a render would: two dependent 8-byte loads from a small (L2-resident)
// table per lane + a little index arithmetic
__global__ void k_one_store_gather(u4 *buf, size_t n16, const double *__restrict__ table, unsigned mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const size_t e = 2 * i;
    const unsigned t0 = (unsigned)((e / 5) * 4 + (e % 5 < 4 ? e % 5 : 0)) & mask, t1 = (unsigned)(((e + 1) / 5) * 4 + ((e + 1) % 5 < 4 ? (e + 1) % 5 : 0)) & mask;
    const double a = table[t0], b = table[t1];
    union { double d[2]; u4 v; } u;
    u.d[0] = a;
    u.d[1] = b;
    buf[i] = u.v;
}

// (2) persistent, flat grid-stride: workgroup w writes chunks w, w + G, w + 2G ... of `per_wg` bytes
template <bool NT>
__global__ __launch_bounds__(256) void k_flat_persistent(u4 *buf, size_t n16, int per_wg16) {
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    const size_t chunks = (n16 + per_wg16 - 1) / per_wg16;
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const size_t base = c * per_wg16;
        for (int i = threadIdx.x; i < per_wg16; i += 256)
            if (base + i < n16) st16<NT>(buf + base + i, v);
    }
}

// (3) the step kernel's pattern: a workgroup owns a TILE of `tile16` x 16 bytes; its 4 wavefronts walk it in rounds of 4 x 5 KiB
// (wave j writes the 5-KiB chunk 4 r + j: five 1-KiB store instructions), tiles grid-strided (persistent) or one per workgroup
template <bool NT>
__global__ __launch_bounds__(256) void k_tiled(u4 *buf, size_t n16, int tile16, int chunk16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    const size_t tiles = (n16 + tile16 - 1) / tile16;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t tb = t * tile16;
        for (int c = wave * chunk16; c < tile16; c += 4 * chunk16)
            for (int i = lane; i < chunk16; i += 64)
                if (tb + c + i < n16 && c + i < tile16) st16<NT>(buf + tb + c + i, v);
    }
}

// (4) the tiled persistent walk with the number of stores a wavefront may have in flight bounded: after each 1-KiB store instruction
// the wave waits until at most `KEEP` of its stores are unacknowledged (s_waitcnt vmcnt: stores count there on gfx9)
template <int KEEP>
__global__ __launch_bounds__(256) void k_tiled_throttled(u4 *buf, size_t n16, int tile16, int chunk16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    const size_t tiles = (n16 + tile16 - 1) / tile16;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t tb = t * tile16;
        for (int c = wave * chunk16; c < tile16; c += 4 * chunk16)
            for (int i = lane; i < chunk16; i += 64) {
                if (tb + c + i < n16 && c + i < tile16) buf[tb + c + i] = v;
                if (KEEP == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 0);       // vmcnt(0) (expcnt / lgkmcnt fields left at "no wait")
                else if (KEEP == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 1);
                else if (KEEP == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
                else if (KEEP == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
                else if (KEEP == 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
            }
    }
}

// (5) persistent workgroups that draw their next chunk from ONE global ticket counter: chunks go out in address order whatever the
// workgroups' speeds are, as with the hardware's own dispatch of a huge grid -- the front stays tight
__global__ __launch_bounds__(256) void k_ticket(u4 *buf, size_t n16, int chunk16, unsigned *ticket) {
    __shared__ unsigned s_t;
    const u4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
    const size_t chunks = (n16 + chunk16 - 1) / chunk16;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1u);
        __syncthreads();
        const size_t c = s_t;
        __syncthreads();
        if (c >= chunks) break;
        const size_t base = c * chunk16;
        for (int i = threadIdx.x; i < chunk16; i += 256)
            if (base + i < n16) buf[base + i] = v;
    }
}

struct Res {
    const char *name;
    double tbps;
};

template <typename F>
static double timeit(F launch, size_t bytes) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 6; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 6);
    }
    std::sort(t.begin(), t.end());
    return bytes / (t[2] * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 20.0;
    const size_t bytes = (size_t)(gb * 1e9) / 1228800 * 1228800;  // whole 1.2 MB tiles
    const size_t n16 = bytes / 16;
    u4 *buf;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    printf("# buffer %.2f GB\n", bytes / 1e9);
    printf("hipMemsetAsync                                           %5.2f TB/s\n", timeit([&] { CHECK(hipMemsetAsync(buf, 1, bytes, 0)); }, bytes));
    for (int per_wg : {4096, 8192, 20480, 81920}) {
        const int p16 = per_wg / 16;
        const unsigned grid = (unsigned)((n16 + p16 - 1) / p16);
        printf("flat, one workgroup per %6d B (grid %9u)          %5.2f TB/s   nt %5.2f\n", per_wg, grid,
               timeit([&] { hipLaunchKernelGGL(k_flat<false>, dim3(grid), dim3(256), 0, 0, buf, n16, p16); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_flat<true>, dim3(grid), dim3(256), 0, 0, buf, n16, p16); }, bytes));
    }
    for (int block : {64, 128, 256, 320, 512, 1024}) {
        const unsigned grid = (unsigned)((n16 + block - 1) / block);
        printf("ONE store per lane, workgroup of %4d lanes = %5d B (grid %9u)   %5.2f TB/s\n", block, block * 16, grid,
               timeit([&] { hipLaunchKernelGGL(k_one_store, dim3(grid), dim3(block), 0, 0, buf, n16); }, bytes));
    }
    {
        double *table;
        const unsigned tn = 1u << 22;  // 32 MB of doubles: the size of the log-return table of config 3
        CHECK(hipMalloc(&table, tn * sizeof(double)));
        CHECK(hipMemset(table, 0, tn * sizeof(double)));
        for (int block : {256, 320}) {
            const unsigned grid = (unsigned)((n16 + block - 1) / block);
            printf("ONE store per lane + 2 gathered 8-B loads from a 32 MB table, %4d lanes   %5.2f TB/s\n", block,
                   timeit([&] { hipLaunchKernelGGL(k_one_store_gather, dim3(grid), dim3(block), 0, 0, buf, n16, table, tn - 1); }, bytes));
        }
        CHECK(hipFree(table));
    }
    for (int per_wg : {4096, 20480, 81920})
        for (unsigned grid : {1536u, 2048u, 4096u}) {
            const int p16 = per_wg / 16;
            printf("flat PERSISTENT, chunks of %6d B, grid %5u            %5.2f TB/s   nt %5.2f\n", per_wg, grid,
                   timeit([&] { hipLaunchKernelGGL(k_flat_persistent<false>, dim3(grid), dim3(256), 0, 0, buf, n16, p16); }, bytes),
                   timeit([&] { hipLaunchKernelGGL(k_flat_persistent<true>, dim3(grid), dim3(256), 0, 0, buf, n16, p16); }, bytes));
        }
    for (int tile : {20480, 76800, 153600, 614400, 1228800})
        for (int persistent : {1, 0}) {
            const int t16 = tile / 16;
            const size_t tiles = (n16 + t16 - 1) / t16;
            const unsigned grid = persistent ? 1536u : (unsigned)tiles;
            printf("TILED (4 waves x 5 KiB rounds), tile %7d B, %s grid %7u   %5.2f TB/s   nt %5.2f\n", tile, persistent ? "persistent    " : "one tile per WG",
                   grid, timeit([&] { hipLaunchKernelGGL(k_tiled<false>, dim3(grid), dim3(256), 0, 0, buf, n16, t16, 320); }, bytes),
                   timeit([&] { hipLaunchKernelGGL(k_tiled<true>, dim3(grid), dim3(256), 0, 0, buf, n16, t16, 320); }, bytes));
        }
    // the tile walked in other chunk sizes per wave (1 KiB = one store instruction ... 20 KiB)
    for (int chunk : {1024, 2048, 5120, 20480, 81920})
        printf("TILED persistent, tile 1228800 B, wave chunk %6d B        %5.2f TB/s   nt %5.2f\n", chunk,
               timeit([&] { hipLaunchKernelGGL(k_tiled<false>, dim3(1536), dim3(256), 0, 0, buf, n16, 1228800 / 16, chunk / 16); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled<true>, dim3(1536), dim3(256), 0, 0, buf, n16, 1228800 / 16, chunk / 16); }, bytes));
    for (unsigned grid : {1536u, 2048u}) {
        printf("TILED persistent, tile 1228800 B, 5 KiB chunks, grid %u, stores in flight per wave <= 0+1 / 1+1 / 2+1 / 4+1 / 8+1:  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f TB/s\n", grid,
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<0>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<1>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<2>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<4>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<8>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes));
    }
    for (unsigned grid : {256u, 512u, 768u, 1024u})
        printf("TILED persistent, tile 1228800 B, 5 KiB chunks, grid %4u (fewer wavefronts), unthrottled / <= 1 / <= 3 in flight:  %5.2f  %5.2f  %5.2f TB/s\n", grid,
               timeit([&] { hipLaunchKernelGGL(k_tiled<false>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<0>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes),
               timeit([&] { hipLaunchKernelGGL(k_tiled_throttled<2>, dim3(grid), dim3(256), 0, 0, buf, n16, 1228800 / 16, 320); }, bytes));
    {
        unsigned *ticket;
        CHECK(hipMalloc(&ticket, 4));
        for (int chunk : {4096, 16384, 65536, 262144, 1228800})
            for (unsigned grid : {1536u, 2048u}) {
                printf("persistent, chunks of %7d B drawn from ONE ticket counter, grid %u:   %5.2f TB/s\n", chunk, grid,
                       timeit([&] {
                           CHECK(hipMemsetAsync(ticket, 0, 4, 0));
                           hipLaunchKernelGGL(k_ticket, dim3(grid), dim3(256), 0, 0, buf, n16, chunk / 16, ticket);
                       }, bytes));
            }
        CHECK(hipFree(ticket));
    }
    CHECK(hipFree(buf));
    return 0;
}
