// Microbenchmark (GPU box): variants of the observation-stream phase in isolation.
//   V0  current kernel's scheme: per 16-B output chunk two 8-B gathers + div-by-5 index math
//   V1  LDS transpose: 16-B coalesced table loads -> wave-private LDS image of 5-tuples ->
//       linear ds_read_b128 -> fully coalesced 16-B stores; no div-by-5
// build: hipcc -O3 --offload-arch=gfx950 tools/copy_variants.hip -o tools/copy_variants
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct FastDiv { uint32_t m, sh1, sh2, d; };
static FastDiv mk(uint32_t d) { FastDiv f; f.d = d; uint32_t l = 0; while ((1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1); f.sh1 = l < 1 ? l : 1; f.sh2 = l > 1 ? l - 1 : 0; return f; }
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) { uint32_t t = __umulhi(f.m, n); return (t + ((n - t) >> f.sh1)) >> f.sh2; }

struct P {
    const double *LR; const int64_t *src; const double *pos; double *obs;
    int64_t N, num_tiles; int32_t W, A, EB; uint32_t env_elems; FastDiv div_chunks, div_A, div_WA;
};

__global__ __launch_bounds__(256) void v0(const P p) {
    extern __shared__ __align__(16) unsigned char smem[];
    int64_t *s_src = (int64_t *)smem; double *s_pos = (double *)(s_src + p.EB);
    const int A = p.A, EB = p.EB, tid = threadIdx.x; const uint32_t chunks = p.env_elems / 2;
    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        int64_t n0 = tile * EB; int ebt = (p.N - n0) < EB ? (int)(p.N - n0) : EB;
        for (int i = tid; i < ebt; i += 256) s_src[i] = p.src[n0 + i];
        for (int i = tid; i < ebt * A; i += 256) s_pos[i] = p.pos[n0 * A + i];
        __syncthreads();
        const uint32_t total = (uint32_t)ebt * chunks;
        double2 *dst = (double2 *)(p.obs + n0 * (int64_t)p.env_elems);
#pragma unroll 4
        for (uint32_t g = tid; g < total; g += 256) {
            uint32_t ee = fdiv(g, p.div_chunks), c = g - ee * chunks;
            const double *src = p.LR + s_src[ee]; const double *posr = s_pos + ee * A;
            double o[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t el = c * 2 + i, t = el / 5u, k = el - 5u * t;
                double x = src[4u * t + (k < 4u ? k : 3u)];
                uint32_t aa = A == 1 ? 0u : t - fdiv(t, p.div_A) * (uint32_t)A;
                o[i] = k < 4u ? x : posr[aa];
            }
            dst[g] = make_double2(o[0], o[1]);
        }
        __syncthreads();
    }
}

// V1: each wave turns G groups of 64 tuples (32 B in, 40 B out each) per iteration
template <int G>
__global__ __launch_bounds__(256) void v1(const P p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = p.A, EB = p.EB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *stage = (double *)smem + wave * (G * 64 * 5);           // wave-private 5-tuple image
    int64_t *s_src = (int64_t *)((double *)smem + 4 * G * 64 * 5); double *s_pos = (double *)(s_src + EB);
    const uint32_t WA = (uint32_t)p.W * A;
    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        int64_t n0 = tile * EB; int ebt = (p.N - n0) < EB ? (int)(p.N - n0) : EB;
        for (int i = tid; i < ebt; i += 256) s_src[i] = p.src[n0 + i];
        for (int i = tid; i < ebt * A; i += 256) s_pos[i] = p.pos[n0 * A + i];
        __syncthreads();
        const uint32_t tuples = (uint32_t)ebt * WA;                  // tuples in this tile
        const uint32_t per_it = G * 64;
        double2 *dst = (double2 *)(p.obs + n0 * (int64_t)p.env_elems);
        for (uint32_t base = wave * per_it; base < tuples; base += 4 * per_it) {
            double4 v[G]; double pz[G];
#pragma unroll
            for (int gI = 0; gI < G; ++gI) {
                uint32_t t = base + gI * 64 + lane;
                uint32_t tc = t < tuples ? t : tuples - 1;
                uint32_t ee = fdiv(tc, p.div_WA), r = tc - ee * WA;
                uint32_t aa = A == 1 ? 0u : r - fdiv(r, p.div_A) * (uint32_t)A;
                v[gI] = *(const double4 *)(p.LR + s_src[ee] + 4u * r);
                pz[gI] = s_pos[ee * A + aa];
            }
#pragma unroll
            for (int gI = 0; gI < G; ++gI) {
                double *w = stage + (gI * 64 + lane) * 5;
                w[0] = v[gI].x; w[1] = v[gI].y; w[2] = v[gI].z; w[3] = v[gI].w; w[4] = pz[gI];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 64*G tuples = 160*G 16-byte chunks, linear
            const uint32_t nvalid = (tuples - base < per_it ? tuples - base : per_it) * 5u / 2u;  // chunks (tile tuple count*5 is even)
            const double2 *rd = (const double2 *)stage;
            double2 *o = dst + (size_t)base * 5u / 2u;
#pragma unroll
            for (uint32_t c = lane; c < (uint32_t)(160 * G); c += 64) {
                if (c < nvalid) o[c] = rd[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
}

int main() {
    struct Cfg { int64_t N; int A, W, EB; } cfgs[] = {{65536, 1, 64, 32}, {262144, 30, 64, 8}, {1048576, 30, 128, 8}, {65536, 1, 64, 64}, {65536,1,64,16}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto c : cfgs) {
        const int D = 64; const int64_t L = c.W + 390;
        size_t tab = (size_t)D * L * 4 * c.A;
        std::vector<double> h(tab); for (size_t i = 0; i < tab; ++i) h[i] = (double)(i % 1000) * 1e-3;
        double *LR; CK(hipMalloc(&LR, tab * 8)); CK(hipMemcpy(LR, h.data(), tab * 8, hipMemcpyHostToDevice));
        std::vector<int64_t> src(c.N); for (int64_t n = 0; n < c.N; ++n) src[n] = ((n % D) * L + 17) * 4 * c.A;
        int64_t *dsrc; CK(hipMalloc(&dsrc, c.N * 8)); CK(hipMemcpy(dsrc, src.data(), c.N * 8, hipMemcpyHostToDevice));
        std::vector<double> pos((size_t)c.N * c.A, 0.25); double *dpos; CK(hipMalloc(&dpos, pos.size() * 8)); CK(hipMemcpy(dpos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice));
        size_t env_elems = (size_t)c.W * 5 * c.A, obs_bytes = (size_t)c.N * env_elems * 8;
        int nb = obs_bytes < (size_t)50e9 ? 2 : 1; double *obs[2];
        for (int i = 0; i < nb; ++i) CK(hipMalloc(&obs[i], obs_bytes));
        P p; p.LR = LR; p.src = dsrc; p.pos = dpos; p.N = c.N; p.W = c.W; p.A = c.A; p.EB = c.EB;
        p.num_tiles = (c.N + c.EB - 1) / c.EB; p.env_elems = (uint32_t)env_elems;
        p.div_chunks = mk((uint32_t)(env_elems / 2)); p.div_A = mk(c.A); p.div_WA = mk(c.W * c.A);
        int reps = obs_bytes < (size_t)1e9 ? 200 : (obs_bytes < (size_t)50e9 ? 20 : 6);
        size_t lds0 = (size_t)c.EB * 8 + (size_t)c.EB * c.A * 8;
        auto run = [&](const char *name, auto launch) {
            for (int i = 0; i < 2; ++i) { p.obs = obs[i % nb]; launch(); }
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) { p.obs = obs[i % nb]; launch(); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
            printf("N=%-8lld A=%-2d W=%-3d EB=%-2d %-22s %8.4f ms  obs-write %5.2f TB/s\n", (long long)c.N, c.A, c.W, c.EB, name, ms / reps, obs_bytes / (ms / reps * 1e-3) / 1e12);
            fflush(stdout);
        };
        for (int g : {1536, 2048}) {
            char nm[64]; snprintf(nm, 64, "V0 gather8 grid=%d", g);
            run(nm, [&] { hipLaunchKernelGGL(v0, dim3(g), dim3(256), lds0, 0, p); });
            snprintf(nm, 64, "V1 lds G=2 grid=%d", g);
            run(nm, [&] { hipLaunchKernelGGL(v1<2>, dim3(g), dim3(256), lds0 + 4 * 2 * 64 * 40, 0, p); });
            snprintf(nm, 64, "V1 lds G=4 grid=%d", g);
            run(nm, [&] { hipLaunchKernelGGL(v1<4>, dim3(g), dim3(256), lds0 + 4 * 4 * 64 * 40, 0, p); });
        }
        // verify V1 == V0 on a sample
        {
            size_t nchk = 1 << 20; std::vector<double> a(nchk), b(nchk);
            p.obs = obs[0]; hipLaunchKernelGGL(v0, dim3(2048), dim3(256), lds0, 0, p); CK(hipDeviceSynchronize());
            CK(hipMemcpy(a.data(), obs[0] + (c.N * env_elems - nchk), nchk * 8, hipMemcpyDeviceToHost));
            CK(hipMemset(obs[0], 0, obs_bytes < (size_t)4e9 ? obs_bytes : (size_t)4e9));
            hipLaunchKernelGGL(v1<2>, dim3(2048), dim3(256), lds0 + 4 * 2 * 64 * 40, 0, p); CK(hipDeviceSynchronize());
            CK(hipMemcpy(b.data(), obs[0] + (c.N * env_elems - nchk), nchk * 8, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < nchk; ++i) bad += a[i] != b[i];
            std::vector<double> a2(nchk), b2(nchk);
            printf("   check V1==V0 tail: %zu mismatches\n", bad);
        }
        for (int i = 0; i < nb; ++i) CK(hipFree(obs[i]));
        CK(hipFree(LR)); CK(hipFree(dsrc)); CK(hipFree(dpos));
    }
    return 0;
}
