// GPU box: (1) is v_mfma_f32_16x16x4_f32 an exact fmaf chain over k = 0..3 like v_mfma_f32_32x32x2_f32 is over k = 0..1?
// (2) how often can a DEPENDENT MFMA issue on an otherwise idle SIMD (one wavefront, one accumulator chain)?
// build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_16x16x4(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;  // A: row l % 16, k = l / 16;  B: col l % 16, k = l / 16;  C/D: rows 4 (l / 16) + i, col l % 16
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C[(4 * (l / 16) + i) * 16 + l % 16];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = acc[i];
}

template <int KIND>
__global__ void k_latency(float a, float b, float *out, long long *cycles, int n) {
    f32x16 acc32;
    f32x4 acc16;
    for (int i = 0; i < 16; ++i) acc32[i] = 0.0f;
    for (int i = 0; i < 4; ++i) acc16[i] = 0.0f;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32, 0, 0, 0);
        else acc16 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = KIND == 0 ? acc32[0] : acc16[0];
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        cycles[0] = t1 - t0;
        cycles[1] = w1 - w0;  // 100 MHz
    }
}

int main() {
    float hA[64], hB[64], hC[256], hD[256];
    srand(7);
    auto rnd = []() { return (float)ldexp((double)(rand() % 2000001 - 1000000) / 1e6, rand() % 25 - 12); };
    int mismatch_seq = 0, mismatch_rev = 0, mismatch_pair = 0;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { hA[i] = rnd(); hB[i] = rnd(); }
        for (int i = 0; i < 256; ++i) hC[i] = rnd();
        hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_16x16x4, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                float s = hC[r * 16 + c], v = hC[r * 16 + c];
                for (int k = 0; k < 4; ++k) s = fmaf(hA[r * 4 + k], hB[k * 16 + c], s);
                for (int k = 3; k >= 0; --k) v = fmaf(hA[r * 4 + k], hB[k * 16 + c], v);
                const float p = (float)((double)hC[r * 16 + c] + ((double)hA[r * 4] * hB[c] + (double)hA[r * 4 + 1] * hB[16 + c]) +
                                        ((double)hA[r * 4 + 2] * hB[32 + c] + (double)hA[r * 4 + 3] * hB[48 + c]));
                const float d = hD[r * 16 + c];
                mismatch_seq += memcmp(&d, &s, 4) != 0;
                mismatch_rev += memcmp(&d, &v, 4) != 0;
                mismatch_pair += memcmp(&d, &p, 4) != 0;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 over 200 random 16x16 tiles (51200 outputs): mismatches vs fmaf chain k=0..3: %d, vs k=3..0: %d, vs one f64 sum: %d\n",
           mismatch_seq, mismatch_rev, mismatch_pair);
    float *dout; long long *dcy, hcy[2];
    hipMalloc(&dout, 256); hipMalloc(&dcy, 16);
    for (int kind = 0; kind < 2; ++kind)
        for (int n : {256, 1024, 16384}) {
            if (kind == 0) hipLaunchKernelGGL(k_latency<0>, dim3(1), dim3(64), 0, 0, 1.0f, 0.5f, dout, dcy, n);
            else hipLaunchKernelGGL(k_latency<1>, dim3(1), dim3(64), 0, 0, 1.0f, 0.5f, dout, dcy, n);
            hipMemcpy(hcy, dcy, 16, hipMemcpyDeviceToHost);
            printf("%s: %d dependent MFMAs on one wavefront: %lld clock64 ticks = %.1f per MFMA; %.2f us wall -> %.0f ns per MFMA (clock64 runs at %.0f MHz)\n",
                   kind == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", n, hcy[0], (double)hcy[0] / n, hcy[1] / 100.0,
                   hcy[1] * 10.0 / n, hcy[0] / (hcy[1] / 100.0));
        }
    return 0;
}
