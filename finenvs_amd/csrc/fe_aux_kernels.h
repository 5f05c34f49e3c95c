// fe_aux_kernels.h -- part of fe_env.hip (one translation unit; see the overview there): descriptor / render kernels, init kernels (log-returns, day tables), trajectory kernels.
#pragma once
#include "fe_device_common.h"
#include "fe_step_kernel.h"

namespace {

// descriptors of the current state's observation (reset() semantics), one lane per sleeve
template <bool SINGLE>
__global__ __launch_bounds__(kBlock) void fe_describe_kernel(const Params p, int64_t *obs_src, double *obs_pos) {
    const int A = SINGLE ? 1 : p.A;
    const int64_t NA = p.N * A;
    const int64_t rs = 4 * (int64_t)A;
    for (int64_t sl = blockIdx.x * (int64_t)kBlock + threadIdx.x; sl < NA; sl += (int64_t)gridDim.x * kBlock) {
        const int64_t n = SINGLE ? sl : sl / A;
        const int a = SINGLE ? 0 : (int)(sl - n * A);
        const int64_t idx = p.env_idx[n], s0 = p.spot0[n];
        int64_t last = s0 + p.W - 1;
        last = last < p.L ? last : p.L - 1;
        const double C = p.P[(idx * p.L + last) * rs + 4 * a + 3];
        obs_pos[sl] = (double)(p.lng[sl] - p.sht[sl]) * C / p.S;
        if (a == 0) {
            const int64_t s0c = s0 + p.W <= p.L ? s0 : p.L - p.W;
            obs_src[n] = (idx * p.L + s0c) * rs;
        }
    }
}

// fe_env_set_day: env_idx[env] = day, stream-ordered; the day travels as a kernel argument (no host buffer to outlive)
__global__ void fe_set_day_kernel(int64_t *env_idx, int64_t env, int64_t day) { env_idx[env] = day; }

// debug check of foreign descriptors (fe_env_check_descriptors): out[0] = how many are invalid, out[1] = the smallest
// invalid index (initialised to count by the host)
__global__ __launch_bounds__(kBlock) void fe_check_descriptors_kernel(const int64_t *__restrict__ obs_src, int64_t count,
                                                                      int64_t row_elems, int64_t W, int64_t rows,
                                                                      unsigned long long *out) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
        const int64_t s = obs_src[i];
        const bool ok = s >= 0 && s % row_elems == 0 && s / row_elems + W <= rows;
        if (!ok) {
            atomicAdd(&out[0], 1ull);
            atomicMin(&out[1], (unsigned long long)i);
        }
    }
}

// materialise the observation a pair of descriptor arrays stands for (phase 2 alone)
template <typename OT, int VEC, bool SINGLE>
__global__ __launch_bounds__(kBlock) void fe_render_kernel(const Params p, const int64_t *obs_src, const double *obs_pos) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const TileLds l = carve_lds(smem + 4 * kStageBytes, EB, EB * A);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    OT *stage = reinterpret_cast<OT *>(smem + wave * kStageBytes);
    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        for (int i = tid; i < ebt; i += kBlock) l.src[i] = obs_src[n0 + i];
        for (int i = tid; i < ebt * A; i += kBlock) l.pos[i] = obs_pos[n0 * A + i];
        __syncthreads();
        stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems,
                                     lane, wave);
        __syncthreads();
    }
}

// ---- a18: 100*ln(H/O, L/O, C/O), 100*ln(O_t/C_{t-1}) over the whole series, TSE:179-194 ----
__global__ __launch_bounds__(kBlock) void fe_logret_kernel(const double *__restrict__ prices,
                                                           double *__restrict__ out, int64_t T, int32_t A) {
    const int64_t total = T * A;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t t = i / A;
        const double4 p = *reinterpret_cast<const double4 *>(prices + 4 * i);
        const double prev = (t == 0) ? p.x : prices[4 * (i - A) + 3];  // previous close, TSE:188-190
        double4 o;
        o.x = 100.0 * log(p.x / prev);
        o.y = 100.0 * log(p.y / p.x);
        o.z = 100.0 * log(p.z / p.x);
        o.w = 100.0 * log(p.w / p.x);
        *reinterpret_cast<double4 *>(out + 4 * i) = o;
    }
}

// ---- a18 on the padded (D, L, 4A) price table, for fe_env_create(logret = NULL) ----
// Same transform per row; the previous close of a day's row 0 lies outside its slice, so row 0 takes
// the rule the reference applies to the first row of the series (open over open = 0, TSE:188-190).
// NaN padding rows stay NaN (log of NaN).
__global__ __launch_bounds__(kBlock) void fe_logret_tables_kernel(const double *__restrict__ P,
                                                                  double *__restrict__ out, int64_t D, int64_t L,
                                                                  int32_t A) {
    const int64_t total = D * L * A;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = i / A;
        const int64_t r = row % L;
        const double4 p = *reinterpret_cast<const double4 *>(P + 4 * i);
        const double prev = (r == 0) ? p.x : P[4 * (i - A) + 3];
        double4 o;
        o.x = 100.0 * log(p.x / prev);
        o.y = 100.0 * log(p.y / p.x);
        o.z = 100.0 * log(p.z / p.x);
        o.w = 100.0 * log(p.w / p.x);
        *reinterpret_cast<double4 *>(out + 4 * i) = o;
    }
}

// ---- a19: per-day slices, NaN-padded, TSE:196-216 ----
__global__ __launch_bounds__(kBlock) void fe_tables_kernel(const double *__restrict__ series,
                                                           const int64_t *__restrict__ starts,
                                                           const int64_t *__restrict__ stops, int64_t D,
                                                           int64_t L, int32_t A, double *__restrict__ out) {
    const int64_t rs = 4 * (int64_t)A;
    const int64_t total = D * L * rs;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = i / rs, k = i - row * rs;
        const int64_t d = row / L, r = row - d * L;
        const int64_t st = starts[d];
        out[i] = (r <= stops[d] - st) ? series[(st + r) * rs + k] : nan;
    }
}

// ---- f4: fixed-order reduction of the per-env episode statistics (fe_env_stats_reduce) ----
// acc is (N, 3) f64: per env the count, sum and sum of squares of its finished training episodes (the step kernels
// add to env n's three slots, one writer per slot).  ONE workgroup of kStatsLanes lanes adds the envs up in an order
// that is part of the contract (the test oracle restates it as fo_stats_reduce): per column k, lane t sums
// x[t][k], x[t + 1024][k], ... in ascending order starting from +0.0, then a halving tree s[t] += s[t + stride],
// stride = 512 ... 1.  Same bits whatever the step kernel's tile walk, launch geometry or form was.  Log-time only
// (PPO_agent.py:146-163).
constexpr int kStatsLanes = 1024;
__global__ __launch_bounds__(kStatsLanes) void fe_stats_reduce_kernel(const double *__restrict__ acc, int64_t N,
                                                                      double *__restrict__ out) {
    __shared__ double sh[3][kStatsLanes];
    const int t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int64_t j = t;
    // two envs' triples in flight, added in index order
    for (; j + (int64_t)kStatsLanes < N; j += 2 * (int64_t)kStatsLanes) {
        const double *x = acc + 3 * j, *y = acc + 3 * (j + kStatsLanes);
        const double a0 = x[0], a1 = x[1], a2 = x[2], b0 = y[0], b1 = y[1], b2 = y[2];
        s0 += a0; s1 += a1; s2 += a2;
        s0 += b0; s1 += b1; s2 += b2;
    }
    for (; j < N; j += kStatsLanes) {
        const double *x = acc + 3 * j;
        s0 += x[0]; s1 += x[1]; s2 += x[2];
    }
    sh[0][t] = s0; sh[1][t] = s1; sh[2][t] = s2;
    __syncthreads();
    for (int stride = kStatsLanes / 2; stride >= 1; stride >>= 1) {
        if (t < stride)
            for (int k = 0; k < 3; ++k) sh[k][t] += sh[k][t + stride];
        __syncthreads();
    }
    if (t < 3) out[t] = sh[t][0];
}

// ---- f1: trajectory slot store ----
__global__ __launch_bounds__(kBlock) void fe_traj_store_kernel(int64_t N, int64_t NA,
                                                               const float *__restrict__ actions,
                                                               const double *__restrict__ rewards,
                                                               const int32_t *__restrict__ dones,
                                                               float *__restrict__ ta, double *__restrict__ tr,
                                                               int32_t *__restrict__ td) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < NA; i += (int64_t)gridDim.x * kBlock) {
        ta[i] = actions[i];
        if (i < N) {
            tr[i] = rewards[i];
            td[i] = dones[i];
        }
    }
}

// ---- f1: discounted returns + advantages, one reverse scan per env (buffer.py:80-100).
// dtype discipline of the reference: (1 - dones) * gamma is f32; the first product with the
// f32 last_values is an f32 product, later ones are f64; returns/advantages are stored f32.
template <int U>
__global__ __launch_bounds__(kBlock) void fe_traj_returns_kernel(const double *__restrict__ rewards,
                                                                 const int32_t *__restrict__ dones,
                                                                 const float *__restrict__ values,
                                                                 const float *__restrict__ last_values,
                                                                 int64_t T, int64_t N, float g32,
                                                                 float *__restrict__ returns,
                                                                 float *__restrict__ adv) {
    // The scan itself is serial per env (the reference's rounding order), but its loads are not: U steps' inputs are
    // fetched together before the U dependent updates, so that at 64k envs (one workgroup per CU) a chunk of T steps
    // costs T / U memory round trips instead of T (64k envs x 128 steps: 49.8 us at U = 1, 34.0 at U = 4; at 1M envs
    // the chip is full anyway and the extra registers cost 7 % at T = 16, so the host picks U = 1 there).
    for (int64_t n = blockIdx.x * (int64_t)kBlock + threadIdx.x; n < N; n += (int64_t)gridDim.x * kBlock) {
        double R = 0.0;
        const float last = last_values[n];
        for (int64_t t0 = T - 1; t0 >= 0; t0 -= U) {
            double rw[U];
            int32_t dn[U];
            float vl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 - u;
                rw[u] = 0.0; dn[u] = 0; vl[u] = 0.0f;
                if (t >= 0) {
                    rw[u] = rewards[t * N + n];
                    dn[u] = dones[t * N + n];
                    if (adv) vl[u] = values[t * N + n];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 - u;
                if (t < 0) break;
                const float factor = (float)(1 - dn[u]) * g32;
                if (t == T - 1)
                    R = rw[u] + (double)(factor * last);
                else
                    R = rw[u] + (double)factor * R;
                const float r32 = (float)R;
                returns[t * N + n] = r32;
                if (adv) adv[t * N + n] = r32 - vl[u];
            }
        }
    }
}

}  // namespace
