/*
 * fe_c_demo.c -- the C ABI of include/finenvs_amd.h driven from plain C + the HIP runtime:
 * no Python, no torch.  Builds a small synthetic price series on the host, lets the library
 * build the log-return transform and the per-day tables on the GPU, creates an env, steps it with
 * a deterministic action pattern and prints checksums (tests/test_c_host_gpu.py compares them with
 * the same run through the Python binding).
 *
 *   gcc -O2 examples/c_host/fe_c_demo.c -I include -I /opt/rocm/include -L finenvs_amd/csrc -lfinenvs_amd \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/finenvs_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm \
 *       -o /tmp/fe_c_demo && /tmp/fe_c_demo 4096 16 200 [notify]
 *
 * With a fourth argument "notify" the steps go through fe_env_step_notify: the host learns whether the evaluation env
 * finished (the reference's per-step `if self.dones[-1].item():`, TSE:510) by polling a coherent host word that the kernel
 * writes a few microseconds into the launch, instead of copying dones back; every printed number must be the same.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "finenvs_amd.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define FECK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, r_, fe_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4096;
    const int32_t W = argc > 2 ? atoi(argv[2]) : 16;
    const int steps = argc > 3 ? atoi(argv[3]) : 200;
    const int notify = argc > 4 && argv[4][0] == 'n';
    const int promoted = argc > 4 && argv[4][0] == 'p'; /* float64 actions: the reference's f64 promotion, fe_env_step_promoted */
    const int32_t A = 1;
    const int64_t days = 6, bars = 50, T = days * bars;
    if (fe_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 1; }

    /* a deterministic market-hours series: day d, bar b */
    double *h_series = (double *)malloc(sizeof(double) * T * 4);
    double px = 100.0;
    for (int64_t t = 0; t < T; ++t) {
        double o = px * (1.0 + 0.0007 * sin(0.37 * (double)t));
        double c = o * (1.0 + 0.0009 * cos(0.11 * (double)t));
        double hi = (o > c ? o : c) * 1.0004, lo = (o < c ? o : c) * 0.9996;
        h_series[4 * t + 0] = o; h_series[4 * t + 1] = hi; h_series[4 * t + 2] = lo; h_series[4 * t + 3] = c;
        px = c;
    }
    /* episode bounds exactly as TSE:127-152: rows [first - W, last] per day, days without history skipped */
    int64_t h_starts[16], h_stops[16], D = 0, L = 0;
    for (int64_t d = 0; d < days; ++d) {
        int64_t first = d * bars, last = first + bars - 1, start = first - W;
        if (start < 0) continue;
        h_starts[D] = start; h_stops[D] = last;
        if (last - start + 1 > L) L = last - start + 1;
        ++D;
    }
    double *d_series, *d_lr, *d_P, *d_LR;
    int64_t *d_starts, *d_stops;
    HIPCK(hipMalloc((void **)&d_series, sizeof(double) * T * 4));
    HIPCK(hipMalloc((void **)&d_lr, sizeof(double) * T * 4));
    HIPCK(hipMalloc((void **)&d_P, sizeof(double) * D * L * 4));
    HIPCK(hipMalloc((void **)&d_LR, sizeof(double) * D * L * 4));
    HIPCK(hipMalloc((void **)&d_starts, sizeof(int64_t) * D));
    HIPCK(hipMalloc((void **)&d_stops, sizeof(int64_t) * D));
    HIPCK(hipMemcpy(d_series, h_series, sizeof(double) * T * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_starts, h_starts, sizeof(int64_t) * D, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_stops, h_stops, sizeof(int64_t) * D, hipMemcpyHostToDevice));
    FECK(fe_build_logret(d_series, d_lr, T, A, NULL));
    FECK(fe_build_tables(d_series, d_starts, d_stops, D, L, A, d_P, NULL));
    FECK(fe_build_tables(d_lr, d_starts, d_stops, D, L, A, d_LR, NULL));

    /* state: env n -> day n mod D, everything else as TSE:258-269 */
    int64_t *h_idx = (int64_t *)malloc(sizeof(int64_t) * N);
    float *h_cash = (float *)malloc(sizeof(float) * N);
    for (int64_t n = 0; n < N; ++n) { h_idx[n] = n % D; h_cash[n] = 10000.0f; }
    int64_t *d_idx, *d_spot, *d_counters;
    float *d_cash, *d_long, *d_short, *d_act;
    double *d_margin, *d_obs, *d_rew;
    int32_t *d_done;
    HIPCK(hipMalloc((void **)&d_idx, sizeof(int64_t) * N));
    HIPCK(hipMalloc((void **)&d_spot, sizeof(int64_t) * N));
    HIPCK(hipMalloc((void **)&d_counters, sizeof(int64_t) * 2));
    HIPCK(hipMalloc((void **)&d_cash, sizeof(float) * N));
    HIPCK(hipMalloc((void **)&d_long, sizeof(float) * N));
    HIPCK(hipMalloc((void **)&d_short, sizeof(float) * N));
    HIPCK(hipMalloc((void **)&d_act, sizeof(float) * N));
    HIPCK(hipMalloc((void **)&d_margin, sizeof(double) * N));
    HIPCK(hipMalloc((void **)&d_obs, sizeof(double) * N * W * 5));
    HIPCK(hipMalloc((void **)&d_rew, sizeof(double) * N));
    HIPCK(hipMalloc((void **)&d_done, sizeof(int32_t) * N));
    HIPCK(hipMemcpy(d_idx, h_idx, sizeof(int64_t) * N, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_cash, h_cash, sizeof(float) * N, hipMemcpyHostToDevice));
    HIPCK(hipMemset(d_spot, 0, sizeof(int64_t) * N));
    HIPCK(hipMemset(d_counters, 0, sizeof(int64_t) * 2));
    HIPCK(hipMemset(d_long, 0, sizeof(float) * N));
    HIPCK(hipMemset(d_short, 0, sizeof(float) * N));
    HIPCK(hipMemset(d_margin, 0, sizeof(double) * N));

    fe_config cfg = {0};
    cfg.N = N; cfg.D = D; cfg.L = L; cfg.W = W; cfg.A = A; cfg.max_shares = 5; cfg.evaluate = 0;
    cfg.starting_balance = 10000.0; cfg.commission = 0.01; cfg.init_margin = 1.5; cfg.maint_margin = 0.25;
    cfg.obs_is_f32 = 0; cfg.redraw_mode = 1; cfg.seed = 42; cfg.eval_env = N - 1;
    fe_env *env = NULL;
    FECK(fe_env_create(&cfg, d_P, d_LR, &env));
    FECK(fe_env_bind_state(env, d_idx, d_spot, d_cash, d_long, d_short, d_margin, NULL, NULL, d_counters));
    FECK(fe_env_reset_obs(env, d_obs, NULL));

    /* episode statistics fused into the step (PPO_agent.py:120-132) + their fixed-order read-out (146-163) */
    float *d_running, *d_evalret;
    double *d_acc, *d_sums, *d_act64;
    HIPCK(hipMalloc((void **)&d_running, sizeof(float) * N));
    HIPCK(hipMalloc((void **)&d_acc, sizeof(double) * 3 * N));
    HIPCK(hipMalloc((void **)&d_evalret, sizeof(float) * 2));
    HIPCK(hipMalloc((void **)&d_sums, sizeof(double) * 3));
    HIPCK(hipMalloc((void **)&d_act64, sizeof(double) * N));
    HIPCK(hipMemset(d_running, 0, sizeof(float) * N));
    HIPCK(hipMemset(d_acc, 0, sizeof(double) * 3 * N));
    HIPCK(hipMemset(d_evalret, 0, sizeof(float) * 2));
    FECK(fe_env_bind_stats(env, d_running, d_acc, d_evalret));
    double *h_act64 = (double *)malloc(sizeof(double) * N);

    float *h_act = (float *)malloc(sizeof(float) * N);
    double *h_rew = (double *)malloc(sizeof(double) * N);
    int32_t *h_done = (int32_t *)malloc(sizeof(int32_t) * N);
    double rew_sum = 0.0;
    long long dones = 0, eval_dones_flag = 0, eval_dones = 0;
    uint64_t *flag = NULL;
    if (notify) FECK(fe_host_flag_create(&flag));
    for (int s = 0; s < steps; ++s) {
        for (int64_t n = 0; n < N; ++n) h_act[n] = (float)sin(0.013 * (double)(n + 1) * (double)(s + 1));
        HIPCK(hipMemcpy(d_act, h_act, sizeof(float) * N, hipMemcpyHostToDevice));
        if (notify) {
            const uint64_t seq = (uint64_t)s + 1;
            FECK(fe_env_step_notify(env, d_act, d_obs, d_rew, d_done, flag, seq, NULL));
            uint64_t v;
            /* the launch is still running; only the flag is awaited.  Bounded: every 2^16 polls ask the stream -- a launch that
             * failed asynchronously, or finished without ever writing the flag, must end the wait with an error, not hang */
            unsigned long spins = 0;
            while (((v = *(volatile uint64_t *)flag) >> 1) != seq) {
                if ((++spins & 0xFFFF) == 0) {
                    hipError_t q = hipStreamQuery(NULL);
                    if (q != hipErrorNotReady) {
                        if (((v = *(volatile uint64_t *)flag) >> 1) == seq) break;
                        fprintf(stderr, "step %d: the stream is %s but the host flag never carried seq %llu\n", s,
                                q == hipSuccess ? "idle" : hipGetErrorString(q), (unsigned long long)seq);
                        return 5;
                    }
                }
            }
            eval_dones_flag += (long long)(v & 1);
        } else if (promoted) {
            /* f64 actions on two steps of three; the f32 steps in between run on the promoted env too (sticky, as in the reference) */
            const int f64 = s % 3 != 2;
            if (f64) {
                for (int64_t n = 0; n < N; ++n) h_act64[n] = sin(0.013 * (double)(n + 1) * (double)(s + 1));
                HIPCK(hipMemcpy(d_act64, h_act64, sizeof(double) * N, hipMemcpyHostToDevice));
            }
            FECK(fe_env_step_promoted(env, f64 ? (const void *)d_act64 : (const void *)d_act, f64, d_obs, d_rew, d_done, NULL, NULL, NULL,
                                      NULL, 0, NULL));
        } else {
            FECK(fe_env_step(env, d_act, d_obs, d_rew, d_done, NULL));
        }
        HIPCK(hipMemcpy(h_rew, d_rew, sizeof(double) * N, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(h_done, d_done, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
        for (int64_t n = 0; n < N; ++n) { rew_sum += h_rew[n]; dones += h_done[n]; }
        eval_dones += h_done[N - 1];
    }
    if (notify && eval_dones_flag != eval_dones) { fprintf(stderr, "host flag saw %lld evaluation-env dones, dones[N-1] %lld\n", eval_dones_flag, eval_dones); return 4; }
    HIPCK(hipMemcpy(h_cash, d_cash, sizeof(float) * N, hipMemcpyDeviceToHost));
    double cash_sum = 0.0;
    for (int64_t n = 0; n < N; ++n) cash_sum += (double)h_cash[n];
    double *h_obs = (double *)malloc(sizeof(double) * W * 5);
    HIPCK(hipMemcpy(h_obs, d_obs + (N - 1) * (int64_t)W * 5, sizeof(double) * W * 5, hipMemcpyDeviceToHost));
    double obs_sum = 0.0;
    for (int i = 0; i < W * 5; ++i) obs_sum += h_obs[i];
    int32_t grid, block, tile, lds;
    FECK(fe_env_launch_info(env, &grid, &block, &tile, &lds));
    printf("abi=%d N=%lld W=%d D=%lld L=%lld steps=%d grid=%d tile=%d\n", fe_version(), (long long)N, W, (long long)D,
           (long long)L, steps, grid, tile);
    double h_sums[3];
    FECK(fe_env_stats_reduce(env, d_sums, NULL));
    HIPCK(hipMemcpy(h_sums, d_sums, sizeof(h_sums), hipMemcpyDeviceToHost));
    printf("reward_sum=%.17g dones=%lld cash_sum=%.17g last_obs_sum=%.17g eval_dones=%lld stat_episodes=%.17g stat_sum=%.17g stat_sumsq=%.17g\n",
           rew_sum, dones, cash_sum, obs_sum, eval_dones, h_sums[0], h_sums[1], h_sums[2]);
    if (flag) FECK(fe_host_flag_destroy(flag));
    FECK(fe_env_destroy(env));
    return 0;
}
