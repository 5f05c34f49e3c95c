// fe_rollout_kernels.h -- part of fe_env.hip (one translation unit; see the overview there): K-step fused rollouts with an in-kernel policy: account_keep, linear window form, table form, MLP head on MFMA.
#pragma once
#include "fe_device_common.h"
#include "fe_step_kernel.h"
#include "fe_activations.h"

namespace {

// Register-resident account state of one sleeve across the K steps of a fused rollout.
struct SleeveReg {
    int64_t idx, spot;  // env_indices[n], env_spots[n][0]
    float cash, lng, sht;
    double margin;
    int64_t obs_row;  // table row (idx * L + window start) of the observation the last step returned
    double obs_pos;   // its position feature for this sleeve
};

// One step of phases 1/1b with the state in registers (the fused rollout): same arithmetic and the
// same global side effects as account_core (rewards, dones, evaluate-mode metrics, statistics,
// redraw counter), but cash/shares/margin/spot/idx are only updated in `st`; cross-lane traffic
// (env-level done, redrawn day) goes through LDS.
template <bool SINGLE>
__device__ __forceinline__ void account_keep(const Params &p, const TileLds &l, int64_t *l_idx, int A, int e, int a,
                                             bool active, int64_t n, SleeveReg &st, float action, double *rew_out,
                                             int32_t *done_out) {
    const int64_t rs = 4 * (int64_t)A;
    const int W = p.W;
    const int64_t L = p.L;
    Sleeve s;
    int64_t s0 = 0;
    bool sdone = false;
    if (active) {
        s0 = st.spot + 1;  // TSE:281-282
        int64_t last = s0 + W - 1;
        last = last < L ? last : L - 1;
        const int64_t nxt = last + 1;
        const double4 bar = *reinterpret_cast<const double4 *>(p.P + (st.idx * L + last) * rs + 4 * a);
        double probe = 0.0;
        if (nxt < L) probe = p.LR[(st.idx * L + nxt) * rs + 4 * a];
        s.cash = st.cash;
        s.lng = st.lng;
        s.sht = st.sht;
        s.margin = st.margin;
        sleeve_step<false>(p, action, bar.x, bar.y, bar.z, bar.w, s);
        sdone = s.bankrupt | (nxt >= L) | (probe != probe);
        l.pos[e * A + a] = s.pos_obs;
        st.obs_pos = s.pos_obs;
        st.obs_row = st.idx * L + (s0 + W <= L ? s0 : L - W);
        if constexpr (!SINGLE) {
            l.rew[e * A + a] = s.rew;
            l.shr[e * A + a] = s.sht + s.lng;
            l.flg[e * A + a] = sdone ? 1 : 0;
        }
        if (a == 0) {
            const int64_t s0c = s0 + W <= L ? s0 : L - W;
            l.src[e] = (st.idx * L + s0c) * rs;
        }
    }
    bool any = sdone;
    int64_t new_idx = st.idx;
    if constexpr (!SINGLE) lds_barrier();
    if (active && a == 0) {
        double rew;
        if constexpr (SINGLE) {
            float fee = ((any ? 1.0f : 0.0f) * (s.sht + s.lng)) * p.c32;
            rew = s.rew - (double)fee;
        } else {
            any = false;
            for (int k = 0; k < A; ++k) any |= l.flg[e * A + k] != 0;
            rew = 0.0;
            for (int k = 0; k < A; ++k) {
                float fee = ((any ? 1.0f : 0.0f) * l.shr[e * A + k]) * p.c32;
                double r = l.rew[e * A + k] - (double)fee;
                rew = (k == 0) ? r : rew + r;
            }
        }
        if (any && !p.evaluate && p.redraw_mode == 1 && n == p.eval_env) {  // TSE:504-513
            unsigned long long ctr = p.counters[1];
            new_idx = (int64_t)(((uint64_t)philox_u32(p.seed, ctr) * (uint64_t)p.D) >> 32);
            p.counters[1] = ctr + 1;
        }
        if constexpr (!SINGLE) {
            l.any[e] = any ? 1 : 0;
            l_idx[e] = new_idx;
        }
        if (p.evaluate) {  // TSE:523-536
            const bool term = p.terminated[n] != 0;
            if (term) rew = 0.0;
            if (any && !term) {
                p.terminated[n] = 1;
                atomicAdd(&p.counters[0], 1ull);
            }
            p.ep_ret[n] = (float)((double)p.ep_ret[n] + rew);
        }
        rew_out[n] = rew;
        done_out[n] = any ? 1 : 0;
        if (p.run_ret) {
            float cr = (float)((double)p.run_ret[n] + rew);
            if (any) {
                if (n == p.eval_env) {
                    p.stat_eval[0] = cr;
                    p.stat_eval[1] += 1.0f;
                } else {
                    double *acc = p.stat_acc + 3 * n;  // per-env partial sums with one writer each, as in fe_step_kernel.h
                    atomicAdd(acc, 1.0);
                    atomicAdd(acc + 1, (double)cr);
                    atomicAdd(acc + 2, (double)cr * (double)cr);
                }
                cr = 0.0f;
            }
            p.run_ret[n] = cr;
        }
    }
    if constexpr (!SINGLE) {
        lds_barrier();
        if (active) {
            any = l.any[e] != 0;
            new_idx = l_idx[e];
        }
    }
    if (active) {  // episodic reset folded in, TSE:498-521
        st.cash = any ? p.S32 : s.cash;
        st.lng = any ? 0.0f : s.lng;
        st.sht = any ? 0.0f : s.sht;
        st.margin = any ? 0.0 : s.margin;
        st.spot = any ? 0 : s0;
        st.idx = new_idx;
    }
}

// A tile's account state moves into registers for the K steps of a fused rollout and goes back to HBM once per launch.
__device__ __forceinline__ SleeveReg rollout_load_state(const Params &p, bool active, int64_t n, int64_t sl) {
    SleeveReg st;
    st.idx = 0; st.spot = 0; st.cash = 0.0f; st.lng = 0.0f; st.sht = 0.0f; st.margin = 0.0;
    st.obs_row = 0; st.obs_pos = 0.0;
    if (active) {
        st.idx = p.env_idx[n];
        st.spot = p.spot0[n];
        st.cash = p.cash[sl];
        st.lng = p.lng[sl];
        st.sht = p.sht[sl];
        st.margin = p.margin[sl];
    }
    return st;
}

__device__ __forceinline__ void rollout_store_state(const Params &p, bool active, int a, int64_t n, int64_t sl,
                                                    const SleeveReg &st) {
    if (!active) return;
    p.cash[sl] = st.cash;
    p.lng[sl] = st.lng;
    p.sht[sl] = st.sht;
    p.margin[sl] = st.margin;
    if (a == 0) {
        p.env_idx[n] = st.idx;
        p.spot0[n] = st.spot;
    }
}

// ---- f2: K env steps per launch with an in-kernel linear policy ----
// The policy is the "observation projection" of the north star reduced to its simplest useful
// form: one weight per (window row, feature), shared by all assets,
//   action[n][a] = clamp(bias + sum_j sum_c obs[n][j][5a+c] * w[j][c], -1, 1)
// evaluated by one wavefront per (env, asset): lane l accumulates rows j = l, l+64, ... in row
// order (c = 0..4 inside a row), then a butterfly (xor 32,16,8,4,2,1) of wavefront shuffles sums
// the 64 partials.  The observation itself is never materialised: the policy reads the window
// straight from the L2-resident table through the same (src, pos) descriptors phase 2 uses.
struct RolloutArgs {
    const double *weights;  // (W, 5) f64
    double bias;
    int32_t K;
    int64_t *obs_src;    // (N)   in/out: descriptor of the current observation
    double *obs_pos;     // (N*A) in/out
    float *actions_out;  // (K, N*A) or null
    double *rew_out;     // (K, N)
    int32_t *done_out;   // (K, N)
};

__host__ __device__ inline size_t rollout_lds_bytes(int EB, int A, int W) {
    size_t S = (size_t)EB * A;
    size_t b = (size_t)EB * 8 + S * 8 + S * 8 + S * 4 + S * 4 + (size_t)EB * 4;  // TileLds
    b = (b + 7) & ~(size_t)7;
    b += (size_t)W * 5 * 8;  // weights
    b += S * 4;              // actions
    b = (b + 7) & ~(size_t)7;
    b += (size_t)EB * 8;     // redrawn day per env (A > 1)
    return (b + 15) & ~(size_t)15;
}

template <bool SINGLE>
__global__ __launch_bounds__(kBlock, 1) void fe_rollout_linear_kernel(const Params p, const RolloutArgs r) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    double *s_w = reinterpret_cast<double *>(smem + off);
    float *s_act = reinterpret_cast<float *>(s_w + (size_t)p.W * 5);
    int64_t *l_idx = reinterpret_cast<int64_t *>(
        smem + ((off + (size_t)p.W * 40 + (size_t)S * 4 + 7) & ~(size_t)7));
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    const int W = p.W;
    const int64_t NA = p.N * A;
    for (int i = tid; i < W * 5; i += kBlock) s_w[i] = r.weights[i];

    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active, n, sl);
        if (active) {
            if (a == 0) l.src[e] = r.obs_src[n];
            l.pos[e * A + a] = r.obs_pos[sl];
        }
        __syncthreads();
        const int pairs = ebt * A;
        for (int k = 0; k < r.K; ++k) {
            // policy: one wavefront per (env, asset) pair of the tile
            for (int q = wave; q < pairs; q += kBlock / 64) {
                const int ee = SINGLE ? q : (int)fdiv((uint32_t)q, p.div_A);
                const int aa = SINGLE ? 0 : q - ee * A;
                const double *src = p.LR + l.src[ee];
                const double pos = l.pos[q];
                double acc = 0.0;
                for (int j = lane; j < W; j += 64) {
                    const double4 v = *reinterpret_cast<const double4 *>(src + ((int64_t)j * A + aa) * 4);
                    const double *wr = s_w + j * 5;
                    acc += v.x * wr[0];
                    acc += v.y * wr[1];
                    acc += v.z * wr[2];
                    acc += v.w * wr[3];
                    acc += pos * wr[4];
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m, 64);
                double a64 = r.bias + acc;
                a64 = a64 < -1.0 ? -1.0 : (a64 > 1.0 ? 1.0 : a64);
                if (lane == 0) s_act[q] = (float)a64;
            }
            lds_barrier();
            const float act = active ? s_act[e * A + a] : 0.0f;
            if (active && r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                 r.done_out + (int64_t)k * p.N);
            lds_barrier();  // the new observation's descriptors are complete
        }
        rollout_store_state(p, active, a, n, sl, st);  // state and descriptors go back to HBM once per launch
        if (active) {
            r.obs_pos[sl] = l.pos[e * A + a];
            if (a == 0) r.obs_src[n] = l.src[e];
        }
        __syncthreads();
    }
}

// ---- f2, table form: the linear policy as a precomputed indicator ----
// For fixed weights the log-return part of the policy is a FIR filter over the day's series: one
// number per (day, window start, asset).  fe_policy_table_kernel evaluates it once per weight update
// (one wavefront per entry, the same lane/butterfly order as above over the four log-return
// features), then a K-step rollout needs two 8-byte lookups per sleeve and step:
//   action = clamp(bias + (table[row][a] + pos * wsum), -1, 1),  wsum = sum_j w[j][4] (same order).
// The split of the sum is part of THIS form's contract (it rounds differently from the window form).
__global__ __launch_bounds__(kBlock) void fe_policy_table_kernel(const Params p, const double *weights,
                                                                double *table, double *wsum) {
    const int A = p.A, W = p.W;
    const int64_t L = p.L;
    const int lane = threadIdx.x & 63;
    const int64_t gw = (blockIdx.x * (int64_t)kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t total = p.D * L * A;
    for (int64_t q = gw; q < total; q += nw) {
        const int64_t row = q / A;
        const int a = (int)(q - row * A);
        const int64_t s = row % L;
        double acc = 0.0;
        if (s + W <= L) {
            for (int j = lane; j < W; j += 64) {
                const double4 v = *reinterpret_cast<const double4 *>(p.LR + ((row + j) * A + a) * 4);
                const double *wr = weights + j * 5;
                acc += v.x * wr[0];
                acc += v.y * wr[1];
                acc += v.z * wr[2];
                acc += v.w * wr[3];
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m, 64);
        } else {
            acc = __longlong_as_double(0x7ff8000000000000ll);  // no window starts here
        }
        if (lane == 0) table[q] = acc;
    }
    if (gw == 0) {
        double acc = 0.0;
        for (int j = lane; j < W; j += 64) acc += weights[j * 5 + 4];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m, 64);
        if (lane == 0) wsum[0] = acc;
    }
}

struct TableRolloutArgs {
    const double *table;  // (D, L, A)
    const double *wsum;   // (1)
    double bias;
    int32_t K;
    int64_t *obs_src;
    double *obs_pos;
    float *actions_out;
    double *rew_out;
    int32_t *done_out;
};

__host__ __device__ inline size_t table_rollout_lds_bytes(int EB, int A) {
    size_t S = (size_t)EB * A;
    size_t b = (size_t)EB * 8 + S * 8 + S * 8 + S * 4 + S * 4 + (size_t)EB * 4;  // TileLds
    b = (b + 7) & ~(size_t)7;
    return ((b + (size_t)EB * 8) + 15) & ~(size_t)15;  // + redrawn day per env
}

template <bool SINGLE>
__global__ __launch_bounds__(kBlock) void fe_rollout_table_kernel(const Params p, const TableRolloutArgs r) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    int64_t *l_idx = reinterpret_cast<int64_t *>(smem + off);
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int64_t NA = p.N * A;
    const double wsum = r.wsum[0];
    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active, n, sl);
        if (active) {
            st.obs_row = r.obs_src[n] / (4 * (int64_t)A);
            st.obs_pos = r.obs_pos[sl];
        }
        for (int k = 0; k < r.K; ++k) {
            float act = 0.0f;
            if (active) {  // the whole policy: two lookups, lane-private
                double a64 = r.bias + (r.table[st.obs_row * A + a] + st.obs_pos * wsum);
                a64 = a64 < -1.0 ? -1.0 : (a64 > 1.0 ? 1.0 : a64);
                act = (float)a64;
                if (r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            }
            account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                 r.done_out + (int64_t)k * p.N);
            if constexpr (!SINGLE) lds_barrier();  // LDS scratch of account_keep is reused next step
        }
        rollout_store_state(p, active, a, n, sl, st);
        if (active) {
            r.obs_pos[sl] = st.obs_pos;
            if (a == 0) r.obs_src[n] = st.obs_row * 4 * (int64_t)A;
        }
        if constexpr (!SINGLE) __syncthreads();
    }
}

// ---- f2, MLP head: the observation projection as a dense GEMM on the matrix cores ----
// For a two-layer perceptron on the flattened window (what an MLP actor of the reference sees after
// states.float(), finenvs/agents/networks/multilayer_perceptron.py:17-25 with its default ELU,
// finenvs/agents/PPO/PPO_agent.py:101) the first layer is a true dense contraction,
//   pre[pair][h] = b1[h] + sum_{j<W} sum_{c<5} (float)obs[pair][j][c] * W1[5j+c][h],     (pairs) x (5W) x (H),
// so it runs on MFMA -- v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate, i.e. exactly an fmaf chain in k order
// (cdna_hip_programming.md section 3), which makes the result bit-reproducible on the CPU:
//   * D = W1t . X^T with the hidden units on the rows and 32 (env, asset) pairs on the columns of a tile, so that
//     after the K loop every lane holds hidden units of ITS pair and the second layer is an in-lane dot product;
//   * the position feature is the same in every window row: its W weights per hidden unit are pre-summed by the
//     host (wpos[h]) and enter as the accumulator's start value fmaf((float)pos, wpos[h], b1[h]);
//   * the remaining K4 = 4W log-return features are contracted in groups of two window rows: lane half 0 supplies
//     row 2g, half 1 row 2g+1 (one 16-byte load per lane from the f32 table, straight from L2); the k order of
//     the chain is therefore g ascending, then c = 0..3, then row 2g before row 2g+1;
//   * W1t lives in LDS for the whole launch (rows padded by 16 bytes: conflict-free ds_read_b128 fragments).
// action = clamp(b2 + [half 0: sum_h w2[h] act(pre[h])] + [half 1: ...], -1, 1); the in-lane order is tile by
// tile, register by register (hidden unit 32t + (r&3) + 8(r>>2) + 4*half).
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpArgs {
    const float *lr32;  // (D, L, 4A) f32 copy of the log-return table
    const float *w1t;   // (H, 4W) f32, w1t[h][4j+c] = W1[5j+c][h], c < 4
    const float *wpos;  // (H) f32, sum_j W1[5j+4][h]
    const float *b1;    // (H)
    const float *w2;    // (H)
    float b2;
    int32_t H, act, K;  // act: 0 ELU (the reference's default), 1 ReLU, 2 tanh
    int64_t *obs_src;
    double *obs_pos;
    float *actions_out;
    double *rew_out;
    int32_t *done_out;
};

constexpr int kMlpChunk = 4;  // row groups (8 k each) per software-pipeline stage of the first layer
// padded row length of W1t in LDS: whole chunks of zero-filled k, plus 16 bytes against bank conflicts
__host__ __device__ inline int mlp_kp(int W) { return ((4 * W + 8 * kMlpChunk - 1) / (8 * kMlpChunk)) * (8 * kMlpChunk) + 4; }

__host__ __device__ inline size_t mlp_lds_bytes(int EB, int A, int W, int H) {
    size_t S = (size_t)EB * A;
    size_t b = (size_t)EB * 8 + S * 8 + S * 8 + S * 4 + S * 4 + (size_t)EB * 4;  // TileLds
    b = (b + 7) & ~(size_t)7;
    b += (size_t)EB * 8;  // redrawn day per env
    b += S * 4;           // actions
    b = (b + 15) & ~(size_t)15;
    b += (size_t)H * mlp_kp(W) * 4;  // W1t
    b += 3 * (size_t)H * 4;          // wpos, b1, w2
    return (b + 15) & ~(size_t)15;
}

template <int ACT>
__device__ __forceinline__ float mlp_act(float z) {
    if constexpr (ACT == 1) return z > 0.0f ? z : (z != z ? z : 0.0f);
    // tanh: the exact-operation form of the LSTM head (fe_activations.h) -- branch-free and bit-reproducible on the CPU.
    // libm's tanhf, inlined 32 * NT times with its branches, left H = 128 no registers for the sleeve state (it went to
    // scratch memory: 72 bytes per lane); absolute error <= 1.2e-7 either way.
    if constexpr (ACT == 2) return lstm_tanh(z);
    // ELU, alpha = 1.  exp through v_exp_f32 (__expf), not expm1f: the second layer is VALU-bound (32 hidden units per
    // lane and block) and expm1f costs ~25 instructions per unit; the absolute error of exp(z) - 1 is <= 2e-7 per
    // unit (an ulp of 1.0), inside the 2e-6 tolerance of the action (tests/test_mlp_rollout_gpu.py)
    return z > 0.0f ? z : __expf(z) - 1.0f;
}

// second layer for one lane: fmaf chain over this lane's hidden units, tile by tile, register by register
template <int ACT, int NT>
__device__ __forceinline__ float mlp_second_layer(const f32x16 (&acc)[NT], const float *s_w2, int half) {
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int h = 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * half;
            part = fmaf(s_w2[h], mlp_act<ACT>(acc[t][rr]), part);
        }
    return part;
}

// Policy of one block of 32 (env, asset) pairs, run by one wavefront: first layer on the matrix cores, second
// layer in-lane, action into s_act[q].  l_src / l_pos are the tile's observation descriptors in LDS.
template <bool SINGLE, int NT>
__device__ __forceinline__ void mlp_policy_block(const Params &p, const MlpArgs &r, const int64_t *l_src,
                                                 const double *l_pos, float *s_act, const float *s_w1t,
                                                 const float *s_wpos, const float *s_b1, const float *s_w2, int KP,
                                                 int blk, int pairs, int lane) {
    const int A = SINGLE ? 1 : p.A;
    const int W = p.W;
    const int col = lane & 31, half = lane >> 5;
    const int ngroups = (4 * W + 7) / 8;  // two window rows per group
    const int64_t rstride = 4 * (int64_t)A;
    struct { const int64_t *src; const double *pos; } l = {l_src, l_pos};

    const int q = blk * 32 + col;
    const int qc = q < pairs ? q : pairs - 1;
    const int ee = SINGLE ? qc : (int)fdiv((uint32_t)qc, p.div_A);
    const int aa = SINGLE ? 0 : qc - ee * A;
    const float *xsrc = r.lr32 + l.src[ee] + 4 * aa;
    const float pos32 = (float)l.pos[qc];
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int h = 32 * t + (rr & 3) + 8 * (rr >> 2) + 4 * half;
            acc[t][rr] = fmaf(pos32, s_wpos[h], s_b1[h]);
        }
    const float *wrow = s_w1t + (size_t)col * KP + 4 * half;
    // rows past the window re-read its last row: their W1t entries are zero padding, so they contribute exactly
    // fmaf(0, x, acc) -- and an unconditional load keeps the chunk loop free of branches (with a branch around
    // the load the compiler waited vmcnt(0) at the top of every chunk, i.e. for the prefetch it had just issued)
    auto load_x = [&](int g) {
        const int row = 2 * g + half;
        return *reinterpret_cast<const float4 *>(xsrc + (int64_t)(row < W ? row : W - 1) * rstride);
    };
    // First layer.  B operands (window rows, from L2) are fetched one chunk of CH row groups ahead -- a
    // chunk is CH * NT * 4 MFMAs of 64 cycles, several L2 round trips --; the chunk body has no control
    // flow (rows past W re-read the last row, W1t is zero-padded to whole chunks), so the compiler is free to
    // hoist the LDS fragment reads over the MFMAs.
    constexpr int CH = kMlpChunk;
    const int nchunks = (ngroups + CH - 1) / CH;
    float4 xc[CH], xn[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) xc[i] = load_x(i);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) xn[i] = load_x((c + 1) * CH + i);
        // keep the next chunk's row loads ahead of this chunk's MFMAs: left alone the scheduler sinks them towards their use,
        // which costs 6 - 7 % at H = 128 (neutral at H = 32 / 64; tools/fused_bench.py with FUSED_LIB=mlppin)
        if constexpr (NT == 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float4 wa[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wa[t] = *reinterpret_cast<const float4 *>(wrow + (size_t)(32 * t) * KP + 8 * (c * CH + i));
            const float xs[4] = {xc[i].x, xc[i].y, xc[i].z, xc[i].w};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float ws = m == 0 ? wa[t].x : (m == 1 ? wa[t].y : (m == 2 ? wa[t].z : wa[t].w));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs[m], acc[t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) xc[i] = xn[i];
    }
    // second layer: in-lane over this lane's hidden units, then the two halves of the pair
    float part;
    if (r.act == 1) part = mlp_second_layer<1, NT>(acc, s_w2, half);
    else if (r.act == 2) part = mlp_second_layer<2, NT>(acc, s_w2, half);
    else part = mlp_second_layer<0, NT>(acc, s_w2, half);
    const float other = __shfl_xor(part, 32, 64);
    const float tot = half == 0 ? part + other : other + part;  // always (half 0) + (half 1)
    float a32 = r.b2 + tot;
    a32 = a32 < -1.0f ? -1.0f : (a32 > 1.0f ? 1.0f : a32);
    if (half == 0 && q < pairs) s_act[q] = a32;
            }

template <bool SINGLE, int NT>
__global__ __launch_bounds__(kBlock, 2) void fe_rollout_mlp_kernel(const Params p, const MlpArgs r) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const int W = p.W;
    constexpr int H = 32 * NT;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    int64_t *l_idx = reinterpret_cast<int64_t *>(smem + off);
    off += (size_t)EB * 8;
    float *s_act = reinterpret_cast<float *>(smem + off);
    off = (off + (size_t)S * 4 + 15) & ~(size_t)15;
    const int KP = mlp_kp(W);
    float *s_w1t = reinterpret_cast<float *>(smem + off);
    float *s_wpos = s_w1t + (size_t)H * KP;
    float *s_b1 = s_wpos + H;
    float *s_w2 = s_b1 + H;
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t NA = p.N * A;
    const int K4 = 4 * W;
    for (int i = tid; i < H * KP; i += kBlock) {
        const int h = i / KP, k = i - h * KP;
        s_w1t[i] = k < K4 ? r.w1t[(size_t)h * K4 + k] : 0.0f;
    }
    for (int i = tid; i < H; i += kBlock) {
        s_wpos[i] = r.wpos[i];
        s_b1[i] = r.b1[i];
        s_w2[i] = r.w2[i];
    }

    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active, n, sl);
        if (active) {
            if (a == 0) l.src[e] = r.obs_src[n];
            l.pos[e * A + a] = r.obs_pos[sl];
        }
        __syncthreads();  // also covers the weight image on the first tile
        const int pairs = ebt * A;
        const int nblk = (pairs + 31) / 32;
        for (int k = 0; k < r.K; ++k) {
            // ---- policy: one wavefront per block of 32 pairs ----
            for (int blk = wave; blk < nblk; blk += kBlock / 64)
                mlp_policy_block<SINGLE, NT>(p, r, l.src, l.pos, s_act, s_w1t, s_wpos, s_b1, s_w2, KP, blk, pairs, lane);
            lds_barrier();
            const float act = active ? s_act[e * A + a] : 0.0f;
            if (active && r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                 r.done_out + (int64_t)k * p.N);
            lds_barrier();  // the new observation's descriptors are complete
        }
        rollout_store_state(p, active, a, n, sl, st);  // state and descriptors go back to HBM once per launch
        if (active) {
            r.obs_pos[sl] = l.pos[e * A + a];
            if (a == 0) r.obs_src[n] = l.src[e];
        }
        __syncthreads();
    }
}

}  // namespace
