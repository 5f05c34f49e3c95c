// fe_lstm_kernel.h -- part of fe_env.hip (one translation unit; see the overview there): K-step fused rollout with the reference's LSTM actor on MFMA, and its exact-operation sigmoid / tanh.
#pragma once
#include "fe_device_common.h"
#include "fe_rollout_kernels.h"
#include "fe_activations.h"

namespace {

// ---- f2, LSTM head: the actor of the reference's own time-series scripts, on the matrix cores ----
// finenvs/agents/networks/lstm.py:28-57 -- nn.LSTM(5, H, batch_first) from a zero state over the W rows of the
// observation, Linear(H, 1) on the last hidden state, Tanh (continuous_actor.py:104-126) -- evaluated on
// states.float() per (env, asset) pair; examples/time_series/PPO_LSTM_testing_SPY.py:43-52 is the loop this fuses.
// Per time step the gates are a (4H) x (H + 8) x (pairs) contraction, G^T = [Whh | Wx] . [h_{t-1} ; x_t]^T:
//   * gate rows on the M side of v_mfma_f32_32x32x2_f32, 32 (env, asset) pairs on the N side; the rows are packed by
//     the host so that an accumulator lane holds all four gates of four hidden units of ITS pair (row
//     R = 32 mt + 8 b + 4 half + gate <-> unit 8 mt + 4 half + b): the cell update is in-lane, c_t never leaves
//     the registers, and h_t goes to LDS as one 16-byte store per lane -- already in the [pair][unit] layout the next
//     step's B operand reads with one ds_read_b128 per four MFMAs;
//   * the recurrent weights stay in REGISTERS for the whole launch: Whh for H = 128 is 256 KiB, more than the LDS,
//     but split over the 8 wavefronts of a 512-thread workgroup it is 128 VGPRs per lane (2 wavefronts per SIMD,
//     256 VGPRs each); every wavefront owns MPW row tiles and runs all the workgroup's 32-pair column tiles;
//   * the input part (K = 8: four log-returns | position, 1 for the bias, 0, 0) is four more MFMAs per tile;
//   * the f32 MFMA is an fmaf chain in k order, and sigmoid / tanh are built from rintf, fmaf, ldexpf and IEEE
//     division only (lstm_exp_nonpos), so the test-side CPU restatement (fo_policy_lstm) reproduces every
//     action BIT FOR BIT; against torch's own nn.LSTM the actions agree to ~1e-7.
// One barrier per time step (h double-buffered in LDS); the last hidden state is reduced by the pair's accounting lane.
constexpr int kLstmBlock = 512;
// large-H kernel: row tiles run together (independent accumulator chains per B fragment) / k groups of weight
// fragments in flight; split kernel: k groups of h in flight (profiles/r02_microbench/lstm_*.txt)
constexpr int kLstmBigRI = 4;
constexpr int kLstmBigAhead = 2;
constexpr int kLstmSplitAhead = 16;

struct LstmArgs {
    const float *lr32;  // (D, L, 4A) f32 copy of the log-return table
    const float *whh;   // (4H, H) f32, packed row order
    const float *wx;    // (4H, 8) f32, packed row order: w_ih[0..3], w_ih[4], b_ih + b_hh, 0, 0
    const float *wout;  // (H)
    float bout;
    int32_t H, out_act, K;  // out_act: 0 tanh (the reference's actor), 1 clamp to [-1, 1], 2 none (a critic's value)
    int64_t *obs_src;
    double *obs_pos;
    float *actions_out;
    double *rew_out;
    int32_t *done_out;
    // training rollouts (PPO_agent.py:98-108): actions = clamp(mean + std * noise, -1, 1), the eval env acts on the mean
    const float *noise;  // (K, N*A) standard normal draws, or null: act on the mean
    float std;
    float *means_out;    // (K, N*A) or null
    int64_t *traj_src;   // (K + 1, N) or null: descriptors of the state the policy sees at every step (+ the last one)
    double *traj_pos;    // (K + 1, N*A)
    int32_t forward_only;  // fe_lstm_forward: evaluate the head on p.N given descriptors, no env step (K = 1, actions_out = the outputs)
};

template <int NT> struct LstmGeom {
    static constexpr int H = 32 * NT;
    static constexpr int MT = H / 8;                        // 32-row gate tiles
    static constexpr int MPW = MT >= 8 ? MT / 8 : 1;        // row tiles per wavefront
    static constexpr int NSPLIT = MT >= 8 ? 1 : 8 / MT;     // wavefronts sharing a row tile split the column tiles
    static constexpr int SP = NT == 4 ? 64 : 128;           // (env, asset) pairs per workgroup tile
    static constexpr int MAXNT = SP / 32 / NSPLIT;          // column tiles per wavefront
    static constexpr int HP = H + 4;                        // LDS row length of h: 16 bytes against bank conflicts
};

__host__ __device__ inline size_t lstm_lds_bytes(int EB, int A, int H, int SP) {
    size_t S = (size_t)EB * A;
    size_t b = (size_t)EB * 8 + S * 8 + S * 8 + S * 4 + S * 4 + (size_t)EB * 4;  // TileLds
    b = (b + 7) & ~(size_t)7;
    b += (size_t)EB * 8;  // redrawn day per env
    b = (b + 15) & ~(size_t)15;
    b += 2 * (size_t)SP * (H + 4) * 4;  // h, double-buffered
    b += (size_t)H * 4;                 // wout
    return (b + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(kBlock) void fe_lstm_activations_kernel(const float *x, float *sig, float *tnh, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const v2f v = lstm_act2<false, true>((v2f){x[i], x[i]});
        sig[i] = v.x;
        tnh[i] = v.y;
    }
}

template <bool SINGLE, int NT>
__global__ __launch_bounds__(kLstmBlock, (NT == 1 ? 4 : 2)) void fe_rollout_lstm_kernel(const Params p, const LstmArgs r) {
    using G = LstmGeom<NT>;
    constexpr int H = G::H, HP = G::HP, MPW = G::MPW, NSPLIT = G::NSPLIT, MAXNT = G::MAXNT, NG = H / 8;
    constexpr int JB = MPW == 1 ? 2 : 1;  // column tiles processed together
    static_assert(MAXNT % JB == 0, "column tiles per wavefront must come in whole groups");
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const int W = p.W;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    int64_t *l_idx = reinterpret_cast<int64_t *>(smem + off);
    off = (off + (size_t)EB * 8 + 15) & ~(size_t)15;
    float *s_h = reinterpret_cast<float *>(smem + off);  // [2][SP][HP]
    float *s_wout = s_h + 2 * (size_t)G::SP * HP;
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int64_t NA = p.N * A;
    const int64_t rstride = 4 * (int64_t)A;
    const int mt0 = NSPLIT == 1 ? wave * MPW : wave % G::MT;  // first gate-row tile of this wavefront
    const int nsub = NSPLIT == 1 ? 0 : wave / G::MT;          // its share of the column tiles

    // this wavefront's slice of the weights: A fragments, lane (row = lane & 31, k half = lane >> 5)
    float4 whh[MPW][NG], wx[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const size_t R = (size_t)32 * (mt0 + i) + col;
        wx[i] = *reinterpret_cast<const float4 *>(r.wx + R * 8 + 4 * half);
#pragma unroll
        for (int g = 0; g < NG; ++g) whh[i][g] = *reinterpret_cast<const float4 *>(r.whh + R * H + 8 * g + 4 * half);
    }
    for (int i = tid; i < H; i += kLstmBlock) s_wout[i] = r.wout[i];

    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active && !r.forward_only, n, sl);
        if (active) {
            const double pos0 = r.obs_pos[sl];
            l.pos[e * A + a] = pos0;
            if (a == 0) l.src[e] = r.obs_src[n];
            if (r.traj_src) {  // row 0: the state the first policy evaluation sees
                r.traj_pos[sl] = pos0;
                if (a == 0) r.traj_src[n] = r.obs_src[n];
            }
        }
        __syncthreads();  // also covers s_wout on the first tile
        const int pairs = ebt * A;
        const int ntiles = (pairs + 31) / 32;
        for (int k = 0; k < r.K; ++k) {
            // ---- policy: W recurrent steps, every wavefront its gate rows for all of its column tiles ----
            const float *xsrc[MAXNT];
            float4 xh[MAXNT], xc[MAXNT];
            float cst[MPW][MAXNT][4];
#pragma unroll
            for (int j = 0; j < MAXNT; ++j) {
                const int q = (nsub + j * NSPLIT) * 32 + col;
                const int qc = q < pairs ? q : pairs - 1;
                const int ee = SINGLE ? qc : (int)fdiv((uint32_t)qc, p.div_A);
                const int aa = SINGLE ? 0 : qc - ee * A;
                xsrc[j] = r.lr32 + l.src[ee] + 4 * aa;
                xh[j] = make_float4((float)l.pos[qc], 1.0f, 0.0f, 0.0f);
                xc[j] = half == 0 ? *reinterpret_cast<const float4 *>(xsrc[j]) : xh[j];
#pragma unroll
                for (int i = 0; i < MPW; ++i)
#pragma unroll
                    for (int b = 0; b < 4; ++b) cst[i][j][b] = 0.0f;
            }
            for (int t = 0; t < W; ++t) {
                const float *hprev = s_h + (size_t)((t + 1) & 1) * G::SP * HP;
                float *hnext = s_h + (size_t)(t & 1) * G::SP * HP;
                float4 xn[MAXNT];
                const int tn = t + 1 < W ? t + 1 : t;  // the next step's rows, one step ahead of their use
#pragma unroll
                for (int j = 0; j < MAXNT; ++j)
                    xn[j] = half == 0 ? *reinterpret_cast<const float4 *>(xsrc[j] + (int64_t)tn * rstride) : xh[j];
                // keep the next time step's row loads up here (the scheduler otherwise sinks them towards their use): -6.5 % at
                // H = 128, -1 % at 64, +1.5 % at 32 (tools/fused_bench.py with FUSED_LIB=lstmpin)
                if constexpr (NT >= 2) __builtin_amdgcn_sched_barrier(0);
                // JB column tiles at a time: with MPW row tiles that is MPW * JB >= 2 independent accumulator chains,
                // so a dependent MFMA never waits for its predecessor's 16 passes
#pragma unroll
                for (int j0 = 0; j0 < MAXNT; j0 += JB) {
                    if (nsub + j0 * NSPLIT < ntiles) {  // (a trailing tile of the group past `pairs` computes on clamped rows)
                        f32x16 acc[MPW][JB];
#pragma unroll
                        for (int i = 0; i < MPW; ++i)
#pragma unroll
                            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr) acc[i][jj][rr] = 0.0f;
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int i = 0; i < MPW; ++i)
#pragma unroll
                                for (int jj = 0; jj < JB; ++jj) {
                                    const float4 xv = xc[j0 + jj];
                                    const float xs = m == 0 ? xv.x : (m == 1 ? xv.y : (m == 2 ? xv.z : xv.w));
                                    const float ws = m == 0 ? wx[i].x : (m == 1 ? wx[i].y : (m == 2 ? wx[i].z : wx[i].w));
                                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs, acc[i][jj], 0, 0, 0);
                                }
                        if (t > 0) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                float4 hb[JB];
#pragma unroll
                                for (int jj = 0; jj < JB; ++jj)
                                    hb[jj] = *reinterpret_cast<const float4 *>(
                                        hprev + (size_t)(32 * (nsub + (j0 + jj) * NSPLIT) + col) * HP + 4 * half + 8 * g);
#pragma unroll
                                for (int m = 0; m < 4; ++m)
#pragma unroll
                                    for (int i = 0; i < MPW; ++i)
#pragma unroll
                                        for (int jj = 0; jj < JB; ++jj) {
                                            const float4 wv = whh[i][g];
                                            const float ws = m == 0 ? wv.x : (m == 1 ? wv.y : (m == 2 ? wv.z : wv.w));
                                            const float hs = m == 0 ? hb[jj].x : (m == 1 ? hb[jj].y : (m == 2 ? hb[jj].z : hb[jj].w));
                                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, hs, acc[i][jj], 0, 0, 0);
                                        }
                            }
                        }
                        // cell update, in-lane: acc[i][jj][4b + gate] belongs to unit 8 (mt0 + i) + 4 half + b
#pragma unroll
                        for (int i = 0; i < MPW; ++i)
#pragma unroll
                            for (int jj = 0; jj < JB; ++jj) {
                                const int j = j0 + jj;
                                float hv[4], og[4];
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const v2f sif = lstm_act2<false, false>((v2f){acc[i][jj][4 * b + 0], acc[i][jj][4 * b + 1]});
                                    const v2f tgo = lstm_act2<true, false>((v2f){acc[i][jj][4 * b + 2], acc[i][jj][4 * b + 3]});
                                    const float t1 = sif.y * cst[i][j][b];
                                    const float t2 = sif.x * tgo.x;
                                    cst[i][j][b] = t1 + t2;
                                    og[b] = tgo.y;
                                }
#pragma unroll
                                for (int b = 0; b < 4; b += 2) {
                                    const v2f tc = lstm_act2<true, true>((v2f){cst[i][j][b], cst[i][j][b + 1]});
                                    hv[b] = og[b] * tc.x;
                                    hv[b + 1] = og[b + 1] * tc.y;
                                }
                                *reinterpret_cast<float4 *>(hnext + (size_t)(32 * (nsub + j * NSPLIT) + col) * HP + 8 * (mt0 + i) + 4 * half) =
                                    make_float4(hv[0], hv[1], hv[2], hv[3]);
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < MAXNT; ++j) xc[j] = xn[j];
                lds_barrier();  // h_t is complete
            }
            // ---- output layer: the pair's accounting lane reduces its last hidden state ----
            float act = 0.0f;
            if (active) {
                const float *hl = s_h + (size_t)((W - 1) & 1) * G::SP * HP + (size_t)(e * A + a) * HP;
                float o = r.bout;
#pragma unroll 8
                for (int u = 0; u < H; ++u) o = fmaf(s_wout[u], hl[u], o);
                act = r.out_act == 0 ? lstm_tanh(o) : (r.out_act == 2 ? o : (o < -1.0f ? -1.0f : (o > 1.0f ? 1.0f : o)));
                if (r.means_out) r.means_out[(int64_t)k * NA + sl] = act;
                if (r.noise && n != p.eval_env) {  // distribution.sample() clamped; the eval env keeps the mean
                    const float dev = r.std * r.noise[(int64_t)k * NA + sl];
                    const float smp = act + dev;
                    act = smp < -1.0f ? -1.0f : (smp > 1.0f ? 1.0f : smp);
                }
                if (r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            }
            if (!r.forward_only) {  // (uniform)
                account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                     r.done_out + (int64_t)k * p.N);
                if (active && r.traj_src) {  // row k + 1: the observation this step returns (own LDS entries: no barrier needed)
                    r.traj_pos[(int64_t)(k + 1) * NA + sl] = l.pos[e * A + a];
                    if (a == 0) r.traj_src[(int64_t)(k + 1) * p.N + n] = l.src[e];
                }
            }
            lds_barrier();  // the new observation's descriptors are complete; everyone is done with h_W
        }
        if (!r.forward_only) {
            rollout_store_state(p, active, a, n, sl, st);  // state and descriptors go back to HBM once per launch
            if (active) {
                r.obs_pos[sl] = l.pos[e * A + a];
                if (a == 0) r.obs_src[n] = l.src[e];
            }
        }
        __syncthreads();
    }
}

// ---- the same head for H = 256, 512, 1024 (the reference example trains hidden_dim = 1024): weights from L2 ----
// Whh no longer fits a workgroup's registers (4 MiB / 16 MiB), so it is STREAMED: the host stores it fragment-major --
// [row tile mt][k group g][lane][4], one contiguous KiB per (mt, g), exactly the A fragment of four MFMAs -- and every
// wavefront loads its fragments with coalesced 16-byte loads two k groups (32 MFMAs) ahead of their use, continuously; the XCD's 4 MiB L2 (and the
// Infinity Cache behind it) serves the 32 CUs that sweep the same matrix.  Per workgroup ONE 32-pair column tile
// (H = 1024: h alone is 128 KiB of LDS, single-buffered: a wavefront keeps its new h values in registers until every
// wavefront has finished reading the old ones -- two barriers per time step); each of the 8 wavefronts owns
// RTW = H / 64 row tiles and runs them four at a time (four independent accumulator chains sharing the B fragment;
// measured against two: +6 % at H = 256, +2 % at H = 1024, tools/fused_bench.py with FUSED_LIB).
// c_t and the pending h_t live in per-lane scratch (RTW x 4 floats each, touched once per row tile).  Same k order, same activations, same cell update as the register-
// resident kernel: the same oracle function pins it bit for bit.
__host__ __device__ inline size_t lstm_big_lds_bytes(int EB, int A, int H) {
    size_t S = (size_t)EB * A;
    size_t b = (size_t)EB * 8 + S * 8 + S * 8 + S * 4 + S * 4 + (size_t)EB * 4;  // TileLds
    b = (b + 7) & ~(size_t)7;
    b += (size_t)EB * 8;  // redrawn day per env
    b = (b + 15) & ~(size_t)15;
    b += (size_t)32 * (H + 4) * 4;  // h, one buffer
    b += (size_t)H * 4;             // wout
    return (b + 15) & ~(size_t)15;
}

template <bool SINGLE, int RTW>
__global__ __launch_bounds__(kLstmBlock, 2) void fe_rollout_lstm_big_kernel(const Params p, const LstmArgs r) {
    constexpr int H = 64 * RTW, HP = H + 4, NG = H / 8, SP = 32;
    constexpr int RI = kLstmBigRI;  // row tiles run together: RI independent accumulator chains share every B fragment
    constexpr int AHEAD = kLstmBigAhead;  // k groups a weight fragment is loaded ahead of its MFMAs
    static_assert(RTW % RI == 0 && (H / 8) % AHEAD == 0, "row tiles / k groups must come in whole groups");
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const int W = p.W;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    int64_t *l_idx = reinterpret_cast<int64_t *>(smem + off);
    off = (off + (size_t)EB * 8 + 15) & ~(size_t)15;
    float *s_h = reinterpret_cast<float *>(smem + off);  // [SP][HP]
    float *s_wout = s_h + (size_t)SP * HP;
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int64_t NA = p.N * A;
    const int64_t rstride = 4 * (int64_t)A;
    const int mt0 = wave * RTW;  // this wavefront's row tiles: mt0 .. mt0 + RTW - 1
    float4 wq[kLstmBigAhead][kLstmBigRI];  // weight fragments in flight (see the k loop)
    bool primed = false;
    for (int i = tid; i < H; i += kLstmBlock) s_wout[i] = r.wout[i];

    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active && !r.forward_only, n, sl);
        if (active) {
            const double pos0 = r.obs_pos[sl];
            l.pos[e * A + a] = pos0;
            if (a == 0) l.src[e] = r.obs_src[n];
            if (r.traj_src) {
                r.traj_pos[sl] = pos0;
                if (a == 0) r.traj_src[n] = r.obs_src[n];
            }
        }
        __syncthreads();  // also covers s_wout on the first tile
        const int pairs = ebt * A;
        for (int k = 0; k < r.K; ++k) {
            const int qc = col < pairs ? col : pairs - 1;
            const int ee = SINGLE ? qc : (int)fdiv((uint32_t)qc, p.div_A);
            const int aa = SINGLE ? 0 : qc - ee * A;
            const float *xsrc = r.lr32 + l.src[ee] + 4 * aa;
            const float4 xh = make_float4((float)l.pos[qc], 1.0f, 0.0f, 0.0f);
            float4 xc = half == 0 ? *reinterpret_cast<const float4 *>(xsrc) : xh;
            float cst[RTW][4], hnew[RTW][4];
#pragma unroll
            for (int i = 0; i < RTW; ++i)
#pragma unroll
                for (int b = 0; b < 4; ++b) cst[i][b] = 0.0f;
            for (int t = 0; t < W; ++t) {
                const int tn = t + 1 < W ? t + 1 : t;
                const float4 xn = half == 0 ? *reinterpret_cast<const float4 *>(xsrc + (int64_t)tn * rstride) : xh;
                const float *hrow = s_h + (size_t)col * HP + 4 * half;
                // a real loop over this wavefront's row-tile pairs: c_t and the pending h_t (RTW x 4 floats each per lane,
                // touched once per 1040 MFMAs) are indexed dynamically, i.e. live in per-lane scratch, not in VGPRs
#pragma unroll 1
                for (int i0 = 0; i0 < RTW; i0 += RI) {
                    f32x16 acc[RI];
#pragma unroll
                    for (int i = 0; i < RI; ++i)
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) acc[i][rr] = 0.0f;
                    // input part: four MFMAs per row tile
                    float4 wxv[RI];
#pragma unroll
                    for (int i = 0; i < RI; ++i)
                        wxv[i] = *reinterpret_cast<const float4 *>(r.wx + ((size_t)32 * (mt0 + i0 + i) + col) * 8 + 4 * half);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int i = 0; i < RI; ++i) {
                            const float xs = m == 0 ? xc.x : (m == 1 ? xc.y : (m == 2 ? xc.z : xc.w));
                            const float ws = m == 0 ? wxv[i].x : (m == 1 ? wxv[i].y : (m == 2 ? wxv[i].z : wxv[i].w));
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs, acc[i], 0, 0, 0);
                        }
                    if (t > 0) {
                        // fragment-major weights: one coalesced KiB per (row tile, k group), AHEAD groups in flight -- across
                        // row-tile groups and time steps too: the tail of one k loop already fetches the head of the next
                        // (the next group's, or after the last group the first one's again: the matrix never changes)
                        const float4 *wbase = reinterpret_cast<const float4 *>(r.whh) + lane;
                        const float4 *wf[RI], *wfn[RI];
#pragma unroll
                        for (int i = 0; i < RI; ++i) {
                            wf[i] = wbase + ((size_t)(mt0 + i0 + i) * NG) * 64;
                            wfn[i] = wbase + ((size_t)(mt0 + (i0 + RI < RTW ? i0 + RI : 0) + i) * NG) * 64;
                        }
                        if (!primed) {
#pragma unroll
                            for (int d = 0; d < AHEAD; ++d)
#pragma unroll
                                for (int i = 0; i < RI; ++i) wq[d][i] = wf[i][(size_t)d * 64];
                            primed = true;
                        }
#pragma unroll 1  // a real loop: unrolled, its hoisted loads spill (NG is up to 128 groups of 4 RI MFMAs)
                        for (int g0 = 0; g0 < NG; g0 += AHEAD) {
#pragma unroll
                            for (int d = 0; d < AHEAD; ++d) {
                                const int g = g0 + d;
                                float4 wv[RI];
                                const int gn = g + AHEAD;
#pragma unroll
                                for (int i = 0; i < RI; ++i) {
                                    wv[i] = wq[d][i];
                                    wq[d][i] = gn < NG ? wf[i][(size_t)gn * 64] : wfn[i][(size_t)(gn - NG) * 64];
                                }
                                const float4 hb = *reinterpret_cast<const float4 *>(hrow + 8 * g);
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    const float hs = m == 0 ? hb.x : (m == 1 ? hb.y : (m == 2 ? hb.z : hb.w));
#pragma unroll
                                    for (int i = 0; i < RI; ++i) {
                                        const float ws = m == 0 ? wv[i].x : (m == 1 ? wv[i].y : (m == 2 ? wv[i].z : wv[i].w));
                                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, hs, acc[i], 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    // cell update, in-lane; the new h waits (in scratch) until everyone has read the old one
#pragma unroll
                    for (int i = 0; i < RI; ++i) {
                        float og[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const v2f sif = lstm_act2<false, false>((v2f){acc[i][4 * b + 0], acc[i][4 * b + 1]});
                            const v2f tgo = lstm_act2<true, false>((v2f){acc[i][4 * b + 2], acc[i][4 * b + 3]});
                            const float t1 = sif.y * cst[i0 + i][b];
                            const float t2 = sif.x * tgo.x;
                            cst[i0 + i][b] = t1 + t2;
                            og[b] = tgo.y;
                        }
#pragma unroll
                        for (int b = 0; b < 4; b += 2) {
                            const v2f tc = lstm_act2<true, true>((v2f){cst[i0 + i][b], cst[i0 + i][b + 1]});
                            hnew[i0 + i][b] = og[b] * tc.x;
                            hnew[i0 + i][b + 1] = og[b + 1] * tc.y;
                        }
                    }
                }
                lds_barrier();  // every wavefront has read h_{t-1}
#pragma unroll
                for (int i = 0; i < RTW; ++i)
                    *reinterpret_cast<float4 *>(s_h + (size_t)col * HP + 8 * (mt0 + i) + 4 * half) =
                        make_float4(hnew[i][0], hnew[i][1], hnew[i][2], hnew[i][3]);
                xc = xn;
                lds_barrier();  // h_t is complete
            }
            // ---- output layer: the pair's accounting lane reduces its last hidden state ----
            float act = 0.0f;
            if (active) {
                const float *hl = s_h + (size_t)(e * A + a) * HP;
                float o = r.bout;
#pragma unroll 8
                for (int u = 0; u < H; ++u) o = fmaf(s_wout[u], hl[u], o);
                act = r.out_act == 0 ? lstm_tanh(o) : (r.out_act == 2 ? o : (o < -1.0f ? -1.0f : (o > 1.0f ? 1.0f : o)));
                if (r.means_out) r.means_out[(int64_t)k * NA + sl] = act;
                if (r.noise && n != p.eval_env) {
                    const float dev = r.std * r.noise[(int64_t)k * NA + sl];
                    const float smp = act + dev;
                    act = smp < -1.0f ? -1.0f : (smp > 1.0f ? 1.0f : smp);
                }
                if (r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            }
            if (!r.forward_only) {  // (uniform)
                account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                     r.done_out + (int64_t)k * p.N);
                if (active && r.traj_src) {
                    r.traj_pos[(int64_t)(k + 1) * NA + sl] = l.pos[e * A + a];
                    if (a == 0) r.traj_src[(int64_t)(k + 1) * p.N + n] = l.src[e];
                }
            }
            lds_barrier();  // the new observation's descriptors are complete; everyone is done with h_W
        }
        if (!r.forward_only) {
            rollout_store_state(p, active, a, n, sl, st);
            if (active) {
                r.obs_pos[sl] = l.pos[e * A + a];
                if (a == 0) r.obs_src[n] = l.src[e];
            }
        }
        __syncthreads();
    }
}

// ---- large H at SMALL env counts: one launch per time step, the row tiles spread over the whole GPU ----
// The fused kernels keep a tile on one CU for a whole env step; at H = 1024 that CU walks 128 row tiles (1.8 ms per step
// however few envs there are).  Below a few thousand pairs the step is better cut the other way: one launch per LSTM
// time step in which workgroup (mt, y) computes gate-row tile mt for four 32-pair column tiles (one per wavefront), with
// h_{t-1}, h_t and c_t in global memory -- FRAGMENT-MAJOR, [column tile][k group][lane][4]: the float4 a lane of row
// tile mt produces (units 8 mt + 4 half + b of its pair) IS the B fragment of k group g = mt of the next time step, so
// both sides are one coalesced KiB per wavefront -- and the launch boundary is the exchange of h.  A last launch per env
// step reduces h_W to the action and runs the accounting (one lane per sleeve).  Same k order, same activations, same
// cell update: the same oracle function, bit for bit.  The floor is the accumulator chain itself: H/2 + 4 DEPENDENT MFMAs
// per time step at 64 cycles each (measured in round 2 with a dependent-MFMA probe, NOTES.md) = 15 us at H = 1024; the launch reaches 25 - 30 us: its L2 is
// cold, so the row tile's weights are staged through LDS by the whole workgroup and h arrives 16 k groups ahead.
struct LstmSplitArgs {
    LstmArgs a;          // weights (whh fragment-major), descriptors, outputs of step k (pointers already offset)
    float *hbuf;         // [2][CT][H/8][64][4] f32: h, double-buffered by time-step parity
    float *cbuf;         // [CT][H/8][64][4] f32
    int32_t t;           // time step of this launch (gates kernel)
    int32_t k;           // env step of this launch (finish kernel: row k + 1 of the trajectory)
    int64_t pairs;       // N * A
};

// PER float4s per lane, all loads in flight before the first LDS write
template <int PER>
__device__ __forceinline__ void lstm_stage_weights(const float4 *src, float4 *dst, int tid) {
    float4 v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = src[j * kBlock + tid];
#pragma unroll
    for (int j = 0; j < PER; ++j) dst[j * kBlock + tid] = v[j];
}

template <bool SINGLE>
__global__ __launch_bounds__(kBlock) void fe_lstm_split_gates_kernel(const Params p, const LstmSplitArgs s) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LstmArgs &r = s.a;
    const int A = SINGLE ? 1 : p.A;
    const int H = r.H, NG = H / 8, t = s.t;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int mt = blockIdx.x;
    const int64_t ct = (int64_t)blockIdx.y * (kBlock / 64) + wave;
    const int64_t CT = (s.pairs + 31) / 32;
    // Every launch starts with a cold L2: everything this wavefront needs first goes out FIRST -- its row of x, the input
    // weights, the first AHEAD k groups of h -- and then the whole workgroup fetches the row tile's recurrent weights (H/8
    // KiB, fragment-major; Infinity Cache / HBM at ~1.5 us) ONCE into LDS with all its loads in flight together (up to 32 per
    // lane), instead of each wavefront trickling them in a few KiB ahead of its accumulator chain.
    const bool valid = ct < CT;
    const int64_t ctc = valid ? ct : CT - 1;
    const int64_t q = ctc * 32 + col;
    const int64_t qc = q < s.pairs ? q : s.pairs - 1;
    const int64_t n = SINGLE ? qc : qc / A;
    const int aa = SINGLE ? 0 : (int)(qc - n * A);
    // B fragment of the input part: the row's four log-returns | position, 1, 0, 0
    float4 xv;
    if (half == 0) xv = *reinterpret_cast<const float4 *>(r.lr32 + r.obs_src[n] + 4 * aa + (int64_t)t * 4 * A);
    else xv = make_float4((float)r.obs_pos[qc], 1.0f, 0.0f, 0.0f);
    const float4 wxv = *reinterpret_cast<const float4 *>(r.wx + ((size_t)32 * mt + col) * 8 + 4 * half);
    const size_t frag = (size_t)CT * NG * 64;  // float4s per h buffer
    float4 *hnext = reinterpret_cast<float4 *>(s.hbuf) + (size_t)(t & 1) * frag;
    const float4 *hprev = reinterpret_cast<const float4 *>(s.hbuf) + (size_t)((t + 1) & 1) * frag + ((size_t)ctc * NG) * 64 + lane;
    constexpr int AHEAD = kLstmSplitAhead;  // k groups of h in flight (h was written by the previous launch: also cold)
    float4 hq[AHEAD];
    if (t > 0) {
#pragma unroll
        for (int d = 0; d < AHEAD; ++d) hq[d] = hprev[(size_t)d * 64];
    }
    float4 *s_w = reinterpret_cast<float4 *>(smem);  // [NG][64]
    if (t > 0) {
        const float4 *wsrc = reinterpret_cast<const float4 *>(r.whh) + ((size_t)mt * NG) * 64;
        const int per = NG * 64 / kBlock;  // float4s per lane: 8 (H = 256), 16, 32 (H = 1024)
        if (per == 32) lstm_stage_weights<32>(wsrc, s_w, tid);
        else if (per == 16) lstm_stage_weights<16>(wsrc, s_w, tid);
        else lstm_stage_weights<8>(wsrc, s_w, tid);
    }
    __syncthreads();
    if (!valid) return;  // (no further barriers)
    f32x16 acc;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float xs = m == 0 ? xv.x : (m == 1 ? xv.y : (m == 2 ? xv.z : xv.w));
        const float ws = m == 0 ? wxv.x : (m == 1 ? wxv.y : (m == 2 ? wxv.z : wxv.w));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, xs, acc, 0, 0, 0);
    }
    if (t > 0) {
        float4 wnext = s_w[lane];  // the A fragment is read from LDS one k group ahead of its MFMAs (a dependent MFMA can
                                   // issue every 64 cycles, measured in round 2, NOTES.md: nothing may wait in between)
#pragma unroll 1
        for (int g0 = 0; g0 < NG; g0 += AHEAD) {
#pragma unroll
            for (int d = 0; d < AHEAD; ++d) {
                const float4 hb = hq[d];
                const int gn = g0 + d + AHEAD < NG ? g0 + d + AHEAD : NG - 1;  // (the last loads are redundant, in range)
                hq[d] = hprev[(size_t)gn * 64];
                const float4 wv = wnext;
                wnext = s_w[(g0 + d + 1 < NG ? g0 + d + 1 : NG - 1) * 64 + lane];
                // keep the refill load HERE, AHEAD groups before its use: left alone, the scheduler sinks it to ~3 groups ahead
                // (and splits it into dword loads), which the cold h cannot cover
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float hs = m == 0 ? hb.x : (m == 1 ? hb.y : (m == 2 ? hb.z : hb.w));
                    const float ws = m == 0 ? wv.x : (m == 1 ? wv.y : (m == 2 ? wv.z : wv.w));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, hs, acc, 0, 0, 0);
                }
            }
        }
    }
    // cell update, in-lane: acc[4b + gate] belongs to unit 8 mt + 4 half + b of pair col
    float4 *cptr = reinterpret_cast<float4 *>(s.cbuf) + ((size_t)ct * NG + mt) * 64 + lane;
    const float4 cold = t > 0 ? *cptr : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float co[4] = {cold.x, cold.y, cold.z, cold.w};
    float cn[4], og[4], hv[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const v2f sif = lstm_act2<false, false>((v2f){acc[4 * b + 0], acc[4 * b + 1]});
        const v2f tgo = lstm_act2<true, false>((v2f){acc[4 * b + 2], acc[4 * b + 3]});
        const float t1 = sif.y * co[b];
        const float t2 = sif.x * tgo.x;
        cn[b] = t1 + t2;
        og[b] = tgo.y;
    }
#pragma unroll
    for (int b = 0; b < 4; b += 2) {
        const v2f tc = lstm_act2<true, true>((v2f){cn[b], cn[b + 1]});
        hv[b] = og[b] * tc.x;
        hv[b + 1] = og[b + 1] * tc.y;
    }
    *cptr = make_float4(cn[0], cn[1], cn[2], cn[3]);
    hnext[((size_t)ct * NG + mt) * 64 + lane] = make_float4(hv[0], hv[1], hv[2], hv[3]);
}

// output layer + accounting of one env step (the tail of fe_rollout_lstm_kernel's step loop as its own launch)
template <bool SINGLE>
__global__ __launch_bounds__(kBlock) void fe_lstm_split_finish_kernel(const Params p, const LstmSplitArgs s) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LstmArgs &r = s.a;
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const int S = EB * A;
    const int H = r.H, NG = H / 8;
    const TileLds l = carve_lds(smem, EB, S);
    size_t off = (size_t)EB * 8 + (size_t)S * 8 + (size_t)S * 8 + (size_t)S * 4 + (size_t)S * 4 + (size_t)EB * 4;
    off = (off + 7) & ~(size_t)7;
    int64_t *l_idx = reinterpret_cast<int64_t *>(smem + off);
    off = (off + (size_t)EB * 8 + 15) & ~(size_t)15;
    float4 *s_hq = reinterpret_cast<float4 *>(smem + off);  // [EB * A][2 NG]
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int64_t NA = p.N * A;
    const int64_t CT = (s.pairs + 31) / 32;
    const float *hW = s.hbuf + (size_t)((p.W - 1) & 1) * CT * NG * 64 * 4;
    for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * EB;
        const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
        const bool active = e < ebt;
        const int64_t n = n0 + e;
        const int64_t sl = n * A + a;
        SleeveReg st = rollout_load_state(p, active && !r.forward_only, n, sl);
        // h_W of the tile's pairs -> LDS, the whole workgroup loading (the launch starts with a cold L2: a lane walking its
        // own 4 KiB alone would pay ~0.3 us per dependent-latency step): [pair][2 NG] float4, units ascending per pair
        {
            const int tp = ebt * A;  // pairs of this tile: sl0 .. sl0 + tp - 1
            const int64_t sl0 = n0 * A;
            const int per = 2 * NG;  // float4s per pair
            for (int i = tid; i < tp * per; i += kBlock) {
                const int pq = i / per, j = i - pq * per;       // pair, (g, half) = (j >> 1, j & 1)
                const int64_t qg = sl0 + pq;
                s_hq[i] = reinterpret_cast<const float4 *>(hW)[(((size_t)(qg >> 5) * NG + (j >> 1)) * 64 + (qg & 31) + 32 * (j & 1))];
            }
            __syncthreads();
        }
        float act = 0.0f;
        if (active) {
            if (s.k == 0 && r.traj_src) {  // row 0: the state the first policy evaluation saw
                r.traj_pos[sl] = r.obs_pos[sl];
                if (a == 0) r.traj_src[n] = r.obs_src[n];
            }
            const float4 *hp = s_hq + (size_t)(e * A + a) * (2 * NG);  // this pair's h_W, units ascending (staged above)
            float o = r.bout;
#pragma unroll 4
            for (int g = 0; g < NG; ++g) {  // units 8g + 4 half + c
                const float4 h0 = hp[2 * g], h1 = hp[2 * g + 1];
                const float *w = r.wout + 8 * g;
                o = fmaf(w[0], h0.x, o); o = fmaf(w[1], h0.y, o); o = fmaf(w[2], h0.z, o); o = fmaf(w[3], h0.w, o);
                o = fmaf(w[4], h1.x, o); o = fmaf(w[5], h1.y, o); o = fmaf(w[6], h1.z, o); o = fmaf(w[7], h1.w, o);
            }
            act = r.out_act == 0 ? lstm_tanh(o) : (r.out_act == 2 ? o : (o < -1.0f ? -1.0f : (o > 1.0f ? 1.0f : o)));
            if (r.means_out) r.means_out[sl] = act;
            if (r.noise && n != p.eval_env) {
                const float dev = r.std * r.noise[sl];
                const float smp = act + dev;
                act = smp < -1.0f ? -1.0f : (smp > 1.0f ? 1.0f : smp);
            }
            if (r.actions_out) r.actions_out[sl] = act;
        }
        if (!r.forward_only) {  // (uniform)
            account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out, r.done_out);
            rollout_store_state(p, active, a, n, sl, st);
            if (active) {
                r.obs_pos[sl] = l.pos[e * A + a];
                if (a == 0) r.obs_src[n] = l.src[e];
                if (r.traj_src) {
                    r.traj_pos[(int64_t)(s.k + 1) * NA + sl] = l.pos[e * A + a];
                    if (a == 0) r.traj_src[(int64_t)(s.k + 1) * p.N + n] = l.src[e];
                }
            }
        }
        __syncthreads();  // s_hq and the accounting scratch are reused by the next tile
    }
}

}  // namespace
