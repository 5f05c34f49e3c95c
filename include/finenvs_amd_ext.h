/*
 * finenvs_amd_ext.h -- EXPERIMENTAL extensions of the C ABI in finenvs_amd.h (same library).
 *
 * Nothing here is needed to replace TimeSeriesEnv.reset()/step() or the 8(f)
 * rows; these entry points fuse a POLICY into the K-step rollout launch (SURVEY.md 8f.2's "policy
 * hook", widened to the reference's MLP and LSTM actor shapes) and expose a tuning hook for tools/.
 * They are pinned by the oracle (oracle/fe_oracle.c: fo_policy_*) and by tests/test_mlp_rollout_gpu.py /
 * tests/test_lstm_rollout_gpu.py, but their signatures may change between ABI versions without the
 * frozen header changing, and no new exports are added here.  Conventions as in finenvs_amd.h.
 */
#ifndef FINENVS_AMD_EXT_H
#define FINENVS_AMD_EXT_H

#include "finenvs_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Table form of the in-kernel linear policy.  For fixed weights the log-return part of the policy
 * is an indicator of the day's series, like the reference's precomputed log-returns (TSE:179-194):
 * fe_policy_table fills table (D, L, A) f64 with
 *   table[d][s][a] = sum_j sum_{c<4} log_return[d][s+j][4a+c] * weights[j][c]   (NaN where s+W > L)
 * and wsum[0] = sum_j weights[j][4], both in the 64-lane partial-sum + butterfly order above;
 * fe_env_rollout_table then runs K steps with
 *   action = clamp(bias + (table[row][a] + pos * wsum), -1, 1)
 * i.e. two 8-byte lookups per sleeve and step instead of re-reading the window.  Same rollout-loop
 * semantics and side effects as fe_env_rollout_linear (examples/time_series/
 * PPO_LSTM_training_SPY.py:22-28); the split of the sum is part of this form's contract.
 */
int fe_policy_table(fe_env *env, const double *weights, double *table, double *wsum, void *stream);
int fe_env_rollout_table(fe_env *env, const double *table, const double *wsum, double bias, int32_t K,
                         int64_t *obs_src, double *obs_pos, float *actions_out, double *rewards_out,
                         int32_t *dones_out, void *stream);

/*
 * The same K-step loop with a two-layer perceptron head on the flattened window of every (env, asset) pair -- the
 * shape of the reference's MLP networks (finenvs/agents/networks/multilayer_perceptron.py:17-25, default ELU) fed
 * with states.float() (finenvs/agents/PPO/PPO_agent.py:101):
 *   action = clamp(b2 + sum_h w2[h] * act(b1[h] + sum_{j<W} sum_{c<5} (float)obs[j][5a+c] * W1[5j+c][h]), -1, 1)
 * The first layer is a dense (pairs x 5W x H) contraction and runs on the matrix cores (v_mfma_f32_32x32x2_f32:
 * f32 in, f32 accumulate, exactly an fmaf chain); the summation order is part of the contract and restated by
 * oracle/fe_oracle.c:fo_policy_mlp, so the pre-activations are bit-reproducible on the CPU:
 *   logret_f32 (D, L, 4*A) f32 = (float) of the log-return table; w1t (H, 4W) f32 with w1t[h][4j+c] = W1[5j+c][h]
 *   for the four log-return features; wpos (H) = sum_j W1[5j+4][h] (the position feature is the same in every
 *   row); b1, w2 (H); H in {32, 64, 128}; activation 0 = ELU, 1 = ReLU, 2 = tanh (the exact-operation
 *   tanh of fe_env_rollout_lstm's head: bit-reproducible, |error| <= 1.2e-7).  W1t must fit the 160 KiB LDS
 *   (H * (4W rounded up to a multiple of 32, + 4) * 4 bytes + a few KiB), else FE_ERR_ARG.
 * Other arguments, loop semantics and side effects as fe_env_rollout_linear
 * (examples/time_series/PPO_LSTM_training_SPY.py:22-28).
 */
int fe_env_rollout_mlp(fe_env *env, const float *logret_f32, const float *w1t, const float *wpos, const float *b1,
                       const float *w2, float b2, int32_t H, int32_t activation, int32_t K, int64_t *obs_src,
                       double *obs_pos, float *actions_out, double *rewards_out, int32_t *dones_out, void *stream);

/*
 * The same loop with the LSTM actor the reference's own time-series scripts use
 * (finenvs/agents/networks/lstm.py:28-57: nn.LSTM(5, H, batch_first) from a zero state over the W rows of the
 * observation, Linear(H, 1) on the last hidden state; Tanh output, finenvs/agents/PPO/continuous_actor.py:104-126;
 * fed states.float(), examples/time_series/PPO_LSTM_testing_SPY.py:43-52), applied per (env, asset) pair.
 * The gate contraction (4H) x (H + 8) x pairs of every time step runs on the matrix cores (v_mfma_f32_32x32x2_f32),
 * the recurrent weights live in registers, c_t in registers, h_t in LDS.  Summation order and the exact-operation
 * sigmoid / tanh are part of the contract and restated by oracle/fe_oracle.c:fo_policy_lstm (bit-reproducible):
 *   whh (4H, H) f32: weight_hh_l0 with its rows PACKED: row 32*mt + 8*b + 4*half + gate = gate (0 i, 1 f, 2 g, 3 o)
 *   of hidden unit 8*mt + 4*half + b;  wx (4H, 8) f32, same row order: weight_ih_l0[row][0..4], bias_ih + bias_hh,
 *   0, 0;  wout (H), bout: the output layer;  out_activation 0 = tanh, 1 = clamp to [-1, 1].
 *   H in {32, 64, 128}: whh row-major as above, held in registers for the whole launch.
 *   H in {256, 512, 1024} (the reference example trains hidden_dim = 1024): the same packed rows stored FRAGMENT-MAJOR,
 *   whh[((mt * (H/8) + g) * 64 + lane) * 4 + c] = packed_whh[32*mt + (lane & 31)][8*g + 4*(lane >> 5) + c] -- one
 *   contiguous KiB per (row tile, k group), streamed from L2 with coalesced loads; same arithmetic, same oracle.
 * A (assets per env) must not exceed the pairs of a workgroup tile (32 for H >= 256, 64 for H = 128, else 128): FE_ERR_ARG.
 * Training rollouts (finenvs/agents/PPO/PPO_agent.py:98-108, agent.step): with noise (K, N*A) f32 standard-normal
 * draws (made by the caller's generator) and std = exp(log_standard_deviation), the action is
 * clamp(mean + std * noise, -1, 1) (one f32 product, one f32 sum) -- except for the eval env of a training-mode env,
 * which acts on the mean (PPO_agent.py:105-107; an evaluate-mode env has no eval env, so all its envs sample -- the
 * reference's agent.step overwrites its last row with the mean in either mode: pass zero noise for that env, which is
 * the same action bit for bit); means_out (K, N*A) receives the means (for log_prob), and
 * states_src_out (K+1, N) i64 / states_pos_out (K+1, N*A) f64 the descriptors of the state the policy saw at every
 * step (row k) and of the last returned one (row K) -- the `states` of agent.store (PPO_LSTM_training_SPY.py:27),
 * see fe_env_render_n.  noise / means_out / states_*_out may be NULL.
 * Other arguments, loop semantics and side effects as fe_env_rollout_linear.
 */
int fe_env_rollout_lstm(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout,
                        float bout, int32_t H, int32_t out_activation, int32_t K, int64_t *obs_src, double *obs_pos,
                        const float *noise, float std, float *actions_out, float *means_out, double *rewards_out,
                        int32_t *dones_out, int64_t *states_src_out, double *states_pos_out, void *stream);

/*
 * fe_env_rollout_lstm for H in {256, 512, 1024} at SMALL env counts (the reference's own evaluation runs one env per
 * trading day with hidden_dim = 1024, examples/time_series/PPO_LSTM_testing_SPY.py:27-29, 41): instead of keeping a tile
 * on one CU for a whole step, every LSTM time step is ONE launch whose workgroups are the 4H/32 gate-row tiles (x groups
 * of four 32-pair column tiles); h and c live in `workspace` (fe_lstm_split_workspace_floats(H, N*A) floats, fragment-
 * major) and the launch boundary is the exchange of h; a last launch per env step reduces h_W and runs the accounting.
 * W + 1 launches per env step, all on `stream` (capturable in a hipGraph).  Arguments, weights layout (whh fragment-major),
 * semantics and results exactly as fe_env_rollout_lstm: the same oracle function pins both bit for bit.
 */
int64_t fe_lstm_split_workspace_floats(int32_t H, int64_t pairs);
int fe_env_rollout_lstm_split(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout,
                              float bout, int32_t H, int32_t out_activation, int32_t K, int64_t *obs_src, double *obs_pos,
                              const float *noise, float std, float *actions_out, float *means_out, double *rewards_out,
                              int32_t *dones_out, int64_t *states_src_out, double *states_pos_out, float *workspace,
                              void *stream);

/*
 * The same LSTM head (same weights layout, same arithmetic) evaluated on ANY `count` observation descriptors without
 * stepping an env and without materialising the observations: out (count*A) f32.  out_activation 2 = none -- a critic
 * (finenvs/agents/PPO/critic.py, CriticLSTM: LSTMNetwork with the Identity output) -- so the values of all K + 1
 * states of a trajectory chunk (PPO_agent.py:99, 171) are one launch over its descriptor rows.  The env's state is not
 * read or written; obs_src (count) i64, obs_pos (count*A) f64 as produced by fe_env_describe / fe_env_step_traj /
 * fe_env_rollout_lstm (states_*_out).
 */
int fe_lstm_forward(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout, float bout,
                    int32_t H, int32_t out_activation, const int64_t *obs_src, const double *obs_pos, int64_t count,
                    float *out, void *stream);

/* Diagnostics: the sigmoid and tanh fe_env_rollout_lstm applies to nn.LSTM's gates (finenvs/agents/networks/lstm.py:28-34)
 * and to the actor's output (continuous_actor.py:112), elementwise on n device floats: pins them against the oracle. */
int fe_lstm_activations(const float *x, float *sigmoid_out, float *tanh_out, int64_t n, void *stream);

/*
 * Tuning only (tools/, never needed for correctness): override the tile size (envs per workgroup
 * tile) and grid of the step / reset kernels and the tile of the fused rollouts; 0 = automatic.
 */
int fe_env_set_launch(fe_env *env, int32_t tile_envs, int32_t grid, int32_t rollout_tile_envs);

#ifdef __cplusplus
}
#endif
#endif /* FINENVS_AMD_EXT_H */
