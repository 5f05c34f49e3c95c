/*
 * finenvs_amd.h -- C ABI of the MI355X-native TimeSeriesEnv hot path.
 *
 * The reference (hmomin/FinEnvs) has no FFI: its boundary is the duck-typed
 * Python protocol of finenvs/environments/time_series_env.py ("TSE").  This
 * header is the C-ABI a binding for that protocol calls; every entry point
 * names the reference method it replaces.  finenvs_amd/environments/
 * time_series_env.py is the ctypes binding.
 *
 * Conventions
 *  - plain C types only; every tensor argument is a DEVICE pointer (HIP
 *    memory, e.g. torch.Tensor.data_ptr()) unless the comment says host;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *    work is enqueued on it and nothing here synchronises with the host;
 *  - return value 0 on success, negative on error; fe_last_error() returns a
 *    thread-local description of the last failure;
 *  - the library creates no threads and owns no tensors: tables, state and
 *    outputs are caller-owned and must outlive the calls that use them (one
 *    exception: the log-return table fe_env_create computes when passed NULL);
 *  - multi-GPU processes: an env runs on the device its tables live on.  Every
 *    entry point makes that device current for the call and restores the
 *    caller's device afterwards, so the reference's `device_id` constructor
 *    argument (TSE:28, 45) works without the caller switching devices.
 *
 * This header is the FROZEN surface: the entry points SURVEY.md 8(b) lists plus the rows of 8(f)
 * (trajectory outputs / host flag, descriptors + render, statistics, tables, CSV).  The in-kernel
 * policy heads (table / MLP / LSTM) and the tuning hook live in finenvs_amd_ext.h, marked
 * experimental; the same library exports both.
 *
 * There is deliberately NO CPU implementation behind this ABI: without a GPU
 * every compute entry point fails with FE_ERR_HIP.
 */
#ifndef FINENVS_AMD_H
#define FINENVS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FE_ABI_VERSION 5
#define FE_MAX_ASSETS 256

#define FE_OK 0
#define FE_ERR_ARG -1  /* invalid argument / configuration */
#define FE_ERR_HIP -2  /* HIP runtime error (no device, launch failure, ...) */
#define FE_ERR_STATE -3 /* call not possible in the env's current state (before fe_env_bind_state; see the entry points) */

/* Constructor arguments of TimeSeriesEnv (TSE:15-29) plus the build's extensions. */
typedef struct fe_config {
    int64_t N;               /* num_envs (reference: days [+1], TSE:246-257)        */
    int64_t D;               /* days = price_environments.shape[0]                  */
    int64_t L;               /* padded day length = shape[1] (max_length, TSE:136)  */
    int32_t W;               /* num_intervals, TSE:19                               */
    int32_t A;               /* assets per env; 1 in the reference                  */
    int32_t max_shares;      /* TSE:20                                              */
    int32_t evaluate;        /* TSE:27                                              */
    double starting_balance; /* TSE:21                                              */
    double commission;       /* per_share_commission, TSE:22                        */
    double init_margin;      /* initial_margin_requirement, TSE:25                  */
    double maint_margin;     /* maintenance_margin_requirement, TSE:26              */
    int32_t obs_is_f32;      /* 0: f64 observations (reference), 1: f32             */
    int32_t redraw_mode;     /* eval-env day redraw (TSE:510-513): 0 = caller draws
                                and calls fe_env_set_day, 1 = device Philox4x32-10  */
    uint64_t seed;           /* Philox key for redraw_mode 1                        */
    int64_t eval_env;        /* local index of the training-mode eval env or -1     */
} fe_config;

typedef struct fe_env fe_env; /* opaque */

/* FE_ABI_VERSION of the loaded library. */
int fe_version(void);

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *fe_last_error(void);

/* Number of HIP devices visible (0 without a GPU); never fails. */
int fe_device_count(void);

/*
 * Replaces TimeSeriesEnv.__init__'s device-side setup (TSE:245-269 minus tensor
 * allocation).  `prices` and `logret` are the (D, L, 4*A) f64 NaN-padded tables
 * price_environments / log_return_environments (TSE:215-216), asset a in
 * columns 4a..4a+3 = O,H,L,C.  `cfg` is a host pointer.  The env is created on the
 * device `prices` lives on (not the caller's current device).
 * logret == NULL: the library computes the table from `prices` with the transform of
 * generate_log_return_dataset (TSE:179-194) applied per day slice and owns it until
 * fe_env_destroy (fe_env_logret returns it).  One entry per day differs from the table
 * the reference builds from the whole series: the open-over-previous-close feature of a
 * slice's row 0, whose previous close lies outside the slice, takes the rule the reference
 * applies to the first row of a series (open over open, i.e. 0; TSE:188-190).  Callers that
 * need the reference's value there pass the table built by fe_build_logret + fe_build_tables.
 */
int fe_env_create(const fe_config *cfg, const double *prices, const double *logret, fe_env **out);

/*
 * Binds the caller-owned state tensors (TSE:246-275):
 *   env_idx (N) i64 = env_indices; spot0 (N) i64 = env_spots[:,0] (env_spots[n][j]
 *   == spot0[n]+j and env_pointers == spot0, so neither is stored);
 *   cash, long_shares, short_shares (N*A) f32; margin (N*A) f64;
 *   terminated (N) u8 and episode_returns (N) f32 = evaluate-mode metrics
 *   (TSE:271-275; may be NULL when cfg.evaluate == 0);
 *   counters (2) i64: [0] number of terminated envs, [1] redraw counter.
 */
int fe_env_bind_state(fe_env *env, int64_t *env_idx, int64_t *spot0, float *cash, float *long_shares,
                      float *short_shares, double *margin, uint8_t *terminated, float *episode_returns,
                      int64_t *counters);

/*
 * Replaces TimeSeriesEnv.reset() (TSE:423-445): renders the observation of the
 * CURRENT state into obs (N, W, 5*A), f64 or f32 per cfg.obs_is_f32.  Changes no state.
 */
int fe_env_reset_obs(fe_env *env, void *obs, void *stream);

/*
 * Replaces TimeSeriesEnv.step() (TSE:277-296 and everything it calls, TSE:298-536):
 * one fused launch.  actions (N*A) f32 in [-1,1]; obs (N, W, 5*A); rewards (N) f64;
 * dones (N) i32.  In evaluate mode the rewards of already-terminated envs are
 * zeroed and episode_returns accumulated exactly as record_evaluation_metrics
 * does (TSE:523-536); the caller reads counters[0] to learn when to emit them.
 */
int fe_env_step(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                void *stream);

/*
 * fe_env_step for the caller that must know on the HOST, every step, whether the evaluation env finished -- the
 * reference's `if self.dones[-1].item():` (TSE:504-513), which decides whether ONE draw is taken from torch's global
 * generator before anything else samples from it (redraw_mode 0: exact RNG-stream parity).  Instead of a device-to-host
 * copy of dones + a stream synchronisation after the whole launch, the kernel accounts the tile that holds the
 * evaluation env (cfg.eval_env) FIRST and stores (seq << 1) | done into *host_flag -- host memory from
 * fe_host_flag_create, system scope -- a few microseconds after the launch starts; the host polls the flag until it
 * carries `seq` (any value that changes per call) and can issue the next launches while this one is still streaming.
 * Tile order is the only difference to fe_env_step: results are identical.  FE_ERR_ARG for a training-mode env without an
 * evaluation env (a shard that does not own it).  fe_env_step_traj_notify is the same with fe_env_step_traj's optional
 * trajectory outputs (each may be NULL); bound episode statistics are updated by both.
 * EVALUATE-mode envs have no evaluation env; their per-step host read is `torch.all(self.terminated_episodes)` (TSE:531),
 * a fact of the whole launch: there the LAST workgroup to finish stores (seq << 32) | counters[0] (how many envs have
 * terminated so far; seq < 2^31) into *host_flag, and the host compares the count with N.
 */
int fe_env_step_notify(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                       uint64_t *host_flag, uint64_t seq, void *stream);
int fe_env_step_traj_notify(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                            float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out, uint64_t *host_flag,
                            uint64_t seq, void *stream);
/*
 * step() of an env whose share tensors the reference has PROMOTED to f64.  TimeSeriesEnv keeps long_shares / short_shares
 * as f32 tensors only until the first step() with float64 actions: sell_long_positions / buy_back_short_positions rebind
 * them to f64 results (TSE:353-361, 367-374), and they stay f64 for the life of the env -- also under later f32 actions.
 * Share counts are small integers, identical in either dtype; what changes is the precision of the products they enter:
 *   shares * per_share_commission is an f64 product added with one rounding into the f32 commission accumulator
 *   (TSE:363-365) -- for the sell / buy-back legs always, for the entry legs when this step's share CHANGES are f64
 *   (actions_are_f64 = 1; with f32 actions they remain f32 products);  the short-entry commission likewise (TSE:401-421);
 *   the share change is scaled, rounded and clamped in the actions' dtype (TSE:298-302);  the liquidation fee
 *   dones * num_shares * per_share_commission is an f64 product (TSE:288-289).
 * The caller owns the "has been promoted" bit (the Python class sets it at the first f64 step and routes every later step
 * here); the state arrays stay f32 / f64 as bound (same values).  actions: (N*A) f64 if actions_are_f64 else f32.
 * Optional outputs as fe_env_step_traj (actions_store_out only with f32 actions); host_flag / seq as fe_env_step_notify
 * (NULL: no flag).  Same kernel structure as fe_env_step (single-asset envs with f32 observations take the tile loop).
 */
int fe_env_step_promoted(fe_env *env, const void *actions, int32_t actions_are_f64, void *obs, double *rewards,
                         int32_t *dones, float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out,
                         uint64_t *host_flag, uint64_t seq, void *stream);

/* A host-resident, device-visible, coherent 8-byte flag for fe_env_step_notify (TSE:510 is the host read it serves); zeroed. */
int fe_host_flag_create(uint64_t **host_flag);
int fe_host_flag_destroy(uint64_t *host_flag);

/*
 * Optional, f32 observations only: an f32 copy (D, L, 4*A) of the log-return table, cast by the
 * caller ((float) of every entry).  Observations then stream from it -- half the L2 traffic of the
 * f64 table, identical values (TSE:423-445 followed by the callers' states.float(),
 * finenvs/agents/PPO/PPO_agent.py:101).  NULL unbinds.
 */
int fe_env_bind_f32_table(fe_env *env, const float *logret_f32);

/*
 * Optional episode statistics fused into fe_env_step: replaces the per-step
 * bookkeeping of the reference's agents (finenvs/agents/PPO/PPO_agent.py:120-132: running
 * return per env, returns of finished training episodes, the eval env's return -- and its
 * per-step .item() sync).  running_returns (N) f32; accumulators (N, 3) f64 = PER-ENV partial
 * sums {finished training episodes of env n, sum of their returns, sum of squares} in columns
 * 0, 1, 2, each slot written only by the lane that owns env n (so the sums do not depend on the
 * tile walk, launch geometry or form of the step kernel); eval_return (2) f32 = {return of
 * the eval env's last finished episode, number it finished}.  All device pointers, caller
 * zero-initialised (and re-zeroed by the caller after a report, PPO_agent.py:165-168); pass
 * running_returns = NULL to unbind.
 * fe_env_stats_reduce: out (3) f64 device = {episodes, sum, sum of squares} over all envs -- what
 * log_progress turns into len / mean / std (PPO_agent.py:146-163) -- added up in a FIXED order
 * (one workgroup of 1024 lanes: per column, lane t adds envs t, t + 1024, ... ascending from +0.0, then a
 * halving tree s[t] += s[t + stride], stride 512 ... 1; restated by oracle/fe_oracle.c:
 * fo_stats_reduce), so mean / std are bit-stable from run to run.  Log-time only.
 */
int fe_env_bind_stats(fe_env *env, float *running_returns, double *accumulators, float *eval_return);
int fe_env_stats_reduce(fe_env *env, double *out, void *stream);

/*
 * K env steps in ONE launch with an in-kernel linear policy (fused rollout with a
 * policy hook).  The policy sees the observation the reference's loop would feed it --
 * states = env.reset(), then the obs returned by the previous step
 * (examples/time_series/PPO_LSTM_training_SPY.py:22-28) -- but the observation is never
 * written to HBM: it is described by obs_src (N) i64 / obs_pos (N*A) f64 (window offset into the
 * log-return table + position feature), which fe_env_describe initialises from the current
 * state (reset() semantics), fe_env_rollout_linear advances, and fe_env_render turns into the
 * (N, W, 5*A) tensor on demand.
 *   action[n][a] = clamp(bias + sum_j sum_c obs[n][j][5a+c] * weights[j][c], -1, 1), cast to f32;
 *   the sum runs per 64-lane wavefront: lane l adds rows j = l, l+64, ... in order (c = 0..4),
 *   then a butterfly over lane xor 32,16,8,4,2,1 -- part of the contract (bit-reproducible).
 * weights (W, 5) f64; actions_out (K, N*A) f32 may be NULL; rewards_out (K, N) f64;
 * dones_out (K, N) i32.  State, evaluate-mode metrics and bound statistics advance exactly as K
 * calls of fe_env_step would.
 */
int fe_env_describe(fe_env *env, int64_t *obs_src, double *obs_pos, void *stream);
int fe_env_render(fe_env *env, const int64_t *obs_src, const double *obs_pos, void *obs, void *stream);
int fe_env_rollout_linear(fe_env *env, const double *weights, double bias, int32_t K, int64_t *obs_src,
                          double *obs_pos, float *actions_out, double *rewards_out, int32_t *dones_out,
                          void *stream);

/*
 * fe_env_step that ALSO writes the trajectory fields a rollout loop keeps per step
 * (examples/time_series/PPO_LSTM_training_SPY.py:26-28: agent.store(states, actions, ...); states = next_states;
 * finenvs/agents/PPO/buffer.py:33-56), each optional (NULL):
 *   actions_store_out (N*A) f32: a copy of `actions` -- the policy's output stays where the policy wrote it (hot), the
 *     trajectory slot gets its copy from the kernel that reads it anyway (no copy launch);
 *   obs_src_out (N) i64 + obs_pos_out (N*A) f64 (together): the observation this step returns as descriptors, the pair
 *     fe_env_render / fe_env_render_n turn back into it (the terminal window on done steps, TSE:321, not the reset state
 *     fe_env_describe would describe): 8 + 8A bytes per env-step for the `states` field.
 * rewards / dones may point into the trajectory as with fe_env_step.
 */
int fe_env_step_traj(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                     float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out, void *stream);

/*
 * fe_env_render for ANY number of descriptors: obs_src (count) i64, obs_pos (count*A) f64 -> obs (count, W, 5*A) in
 * the env's observation dtype.  With fe_env_describe this is the `states` field of the reference's PPO buffer without
 * its bytes: the buffer keeps every step's observation (finenvs/agents/PPO/buffer.py:33-56, one torch.cat per step;
 * 40*W*A bytes per env-step) and training indexes minibatches out of it (finenvs/agents/PPO/PPO_agent.py:175-188);
 * a trajectory of descriptors costs 8 + 8*A bytes per env-step and is rendered per minibatch -- on any rank, since the
 * tables are replicated (finenvs_amd/trajectory.py, states=True).
 */
int fe_env_render_n(fe_env *env, const int64_t *obs_src, const double *obs_pos, int64_t count, void *obs, void *stream);

/*
 * Debug check for descriptors that did not come from THIS env object -- a chunk all-gathered from other ranks
 * (finenvs_amd/trajectory.py), a caller-filled buffer: fe_env_render_n / fe_lstm_forward / the fused rollouts use
 * obs_src as an element offset into the log-return table without a range check (a rank built with another W, D or L,
 * or an uninitialised row, would read out of bounds -- a GPU memory fault, not an error code).  A descriptor is valid
 * iff  obs_src >= 0,  obs_src % (4*A) == 0  and  obs_src / (4*A) + W <= D*L  (the window lies inside the table; the
 * reference has no counterpart: its buffer stores the observations themselves, finenvs/agents/PPO/buffer.py:33-56).
 * Synchronises `stream`.  Returns FE_OK and *first_bad = -1 if all `count` descriptors are valid; FE_ERR_ARG,
 * *first_bad = index of the first invalid one and fe_last_error() naming its value otherwise.  first_bad: HOST pointer.
 */
int fe_env_check_descriptors(fe_env *env, const int64_t *obs_src, int64_t count, int64_t *first_bad, void *stream);

/* env_indices[env] = day (the host half of TSE:510-513 when redraw_mode == 0): a one-lane launch on `stream`, no host synchronisation. */
int fe_env_set_day(fe_env *env, int64_t env_index, int64_t day, void *stream);

/* Launch geometry chosen for this env (diagnostics / bench reporting). Host pointers. */
int fe_env_launch_info(const fe_env *env, int32_t *grid, int32_t *block, int32_t *tile_envs,
                       int32_t *lds_bytes);

/* "" for the product build; experiment builds report their -D set (and are refused by the default loader). */
const char *fe_build_tag(void);

int fe_env_destroy(fe_env *env);

/* HIP device index the env runs on (that of its price table). */
int fe_env_device(const fe_env *env);

/* The (D, L, 4*A) log-return table the env reads: the caller's, or the one fe_env_create computed. */
const double *fe_env_logret(const fe_env *env);

/*
 * Replaces generate_log_return_dataset (TSE:179-194): whole-series transform,
 * prices/out (T, 4*A) f64.
 */
int fe_build_logret(const double *prices, double *out, int64_t T, int32_t A, void *stream);

/*
 * The same transform (TSE:179-194) applied to an already sliced, NaN-padded price table
 * (D, L, 4*A) -> out (D, L, 4*A): what fe_env_create(logret = NULL) computes.  Row 0 of every day
 * takes the series-row-0 rule for its open-over-previous-close entry (see fe_env_create).
 */
int fe_build_logret_tables(const double *prices, double *out, int64_t D, int64_t L, int32_t A, void *stream);

/*
 * Replaces generate_environments (TSE:196-216): out[d][r] = series[starts[d]+r]
 * for r <= stops[d]-starts[d], NaN beyond; series (T, 4*A), out (D, L, 4*A),
 * starts/stops (D) i64 device pointers.
 */
int fe_build_tables(const double *series, const int64_t *starts, const int64_t *stops, int64_t D,
                    int64_t L, int32_t A, double *out, void *stream);

/*
 * Trajectory ring (replaces the per-step torch.cat of
 * finenvs/agents/PPO/buffer.py:33-56): copies one step's compact fields into
 * slot t of time-major buffers actions (T, N*A) f32, rewards (T, N) f64,
 * dones (T, N) i32.
 */
int fe_traj_store(int64_t t, int64_t N, int32_t A, const float *actions, const double *rewards,
                  const int32_t *dones, float *traj_actions, double *traj_rewards, int32_t *traj_dones,
                  void *stream);

/*
 * Replaces Buffer.compute_returns_and_advantages (buffer.py:80-100) on time-major
 * (T, N) buffers: rewards f64, dones i32, values f32 (T, N), last_values f32 (N);
 * writes returns and advantages f32 (T, N).  One reverse scan per env.
 */
int fe_traj_returns(const double *rewards, const int32_t *dones, const float *values,
                    const float *last_values, int64_t T, int64_t N, double gamma, float *returns,
                    float *advantages, void *stream);

/*
 * Native CSV reader: replaces read_data + force_market_hours (TSE:80-91).
 * Rows are Date,Time,Open,High,Low,Close,Volume.  ALL POINTERS HERE ARE HOST POINTERS.
 * fe_csv_count_lines gives an upper bound for `capacity`.  fe_csv_read fills
 * prices (rows, 4) f64 = O,H,L,C, day_id (rows) = index of the row's date in order of
 * first appearance, date_key (rows; may be NULL) = join key of the row's date, < 2^45: yyyymmdd when
 * the text is a calendar date (YYYY-MM-DD or MM/DD/YYYY, either separator -- so files that spell
 * dates differently still join), else 2^44 + 44 bits of the text's FNV-1a hash; date_key * 2^17 +
 * second_of_day is a collision-free-by-construction row key for calendar dates; second_of_day (rows); with
 * market_hours_only != 0 only rows with 09:30:00 <= time <= 15:59:00 are kept.
 * Both return the row count, or a negative FE_ERR_* code.
 */
int64_t fe_csv_count_lines(const char *path);
int64_t fe_csv_read(const char *path, int64_t capacity, int32_t market_hours_only, double *prices,
                    int64_t *day_id, int64_t *date_key, int64_t *second_of_day);

#ifdef __cplusplus
}
#endif
#endif /* FINENVS_AMD_H */
