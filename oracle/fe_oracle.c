/*
 * fe_oracle.c -- CPU restatement of the reference's TimeSeriesEnv hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a scalar, plain-C
 * restatement of hmomin/FinEnvs finenvs/environments/time_series_env.py
 * (abbreviated TSE below).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (finenvs_amd/) never
 * links, imports or calls anything in oracle/; it has no CPU fallback.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here
 * bit-for-bit against fixtures under tests/golden/ that were produced by
 * running the reference itself in the build container
 * (oracle/make_goldens.py is the generating script).
 *
 * Every (float)/(double) cast below is a real rounding point of the
 * reference's mixed f32/f64 tensor arithmetic; build with -ffp-contract=off.
 *
 * Multi-asset (A > 1) has no reference: it is the "sleeve" contract of
 * DESIGN.md -- each (env, asset) pair is one independent reference account on
 * a shared calendar; reward = sum over assets in asset order; done = OR.
 * With A == 1 every formula below is exactly the reference's.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct fo_config {
    int64_t N, D, L;          /* envs, days, padded day length            */
    int32_t W, A;             /* window (num_intervals), assets           */
    int32_t max_shares;       /* TSE:20                                   */
    int32_t evaluate;         /* TSE:27                                   */
    double starting_balance;  /* TSE:21                                   */
    double commission;        /* TSE:22                                   */
    double init_margin;       /* TSE:25                                   */
    double maint_margin;      /* TSE:26                                   */
    int32_t obs_is_f32;       /* build extension: observation dtype       */
    int32_t redraw_mode;      /* 0 host (torch RNG), 1 device Philox      */
    uint64_t seed;            /* Philox key for redraw_mode 1             */
    int64_t eval_env;         /* index of the training-mode eval env, -1  */
} fo_config;

/* ---- Philox4x32-10 (Salmon et al. 2011), the device redraw generator ---- */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

uint32_t fo_philox_u32(uint64_t seed, uint64_t counter) {
    uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0x46454e56u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    return c[0];
}

int64_t fo_redraw_day(uint64_t seed, uint64_t counter, int64_t D) {
    return (int64_t)(((uint64_t)fo_philox_u32(seed, counter) * (uint64_t)D) >> 32);
}

/* ---- a18: log-return transform over the whole filtered series, TSE:179-194 ---- */
void fo_build_logret(const double *prices, double *out, int64_t T, int32_t A) {
    const int64_t stride = 4 * (int64_t)A;
    for (int64_t t = 0; t < T; ++t) {
        for (int32_t a = 0; a < A; ++a) {
            const double *p = prices + t * stride + 4 * a;
            double *o = out + t * stride + 4 * a;
            double open = p[0];
            /* previous close; the first row uses its own open (TSE:188-190) */
            double prev = (t == 0) ? open : prices[(t - 1) * stride + 4 * a + 3];
            o[0] = 100.0 * log(open / prev);
            for (int k = 1; k < 4; ++k) o[k] = 100.0 * log(p[k] / open);
        }
    }
}

/* ---- a19: per-day slices [start, stop], NaN-padded to L, TSE:196-216 ---- */
void fo_build_tables(const double *series, int64_t T, int32_t A, const int64_t *starts,
                     const int64_t *stops, int64_t D, int64_t L, double *out) {
    const int64_t stride = 4 * (int64_t)A;
    (void)T;
    for (int64_t d = 0; d < D; ++d) {
        int64_t len = stops[d] - starts[d] + 1;
        for (int64_t r = 0; r < L; ++r) {
            double *o = out + (d * L + r) * stride;
            if (r < len) {
                memcpy(o, series + (starts[d] + r) * stride, (size_t)stride * sizeof(double));
            } else {
                for (int64_t k = 0; k < stride; ++k) o[k] = NAN;
            }
        }
    }
}

/* ---- a20: episode bounds, TSE:127-152.  day_id is the per-row date label of the
 * market-hours-filtered frame, in file order. ---- */
int64_t fo_bounds(const int64_t *day_id, int64_t T, int32_t W, int64_t *starts, int64_t *stops,
                  int64_t *max_length) {
    int64_t D = 0, maxlen = 0, t = 0;
    while (t < T) {
        int64_t first = t, d = day_id[t];
        while (t < T && day_id[t] == d) ++t;
        int64_t last = t - 1;
        int64_t start = first - W; /* backtrack by the window, TSE:151 */
        if (start < 0) continue;   /* TSE:134-135 */
        if (last - start + 1 > maxlen) maxlen = last - start + 1;
        starts[D] = start;
        stops[D] = last;
        ++D;
    }
    *max_length = maxlen;
    return D;
}

#define FO_MAX_ASSETS 256

/* torch.relu lets a NaN through (clamp_min semantics) */
static inline float relu32(float x) { return x > 0.0f ? x : (x != x ? x : 0.0f); }
static inline double relu64(double x) { return x > 0.0 ? x : (x != x ? x : 0.0); }

/* ---- a3: TSE:298-302 ---- */
static inline float share_change(const fo_config *c, float action) {
    float scaled = action * (float)((double)c->max_shares + 0.5);
    float sc = rintf(scaled); /* round-half-even, as torch.round */
    float ms = (float)c->max_shares;
    if (sc < -ms) sc = -ms;
    if (sc > ms) sc = ms;
    return sc;
}

typedef struct sleeve_out {
    double pos_obs;  /* position feature of the observation  */
    double rew;      /* reward before the liquidation fee    */
    int bankrupt;    /* cash < 0 anywhere in the reward step */
} sleeve_out;

/* a3 with float64 actions: the same three operations in f64 (TSE:298-302); the result is an integer in
 * [-max_shares, max_shares] (or NaN), exact as a float */
static inline float share_change64(const fo_config *c, double action) {
    double scaled = action * ((double)c->max_shares + 0.5);
    double sc = rint(scaled);
    double ms = (double)c->max_shares;
    if (sc < -ms) sc = -ms;
    if (sc > ms) sc = ms;
    return (float)sc;
}

/* one (env, asset) account: steps 1-6 of SURVEY Appendix A.
 * sh64 / a64: the dtype promotion the reference goes through once step() has been given float64 actions.  Its
 * long_shares / short_shares tensors start as f32 (TSE:246-251) and are REBOUND to the results of expressions that involve
 * the share changes (TSE:353-361, 367-374): f64 from the first f64-action step on, for the life of the env (sh64).  Share
 * counts are small integers -- identical values in either dtype, kept as floats here -- so only products change:
 *   num_shares * per_share_commission (TSE:363-365) is an f64 product, added into the f32 `commissions` tensor in place
 *   (one rounding of the exact f64 sum), when num_shares is f64: the sold / bought-back counts under sh64, the entry
 *   counts only when THIS step's share changes are f64 (a64: f64 actions; f32 actions give f32 share-change tensors);
 *   short_commission (TSE:401-421) is in the share changes' dtype as well. */
static inline void sleeve_step(const fo_config *c, float action, double action64, int a64, int sh64, const double bar[4],
                               float *cash_io, float *long_io, float *short_io, double *margin_io, sleeve_out *out) {
    const double O = bar[0], H = bar[1], Lo = bar[2], C = bar[3];
    const double comm_d = c->commission;
    const float c32 = (float)c->commission;
    const double imr = c->init_margin;
    const float imr32 = (float)c->init_margin;
    const double one_mmr = 1.0 + c->maint_margin;
    float cash = *cash_io, lng = *long_io, sht = *short_io;
    double margin = *margin_io;
    float comm = 0.0f; /* TSE:305 */

#define FO_ADD_COMMISSION(shares, wide) \
    do { if (wide) comm = (float)((double)comm + (double)(shares) * comm_d); else comm += (shares) * c32; } while (0)
    float sc = a64 ? share_change64(c, action64) : share_change(c, action);
    float pos = sc < 0.0f ? 0.0f : sc; /* TSE:344-351 */
    float neg = sc > 0.0f ? 0.0f : sc;

    /* 1 sell long, TSE:353-361 */
    float nl = relu32(lng + neg);
    float sell = lng - nl;
    neg += sell;
    FO_ADD_COMMISSION(sell, sh64);
    cash = (float)((double)cash + (double)sell * (O - comm_d));
    lng = nl;

    /* 2 buy back short, TSE:367-383 */
    float ns = relu32(sht - pos);
    float bb = sht - ns;
    pos -= bb;
    FO_ADD_COMMISSION(bb, sh64);
    cash = (float)((double)cash - (double)bb * (O + comm_d));
    sht = ns;
    /* new_margin = imr * short_shares * open (TSE:376-379): the first product is in short_shares' dtype -- an f32 product
     * with the f32-rounded imr until the promotion, an f64 product with the full imr after it */
    double nm = sh64 ? (imr * (double)sht) * O : (double)(imr32 * sht) * O;
    cash = (float)((double)cash - (nm - margin));
    margin = nm;

    /* 3 long entry, TSE:385-399 */
    if ((double)cash - (double)pos * (O + comm_d) < 0.0) pos = 0.0f;
    FO_ADD_COMMISSION(pos, a64);
    cash = (float)((double)cash - (double)pos * (O + comm_d));
    lng += pos;

    /* 4 short entry, TSE:401-421 */
    float q = -neg;
    if (((double)cash - imr * ((double)q * O)) - (a64 ? (double)q * comm_d : (double)(q * c32)) < 0.0) {
        neg = 0.0f;
        q = -neg;
    }
    FO_ADD_COMMISSION(q, a64);
    double req = imr * ((double)q * O);
    cash = (float)((double)cash - (req + (a64 ? (double)q * comm_d : (double)(q * c32))));
#undef FO_ADD_COMMISSION
    margin += req;
    sht += q;

    /* 5 observation feature, rendered post-trade, TSE:428-431 */
    out->pos_obs = (double)(lng - sht) * C / c->starting_balance;

    /* 6 reward, TSE:447-475 */
    int done = cash < 0.0f;
    double rew;
    {   /* maintenance margin check at the High */
        double call = relu64((double)sht * H * one_mmr - margin);
        cash = (float)((double)cash - call);
        margin += call;
        done |= cash < 0.0f;
        rew = -call;
    }
    {   /* margin release at the Low */
        double rel = relu64(margin - (double)sht * Lo * imr);
        margin -= rel;
        cash = (float)((double)cash + rel);
    }
    {   /* maintenance margin check at the Close */
        double call = relu64((double)sht * C * one_mmr - margin);
        cash = (float)((double)cash - call);
        margin += call;
        done |= cash < 0.0f;
        rew += -call;
    }
    if (done) { lng = 0.0f; sht = 0.0f; }
    rew += (double)(lng - sht) * (C - O);
    rew -= (double)comm;

    *cash_io = cash; *long_io = lng; *short_io = sht; *margin_io = margin;
    out->rew = rew;
    out->bankrupt = done;
}

/*
 * a2: one step() of all N envs, TSE:277-296.
 *
 * State (all host pointers): env_idx (N) i64, spot0 (N) i64 (first window
 * row; env_spots[n][j] == spot0[n]+j and env_pointers == spot0 in the
 * reference), cash/long/short (N*A) f32, margin (N*A) f64.
 * Evaluate-mode state (may be NULL unless cfg->evaluate): terminated (N) u8,
 * episode_returns (N) f32, n_terminated (1) i64.
 * redraw_counter (1) u64 is used by redraw_mode 1 only.
 * obs is (N, W, 5A) f64 (or f32 when cfg->obs_is_f32).
 */
/* fo_step_ex: `actions` is (N*A) f64 when act_f64 else f32; shares_f64: the env's share tensors have been promoted to f64
 * (see sleeve_step; act_f64 implies it).  fo_step below = the all-f32 case every in-repo caller of the reference runs. */
int fo_step_ex(const fo_config *c, const double *P, const double *LR, int64_t *env_idx, int64_t *spot0,
               float *cash, float *lng, float *sht, double *margin, uint8_t *terminated,
               float *episode_returns, int64_t *n_terminated, uint64_t *redraw_counter,
               const void *actions_any, int act_f64, int shares_f64, void *obs, double *rew_out, int32_t *done_out,
               int nthreads) {
    const float *actions = (const float *)actions_any;
    const double *actions64 = (const double *)actions_any;
    if (act_f64) shares_f64 = 1;
    const int64_t N = c->N, L = c->L;
    const int32_t W = c->W, A = c->A;
    const int64_t rs = 4 * (int64_t)A; /* table row stride in doubles */
    const float c32 = (float)c->commission;
    if (A > FO_MAX_ASSETS) return -1;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t n = 0; n < N; ++n) {
        double posv[FO_MAX_ASSETS], rewv[FO_MAX_ASSETS];
        int64_t idx = env_idx[n];
        int64_t s0 = spot0[n] + 1;          /* TSE:281-282 */
        int64_t last = s0 + W - 1;          /* env_spots[:, -1] */
        int64_t nxt = last + 1;             /* TSE:480 */
        const double *Pd = P + idx * L * rs;
        const double *LRd = LR + idx * L * rs;
        int any_done = 0;
        for (int32_t a = 0; a < A; ++a) {
            sleeve_out so;
            sleeve_step(c, act_f64 ? 0.0f : actions[n * A + a], act_f64 ? actions64[n * A + a] : 0.0, act_f64, shares_f64,
                        Pd + last * rs + 4 * a, &cash[n * A + a], &lng[n * A + a], &sht[n * A + a], &margin[n * A + a], &so);
            posv[a] = so.pos_obs;
            rewv[a] = so.rew;
            /* 7 termination, TSE:477-496 */
            int done = so.bankrupt;
            done |= (nxt >= L);
            if (!done) done = isnan(LRd[nxt * rs + 4 * a]);
            any_done |= done;
        }
        /* liquidation fee on every sleeve of a finished env, TSE:288-289 */
        double rew = 0.0;
        for (int32_t a = 0; a < A; ++a) {
            double r;
            if (shares_f64) {  /* num_shares is an f64 tensor: dones * num_shares * per_share_commission in f64 */
                r = rewv[a] - ((any_done ? 1.0 : 0.0) * ((double)sht[n * A + a] + (double)lng[n * A + a])) * c->commission;
            } else {
                float fee = ((any_done ? 1.0f : 0.0f) * (sht[n * A + a] + lng[n * A + a])) * c32;
                r = rewv[a] - (double)fee;
            }
            rew = (a == 0) ? r : rew + r;
        }
        /* observation: terminal window + position column, TSE:423-445 */
        if (c->obs_is_f32) {
            float *o = (float *)obs + n * (int64_t)W * 5 * A;
            for (int32_t j = 0; j < W; ++j)
                for (int32_t a = 0; a < A; ++a) {
                    const double *src = LRd + (s0 + j) * rs + 4 * a;
                    float *dst = o + ((int64_t)j * A + a) * 5;
                    for (int k = 0; k < 4; ++k) dst[k] = (float)src[k];
                    dst[4] = (float)posv[a];
                }
        } else {
            double *o = (double *)obs + n * (int64_t)W * 5 * A;
            for (int32_t j = 0; j < W; ++j)
                for (int32_t a = 0; a < A; ++a) {
                    const double *src = LRd + (s0 + j) * rs + 4 * a;
                    double *dst = o + ((int64_t)j * A + a) * 5;
                    for (int k = 0; k < 4; ++k) dst[k] = src[k];
                    dst[4] = posv[a];
                }
        }
        /* 8 episodic reset, TSE:498-521 */
        if (any_done) {
            for (int32_t a = 0; a < A; ++a) {
                cash[n * A + a] = (float)c->starting_balance;
                margin[n * A + a] = 0.0;
                lng[n * A + a] = 0.0f;
                sht[n * A + a] = 0.0f;
            }
            s0 = 0;
            if (!c->evaluate && c->redraw_mode == 1 && n == c->eval_env) {
                /* only one env ever draws, so no race under OpenMP */
                env_idx[n] = fo_redraw_day(c->seed, *redraw_counter, c->D);
                *redraw_counter += 1;
            }
        }
        spot0[n] = s0;
        /* a17 evaluate-mode bookkeeping, TSE:523-536 */
        if (c->evaluate) {
            if (terminated[n]) rew = 0.0;
            if (any_done && !terminated[n]) {
                terminated[n] = 1;
#ifdef _OPENMP
#pragma omp atomic
#endif
                *n_terminated += 1;
            }
            episode_returns[n] = (float)((double)episode_returns[n] + rew);
        }
        rew_out[n] = rew;
        done_out[n] = any_done;
    }
    return 0;
}

int fo_step(const fo_config *c, const double *P, const double *LR, int64_t *env_idx, int64_t *spot0,
            float *cash, float *lng, float *sht, double *margin, uint8_t *terminated,
            float *episode_returns, int64_t *n_terminated, uint64_t *redraw_counter,
            const float *actions, void *obs, double *rew_out, int32_t *done_out, int nthreads) {
    return fo_step_ex(c, P, LR, env_idx, spot0, cash, lng, sht, margin, terminated, episode_returns, n_terminated,
                      redraw_counter, actions, 0, 0, obs, rew_out, done_out, nthreads);
}

/* a10/a11: reset() only renders the observation of the current state, TSE:423-445 */
int fo_reset_obs(const fo_config *c, const double *P, const double *LR, const int64_t *env_idx,
                 const int64_t *spot0, const float *lng, const float *sht, void *obs) {
    const int64_t N = c->N, L = c->L;
    const int32_t W = c->W, A = c->A;
    const int64_t rs = 4 * (int64_t)A;
    for (int64_t n = 0; n < N; ++n) {
        int64_t idx = env_idx[n], s0 = spot0[n], last = s0 + W - 1;
        const double *Pd = P + idx * L * rs;
        const double *LRd = LR + idx * L * rs;
        for (int32_t j = 0; j < W; ++j)
            for (int32_t a = 0; a < A; ++a) {
                double C = Pd[last * rs + 4 * a + 3];
                double pos = (double)(lng[n * A + a] - sht[n * A + a]) * C / c->starting_balance;
                const double *src = LRd + (s0 + j) * rs + 4 * a;
                if (c->obs_is_f32) {
                    float *dst = (float *)obs + (n * (int64_t)W * A + (int64_t)j * A + a) * 5;
                    for (int k = 0; k < 4; ++k) dst[k] = (float)src[k];
                    dst[4] = (float)pos;
                } else {
                    double *dst = (double *)obs + (n * (int64_t)W * A + (int64_t)j * A + a) * 5;
                    for (int k = 0; k < 4; ++k) dst[k] = src[k];
                    dst[4] = pos;
                }
            }
    }
    return 0;
}

/*
 * f1: discounted returns over a (T, N) trajectory chunk, the loop of
 * finenvs/agents/PPO/buffer.py:80-100 with its tensor dtypes spelled out:
 * rewards f64, dones int32, last_values f32, gamma a Python float.
 * (1 - dones) * gamma is an int32 tensor times a Python scalar -> f32, so the
 * discount factor is gamma rounded to f32; the first product (with f32
 * last_values) is an f32 product, later ones are f64; returns are stored f32.
 * Layout here is time-major (T, N).
 */
void fo_discounted_returns(const double *rew, const int32_t *done, const float *last_values,
                           int64_t T, int64_t N, double gamma, float *returns_out) {
    const float g32 = (float)gamma;
    for (int64_t n = 0; n < N; ++n) {
        double R = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            float factor = (float)(1 - done[t * N + n]) * g32;
            if (t == T - 1)
                R = rew[t * N + n] + (double)(factor * last_values[n]);
            else
                R = rew[t * N + n] + (double)factor * R;
            returns_out[t * N + n] = (float)R;
        }
    }
}

/*
 * The dtype rule every cash update of TSE:353-475 rests on: an in-place `f32 op= f64` computes in f64
 * and rounds once on store.  Exposed on its own so that the known-answer probe of tests/golden/
 * rounding.npz (e.g. 1f += 2^-24 + 2^-50 -> 1.00000012) pins it outside of any rollout.
 */
void fo_f32_iadd_f64(const float *base, const double *delta, int64_t n, int32_t subtract, float *out) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = subtract ? (float)((double)base[i] - delta[i]) : (float)((double)base[i] + delta[i]);
}

/*
 * f4: the agents' per-step return bookkeeping, finenvs/agents/PPO/PPO_agent.py:120-132 (the same
 * lines exist in the TD3 / SAC agents): current_returns (f32) += rewards (f64); returns of finished
 * TRAINING episodes (all envs but the last) are collected -- here as count / sum / sum of squares in
 * f64 --; when the last env (the evaluation env) finishes, its return is recorded; finished envs'
 * running returns restart at 0.  acc = {count, sum, sum of squares}, eval = {last return, how many}.
 */
void fo_episode_stats_step(int64_t N, int64_t eval_env, const double *rew, const int32_t *done, float *running,
                           double *acc, float *eval) {
    /* acc is (N, 3): PER-ENV partial sums (count, sum, sum of squares of env n's finished training episodes) --
     * the layout fe_env_bind_stats documents; fo_stats_reduce adds the envs up in the build's fixed order. */
    for (int64_t n = 0; n < N; ++n) {
        float cr = (float)((double)running[n] + rew[n]);  /* PPO_agent.py:121 */
        if (done[n]) {
            if (n == eval_env) {                           /* PPO_agent.py:129-131 */
                eval[0] = cr;
                eval[1] += 1.0f;
            } else {                                       /* PPO_agent.py:122-128, 132 */
                acc[3 * n] += 1.0;
                acc[3 * n + 1] += (double)cr;
                acc[3 * n + 2] += (double)cr * (double)cr;
            }
            cr = 0.0f;
        }
        running[n] = cr;
    }
}

/*
 * The summation order of fe_env_stats_reduce (include/finenvs_amd.h), which turns the per-env partials into what
 * log_progress reports (len / mean / std of the finished-return list, PPO_agent.py:146-163): per column k of the
 * (N, 3) array, 1024 lanes, lane t adds envs t, t + 1024, ... in ascending order starting from +0.0, then a halving
 * tree s[t] += s[t + stride], stride = 512 ... 1.  out[k] = the reduced column k.
 */
void fo_stats_reduce(const double *acc, int64_t N, double *out) {
    enum { LANES = 1024 };
    double s[LANES];
    for (int k = 0; k < 3; ++k) {
        for (int t = 0; t < LANES; ++t) {
            double v = 0.0;
            for (int64_t j = t; j < N; j += LANES) v += acc[3 * j + k];
            s[t] = v;
        }
        for (int stride = LANES / 2; stride >= 1; stride >>= 1)
            for (int t = 0; t < stride; ++t) s[t] += s[t + stride];
        out[k] = s[0];
    }
}

/*
 * f2: the in-kernel linear policy of fe_env_rollout_linear, restated on a materialised
 * observation (N, W, 5A) f64.  The summation order is part of the contract: 64 partial sums
 * (lane l takes rows l, l+64, ... in order, features 0..4 inside a row), then a butterfly over
 * lane xor 32,16,8,4,2,1; action = (float)clamp(bias + sum, -1, 1); NaN passes through.
 */
void fo_policy_linear(const double *obs, const double *weights, double bias, int64_t N, int32_t W,
                      int32_t A, float *actions_out) {
    for (int64_t n = 0; n < N; ++n)
        for (int32_t a = 0; a < A; ++a) {
            double part[64];
            for (int l = 0; l < 64; ++l) {
                double acc = 0.0;
                for (int32_t j = l; j < W; j += 64) {
                    const double *row = obs + ((n * W + j) * (int64_t)A + a) * 5;
                    const double *wr = weights + (int64_t)j * 5;
                    acc += row[0] * wr[0];
                    acc += row[1] * wr[1];
                    acc += row[2] * wr[2];
                    acc += row[3] * wr[3];
                    acc += row[4] * wr[4];
                }
                part[l] = acc;
            }
            for (int m = 32; m >= 1; m >>= 1) {
                double nxt[64];
                for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ m];
                memcpy(part, nxt, sizeof(part));
            }
            double a64 = bias + part[0];
            a64 = a64 < -1.0 ? -1.0 : (a64 > 1.0 ? 1.0 : a64);
            actions_out[n * A + a] = (float)a64;
        }
}

/*
 * f2, table form (fe_policy_table + fe_env_rollout_table): the log-return part of the linear
 * policy as a per-(day, window start, asset) indicator table, same 64-partials + butterfly order
 * over the four log-return features; wsum = sum_j w[j][4] in that order.
 */
static double butterfly64(double part[64]) {
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ m];
        memcpy(part, nxt, sizeof(nxt));
    }
    return part[0];
}

void fo_policy_table(const double *LR, const double *weights, int64_t D, int64_t L, int32_t W, int32_t A,
                     double *table, double *wsum) {
    for (int64_t d = 0; d < D; ++d)
        for (int64_t s = 0; s < L; ++s)
            for (int32_t a = 0; a < A; ++a) {
                double part[64];
                if (s + W > L) {
                    table[(d * L + s) * A + a] = NAN;
                    continue;
                }
                for (int l = 0; l < 64; ++l) {
                    double acc = 0.0;
                    for (int32_t j = l; j < W; j += 64) {
                        const double *row = LR + (((d * L + s + j) * (int64_t)A) + a) * 4;
                        const double *wr = weights + (int64_t)j * 5;
                        acc += row[0] * wr[0];
                        acc += row[1] * wr[1];
                        acc += row[2] * wr[2];
                        acc += row[3] * wr[3];
                    }
                    part[l] = acc;
                }
                table[(d * L + s) * A + a] = butterfly64(part);
            }
    double part[64];
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int32_t j = l; j < W; j += 64) acc += weights[(int64_t)j * 5 + 4];
        part[l] = acc;
    }
    wsum[0] = butterfly64(part);
}

/* actions of the table form for observations described by (row = idx*L + window start, pos) */
void fo_policy_table_actions(const double *table, double wsum, double bias, const int64_t *obs_row,
                             const double *obs_pos, int64_t N, int32_t A, float *actions_out) {
    for (int64_t n = 0; n < N; ++n)
        for (int32_t a = 0; a < A; ++a) {
            double a64 = bias + (table[obs_row[n] * A + a] + obs_pos[n * A + a] * wsum);
            a64 = a64 < -1.0 ? -1.0 : (a64 > 1.0 ? 1.0 : a64);
            actions_out[n * A + a] = (float)a64;
        }
}

/*
 * f2, MLP head: CPU restatement of fe_env_rollout_mlp's policy on a materialised observation (N, W, 5A) f64.
 * Two-layer perceptron on the flattened window of each (env, asset) pair, all in f32 (the reference's agents feed
 * states.float() to their networks, PPO_agent.py:101; layer shape of multilayer_perceptron.py:17-25):
 *   pre[h]  = fmaf chain, start fmaf((float)pos, wpos[h], b1[h]), then the 4W log-return features in the order
 *             g = 0.. (two window rows per group, groups padded to a multiple of four), c = 0..3, row 2g before
 *             row 2g+1  (rows past W contribute fmaf(0, last row, acc)) -- the k order of the v_mfma_f32_32x32x2_f32
 *             chain of the HIP kernel;
 *   action  = clamp(b2 + (P0 + P1), -1, 1), P_half = fmaf chain over the hidden units
 *             32t + (r&3) + 8(r>>2) + 4*half, t and r ascending, of w2[h] * act(pre[h]).
 * act: 0 ELU (alpha 1, expm1f), 1 ReLU (NaN passes), 2 tanh (the build's exact-operation form, fo_lstm_tanh).  w1t is (H, 4W): w1t[h][4j+c] = W1[5j+c][h].
 */
float fo_lstm_tanh(float x);  /* the exact-operation tanh of the build's heads, below */
static float fo_mlp_act(float z, int act) {
    if (act == 1) return z > 0.0f ? z : (z != z ? z : 0.0f);
    if (act == 2) return fo_lstm_tanh(z);  /* bit-reproducible (ELU is not: the kernel's v_exp_f32 vs expm1f) */
    return z > 0.0f ? z : expm1f(z);
}

void fo_policy_mlp(const double *obs, const float *w1t, const float *wpos, const float *b1, const float *w2,
                   float b2, int32_t H, int32_t act, int64_t N, int32_t W, int32_t A, float *actions_out,
                   float *pre_out /* (N, A, H) or NULL */) {
    const int K4 = 4 * W, ngroups = ((K4 + 7) / 8 + 3) / 4 * 4;
    float *pre = (float *)malloc(sizeof(float) * (size_t)H);
    for (int64_t n = 0; n < N; ++n)
        for (int a = 0; a < A; ++a) {
            const double *o = obs + (size_t)n * W * 5 * A + 5 * a;
            const float pos32 = (float)o[4];
            for (int h = 0; h < H; ++h) {
                float acc = fmaf(pos32, wpos[h], b1[h]);
                for (int g = 0; g < ngroups; ++g)
                    for (int c = 0; c < 4; ++c)
                        for (int half = 0; half < 2; ++half) {
                            const int row = 2 * g + half;
                            const int rc = row < W ? row : W - 1;  /* padding rows: weight 0 times the last row */
                            const float x = (float)o[(size_t)rc * 5 * A + c];
                            const float w = row < W ? w1t[(size_t)h * K4 + 4 * row + c] : 0.0f;
                            acc = fmaf(w, x, acc);
                        }
                pre[h] = acc;
                if (pre_out) pre_out[((size_t)n * A + a) * H + h] = acc;
            }
            float part[2] = {0.0f, 0.0f};
            for (int half = 0; half < 2; ++half)
                for (int t = 0; t < H / 32; ++t)
                    for (int r = 0; r < 16; ++r) {
                        const int h = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                        part[half] = fmaf(w2[h], fo_mlp_act(pre[h], act), part[half]);
                    }
            float a32 = b2 + (part[0] + part[1]);
            a32 = a32 < -1.0f ? -1.0f : (a32 > 1.0f ? 1.0f : a32);
            actions_out[(size_t)n * A + a] = a32;
        }
    free(pre);
}

/* ---- LSTM head of the fused rollout (finenvs_amd fe_env_rollout_lstm) ----
 * The actor the reference's own time-series scripts use: finenvs/agents/networks/lstm.py:28-57 (nn.LSTM(5, H),
 * batch_first, zero initial state, then Linear(H, 1) on the LAST hidden state) with the Tanh output activation of
 * finenvs/agents/PPO/continuous_actor.py:104-126, applied to states.float()
 * (examples/time_series/PPO_LSTM_testing_SPY.py:46) per (env, asset) pair.  This is the order-exact CPU form of the
 * contract fe_env_rollout_lstm documents:
 *   * gate rows are PACKED: row R = 32*mt + 8*b + 4*half + gate holds gate (0 i, 1 f, 2 g, 3 o; torch's order)
 *     of hidden unit u = 8*mt + 4*half + b;  whh (4H, H) and wx (4H, 8) = [w_ih[.][0..3], w_ih[.][4], b_ih+b_hh, 0, 0]
 *   * pre-activation of a row at time t: an fmaf chain from 0: for m = 0..3: wx[m]*x[m] then wx[4+m]*xh[m]
 *     (x = the row's four log-returns as f32, xh = (position feature as f32, 1, 0, 0)); for t > 0 then, for
 *     g = 0..H/8-1, m = 0..3: whh[8g+m]*h[8g+m] then whh[8g+4+m]*h[8g+4+m]
 *   * sigmoid / tanh are the exact-operation forms below (rintf, fmaf, ldexpf, IEEE division only), so that the
 *     device reproduces them bit for bit
 *   * c = f*c + i*g (two products, one sum; c = i*g at t = 0), h = o*tanh(c)
 *   * action = tanh(fmaf chain over u ascending of wout[u]*h_W[u] starting from bout)   (out_act 0), or that chain
 *     clamped to [-1, 1] (out_act 1).
 */
static float fo_exp_nonpos(float y) { /* exp(y), y <= 0 (clamped at -60; a NaN acts like -60); Cephes expf's
                                       * reduction and polynomial */
    y = fmaxf(y, -60.0f);
    const float n = rintf(y * 1.44269504f);
    float r = fmaf(n, -0.693359375f, y);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    return ldexpf(p, (int)n);
}

float fo_lstm_sigmoid(float x) {
    const float e = fo_exp_nonpos(-fabsf(x));
    const float d = 1.0f + e;
    return x >= 0.0f ? 1.0f / d : e / d;
}

float fo_lstm_tanh(float x) {
    const float e = fo_exp_nonpos(-2.0f * fabsf(x));
    const float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

void fo_lstm_activations(const float *x, float *sig, float *tnh, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        sig[i] = fo_lstm_sigmoid(x[i]);
        tnh[i] = fo_lstm_tanh(x[i]);
    }
}

void fo_policy_lstm(const double *obs, const float *whh, const float *wx, const float *wout, float bout, int32_t H,
                    int32_t out_act, int64_t N, int32_t W, int32_t A, float *actions_out,
                    float *h_out /* (N, A, H) or NULL */) {
    float *h = (float *)calloc((size_t)H, sizeof(float)), *hn = (float *)calloc((size_t)H, sizeof(float));
    float *c = (float *)calloc((size_t)H, sizeof(float));
    for (int64_t n = 0; n < N; ++n)
        for (int a = 0; a < A; ++a) {
            const double *o = obs + (size_t)n * W * 5 * A + 5 * a;
            for (int t = 0; t < W; ++t) {
                const double *row = o + (size_t)t * 5 * A;
                const float x[4] = {(float)row[0], (float)row[1], (float)row[2], (float)row[3]};
                const float xh[4] = {(float)row[4], 1.0f, 0.0f, 0.0f};
                for (int u = 0; u < H; ++u) {
                    const int mt = u / 8, half = (u % 8) / 4, b = u % 4;
                    float gate[4];
                    for (int q = 0; q < 4; ++q) {
                        const size_t R = (size_t)32 * mt + 8 * b + 4 * half + q;
                        float acc = 0.0f;
                        for (int m = 0; m < 4; ++m) {
                            acc = fmaf(wx[R * 8 + m], x[m], acc);
                            acc = fmaf(wx[R * 8 + 4 + m], xh[m], acc);
                        }
                        if (t > 0)
                            for (int g = 0; g < H / 8; ++g)
                                for (int m = 0; m < 4; ++m) {
                                    acc = fmaf(whh[R * H + 8 * g + m], h[8 * g + m], acc);
                                    acc = fmaf(whh[R * H + 8 * g + 4 + m], h[8 * g + 4 + m], acc);
                                }
                        gate[q] = acc;
                    }
                    const float ig = fo_lstm_sigmoid(gate[0]), fg = fo_lstm_sigmoid(gate[1]);
                    const float gg = fo_lstm_tanh(gate[2]), og = fo_lstm_sigmoid(gate[3]);
                    const float t1 = fg * (t > 0 ? c[u] : 0.0f), t2 = ig * gg;
                    c[u] = t1 + t2;
                    hn[u] = og * fo_lstm_tanh(c[u]);
                }
                float *sw = h; h = hn; hn = sw;
            }
            float acc = bout;
            for (int u = 0; u < H; ++u) acc = fmaf(wout[u], h[u], acc);
            if (h_out) memcpy(h_out + ((size_t)n * A + a) * H, h, sizeof(float) * (size_t)H);
            if (out_act == 0) acc = fo_lstm_tanh(acc);
            else if (out_act == 1) acc = acc < -1.0f ? -1.0f : (acc > 1.0f ? 1.0f : acc);  /* 2: none (a critic's value) */
            actions_out[(size_t)n * A + a] = acc;
        }
    free(h); free(hn); free(c);
}
