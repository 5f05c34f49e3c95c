// fe_env.hip -- MI355X (gfx950) implementation of the TimeSeriesEnv hot path.
//
// What the reference does with ~660 eager ATen ops and four full-day
// advanced-index copies per step (finenvs/environments/time_series_env.py,
// "TSE", lines 277-536) is ONE kernel launch here:
//
//   phase 1  one lane per (env, asset) account ("sleeve"): action -> share
//            delta, six-stage trade, margin checks, reward, done      TSE:298-421,447-496
//   phase 1b one lane per env: OR of sleeve dones, liquidation fee, reward sum,
//            evaluate-mode bookkeeping, eval-env redraw                TSE:288-289,498-536
//   phase 2  the whole workgroup streams the tile's observations: each wavefront
//            loads 32-byte (O,H,L,C log-return) tuples of the (W, 4A) window from
//            the L2/MALL-resident table with coalesced 16-byte loads, lays them
//            out as 5-tuples (+ position feature) in a wave-private LDS image,
//            reads the image back linearly (ds_read_b128) and writes the
//            (W, 5A) observation as full 1-KiB-per-instruction stores  TSE:423-445
//
// A workgroup (256 threads = 4 wavefronts of 64) owns a TILE of EB consecutive
// envs; tiles are grid-strided.  The observation of a tile is one contiguous
// region of HBM, so phase 2 is a flat, perfectly coalesced store stream.
// Round-2 structure (measurements: DESIGN.md section 5, profiles/r02_microbench/):
// workgroup barriers order LDS only (no store drain), observation stores are
// write-through (sc1, single asset) or non-temporal (multi asset) so that state
// and tables stay in L2, a single-asset tile is a whole number of workgroup
// iterations with 4 (f64) / 6 (f32) workgroups per CU, and the first tile's table
// loads are issued before its accounting.
//
// Also here: K-step fused rollouts with an in-kernel policy (linear window /
// table form; two-layer MLP with the first layer on v_mfma_f32_32x32x2_f32),
// init kernels (log-returns, day tables), the trajectory kernels, and the C ABI.
//
// Arithmetic contract: every (float)/(double) cast is a rounding point of the
// reference's mixed f32/f64 tensor arithmetic; this file
// must be compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <type_traits>

#include "finenvs_amd.h"

namespace {

constexpr int kBlock = 256;
// Single-asset kernels: 7 waves per SIMD = 72 VGPRs, the most they reach without scratch spills
// (8 spills 12-36 B/lane and measured slower); the 20 KiB LDS stage allows 7 workgroups per CU too.
// Multi-asset kernels carry the per-sleeve LDS arrays (26 KiB at 8 envs x 30 assets -> 6 per CU),
// so they are built for 6 waves per SIMD (80 VGPRs, no spills).
#ifndef FE_MIN_WAVES_PER_EU
#define FE_MIN_WAVES_PER_EU 7
#endif
// Cache policy of the observation stores (raw buffer stores, aux bits: 1 = sc0, 2 = nt, 16 = sc1).
// -2 (default) = chosen per kernel variant; in both cases the point is that a 0.17-150 GB store stream must not
// evict what phase 1 and phase 2 re-read every step from the 4 MiB L2s:
//   * single-asset envs: sc1 (write-through, the line is dropped from L2).  The per-env state, the action and
//     the tables then stay L2-resident, which shortens the kernel's start-up chain (index load -> bar gather ->
//     accounting -> first store): measured at 64k envs (tools/ab_step.py, interleaved in one process, three
//     boxes) 34.8 -> 31.1, 34.5 -> 32.0, 34.8 -> 33.6 us per step; nt gains about 1 % less, sc0|sc1 the same.
//   * multi-asset envs: nt.  Measured at 1M envs x 30 assets (round 1, profiles/r01_microbench/store_policy.txt): FETCH_SIZE
//     6.9 GiB -> 0.4 GiB per launch and 26.4 -> 25.0 ms; sc1 gives the same fetch reduction but 25.6 ms.
// -1 = plain everywhere; -3 = the round-1 choice (plain for single-asset, nt for multi-asset); >= 0 = that aux
// everywhere (experiment builds).
// "" for the product library; experiment builds (finenvs_amd/csrc/build.py build_variant) carry their
// -D set here and are only ever loaded by explicit path
#ifndef FE_NO_DESC   /* experiment builds only: compile the step kernel's descriptor outputs out (A/B of their cost) */
#define FE_NO_DESC 0
#endif
#ifndef FE_BUILD_TAG
#define FE_BUILD_TAG ""
#endif
#ifndef FE_STORE_AUX
#define FE_STORE_AUX -2
#endif
// Structure of the single-asset step kernel (experiments; tools/ab_step.py):
//   0  software pipeline: per tile [account (wave 0) | barrier | stream (4 waves) | barrier], next tile's
//      loads prefetched under the stream
//   1  up-front accounting: wave w accounts the workgroup's w-th tile, all four at once; one barrier; then the
//      workgroup streams its tiles back to back with no further barriers
//   2  the north star's literal "one wavefront per env": lane 0 of a wavefront accounts one env, then the
//      wavefront streams that env's observation; no workgroup barriers at all (measured A/B for DESIGN.md)
#ifndef FE_STEP_VARIANT
#define FE_STEP_VARIANT 0
#endif
// Timing-only ablations of the step kernel (WRONG outputs; experiment builds only -- tools/ab_step.py):
//   bit 0  no phase 1: descriptors fabricated from the env number, no state / bar loads, no write-back
//   bit 1  no table loads in phase 2 (the image is built from constants)
//   bit 2  no LDS transpose in phase 2 (registers stored directly)
//   bit 3  phase 1 without its global stores (state write-back, reward, done)
//   bit 4  phase 1 without its global loads (constants instead)
//   bit 5  phase 1 without the accounting arithmetic
//   (bits 6 / 7 -- phase 1 skipped on the first tile only / on all but the first -- were used once and removed)
#ifndef FE_ABLATE
#define FE_ABLATE 0
#endif
// Diagnostic build: every workgroup of the single-asset step kernel writes four s_memrealtime stamps (100 MHz)
// -- start, first tile accounted, first tile streamed, end -- into the buffer bound as fe_env_bind_stats'
// eval_return argument (grid * 8 u64; the statistics themselves are off in this build).  tools/stamp_step.py.
#ifndef FE_STAMP
#define FE_STAMP 0
#endif
// 1 (default): the single-asset f64 step kernel issues the first tile's table loads before its accounting
// (0 = A/B arm).  Measured on a shared ring (profiles/r02_microbench/ab_hoist.txt): 30.57 -> 29.32 us at config 2.
// Not for f32 observations: they run 6 workgroups per CU (80 VGPRs) and the 16 extra live registers spill.
#ifndef FE_HOIST_FIRST
#define FE_HOIST_FIRST 1
#endif
#ifndef FE_F32_WAVES
#define FE_F32_WAVES 6
#endif
template <typename OT>
constexpr bool kHoistFirst = FE_HOIST_FIRST != 0 && (sizeof(OT) == 8 || FE_F32_WAVES <= 5);

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return FE_ERR_HIP;
}

// Exact unsigned 32-bit division by a launch-time constant (Granlund & Montgomery 1994).
struct FastDiv {
    uint32_t m, sh1, sh2, d;
};

FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct Params {
    const double *P;
    const double *LR;
    const float *LR32;  // optional f32 copy of LR for f32 observations (fe_env_bind_f32_table)
    int64_t *env_idx;
    int64_t *spot0;
    float *cash;
    float *lng;
    float *sht;
    double *margin;
    uint8_t *terminated;
    float *ep_ret;
    unsigned long long *counters;
    float *run_ret;      // optional episode statistics (fe_env_bind_stats): running return per env
    double *stat_acc;    // [0] finished training episodes, [1] sum of their returns, [2] sum of squares
    float *stat_eval;    // [0] return of the eval env's last finished episode, [1] how many it finished
    const float *actions;
    void *obs;
    double *rew;
    int32_t *done;
    int64_t *desc_src;   // optional (fe_env_step_traj): descriptors of the observation this step returns
    double *desc_pos;
    float *act_store;    // optional (fe_env_step_traj): the actions, copied into a trajectory slot
    int64_t N, D, L;
    int64_t num_tiles;
    int64_t eval_env;
    uint64_t seed;
    int32_t W, A, EB;
    int32_t evaluate, redraw_mode;
    uint32_t env_elems;  // W * 5 * A, observation elements per env
    FastDiv div_WA;  // by tuples per env (W * A)
    FastDiv div_A;
    float scale32, ms32, c32, imr32, S32;
    double comm, imr, one_mmr, S;
};

// ---- Philox4x32-10, the redraw generator of redraw_mode 1 ----
__device__ __forceinline__ uint32_t philox_u32(uint64_t seed, uint64_t counter) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0x46454e56u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

// max(x, 0) that lets a NaN through, as torch.relu does
__device__ __forceinline__ float relu32(float x) { return x > 0.0f ? x : (x != x ? x : 0.0f); }
__device__ __forceinline__ double relu64(double x) { return x > 0.0 ? x : (x != x ? x : 0.0); }

// Workgroup barrier that orders LDS traffic only.  Everything the waves of a workgroup hand to each other inside
// these kernels goes through LDS (descriptors, sleeve rewards / flags, actions); their global stores are
// fire-and-forget and nothing in the same launch reads them back.  __syncthreads() would also wait for every
// outstanding global store of the wave (s_waitcnt vmcnt(0)): in the step kernel that drains the observation
// store stream at every tile boundary and puts a store acknowledgement on the start-up chain.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// the step kernel's barriers (FE_SYNC=1: the round-1 form, for A/B)
#ifndef FE_SYNC
#define FE_SYNC 0
#endif
__device__ __forceinline__ void tile_barrier() {
#if FE_SYNC
    __syncthreads();
#else
    lds_barrier();
#endif
}

struct Sleeve {
    float cash, lng, sht;
    double margin;
    double pos_obs;
    double rew;
    bool bankrupt;
};

// One (env, asset) account for one bar: TSE:298-421 (trade), TSE:428-431
// (position feature), TSE:447-475 (reward).  Pure register arithmetic.
__device__ __forceinline__ void sleeve_step(const Params &p, float action, double O, double H, double Lo,
                                            double C, Sleeve &s) {
    float cash = s.cash, lng = s.lng, sht = s.sht;
    double margin = s.margin;
    float comm = 0.0f;  // TSE:305

    // TSE:298-302  round-half-even then clamp
    float sc = rintf(action * p.scale32);
    sc = sc < -p.ms32 ? -p.ms32 : sc;
    sc = sc > p.ms32 ? p.ms32 : sc;
    float pos = sc < 0.0f ? 0.0f : sc;  // TSE:344-351
    float neg = sc > 0.0f ? 0.0f : sc;

    // sell long positions first, TSE:353-361
    float nl = relu32(lng + neg);
    float sell = lng - nl;
    neg += sell;
    comm += sell * p.c32;
    cash = (float)((double)cash + (double)sell * (O - p.comm));
    lng = nl;

    // buy back shorts and re-mark the margin account, TSE:367-383
    float ns = relu32(sht - pos);
    float bb = sht - ns;
    pos -= bb;
    comm += bb * p.c32;
    cash = (float)((double)cash - (double)bb * (O + p.comm));
    sht = ns;
    double nm = (double)(p.imr32 * sht) * O;
    cash = (float)((double)cash - (nm - margin));
    margin = nm;

    // long entry unless unaffordable, TSE:385-399
    if ((double)cash - (double)pos * (O + p.comm) < 0.0) pos = 0.0f;
    comm += pos * p.c32;
    cash = (float)((double)cash - (double)pos * (O + p.comm));
    lng += pos;

    // short entry unless the 150% margin is unaffordable, TSE:401-421
    float q = -neg;
    if (((double)cash - p.imr * ((double)q * O)) - (double)(q * p.c32) < 0.0) {
        neg = 0.0f;
        q = -neg;
    }
    comm += q * p.c32;
    double req = p.imr * ((double)q * O);
    cash = (float)((double)cash - (req + (double)(q * p.c32)));
    margin += req;
    sht += q;

    // observation feature, rendered post-trade / pre-margin-check, TSE:428-431
    s.pos_obs = (double)(lng - sht) * C / p.S;

    // reward, TSE:447-475
    bool done = cash < 0.0f;
    double rew;
    {
        double call = relu64((double)sht * H * p.one_mmr - margin);
        cash = (float)((double)cash - call);
        margin += call;
        done |= cash < 0.0f;
        rew = -call;
    }
    {
        double rel = relu64(margin - (double)sht * Lo * p.imr);
        margin -= rel;
        cash = (float)((double)cash + rel);
    }
    {
        double call = relu64((double)sht * C * p.one_mmr - margin);
        cash = (float)((double)cash - call);
        margin += call;
        done |= cash < 0.0f;
        rew += -call;
    }
    if (done) {
        lng = 0.0f;
        sht = 0.0f;
    }
    rew += (double)(lng - sht) * (C - O);
    rew -= (double)comm;

    s.cash = cash; s.lng = lng; s.sht = sht; s.margin = margin;
    s.rew = rew;
    s.bankrupt = done;
}

template <typename OT, int VEC>
struct alignas(sizeof(OT) * VEC) Pack {
    OT v[VEC];
};

// Tuples one wavefront turns per phase-2 iteration: 5120 bytes of observation = five full
// 1-KiB store instructions (128 f64 tuples of 40 B, or 256 f32 tuples of 20 B).
constexpr int kStageBytes = 5120;

// LDS carve-up for a tile of EB envs x A assets (S = EB*A sleeves):
//   stage[4][5120 B] wave-private 5-tuple images (phase 2)
//   int64 src[EB]  element offset of the window's first row in the LR table
//   double pos[S]  position feature per sleeve
//   double rew[S]  sleeve reward before the liquidation fee   (A > 1 only)
//   float  shr[S]  long+short after the reward step            (A > 1 only)
//   int    flg[S]  sleeve done flag                            (A > 1 only)
//   int    any[EB] env-level done                              (A > 1 only)
__host__ __device__ inline size_t lds_bytes(int EB, int A) {
    size_t S = (size_t)EB * A;
    size_t b = 4 * (size_t)kStageBytes + (size_t)EB * 8 + S * 8;
    if (A > 1) b += S * 8 + S * 4 + S * 4 + (size_t)EB * 4;
#if FE_STEP_VARIANT == 1
    if (A == 1) b = 4 * (size_t)kStageBytes + 4 * ((size_t)EB * 16);  // descriptors of four tiles at once
#elif FE_STEP_VARIANT == 2
    if (A == 1 && b < 4 * (size_t)kStageBytes + 64) b = 4 * (size_t)kStageBytes + 64;  // one descriptor slot per wavefront
#endif
    return (b + 15) & ~(size_t)15;
}

struct TileLds {
    int64_t *src;  // [EB]  element offset of the observation window's first row in the LR table
    double *pos;   // [S]   position feature per sleeve
    double *rew;   // [S]   sleeve reward before the liquidation fee   (A > 1 only)
    float *shr;    // [S]   long+short after the reward step            (A > 1 only)
    int *flg;      // [S]   sleeve done flag                            (A > 1 only)
    int *any;      // [EB]  env-level done                              (A > 1 only)
};

__device__ __forceinline__ TileLds carve_lds(unsigned char *base, int EB, int S) {
    TileLds l;
    l.src = reinterpret_cast<int64_t *>(base);
    l.pos = reinterpret_cast<double *>(l.src + EB);
    l.rew = l.pos + S;
    l.shr = reinterpret_cast<float *>(l.rew + S);
    l.flg = reinterpret_cast<int *>(l.shr + S);
    l.any = l.flg + S;
    return l;
}

// What phase 1 reads for one sleeve.  Loading is split in two dependent stages so that the step
// kernel can prefetch them for the NEXT tile while the current tile's observation streams out:
//   head: env_idx, spot0 (coalesced)          body: state + the bar/probe gathers that need the head
struct SleeveIn {
    int64_t idx, s0, nxt;
    double4 bar;
    double probe, margin;
    float cash, lng, sht;
};

__device__ __forceinline__ void load_head(const Params &p, bool active, int64_t n, int64_t &idx, int64_t &spot) {
    idx = 0;
    spot = 0;
#if FE_ABLATE & 16
    idx = n % p.D;
    spot = 1;
    return;
#endif
    if (active) {
        idx = p.env_idx[n];
        spot = p.spot0[n];
    }
}

// the part of the body that needs no index: account state of the sleeve (issued together with the head for a
// workgroup's first tile, so that only the L2-resident bar gather sits behind the index load)
__device__ __forceinline__ void load_state(const Params &p, bool active, int64_t sl, SleeveIn &in) {
    if (!active) return;
#if FE_ABLATE & 16
    in.cash = 1e4f; in.lng = (float)(sl & 3); in.sht = 0.0f; in.margin = 0.0;
    return;
#endif
    in.cash = p.cash[sl];
    in.lng = p.lng[sl];
    in.sht = p.sht[sl];
    in.margin = p.margin[sl];
}

// the part that does: the bar at the window's last row and the NaN probe of the next row
__device__ __forceinline__ void load_bar(const Params &p, int A, int a, bool active, int64_t idx, int64_t spot,
                                         SleeveIn &in) {
    if (!active) return;
#if FE_ABLATE & 16
    in.idx = idx; in.s0 = spot + 1; in.nxt = spot + p.W + 1; in.bar = make_double4(100.0, 101.0, 99.0, 100.5);
    in.probe = 0.0;
    return;
#endif
    const int64_t rs = 4 * (int64_t)A;
    const int64_t L = p.L;
    in.idx = idx;
    in.s0 = spot + 1;  // TSE:281-282
    int64_t last = in.s0 + p.W - 1;
    last = last < L ? last : L - 1;  // memory safety only; the done logic keeps last < L
    in.nxt = last + 1;               // TSE:480
    in.bar = *reinterpret_cast<const double4 *>(p.P + (idx * L + last) * rs + 4 * a);
    in.probe = 0.0;
    if (in.nxt < L) in.probe = p.LR[(idx * L + in.nxt) * rs + 4 * a];
}

__device__ __forceinline__ void load_body(const Params &p, int A, int a, bool active, int64_t sl, int64_t idx,
                                          int64_t spot, SleeveIn &in) {
    load_bar(p, A, a, active, idx, spot, in);
    load_state(p, active, sl, in);
}

// Phases 1 and 1b for one tile from preloaded inputs: every thread of the workgroup must call it
// (it holds barriers).  On return l.src / l.pos describe the observation of this step (terminal
// window on done steps, exactly what step() returns, TSE:321) and the state arrays hold the
// post-step (post-reset) state.
template <bool SINGLE>
__device__ __forceinline__ void account_core(const Params &p, const TileLds &l, int A, int e, int a, bool active,
                                             int64_t n, int64_t sl, const SleeveIn &in, float action,
                                             double *rew_out, int32_t *done_out) {
    const int64_t rs = 4 * (int64_t)A;
    const int W = p.W;
    const int64_t L = p.L;
    Sleeve s;
    int64_t s0 = 0;
    bool sdone = false;
    // ---------------- phase 1: one lane per sleeve ----------------
    if (active) {
        s0 = in.s0;
        s.cash = in.cash;
        s.lng = in.lng;
        s.sht = in.sht;
        s.margin = in.margin;
#if FE_ABLATE & 32
        s.pos_obs = in.bar.w + (double)action; s.rew = in.bar.x; s.bankrupt = false;
#else
        sleeve_step(p, action, in.bar.x, in.bar.y, in.bar.z, in.bar.w, s);
#endif
        // termination: bankrupt | end of buffer | next open log-return is NaN, TSE:477-496
        sdone = s.bankrupt | (in.nxt >= L) | (in.probe != in.probe);
        l.pos[e * A + a] = s.pos_obs;
        if constexpr (!SINGLE) {
            l.rew[e * A + a] = s.rew;
            l.shr[e * A + a] = s.sht + s.lng;  // num_shares, TSE:288
            l.flg[e * A + a] = sdone ? 1 : 0;
        }
        if (a == 0) {
            // memory safety only: a well-formed state always has s0 + W <= L
            const int64_t s0c = s0 + W <= L ? s0 : L - W;
            l.src[e] = (in.idx * L + s0c) * rs;
        }
#if !FE_NO_DESC
        if (p.desc_src) {  // the returned observation as descriptors, 8 + 8A bytes per env (a trajectory's `states`)
            p.desc_pos[sl] = s.pos_obs;
            if (a == 0) p.desc_src[n] = l.src[e];
        }
        if (p.act_store) p.act_store[sl] = action;  // agent.store's `actions` field, no copy kernel
#endif
    }
    // ---------------- phase 1b: one lane per env ----------------
    bool any = sdone;
    if constexpr (!SINGLE) tile_barrier();
    if (active && a == 0) {
        double rew;
        if constexpr (SINGLE) {
            float fee = ((any ? 1.0f : 0.0f) * (s.sht + s.lng)) * p.c32;  // TSE:288-289
            rew = s.rew - (double)fee;
        } else {
            any = false;
            for (int k = 0; k < A; ++k) any |= l.flg[e * A + k] != 0;
            rew = 0.0;
            for (int k = 0; k < A; ++k) {  // sleeve contract: sum in asset order
                float fee = ((any ? 1.0f : 0.0f) * l.shr[e * A + k]) * p.c32;
                double r = l.rew[e * A + k] - (double)fee;
                rew = (k == 0) ? r : rew + r;
            }
            l.any[e] = any ? 1 : 0;
        }
        if (any) {
            s0 = 0;  // window rewinds to rows 0..W-1, TSE:514-521
            if (!p.evaluate && p.redraw_mode == 1 && n == p.eval_env) {  // TSE:504-513
                unsigned long long ctr = p.counters[1];
                p.env_idx[n] = (int64_t)(((uint64_t)philox_u32(p.seed, ctr) * (uint64_t)p.D) >> 32);
                p.counters[1] = ctr + 1;
            }
        }
#if !(FE_ABLATE & 8)
        p.spot0[n] = s0;
#endif
        if (p.evaluate) {  // TSE:523-536
            const bool term = p.terminated[n] != 0;
            if (term) rew = 0.0;
            if (any && !term) {
                p.terminated[n] = 1;
                atomicAdd(&p.counters[0], 1ull);
            }
            p.ep_ret[n] = (float)((double)p.ep_ret[n] + rew);
        }
#if FE_ABLATE & 8
        if (rew == 123.456) done_out[n] = 7;  // keeps the arithmetic alive
#else
        rew_out[n] = rew;
        done_out[n] = any ? 1 : 0;
#endif
        if (p.run_ret) {  // PPO_agent.py:120-132 without its per-step host sync
            float cr = (float)((double)p.run_ret[n] + rew);
            if (any) {
                if (n == p.eval_env) {
                    p.stat_eval[0] = cr;
                    p.stat_eval[1] += 1.0f;
                } else {
                    atomicAdd(&p.stat_acc[0], 1.0);
                    atomicAdd(&p.stat_acc[1], (double)cr);
                    atomicAdd(&p.stat_acc[2], (double)cr * (double)cr);
                }
                cr = 0.0f;
            }
            p.run_ret[n] = cr;
        }
    }
    if constexpr (!SINGLE) {
        tile_barrier();
        if (active) any = l.any[e] != 0;
    }
#if FE_ABLATE & 8
    if (active && s.cash == 123.456f && s.margin == 7.0) p.cash[sl] = s.lng + s.sht;
    if (false)
#endif
    if (active) {  // state write-back with the episodic reset folded in, TSE:498-502
        p.cash[sl] = any ? p.S32 : s.cash;
        p.lng[sl] = any ? 0.0f : s.lng;
        p.sht[sl] = any ? 0.0f : s.sht;
        p.margin[sl] = any ? 0.0 : s.margin;
    }
}

// unpipelined form: load, then account (the fused rollout kernel revisits the same tile every step)
template <bool SINGLE>
__device__ __forceinline__ void account_tile(const Params &p, const TileLds &l, int A, int e, int a, bool active,
                                             int64_t n, int64_t sl, float action, double *rew_out,
                                             int32_t *done_out) {
    int64_t idx, spot;
    SleeveIn in;
    load_head(p, active, n, idx, spot);
    load_body(p, A, a, active, sl, idx, spot, in);
    account_core<SINGLE>(p, l, A, e, a, active, n, sl, in, action, rew_out, done_out);
}

// reset(): the observation descriptors of the CURRENT state (TSE:423-435); changes no state.
__device__ __forceinline__ void describe_tile(const Params &p, const TileLds &l, int A, int e, int a, bool active,
                                              int64_t n, int64_t sl) {
    if (!active) return;
    const int64_t rs = 4 * (int64_t)A;
    const int64_t idx = p.env_idx[n];
    const int64_t s0 = p.spot0[n];
    int64_t last = s0 + p.W - 1;
    last = last < p.L ? last : p.L - 1;
    const double C = p.P[(idx * p.L + last) * rs + 4 * a + 3];
    l.pos[e * A + a] = (double)(p.lng[sl] - p.sht[sl]) * C / p.S;
    if (a == 0) {
        const int64_t s0c = s0 + p.W <= p.L ? s0 : p.L - p.W;
        l.src[e] = (idx * p.L + s0c) * rs;
    }
}

template <typename OT>
using TupleOf = typename std::conditional<sizeof(OT) == 4, float4, double4>::type;
template <typename OT>
constexpr int kTuplesPerIter = kStageBytes / (5 * (int)sizeof(OT));  // tuples one wavefront turns per iteration

// Table tuples of one phase-2 iteration held across other work: named members, passed by value -- an array that is
// selected against a freshly loaded one ends up in scratch memory behind flat loads (measured: 31 -> 42 us).
template <typename OT>
struct PreTuples {
    TupleOf<OT> v0, v1, v2, v3;  // G = 2 (f64) uses v0, v1; G = 4 (f32) all four
};

// The table loads of one phase-2 iteration of one wavefront (they need l.src only, not the position feature).
template <typename OT, bool SINGLE>
__device__ __forceinline__ void stream_load(const Params &p, const TileLds &l, int A, uint32_t tuples, uint32_t base,
                                            int lane, TupleOf<OT> (&v)[kTuplesPerIter<OT> / 64], bool skip = false) {
    constexpr int G = kTuplesPerIter<OT> / 64;
    const uint32_t WA = (uint32_t)p.W * (uint32_t)A;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) v[gi] = TupleOf<OT>{};
    if (skip) return;
    // f32 observations read a pre-cast f32 copy of the table when one is bound: half the L2 traffic,
    // same values ((float) of the f64 entry either way)
    const bool narrow = sizeof(OT) == 4 && p.LR32 != nullptr;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const uint32_t t = base + gi * 64 + lane;
        const uint32_t tc = t < tuples ? t : tuples - 1;  // tail lanes re-read the last tuple
        const uint32_t ee = fdiv(tc, p.div_WA);           // env within the tile
        const uint32_t r = tc - ee * WA;                  // (row j, asset a) = r / A, r % A
#if FE_ABLATE & 2
        v[gi].x = (decltype(v[gi].x))tc; v[gi].y = v[gi].x; v[gi].z = v[gi].x; v[gi].w = v[gi].x;
        if (false)
#endif
        if constexpr (sizeof(OT) == 4) {
            if (narrow) {
                v[gi] = *reinterpret_cast<const float4 *>(p.LR32 + l.src[ee] + 4u * r);
            } else {
                const double4 d = *reinterpret_cast<const double4 *>(p.LR + l.src[ee] + 4u * r);
                v[gi] = make_float4((float)d.x, (float)d.y, (float)d.z, (float)d.w);
            }
        } else {
            v[gi] = *reinterpret_cast<const double4 *>(p.LR + l.src[ee] + 4u * r);
        }
    }
}

template <typename OT, int VEC, bool SINGLE>
__device__ __forceinline__ void stream_tile(const Params &p, const TileLds &l, OT *stage, int A, int ebt, OT *dst,
                                            int lane, int wave, int nwaves, bool use_pre, PreTuples<OT> pre);
template <typename OT, int VEC, bool SINGLE>
__device__ __forceinline__ void stream_tile(const Params &p, const TileLds &l, OT *stage, int A, int ebt, OT *dst,
                                            int lane, int wave, int nwaves = kBlock / 64) {
    stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt, dst, lane, wave, nwaves, false, PreTuples<OT>{});
}

// Phase 2 for one tile: l.src / l.pos -> (ebt, W, 5A) observation at dst, through this wavefront's
// private LDS image.  No workgroup barrier inside.  `pre` (optional): the table tuples of this wavefront's FIRST
// iteration, loaded earlier by stream_load (the start-up chain of a workgroup's first tile).
template <typename OT, int VEC, bool SINGLE>
__device__ __forceinline__ void stream_tile(const Params &p, const TileLds &l, OT *stage, int A, int ebt, OT *dst,
                                            int lane, int wave, int nwaves, bool use_pre, PreTuples<OT> pre) {
    constexpr int TPI = kTuplesPerIter<OT>;  // tuples per wave iteration
    constexpr int G = TPI / 64;              // tuples per lane per iteration
    const uint32_t WA = (uint32_t)p.W * (uint32_t)A;           // 32-byte table tuples per env
    const uint32_t tuples = (uint32_t)ebt * WA;
    for (uint32_t base = wave * TPI; base < tuples; base += nwaves * TPI) {
        using TupleT = TupleOf<OT>;
        TupleT v[G];
        double pz[G];
        stream_load<OT, SINGLE>(p, l, A, tuples, base, lane, v, /*skip=*/use_pre && base == (uint32_t)wave * TPI);
        if (use_pre && base == (uint32_t)wave * TPI) {
            v[0] = pre.v0;
            v[1] = pre.v1;
            if constexpr (G == 4) {
                v[2] = pre.v2;
                v[3] = pre.v3;
            }
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const uint32_t t = base + gi * 64 + lane;
            const uint32_t tc = t < tuples ? t : tuples - 1;
            const uint32_t ee = fdiv(tc, p.div_WA);
            const uint32_t r = tc - ee * WA;
            const uint32_t aa = SINGLE ? 0u : r - fdiv(r, p.div_A) * (uint32_t)A;
            pz[gi] = l.pos[ee * A + aa];
        }
#if FE_ABLATE & 4
        {
            const uint32_t left_ = tuples - base;
            const uint32_t nvalid_ = (left_ < (uint32_t)TPI ? left_ : (uint32_t)TPI) * 5u / VEC;
            Pack<OT, VEC> *o_ = reinterpret_cast<Pack<OT, VEC> *>(dst + (size_t)base * 5u);
#pragma unroll
            for (int i = 0; i < TPI * 5 / VEC / 64; ++i) {
                const uint32_t c = (uint32_t)lane + 64u * i;
                Pack<OT, VEC> pk;
                for (int q = 0; q < VEC; ++q) pk.v[q] = (OT)(q & 1 ? v[i % G].y : v[i % G].x) + (OT)pz[i % G];
                if (c < nvalid_) o_[c] = pk;
            }
            continue;
        }
#endif
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            OT *w = stage + (gi * 64 + lane) * 5;
            w[0] = (OT)v[gi].x; w[1] = (OT)v[gi].y; w[2] = (OT)v[gi].z; w[3] = (OT)v[gi].w;
            w[4] = (OT)pz[gi];
        }
        // the image is private to this wavefront: order its LDS writes before the reads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t left = tuples - base;
        const uint32_t nvalid = (left < (uint32_t)TPI ? left : (uint32_t)TPI) * 5u / VEC;  // packs to store
        const Pack<OT, VEC> *rd = reinterpret_cast<const Pack<OT, VEC> *>(stage);
        Pack<OT, VEC> *o = reinterpret_cast<Pack<OT, VEC> *>(dst + (size_t)base * 5u);
        constexpr int kStores = TPI * 5 / VEC / 64;  // 5 full-width store instructions at 16 B/lane
        constexpr int kAux = FE_STORE_AUX == -2 ? (SINGLE ? 16 : 2) : (FE_STORE_AUX == -3 ? (SINGLE ? -1 : 2) : FE_STORE_AUX);
        if constexpr (kAux >= 0 && sizeof(OT) * VEC == 16) {
            // observation stores with explicit cache bits (aux: 1 = sc0, 2 = nt, 16 = sc1) through a
            // buffer descriptor over this wavefront's 5-KiB slab; the descriptor is wave-uniform
            using u4 = __attribute__((ext_vector_type(4))) unsigned int;
            const uint64_t basep = reinterpret_cast<uint64_t>(o);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)basep);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(basep >> 32));
            const unsigned nb = __builtin_amdgcn_readfirstlane(nvalid * 16u);
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, nb,
                                                          0x00020000);
            const u4 *rd4 = reinterpret_cast<const u4 *>(stage);
#pragma unroll
            for (int i = 0; i < kStores; ++i) {
                const uint32_t c = (uint32_t)lane + 64u * i;
                if (c < nvalid) __builtin_amdgcn_raw_buffer_store_b128(rd4[c], rsrc, c * 16u, 0, kAux);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kStores; ++i) {
                const uint32_t c = (uint32_t)lane + 64u * i;
                if (c < nvalid) o[c] = rd[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // reads done before the next iteration overwrites the image
    }
}

// Software pipeline state of the single-asset step kernel: inputs of the current tile, prefetched inputs of the
// next one, indices of the one after.
struct PipeState {
    SleeveIn in_cur, in_nxt;
    float action_cur, action_nxt, action_nn;  // actions run TWO tiles ahead: a caller's action buffer may be cold (a new
                                              // trajectory slot every step costs a TLB walk + an HBM round trip, ~5 us)
    int64_t n_cur, n_nxt, n_nn, idx1, spot1, idx2, spot2;
    bool act0, act1, act2;
    unsigned long long t_accounted;  // FE_STAMP builds
};

__device__ __forceinline__ int64_t pipe_env_of(const Params &p, int EB, int e, int64_t t, bool &act) {
    const int64_t n0 = t * EB;
    const int64_t left = p.N - n0;
    act = t < p.num_tiles && (int64_t)e < (left < (int64_t)EB ? left : (int64_t)EB);
    return n0 + e;
}

// Actions are fetched two tiles ahead with f64 observations (measured on a shared ring, tools/ab_step.py: -2.6 % with hot
// action buffers, 41.1 -> 37.8 us/step with cold ones, tools/cold_slots.py); with f32 observations the extra live
// register spills at the 6 workgroups per CU that shape wants (+2.7 %), so f32 keeps one tile ahead.
template <typename OT>
constexpr bool kActionsTwoAhead = sizeof(OT) == 8;

// One tile of the single-asset pipeline: account it (inputs already in registers), prefetch the next tile's body and
// the head of the one after, stream its observation.  FIRST: the workgroup's first tile, whose first phase-2
// iteration may use table tuples loaded before the accounting (`pre`).
template <typename OT, int VEC, bool FIRST>
__device__ __forceinline__ void single_tile(const Params &p, const TileLds &l, OT *stage, PipeState &ps, int64_t tile,
                                            int64_t G, int EB, int e, int lane, int wave, PreTuples<OT> pre) {
    const int64_t n0 = tile * EB;
    const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
    account_core<true>(p, l, 1, e, 0, ps.act0, ps.n_cur, ps.n_cur, ps.in_cur, ps.action_cur, p.rew, p.done);
    tile_barrier();
#if FE_STAMP
    if (FIRST) ps.t_accounted = __builtin_amdgcn_s_memrealtime();
#endif
    // prefetch: body of the next tile (its head arrived during the previous phase 2), head of the one after
    load_body(p, 1, 0, ps.act1, ps.n_nxt, ps.idx1, ps.spot1, ps.in_nxt);
    if constexpr (!kActionsTwoAhead<OT>)
        if (ps.act1) ps.action_nxt = p.actions[ps.n_nxt];
    ps.n_nn = pipe_env_of(p, EB, e, tile + 2 * G, ps.act2);
    load_head(p, ps.act2, ps.n_nn, ps.idx2, ps.spot2);
    if constexpr (kActionsTwoAhead<OT>)
        if (ps.act2) ps.action_nn = p.actions[ps.n_nn];
    stream_tile<OT, VEC, true>(p, l, stage, 1, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave,
                               kBlock / 64, FIRST && kHoistFirst<OT>, pre);
    tile_barrier();  // LDS is reused by the next tile
    ps.in_cur = ps.in_nxt;
    ps.action_cur = ps.action_nxt;
    if constexpr (kActionsTwoAhead<OT>) ps.action_nxt = ps.action_nn;
    ps.n_cur = ps.n_nxt; ps.act0 = ps.act1;
    ps.n_nxt = ps.n_nn; ps.act1 = ps.act2;
    ps.idx1 = ps.idx2; ps.spot1 = ps.spot2;
}

// Wavefronts per SIMD the kernels are built for = workgroups per CU they are launched with (configure_launch): the
// single-asset step kernel runs 4 per CU with f64 observations (128 VGPRs: room for the hoisted first-iteration
// tuples) and 6 with f32 (80 VGPRs: no more spills -- at 7 / 72 VGPRs it spilled 28 bytes per lane); reset / render
// and the multi-asset kernels keep round 1's 7 and 6.
template <typename OT, bool SINGLE, bool RESET_ONLY>
constexpr int kEnvKernelWaves = !SINGLE ? FE_MIN_WAVES_PER_EU - 1
                                : (RESET_ONLY ? FE_MIN_WAVES_PER_EU : (sizeof(OT) == 8 ? (kHoistFirst<OT> ? 4 : FE_MIN_WAVES_PER_EU) : FE_F32_WAVES));
template <typename OT, int VEC, bool SINGLE, bool RESET_ONLY>
__global__ __launch_bounds__(kBlock, (kEnvKernelWaves<OT, SINGLE, RESET_ONLY>)) void fe_env_kernel(const Params p) {
    extern __shared__ __align__(16) unsigned char smem[];
#if FE_STAMP
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();  // before any kernel argument is needed
    __builtin_amdgcn_sched_barrier(0);
#endif
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const TileLds l = carve_lds(smem + 4 * kStageBytes, EB, EB * A);
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    OT *stage = reinterpret_cast<OT *>(smem + wave * kStageBytes);

    if constexpr (RESET_ONLY) {
        for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
            const int64_t n0 = tile * EB;
            const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
            const int64_t n = n0 + e;
            describe_tile(p, l, A, e, a, e < ebt, n, n * A + a);
            tile_barrier();
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt,
                                         reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            tile_barrier();  // LDS is reused by the next tile
        }
    } else if constexpr (!SINGLE) {
        // multi-asset tiles stream hundreds of KiB each: phase 1 is <1 % of a tile, no pipelining needed
        for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
            const int64_t n0 = tile * EB;
            const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
            const bool active = e < ebt;
            const int64_t n = n0 + e;
            const int64_t sl = n * A + a;
            account_tile<SINGLE>(p, l, A, e, a, active, n, sl, active ? p.actions[sl] : 0.0f, p.rew, p.done);
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt,
                                         reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            tile_barrier();  // LDS is reused by the next tile
        }
#if FE_ABLATE & 1
    } else if constexpr (SINGLE) {
        for (int64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
            const int64_t n0 = tile * EB;
            const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
            if (e < ebt) {
                l.src[e] = (((n0 + e) % p.D) * p.L + 1) * 4;
                l.pos[e] = (double)e;
            }
            tile_barrier();
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            tile_barrier();
        }
#elif FE_STEP_VARIANT == 2
    } else if constexpr (SINGLE) {
        // one wavefront per env: env = global wavefront index, grid-strided
        int64_t *w_src = reinterpret_cast<int64_t *>(smem + 4 * kStageBytes) + wave;  // wave-private descriptor slot
        double *w_pos = reinterpret_cast<double *>(smem + 4 * kStageBytes + 4 * 8) + wave;
        TileLds lw;
        lw.src = w_src; lw.pos = w_pos; lw.rew = nullptr; lw.shr = nullptr; lw.flg = nullptr; lw.any = nullptr;
        const int64_t nw = (int64_t)gridDim.x * (kBlock / 64);
        for (int64_t n = (int64_t)blockIdx.x * (kBlock / 64) + wave; n < p.N; n += nw) {
            const bool act = lane == 0;
            account_tile<true>(p, lw, 1, 0, 0, act, n, n, act ? p.actions[n] : 0.0f, p.rew, p.done);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            stream_tile<OT, VEC, SINGLE>(p, lw, stage, 1, 1, reinterpret_cast<OT *>(p.obs) + n * (int64_t)p.env_elems, lane, 0, 1);
        }
#elif FE_STEP_VARIANT == 1
    } else if constexpr (SINGLE) {
        // Up-front accounting.  A round = up to four of this workgroup's tiles: wave w runs phase 1 for tile w of
        // the round on its first EB lanes (EB <= 64), all four waves at once -- one latency chain (index load ->
        // bar gather -> accounting) per ROUND instead of per tile --, then one barrier, then the tiles stream out
        // back to back.  Most shapes need a single round.
        const int64_t G = gridDim.x;
        int64_t *s_src = reinterpret_cast<int64_t *>(smem + 4 * kStageBytes);  // [4][EB]
        double *s_pos = reinterpret_cast<double *>(s_src + 4 * EB);            // [4][EB]
        for (int64_t t0 = blockIdx.x; t0 < p.num_tiles; t0 += 4 * G) {
            {
                const int64_t tile = t0 + (int64_t)wave * G;
                const int64_t n = tile * EB + lane;
                const bool act = tile < p.num_tiles && lane < EB && n < p.N;
                TileLds lw;
                lw.src = s_src + wave * EB;
                lw.pos = s_pos + wave * EB;
                lw.rew = nullptr; lw.shr = nullptr; lw.flg = nullptr; lw.any = nullptr;
                account_tile<true>(p, lw, 1, lane, 0, act, n, n, act ? p.actions[n] : 0.0f, p.rew, p.done);
            }
            tile_barrier();
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int64_t tile = t0 + (int64_t)j * G;
                if (tile >= p.num_tiles) break;
                const int64_t n0 = tile * EB;
                const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
                TileLds lj;
                lj.src = s_src + j * EB;
                lj.pos = s_pos + j * EB;
                lj.rew = nullptr; lj.shr = nullptr; lj.flg = nullptr; lj.any = nullptr;
                stream_tile<OT, VEC, SINGLE>(p, lj, stage, 1, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems,
                                             lane, wave);
            }
            tile_barrier();  // descriptors are reused by the next round
        }
#endif
    } else {
        // Software pipeline over this workgroup's tiles: while tile i streams its observation
        // (phase 2, the long part), the state + bar gathers of tile i+1 and the index loads of tile
        // i+2 are already in flight, so only the very first tile pays phase 1's two dependent
        // memory round trips.
        const int64_t G = gridDim.x;
        int64_t tile = blockIdx.x;
#if FE_STAMP
        unsigned long long *stamps = reinterpret_cast<unsigned long long *>(p.stat_eval);
        if (stamps && tid == 0) {
            stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
            stamps[blockIdx.x * 8 + 6] = t_entry;
        }
#endif
        PipeState ps;
        ps.action_cur = 0.0f; ps.action_nxt = 0.0f; ps.action_nn = 0.0f;
        ps.n_cur = pipe_env_of(p, EB, e, tile, ps.act0);
        ps.n_nxt = pipe_env_of(p, EB, e, tile + G, ps.act1);
        // first tile: everything that needs no index goes out with the index loads (one round trip), only the
        // bar gather (an L2 hit) waits for them
        load_head(p, ps.act0, ps.n_cur, ps.idx1, ps.spot1);
        load_state(p, ps.act0, ps.n_cur, ps.in_cur);
        if (ps.act0) ps.action_cur = p.actions[ps.n_cur];
        if constexpr (kActionsTwoAhead<OT>)
            if (ps.act1) ps.action_nxt = p.actions[ps.n_nxt];  // the second tile's action leaves with the first one's
#if FE_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamps && tid == 0) stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif
        load_bar(p, 1, 0, ps.act0, ps.idx1, ps.spot1, ps.in_cur);
        PreTuples<OT> pre{};  // table tuples of this wavefront's first phase-2 iteration (FE_HOIST_FIRST)
        // Start-up chain of the first tile: the window descriptors need the index loads only, so they are published
        // now and every wavefront issues the table loads of its first phase-2 iteration BEFORE the accounting --
        // one L2/MALL round trip less between kernel entry and the first observation store.
        if constexpr (kHoistFirst<OT>) {
            if (ps.act0) {
                const int64_t s0 = ps.spot1 + 1;
                const int64_t s0c = s0 + p.W <= p.L ? s0 : p.L - p.W;
                l.src[e] = (ps.idx1 * p.L + s0c) * 4;  // A == 1
            }
            tile_barrier();
            const int64_t left0 = p.N - tile * EB;
            const uint32_t tuples0 = (uint32_t)(left0 < (int64_t)EB ? left0 : (int64_t)EB) * (uint32_t)p.W;
            if ((uint32_t)wave * kTuplesPerIter<OT> < tuples0 && tile < p.num_tiles) {
                TupleOf<OT> v0[kTuplesPerIter<OT> / 64];
                stream_load<OT, true>(p, l, 1, tuples0, (uint32_t)wave * kTuplesPerIter<OT>, lane, v0);
                pre.v0 = v0[0];
                pre.v1 = v0[1];
                if constexpr (kTuplesPerIter<OT> / 64 == 4) {
                    pre.v2 = v0[2];
                    pre.v3 = v0[3];
                }
            }
        }
#if FE_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamps && tid == 0) stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
        load_head(p, ps.act1, ps.n_nxt, ps.idx1, ps.spot1);
        // one tile per call; the workgroup's first tile is peeled (FIRST) so that `pre` dies before the loop
        if (tile < p.num_tiles) {
            single_tile<OT, VEC, true>(p, l, stage, ps, tile, G, EB, e, lane, wave, pre);
#if FE_STAMP
            if (stamps && tid == 0) {
                stamps[blockIdx.x * 8 + 1] = ps.t_accounted;
                stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            }
#endif
            for (tile += G; tile < p.num_tiles; tile += G)
                single_tile<OT, VEC, false>(p, l, stage, ps, tile, G, EB, e, lane, wave, PreTuples<OT>{});
        }
#if FE_STAMP
        if (stamps && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);  // this wavefront's stores have left
            stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
}

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
        sleeve_step(p, action, bar.x, bar.y, bar.z, bar.w, s);
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
                    atomicAdd(&p.stat_acc[0], 1.0);
                    atomicAdd(&p.stat_acc[1], (double)cr);
                    atomicAdd(&p.stat_acc[2], (double)cr * (double)cr);
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

#ifndef FE_ROLLOUT_WAVES
#define FE_ROLLOUT_WAVES 1
#endif
template <bool SINGLE>
__global__ __launch_bounds__(kBlock, FE_ROLLOUT_WAVES) void fe_rollout_linear_kernel(const Params p, const RolloutArgs r) {
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
        // the tile's account state moves into registers for the whole K-step loop
        SleeveReg st;
        st.idx = 0; st.spot = 0; st.cash = 0.0f; st.lng = 0.0f; st.sht = 0.0f; st.margin = 0.0;
        if (active) {
            st.idx = p.env_idx[n];
            st.spot = p.spot0[n];
            st.cash = p.cash[sl];
            st.lng = p.lng[sl];
            st.sht = p.sht[sl];
            st.margin = p.margin[sl];
            if (a == 0) l.src[e] = r.obs_src[n];
            l.pos[e * A + a] = r.obs_pos[sl];
        }
        __syncthreads();
        const int pairs = ebt * A;
        for (int k = 0; k < r.K; ++k) {
            // policy: one wavefront per (env, asset) pair of the tile
#ifdef FE_ROLLOUT_NOPOLICY  /* diagnostic build: how long is a step without the policy? */
            for (int q = tid; q < pairs; q += kBlock) s_act[q] = (float)r.bias;
            if (false)
#endif
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
        if (active) {  // state and descriptors go back to HBM once per launch
            p.cash[sl] = st.cash;
            p.lng[sl] = st.lng;
            p.sht[sl] = st.sht;
            p.margin[sl] = st.margin;
            r.obs_pos[sl] = l.pos[e * A + a];
            if (a == 0) {
                p.env_idx[n] = st.idx;
                p.spot0[n] = st.spot;
                r.obs_src[n] = l.src[e];
            }
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
        if (active) {
            p.cash[sl] = st.cash;
            p.lng[sl] = st.lng;
            p.sht[sl] = st.sht;
            p.margin[sl] = st.margin;
            r.obs_pos[sl] = st.obs_pos;
            if (a == 0) {
                p.env_idx[n] = st.idx;
                p.spot0[n] = st.spot;
                r.obs_src[n] = st.obs_row * 4 * (int64_t)A;
            }
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
    if constexpr (ACT == 2) return tanhf(z);
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
        if (active) {  // state and descriptors go back to HBM once per launch
            p.cash[sl] = st.cash;
            p.lng[sl] = st.lng;
            p.sht[sl] = st.sht;
            p.margin[sl] = st.margin;
            r.obs_pos[sl] = l.pos[e * A + a];
            if (a == 0) {
                p.env_idx[n] = st.idx;
                p.spot0[n] = st.spot;
                r.obs_src[n] = l.src[e];
            }
        }
        __syncthreads();
    }
}

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

struct LstmArgs {
    const float *lr32;  // (D, L, 4A) f32 copy of the log-return table
    const float *whh;   // (4H, H) f32, packed row order
    const float *wx;    // (4H, 8) f32, packed row order: w_ih[0..3], w_ih[4], b_ih + b_hh, 0, 0
    const float *wout;  // (H)
    float bout;
    int32_t H, out_act, K;  // out_act: 0 tanh (the reference's actor), 1 clamp to [-1, 1]
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

// ---- the LSTM head's sigmoid / tanh: exactly-rounded operations only, two activations per instruction ----
//   e = exp(-s |x|) (s = 1 sigmoid, 2 tanh; argument clamped at -60), Cephes expf's reduction and polynomial;
//   sigmoid = (x >= 0 ? 1 : e) / (1 + e),   tanh = copysign((1 - e) / (1 + e), x).
// Every step is an IEEE-exact f32 operation (mul, fma, rint, ldexp, and a division), so the same sequence on the CPU
// (fo_lstm_sigmoid / fo_lstm_tanh of the tests' restatement) gives the same bits.  The f32 MFMA shares the vector
// ALUs with these (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so their instruction count is kernel time: the chains run on
// pairs of activations with packed-f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), and the division
// is the correctly-rounded rcp + fma sequence the compiler itself emits for `/`, minus its v_div_scale / v_div_fixup
// range handling -- the denominator is in [1, 2] and the numerator in {0} U [2^-87, 1], where that handling is the
// identity (this is why the argument clamp is -60: a smaller numerator would need the scaling).
// NaN is not propagated (a NaN pre-activation acts like -60); the host refuses non-finite weights.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float v) { return (v2f){v, v}; }

__device__ __forceinline__ v2f lstm_exp_nonpos2(v2f y0) {
    v2f y = {fmaxf(y0.x, -60.0f), fmaxf(y0.y, -60.0f)};
    v2f n = y * pk_splat(1.44269504f);
    n = (v2f){rintf(n.x), rintf(n.y)};
    v2f r = pk_fma(n, pk_splat(-0.693359375f), y);
    r = pk_fma(n, pk_splat(2.12194440e-4f), r);
    v2f q = pk_splat(1.9875691500e-4f);
    q = pk_fma(q, r, pk_splat(1.3981999507e-3f));
    q = pk_fma(q, r, pk_splat(8.3334519073e-3f));
    q = pk_fma(q, r, pk_splat(4.1665795894e-2f));
    q = pk_fma(q, r, pk_splat(1.6666665459e-1f));
    q = pk_fma(q, r, pk_splat(5.0000001201e-1f));
    const v2f r2 = r * r;
    q = pk_fma(q, r2, r);
    q = q + pk_splat(1.0f);
    return (v2f){ldexpf(q.x, (int)n.x), ldexpf(q.y, (int)n.y)};
}

// num / den, correctly rounded, for den in [1, 2] and num in {0} U [2^-87, 1] (see above)
__device__ __forceinline__ v2f lstm_div2(v2f num, v2f den) {
    v2f r = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const v2f e0 = pk_fma(-den, r, pk_splat(1.0f));
    r = pk_fma(e0, r, r);
    v2f q = num * r;
    v2f rem = pk_fma(-den, q, num);
    q = pk_fma(rem, r, q);
    rem = pk_fma(-den, q, num);
    return pk_fma(rem, r, q);
}

// two activations at once; T0 / T1: the element is a tanh (else a sigmoid)
template <bool T0, bool T1>
__device__ __forceinline__ v2f lstm_act2(v2f x) {
    const v2f ax = {fabsf(x.x), fabsf(x.y)};
    const v2f e = lstm_exp_nonpos2(ax * (v2f){T0 ? -2.0f : -1.0f, T1 ? -2.0f : -1.0f});
    const v2f den = pk_splat(1.0f) + e;
    v2f num;
    num.x = T0 ? 1.0f - e.x : (x.x >= 0.0f ? 1.0f : e.x);
    num.y = T1 ? 1.0f - e.y : (x.y >= 0.0f ? 1.0f : e.y);
    v2f v = lstm_div2(num, den);
    if (T0) v.x = copysignf(v.x, x.x);
    if (T1) v.y = copysignf(v.y, x.y);
    return v;
}

__device__ __forceinline__ float lstm_tanh(float x) { return lstm_act2<true, true>((v2f){x, x}).x; }

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
                act = r.out_act == 0 ? lstm_tanh(o) : (o < -1.0f ? -1.0f : (o > 1.0f ? 1.0f : o));
                if (r.means_out) r.means_out[(int64_t)k * NA + sl] = act;
                if (r.noise && n != p.eval_env) {  // distribution.sample() clamped; the eval env keeps the mean
                    const float dev = r.std * r.noise[(int64_t)k * NA + sl];
                    const float smp = act + dev;
                    act = smp < -1.0f ? -1.0f : (smp > 1.0f ? 1.0f : smp);
                }
                if (r.actions_out) r.actions_out[(int64_t)k * NA + sl] = act;
            }
            account_keep<SINGLE>(p, l, l_idx, A, e, a, active, n, st, act, r.rew_out + (int64_t)k * p.N,
                                 r.done_out + (int64_t)k * p.N);
            if (active && r.traj_src) {  // row k + 1: the observation this step returns (own LDS entries: no barrier needed)
                r.traj_pos[(int64_t)(k + 1) * NA + sl] = l.pos[e * A + a];
                if (a == 0) r.traj_src[(int64_t)(k + 1) * p.N + n] = l.src[e];
            }
            lds_barrier();  // the new observation's descriptors are complete; everyone is done with h_W
        }
        if (active) {  // state and descriptors go back to HBM once per launch
            p.cash[sl] = st.cash;
            p.lng[sl] = st.lng;
            p.sht[sl] = st.sht;
            p.margin[sl] = st.margin;
            r.obs_pos[sl] = l.pos[e * A + a];
            if (a == 0) {
                p.env_idx[n] = st.idx;
                p.spot0[n] = st.spot;
                r.obs_src[n] = l.src[e];
            }
        }
        __syncthreads();
    }
}

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
__global__ __launch_bounds__(kBlock) void fe_traj_returns_kernel(const double *__restrict__ rewards,
                                                                 const int32_t *__restrict__ dones,
                                                                 const float *__restrict__ values,
                                                                 const float *__restrict__ last_values,
                                                                 int64_t T, int64_t N, float g32,
                                                                 float *__restrict__ returns,
                                                                 float *__restrict__ adv) {
    for (int64_t n = blockIdx.x * (int64_t)kBlock + threadIdx.x; n < N; n += (int64_t)gridDim.x * kBlock) {
        double R = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            const float factor = (float)(1 - dones[t * N + n]) * g32;
            if (t == T - 1)
                R = rewards[t * N + n] + (double)(factor * last_values[n]);
            else
                R = rewards[t * N + n] + (double)factor * R;
            const float r32 = (float)R;
            returns[t * N + n] = r32;
            if (adv) adv[t * N + n] = r32 - values[t * N + n];
        }
    }
}

int grid_for(int64_t work_items) {
    int64_t g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

}  // namespace

struct fe_env {
    fe_config cfg;
    Params p;
    int grid;
    int vec;  // observation elements per 16-byte store (1 when the env size is odd)
    size_t lds;
    bool bound;
    int cus;              // compute units of that device
    int tile_override, grid_override, rollout_tile_override;  // fe_env_set_launch (tuning), 0 = automatic
    int device;           // HIP device the tables live on; every launch runs there
    double *owned_logret; // log-return table computed by fe_env_create(logret = NULL), else null
};

// Makes the env's device current for the duration of a call and restores the caller's device
// afterwards: the reference's `device_id` argument works without torch.cuda.set_device (TSE:28, 45),
// so a process driving several envs on several GPUs must not have to juggle the current device.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        if (device < 0) {
            err = hipErrorInvalidDevicePointer;
            return;
        }
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Device a caller-owned pointer lives on (-1 if it is not device memory).
static int device_of(const void *ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) return -1;
    return attr.device;
}

// The kernel instantiation a given env dispatches to (shared by launch and occupancy query).
template <bool RESET_ONLY>
static const void *kernel_for(bool f32, int vec, bool single) {
#define FE_PICK(OT, VEC) \
    (single ? (const void *)fe_env_kernel<OT, VEC, true, RESET_ONLY> : (const void *)fe_env_kernel<OT, VEC, false, RESET_ONLY>)
    if (f32) return vec == 4 ? FE_PICK(float, 4) : (vec == 2 ? FE_PICK(float, 2) : FE_PICK(float, 1));
    return vec == 2 ? FE_PICK(double, 2) : FE_PICK(double, 1);
#undef FE_PICK
}

// Per-call pointers go into a local copy of the parameter block: the env object itself is not
// modified by reset/step, so concurrent calls on different streams do not race on the host side.
template <bool RESET_ONLY>
static int launch_env(const fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                      hipStream_t st, int64_t *desc_src = nullptr, double *desc_pos = nullptr, float *act_store = nullptr) {
    Params p = env->p;
    p.actions = actions;
    p.obs = obs;
    p.rew = rewards;
    p.done = dones;
    p.desc_src = desc_src;
    p.desc_pos = desc_pos;
    p.act_store = act_store;
    void *args[] = {&p};
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipError_t he = hipLaunchKernel(kernel_for<RESET_ONLY>(env->cfg.obs_is_f32 != 0, env->vec, p.A == 1),
                                    dim3(env->grid), dim3(kBlock), args, env->lds, st);
    if (he != hipSuccess) return hip_fail(he, RESET_ONLY ? "fe_env_reset_obs launch" : "fe_env_step launch");
    return FE_OK;
}

// Launch geometry of the step / reset kernels: tile size EB, tile count, grid, dynamic LDS.
static int configure_launch(fe_env *env) {
    const fe_config &cfg = env->cfg;
    const int A = cfg.A;
    const void *kern = kernel_for<false>(cfg.obs_is_f32 != 0, env->vec, A == 1);
    // How many workgroups the chip holds at once for this kernel variant (registers + LDS).
    int64_t cap = kBlock / A > 0 ? kBlock / A : 1;  // one sleeve per lane in phase 1
#if FE_STEP_VARIANT == 1
    if (A == 1) cap = 64;  // a tile is accounted by one wavefront
#endif
    int per_cu = 0;
    hipError_t he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds_bytes((int)cap, A));
    if (he != hipSuccess) return hip_fail(he, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int64_t resident = (int64_t)env->cus * per_cu;
    if (resident > 8) resident -= resident % 8;  // keeps tile % 8 (the XCD label) constant per workgroup
    // Tile = EB consecutive envs.  Aim for ~8/3 tiles per resident workgroup: measured on
    // MI355X (round 1, profiles/r01_microbench/sweep_c2.txt, 64k envs) a few short tiles per workgroup beat one long
    // tile (workgroups drift apart, so phase 1 of one hides under phase 2 of its CU-mates).
    int64_t EB = (3 * cfg.N + 4 * resident) / (8 * resident);
    if (EB < 1) EB = 1;
    if (EB > cap) EB = cap;
    int wgs_per_cu = 0;  // 0 = whatever the occupancy query allows
#ifndef FE_GEOM_R01   /* experiment builds only: the round-1 geometry rule for every shape */
#define FE_GEOM_R01 0
#endif
    if (A == 1 && !FE_GEOM_R01) {
        // Single-asset envs (measured at 64k envs x W64 on a shared observation ring, tools/ab_step.py,
        // profiles/r02_microbench/sweep{3,4}_c2.txt, sweep_f32_c2.txt).  A tile must be a whole number of workgroup
        // iterations of phase 2 (4 wavefronts x one 5-KiB image = 512 f64 / 1024 f32 tuples): with f64 observations
        // 8 envs of W = 64 run 31.2 us, 12 envs 35.2 us, 6 envs 42.2 us.  Fewer workgroups than the occupancy limit
        // start faster (the dispatch ramp of 1792 workgroups costs up to 5 us of a 31 us launch): f64 observations
        // are fastest with 4 workgroups per CU and ~8 short tiles each (31.2-31.8 us vs 33.0-34.2 us for the round-1
        // geometry), f32 observations (half the bytes, a 17-19 us launch) with 6 per CU and 1-2 longer tiles each
        // (17.4 us vs 19.4 us).
        const int64_t wg_tuples = 4 * (kStageBytes / (5 * (cfg.obs_is_f32 ? 4 : 8)));
        int64_t g = wg_tuples, w = cfg.W;
        while (w) { const int64_t t = g % w; g = w; w = t; }  // gcd(wg_tuples, W)
        const int64_t unit = wg_tuples / g;                  // envs per whole workgroup iteration
        if (unit <= cap) {
            wgs_per_cu = cfg.obs_is_f32 ? (FE_F32_WAVES < 6 ? FE_F32_WAVES : 6) : 4;
            const int64_t res = (int64_t)env->cus * wgs_per_cu;
            if (resident > res) resident = res;
            // tiles per workgroup aimed at: 8 (f64) resp. 4/3 (f32)
            const int64_t num = cfg.obs_is_f32 ? 3 * cfg.N : cfg.N, den = (cfg.obs_is_f32 ? 4 : 8) * resident * unit;
            int64_t m = (num + den / 2) / den;
            if (m < 1) m = 1;
            EB = unit * m;
            if (EB > cap) EB = cap - cap % unit;
        }
    }
    if (env->tile_override > 0) EB = env->tile_override < cap ? env->tile_override : cap;
    const int64_t num_tiles = (cfg.N + EB - 1) / EB;
    // the LDS footprint depends on EB: ask again with the real size before fixing the grid
    he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds_bytes((int)EB, A));
    if (he == hipSuccess && per_cu >= 1) {
        if (per_cu > 8) per_cu = 8;
        if (wgs_per_cu > 0 && per_cu > wgs_per_cu) per_cu = wgs_per_cu;  // see above
        resident = (int64_t)env->cus * per_cu;
        if (resident > 8) resident -= resident % 8;
    }
    int64_t grid = num_tiles < resident ? num_tiles : resident;
    if (env->grid_override > 0) grid = env->grid_override;
    env->grid = (int)grid;
    env->lds = lds_bytes((int)EB, A);
    env->p.EB = (int)EB;
    env->p.num_tiles = num_tiles;
    return FE_OK;
}

extern "C" {

// shared with fe_csv.cpp (not part of the public header)
int fe_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int fe_version(void) { return FE_ABI_VERSION; }

const char *fe_last_error(void) { return g_err; }

int fe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int fe_env_create(const fe_config *cfg, const double *prices, const double *logret, fe_env **out) {
    if (!cfg || !out) return fail(FE_ERR_ARG, "fe_env_create: null argument");
    if (!prices) return fail(FE_ERR_ARG, "fe_env_create: the price table is required");
    if (cfg->N < 1 || cfg->D < 1) return fail(FE_ERR_ARG, "fe_env_create: N=%lld D=%lld must be >= 1", (long long)cfg->N, (long long)cfg->D);
    if (cfg->W < 1 || cfg->L <= cfg->W)
        return fail(FE_ERR_ARG, "fe_env_create: need 1 <= W < L (W=%lld, L=%lld)", (long long)cfg->W, (long long)cfg->L);
    if (cfg->A < 1 || cfg->A > FE_MAX_ASSETS)
        return fail(FE_ERR_ARG, "fe_env_create: A=%lld outside 1..%lld", (long long)cfg->A, (long long)FE_MAX_ASSETS);
    if (cfg->max_shares < 0) return fail(FE_ERR_ARG, "fe_env_create: max_shares < 0");
    if (cfg->redraw_mode != 0 && cfg->redraw_mode != 1) return fail(FE_ERR_ARG, "fe_env_create: redraw_mode must be 0 or 1");
    if (cfg->eval_env >= cfg->N) return fail(FE_ERR_ARG, "fe_env_create: eval_env out of range");
    const int64_t env_elems = (int64_t)cfg->W * 5 * cfg->A;
    if (env_elems > (1ll << 24)) return fail(FE_ERR_ARG, "fe_env_create: W*5*A too large");
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev < 1) {
        (void)hipGetLastError();
        return fail(FE_ERR_HIP, "fe_env_create: no HIP device (this library has no CPU path)");
    }
    // the env lives where its tables live, whatever the caller's current device is
    const int dev = device_of(prices);
    if (dev < 0 || dev >= ndev) return fail(FE_ERR_ARG, "fe_env_create: prices is not a device pointer");
    if (logret && device_of(logret) != dev)
        return fail(FE_ERR_ARG, "fe_env_create: prices and logret live on different devices");
    DeviceGuard guard(dev);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((he = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return hip_fail(he, "hipGetDeviceProperties");

    fe_env *env = new (std::nothrow) fe_env();
    if (!env) return fail(FE_ERR_ARG, "fe_env_create: out of host memory");
    env->cfg = *cfg;
    env->bound = false;
    env->device = dev;
    env->owned_logret = nullptr;
    if (!logret) {
        // logret = NULL: compute the table from the prices (the one allocation this library owns)
        const int64_t tuples = cfg->D * cfg->L * (int64_t)cfg->A;
        if ((he = hipMalloc(&env->owned_logret, (size_t)tuples * 32)) != hipSuccess) {
            delete env;
            return hip_fail(he, "fe_env_create: hipMalloc(logret)");
        }
        hipLaunchKernelGGL(fe_logret_tables_kernel, dim3(grid_for(tuples)), dim3(kBlock), 0, (hipStream_t) nullptr,
                           prices, env->owned_logret, cfg->D, cfg->L, cfg->A);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
        if (he != hipSuccess) {
            (void)hipFree(env->owned_logret);
            delete env;
            return hip_fail(he, "fe_env_create: log-return table");
        }
        logret = env->owned_logret;
    }
    const int A = cfg->A;
    const int elem_bytes = cfg->obs_is_f32 ? 4 : 8;
    int vec = 16 / elem_bytes;
    while (vec > 1 && env_elems % vec != 0) vec /= 2;
    env->vec = vec;
    env->cus = prop.multiProcessorCount;
    env->tile_override = 0;
    env->grid_override = 0;
    env->rollout_tile_override = 0;
    Params &p = env->p;
    memset(&p, 0, sizeof(p));
    p.N = cfg->N;
    p.A = A;
    if (int rc = configure_launch(env)) {
        if (env->owned_logret) (void)hipFree(env->owned_logret);
        delete env;
        return rc;
    }
    p.P = prices;
    p.LR = logret;
    p.N = cfg->N; p.D = cfg->D; p.L = cfg->L;
    p.W = cfg->W; p.A = A;
    p.eval_env = cfg->evaluate ? -1 : cfg->eval_env;
    p.seed = cfg->seed;
    p.evaluate = cfg->evaluate ? 1 : 0;
    p.redraw_mode = cfg->redraw_mode;
    p.env_elems = (uint32_t)env_elems;
    p.div_WA = make_fastdiv((uint32_t)((int64_t)cfg->W * cfg->A));
    p.div_A = make_fastdiv((uint32_t)A);
    p.scale32 = (float)((double)cfg->max_shares + 0.5);
    p.ms32 = (float)cfg->max_shares;
    p.c32 = (float)cfg->commission;
    p.imr32 = (float)cfg->init_margin;
    p.S32 = (float)cfg->starting_balance;
    p.comm = cfg->commission;
    p.imr = cfg->init_margin;
    p.one_mmr = 1.0 + cfg->maint_margin;
    p.S = cfg->starting_balance;
    *out = env;
    return FE_OK;
}

int fe_env_bind_state(fe_env *env, int64_t *env_idx, int64_t *spot0, float *cash, float *long_shares,
                      float *short_shares, double *margin, uint8_t *terminated, float *episode_returns,
                      int64_t *counters) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_bind_state: null env");
    if (!env_idx || !spot0 || !cash || !long_shares || !short_shares || !margin || !counters)
        return fail(FE_ERR_ARG, "fe_env_bind_state: null state pointer");
    if (env->cfg.evaluate && (!terminated || !episode_returns))
        return fail(FE_ERR_ARG, "fe_env_bind_state: evaluate mode needs terminated and episode_returns");
    Params &p = env->p;
    p.env_idx = env_idx; p.spot0 = spot0; p.cash = cash; p.lng = long_shares; p.sht = short_shares;
    p.margin = margin; p.terminated = terminated; p.ep_ret = episode_returns;
    p.counters = reinterpret_cast<unsigned long long *>(counters);
    env->bound = true;
    return FE_OK;
}

int fe_env_bind_f32_table(fe_env *env, const float *logret_f32) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_bind_f32_table: null env");
    if (logret_f32 && !env->cfg.obs_is_f32)
        return fail(FE_ERR_ARG, "fe_env_bind_f32_table: only meaningful with f32 observations");
    env->p.LR32 = logret_f32;
    return FE_OK;
}

int fe_env_bind_stats(fe_env *env, float *running_returns, double *accumulators, float *eval_return) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_bind_stats: null env");
    if (!running_returns) {  // unbind
        env->p.run_ret = nullptr;
        env->p.stat_acc = nullptr;
        env->p.stat_eval = nullptr;
        return FE_OK;
    }
    if (!accumulators || !eval_return) return fail(FE_ERR_ARG, "fe_env_bind_stats: null accumulator pointer");
#if FE_STAMP
    env->p.run_ret = nullptr;  // statistics off; eval_return carries the stamp buffer
    env->p.stat_acc = nullptr;
    env->p.stat_eval = eval_return;
#else
    env->p.run_ret = running_returns;
    env->p.stat_acc = accumulators;
    env->p.stat_eval = eval_return;
#endif
    return FE_OK;
}

int fe_env_reset_obs(fe_env *env, void *obs, void *stream) {
    if (!env || !obs) return fail(FE_ERR_ARG, "fe_env_reset_obs: null argument");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_reset_obs: state not bound");
    return launch_env<true>(env, nullptr, obs, nullptr, nullptr, (hipStream_t)stream);
}

int fe_env_step(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones, void *stream) {
    if (!env || !actions || !obs || !rewards || !dones) return fail(FE_ERR_ARG, "fe_env_step: null argument");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_step: state not bound");
    return launch_env<false>(env, actions, obs, rewards, dones, (hipStream_t)stream);
}

int fe_env_step_traj(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                     float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out, void *stream) {
    if (!env || !actions || !obs || !rewards || !dones) return fail(FE_ERR_ARG, "fe_env_step_traj: null argument");
    if ((obs_src_out == nullptr) != (obs_pos_out == nullptr))
        return fail(FE_ERR_ARG, "fe_env_step_traj: obs_src_out and obs_pos_out go together");
    if (actions_store_out == actions) actions_store_out = nullptr;  // already where they belong
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_step_traj: state not bound");
    return launch_env<false>(env, actions, obs, rewards, dones, (hipStream_t)stream, obs_src_out, obs_pos_out, actions_store_out);
}

int fe_env_describe(fe_env *env, int64_t *obs_src, double *obs_pos, void *stream) {
    if (!env || !obs_src || !obs_pos) return fail(FE_ERR_ARG, "fe_env_describe: null argument");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_describe: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    const Params &p = env->p;
    dim3 g(grid_for(p.N * p.A)), b(kBlock);
    if (p.A == 1)
        hipLaunchKernelGGL(fe_describe_kernel<true>, g, b, 0, (hipStream_t)stream, p, obs_src, obs_pos);
    else
        hipLaunchKernelGGL(fe_describe_kernel<false>, g, b, 0, (hipStream_t)stream, p, obs_src, obs_pos);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_describe launch");
    return FE_OK;
}

int fe_env_render(fe_env *env, const int64_t *obs_src, const double *obs_pos, void *obs, void *stream) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_render: null argument");
    return fe_env_render_n(env, obs_src, obs_pos, env->cfg.N, obs, stream);
}

int fe_env_render_n(fe_env *env, const int64_t *obs_src, const double *obs_pos, int64_t count, void *obs, void *stream) {
    if (!env || !obs_src || !obs_pos || !obs || count < 0) return fail(FE_ERR_ARG, "fe_env_render_n: bad argument");
    if (count == 0) return FE_OK;
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    p.obs = obs;
    p.N = count;  // any number of descriptors, e.g. a minibatch drawn from a trajectory of them
    p.num_tiles = (count + p.EB - 1) / p.EB;
    const bool f32 = env->cfg.obs_is_f32 != 0, single = p.A == 1;
    dim3 g((unsigned)(p.num_tiles < (int64_t)env->grid ? p.num_tiles : (int64_t)env->grid)), b(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = env->lds;
#define FE_RENDER(OT, VEC)                                                                                 \
    do {                                                                                                   \
        if (single) hipLaunchKernelGGL((fe_render_kernel<OT, VEC, true>), g, b, lds, st, p, obs_src, obs_pos);  \
        else hipLaunchKernelGGL((fe_render_kernel<OT, VEC, false>), g, b, lds, st, p, obs_src, obs_pos);        \
    } while (0)
    if (f32) {
        if (env->vec == 4) FE_RENDER(float, 4);
        else if (env->vec == 2) FE_RENDER(float, 2);
        else FE_RENDER(float, 1);
    } else {
        if (env->vec == 2) FE_RENDER(double, 2);
        else FE_RENDER(double, 1);
    }
#undef FE_RENDER
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_render_n launch");
    return FE_OK;
}

int fe_env_rollout_linear(fe_env *env, const double *weights, double bias, int32_t K, int64_t *obs_src,
                          double *obs_pos, float *actions_out, double *rewards_out, int32_t *dones_out,
                          void *stream) {
    if (!env || !weights || !obs_src || !obs_pos || !rewards_out || !dones_out || K < 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_linear: bad argument");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_linear: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    // the rollout is latency-bound (policy -> accounting -> policy ...): 64 sleeves per workgroup measured
    // best at 64k envs (tools/fused_bench.py), independent of the tile the streaming step kernel uses
    int64_t cap = kBlock / p.A > 0 ? kBlock / p.A : 1;
    int64_t eb = 64 / p.A;
    if (eb < 8) eb = 8;  // but never fewer than 8 envs per workgroup when they fit
    if (eb > cap) eb = cap;
    if (env->rollout_tile_override > 0) eb = env->rollout_tile_override < cap ? env->rollout_tile_override : cap;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    const size_t lds = rollout_lds_bytes(p.EB, p.A, p.W);
    RolloutArgs r;
    r.weights = weights; r.bias = bias; r.K = K; r.obs_src = obs_src; r.obs_pos = obs_pos;
    r.actions_out = actions_out; r.rew_out = rewards_out; r.done_out = dones_out;
    // state lives in HBM between steps but every tile is revisited by the same workgroup, so a
    // grid of one workgroup per tile (capped) keeps the K-step loop entirely inside the launch
    int64_t grid = p.num_tiles < 8 * 256 ? p.num_tiles : 8 * 256;
    dim3 g((unsigned)grid), b(kBlock);
    if (p.A == 1)
        hipLaunchKernelGGL(fe_rollout_linear_kernel<true>, g, b, lds, (hipStream_t)stream, p, r);
    else
        hipLaunchKernelGGL(fe_rollout_linear_kernel<false>, g, b, lds, (hipStream_t)stream, p, r);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_linear launch");
    return FE_OK;
}

int fe_policy_table(fe_env *env, const double *weights, double *table, double *wsum, void *stream) {
    if (!env || !weights || !table || !wsum) return fail(FE_ERR_ARG, "fe_policy_table: null argument");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    const Params &p = env->p;
    const int64_t entries = p.D * p.L * p.A;
    int64_t blocks = (entries * 64 + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fe_policy_table_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, p, weights,
                       table, wsum);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_policy_table launch");
    return FE_OK;
}

int fe_env_rollout_table(fe_env *env, const double *table, const double *wsum, double bias, int32_t K,
                         int64_t *obs_src, double *obs_pos, float *actions_out, double *rewards_out,
                         int32_t *dones_out, void *stream) {
    if (!env || !table || !wsum || !obs_src || !obs_pos || !rewards_out || !dones_out || K < 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_table: bad argument");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_table: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    // lane-private loop (no LDS traffic at one asset): full workgroups of sleeves
    int64_t eb = kBlock / p.A > 0 ? kBlock / p.A : 1;
    if (env->rollout_tile_override > 0 && env->rollout_tile_override < eb) eb = env->rollout_tile_override;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    TableRolloutArgs r;
    r.table = table; r.wsum = wsum; r.bias = bias; r.K = K; r.obs_src = obs_src; r.obs_pos = obs_pos;
    r.actions_out = actions_out; r.rew_out = rewards_out; r.done_out = dones_out;
    int64_t grid = p.num_tiles < 8 * 256 ? p.num_tiles : 8 * 256;
    const size_t lds = table_rollout_lds_bytes(p.EB, p.A);
    dim3 g((unsigned)grid), b(kBlock);
    if (p.A == 1)
        hipLaunchKernelGGL(fe_rollout_table_kernel<true>, g, b, lds, (hipStream_t)stream, p, r);
    else
        hipLaunchKernelGGL(fe_rollout_table_kernel<false>, g, b, lds, (hipStream_t)stream, p, r);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_table launch");
    return FE_OK;
}

int fe_env_rollout_mlp(fe_env *env, const float *logret_f32, const float *w1t, const float *wpos, const float *b1,
                       const float *w2, float b2, int32_t H, int32_t activation, int32_t K, int64_t *obs_src,
                       double *obs_pos, float *actions_out, double *rewards_out, int32_t *dones_out, void *stream) {
    if (!env || !logret_f32 || !w1t || !wpos || !b1 || !w2 || !obs_src || !obs_pos || !rewards_out || !dones_out || K < 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_mlp: bad argument");
    if (H != 32 && H != 64 && H != 128) return fail(FE_ERR_ARG, "fe_env_rollout_mlp: H must be 32, 64 or 128 (got %d)", (int)H);
    if (activation < 0 || activation > 2) return fail(FE_ERR_ARG, "fe_env_rollout_mlp: activation must be 0 (ELU), 1 (ReLU) or 2 (tanh)");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_mlp: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    MlpArgs r;
    r.lr32 = logret_f32; r.w1t = w1t; r.wpos = wpos; r.b1 = b1; r.w2 = w2; r.b2 = b2; r.H = H; r.act = activation; r.K = K;
    r.obs_src = obs_src; r.obs_pos = obs_pos; r.actions_out = actions_out; r.rew_out = rewards_out; r.done_out = dones_out;
    // 128 pairs per workgroup = four 32-pair MFMA column blocks, one per wavefront, two workgroups per CU.
    // (A 512-thread form running policy and accounting of two sub-tiles in antiphase was tried and dropped: on
    // gfx950 the f32-input MFMA executes on the vector ALUs -- SQ_VALU_MFMA_COEXEC_CYCLES = 0 -- so there is
    // nothing for the accounting to hide behind; profiles/r02_microbench/mlp_prof.txt.)
    int64_t cap = kBlock / p.A > 0 ? kBlock / p.A : 1;
    int64_t eb = 128 / p.A;
    if (eb < 1) eb = 1;
    if (eb > cap) eb = cap;
    if (env->rollout_tile_override > 0) eb = env->rollout_tile_override < cap ? env->rollout_tile_override : cap;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    const size_t lds = mlp_lds_bytes(p.EB, p.A, p.W, H);
    const bool single = p.A == 1;
#define FE_MLP(NT) (single ? (const void *)fe_rollout_mlp_kernel<true, NT> : (const void *)fe_rollout_mlp_kernel<false, NT>)
    const void *kern = H == 32 ? FE_MLP(1) : (H == 64 ? FE_MLP(2) : FE_MLP(4));
#undef FE_MLP
    const int64_t grid = p.num_tiles < 8 * 256 ? p.num_tiles : 8 * 256;
    const int block = kBlock;
    if (lds > 160 * 1024)
        return fail(FE_ERR_ARG, "fe_env_rollout_mlp: W1 (%d x %d) does not fit the 160 KiB LDS (%zu bytes needed)", (int)H, 4 * p.W, lds);
    hipError_t he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_mlp: hipFuncSetAttribute");
    void *args[] = {&p, &r};
    he = hipLaunchKernel(kern, dim3((unsigned)grid), dim3(block), args, lds, (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_mlp launch");
    return FE_OK;
}

int fe_env_rollout_lstm(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout,
                        float bout, int32_t H, int32_t out_activation, int32_t K, int64_t *obs_src, double *obs_pos,
                        const float *noise, float std, float *actions_out, float *means_out, double *rewards_out,
                        int32_t *dones_out, int64_t *states_src_out, double *states_pos_out, void *stream) {
    if ((states_src_out == nullptr) != (states_pos_out == nullptr))
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm: states_src_out and states_pos_out go together");
    if (noise && !(std >= 0.0f)) return fail(FE_ERR_ARG, "fe_env_rollout_lstm: std must be >= 0 when noise is given");
    if (!env || !logret_f32 || !whh || !wx || !wout || !obs_src || !obs_pos || !rewards_out || !dones_out || K < 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm: bad argument");
    if (H != 32 && H != 64 && H != 128) return fail(FE_ERR_ARG, "fe_env_rollout_lstm: H must be 32, 64 or 128 (got %d)", (int)H);
    if (out_activation < 0 || out_activation > 1) return fail(FE_ERR_ARG, "fe_env_rollout_lstm: out_activation must be 0 (tanh) or 1 (clamp)");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_lstm: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    LstmArgs r;
    r.lr32 = logret_f32; r.whh = whh; r.wx = wx; r.wout = wout; r.bout = bout; r.H = H; r.out_act = out_activation; r.K = K;
    r.obs_src = obs_src; r.obs_pos = obs_pos; r.actions_out = actions_out; r.rew_out = rewards_out; r.done_out = dones_out;
    r.noise = noise; r.std = std; r.means_out = means_out; r.traj_src = states_src_out; r.traj_pos = states_pos_out;
    // SP (env, asset) pairs per workgroup: 2 (H = 128) or 4 column tiles of 32; an env's sleeves stay together
    const int SP = H == 128 ? LstmGeom<4>::SP : LstmGeom<2>::SP;
    if (p.A > SP)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm: %d assets per env exceed the %d pairs of a workgroup tile (H = %d)", (int)p.A, SP, (int)H);
    int64_t eb = SP / p.A;
    if (env->rollout_tile_override > 0 && env->rollout_tile_override < eb) eb = env->rollout_tile_override;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    const size_t lds = lstm_lds_bytes(p.EB, p.A, H, SP);
    const bool single = p.A == 1;
#define FE_LSTM(NT) (single ? (const void *)fe_rollout_lstm_kernel<true, NT> : (const void *)fe_rollout_lstm_kernel<false, NT>)
    const void *kern = H == 32 ? FE_LSTM(1) : (H == 64 ? FE_LSTM(2) : FE_LSTM(4));
#undef FE_LSTM
    hipError_t he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_lstm: hipFuncSetAttribute");
    int per_cu = 0;
    he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kLstmBlock, lds);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_lstm: hipOccupancyMaxActiveBlocksPerMultiprocessor");
    if (per_cu < 1) per_cu = 1;
    const int64_t resident = (int64_t)env->cus * per_cu;  // the weights sit in registers: one pass of resident workgroups
    const int64_t grid = p.num_tiles < resident ? p.num_tiles : resident;
    void *args[] = {&p, &r};
    he = hipLaunchKernel(kern, dim3((unsigned)grid), dim3(kLstmBlock), args, lds, (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_lstm launch");
    return FE_OK;
}

int fe_lstm_activations(const float *x, float *sigmoid_out, float *tanh_out, int64_t n, void *stream) {
    if (!x || !sigmoid_out || !tanh_out || n < 0) return fail(FE_ERR_ARG, "fe_lstm_activations: bad argument");
    if (n == 0) return FE_OK;
    DeviceGuard guard(device_of(x));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipError_t he;
    int64_t grid = (n + kBlock - 1) / kBlock;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fe_lstm_activations_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, x, sigmoid_out, tanh_out, n);
    he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_lstm_activations launch");
    return FE_OK;
}

int fe_env_set_day(fe_env *env, int64_t env_index, int64_t day, void *stream) {
    if (!env || !env->bound) return fail(FE_ERR_STATE, "fe_env_set_day: env not bound");
    if (env_index < 0 || env_index >= env->cfg.N || day < 0 || day >= env->cfg.D)
        return fail(FE_ERR_ARG, "fe_env_set_day: env %lld / day %lld out of range", (long long)env_index, (long long)day);
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipError_t he = hipMemcpyAsync(env->p.env_idx + env_index, &day, sizeof(int64_t), hipMemcpyHostToDevice,
                                   (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "fe_env_set_day");
    // the source is a stack variable: make the copy complete before it goes away
    he = hipStreamSynchronize((hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "fe_env_set_day sync");
    return FE_OK;
}

int fe_env_launch_info(const fe_env *env, int32_t *grid, int32_t *block, int32_t *tile_envs, int32_t *lds) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_launch_info: null env");
    if (grid) *grid = env->grid;
    if (block) *block = kBlock;
    if (tile_envs) *tile_envs = env->p.EB;
    if (lds) *lds = (int32_t)env->lds;
    return FE_OK;
}

int fe_env_set_launch(fe_env *env, int32_t tile_envs, int32_t grid, int32_t rollout_tile_envs) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_set_launch: null env");
    if (tile_envs < 0 || grid < 0 || rollout_tile_envs < 0) return fail(FE_ERR_ARG, "fe_env_set_launch: negative value");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    env->tile_override = tile_envs;
    env->grid_override = grid;
    env->rollout_tile_override = rollout_tile_envs;
    return configure_launch(env);
}

const char *fe_build_tag(void) { return FE_BUILD_TAG; }

int fe_env_destroy(fe_env *env) {
    if (env && env->owned_logret) {
        DeviceGuard guard(env->device);
        (void)hipFree(env->owned_logret);
    }
    delete env;
    return FE_OK;
}

int fe_env_device(const fe_env *env) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_device: null env");
    return env->device;
}

const double *fe_env_logret(const fe_env *env) { return env ? env->p.LR : nullptr; }

int fe_build_logret(const double *prices, double *out, int64_t T, int32_t A, void *stream) {
    if (!prices || !out || T < 1 || A < 1) return fail(FE_ERR_ARG, "fe_build_logret: bad argument");
    DeviceGuard guard(device_of(prices));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipLaunchKernelGGL(fe_logret_kernel, dim3(grid_for(T * A)), dim3(kBlock), 0, (hipStream_t)stream, prices, out, T, A);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_build_logret launch");
    return FE_OK;
}

int fe_build_logret_tables(const double *prices, double *out, int64_t D, int64_t L, int32_t A, void *stream) {
    if (!prices || !out || D < 1 || L < 1 || A < 1) return fail(FE_ERR_ARG, "fe_build_logret_tables: bad argument");
    DeviceGuard guard(device_of(prices));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipLaunchKernelGGL(fe_logret_tables_kernel, dim3(grid_for(D * L * A)), dim3(kBlock), 0, (hipStream_t)stream, prices,
                       out, D, L, A);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_build_logret_tables launch");
    return FE_OK;
}

int fe_build_tables(const double *series, const int64_t *starts, const int64_t *stops, int64_t D, int64_t L,
                    int32_t A, double *out, void *stream) {
    if (!series || !starts || !stops || !out || D < 1 || L < 1 || A < 1)
        return fail(FE_ERR_ARG, "fe_build_tables: bad argument");
    DeviceGuard guard(device_of(series));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipLaunchKernelGGL(fe_tables_kernel, dim3(grid_for(D * L * 4 * A)), dim3(kBlock), 0, (hipStream_t)stream, series,
                       starts, stops, D, L, A, out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_build_tables launch");
    return FE_OK;
}

int fe_traj_store(int64_t t, int64_t N, int32_t A, const float *actions, const double *rewards,
                  const int32_t *dones, float *traj_actions, double *traj_rewards, int32_t *traj_dones,
                  void *stream) {
    if (t < 0 || N < 1 || A < 1 || !actions || !rewards || !dones || !traj_actions || !traj_rewards || !traj_dones)
        return fail(FE_ERR_ARG, "fe_traj_store: bad argument");
    DeviceGuard guard(device_of(traj_rewards));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    const int64_t NA = N * A;
    hipLaunchKernelGGL(fe_traj_store_kernel, dim3(grid_for(NA)), dim3(kBlock), 0, (hipStream_t)stream, N, NA, actions,
                       rewards, dones, traj_actions + t * NA, traj_rewards + t * N, traj_dones + t * N);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_traj_store launch");
    return FE_OK;
}

int fe_traj_returns(const double *rewards, const int32_t *dones, const float *values, const float *last_values,
                    int64_t T, int64_t N, double gamma, float *returns, float *advantages, void *stream) {
    if (!rewards || !dones || !last_values || !returns || T < 1 || N < 1 || (advantages && !values))
        return fail(FE_ERR_ARG, "fe_traj_returns: bad argument");
    DeviceGuard guard(device_of(rewards));
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipLaunchKernelGGL(fe_traj_returns_kernel, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, rewards, dones,
                       values, last_values, T, N, (float)gamma, returns, advantages);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_traj_returns launch");
    return FE_OK;
}

}  // extern "C"
