// fe_device_common.h -- part of fe_env.hip (one translation unit; see the overview there): constants and build knobs, the kernel parameter block, Philox, the sleeve accounting (TSE:298-421, 447-475), LDS tile layout, input loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "finenvs_amd.h"

namespace {

constexpr int kBlock = 256;
// Wavefronts per SIMD the streaming kernels are built for.  Reset / render: 7 (72 VGPRs; the 20 KiB LDS stage allows
// 7 workgroups per CU too).  Multi-asset step: 6 (80 VGPRs; the per-sleeve LDS arrays -- 26 KiB at 8 envs x 30 assets
// -- allow 6 per CU).  Single-asset step: see kEnvKernelWaves (fe_step_kernel.h).
constexpr int kRenderWaves = 7;
constexpr int kMultiAssetWaves = 6;
// Cache policy of the observation stores (raw buffer stores, aux bits: 1 = sc0, 2 = nt, 16 = sc1).  A 0.17-150 GB store
// stream must not evict what phase 1 and phase 2 re-read every step from the 4 MiB L2s:
//   * single-asset envs: sc1 (write-through, the line is dropped from L2).  The per-env state, the action and the
//     tables then stay L2-resident, which shortens the kernel's start-up chain (index load -> bar gather -> accounting
//     -> first store): measured at 64k envs (round 2, interleaved in one process, three boxes) 34.8 -> 31.1,
//     34.5 -> 32.0, 34.8 -> 33.6 us per step; nt gains about 1 % less, sc0|sc1 the same.
//   * multi-asset envs: nt.  Measured at 1M envs x 30 assets (profiles/r01_microbench/store_policy.txt): FETCH_SIZE
//     6.9 GiB -> 0.4 GiB per launch and 26.4 -> 25.0 ms; sc1 gives the same fetch reduction but 25.6 ms.
//   * single-asset envs whose observation RING does not fit the 256 MiB Infinity Cache (round 4): sc1 | nt.  Back-to-back
//     launches into ONE 168 MB buffer run 27.5 us, alternating over two (336 MB) 29.1 us, A,A,B,B 27.7 us
//     (profiles/r04_microbench/ring_alternation.txt): the memory-side cache absorbs a rewritten buffer and thrashes on a
//     ring larger than itself.  nt on top of sc1 keeps the stream out of it: 64k envs x W64 f64 (ring 336 MB) 29.73 ->
//     28.93 and 28.84 -> 28.21 us on two boxes; with f32 observations (ring 168 MB, fits) the same bits cost +8.9 %, and
//     at 256k x 30 assets sc1 | nt instead of nt is +0.5 % (profiles/r04_microbench/ab_store_aux.txt).  The host decides per
//     env (fe_env.hip: Params::obs_stream = one observation buffer >= 128 MiB, i.e. two of them overflow the cache).
constexpr int kStoreAuxSingle = 16;
constexpr int kStoreAuxSingleStream = 16 | 2;
constexpr int kStoreAuxMulti = 2;
// "" for the product library; experiment builds (finenvs_amd/csrc/build.py build_variant) carry their -D set here
// and are only ever loaded by explicit path
#ifndef FE_BUILD_TAG
#define FE_BUILD_TAG ""
#endif
// Build knobs (experiment builds only; the product uses the defaults):
//   FE_HOIST_FIRST  1: the single-asset f64 step kernel issues the first tile's table loads before its accounting
//                   (profiles/r02_microbench/ab_hoist.txt: 30.57 -> 29.32 us at config 2 on a shared ring)
//   FE_F32_WAVES    wavefronts per SIMD the single-asset f32-observation step kernel is built for
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
    double *stat_acc;    // (N, 3) per-env partials: [3n] finished training episodes of env n, [3n + 1] sum of their returns, [3n + 2] sum of squares
    float *stat_eval;    // [0] return of the eval env's last finished episode, [1] how many it finished
    const float *actions;  // f32 actions; PROMO launches with act_f64 read them as const double * instead
    void *obs;
    double *rew;
    int32_t *done;
    int64_t *desc_src;   // optional (fe_env_step_traj): descriptors of the observation this step returns
    double *desc_pos;
    float *act_store;    // optional (fe_env_step_traj): the actions, copied into a trajectory slot
    unsigned long long *host_flag;  // optional (fe_env_step_notify): host memory that learns early whether the eval env finished
    unsigned long long flag_seq;
    unsigned int *ticket;           // evaluate-mode envs: workgroups-finished counter of the notify form (env-owned, zero between launches)
    int64_t N, D, L;
    int64_t num_tiles;
    int64_t eval_env;
    uint64_t seed;
    int32_t W, A, EB;
    int32_t evaluate, redraw_mode;
    uint32_t env_elems;  // W * 5 * A, observation elements per env
    int32_t obs_stream;  // single-asset envs: 1 = the observation ring is larger than the Infinity Cache (stores sc1 | nt)
    FastDiv div_WA;  // by tuples per env (W * A)
    FastDiv div_A;
    float scale32, ms32, c32, imr32, S32;
    double comm, imr, one_mmr, S;
    // PROMO launches only (fe_env_step_promoted), kept at the END of the block: the other kernels' scalar loads of the
    // fields above stay as they were
    double scale64, ms64;  // max_shares + 0.5 and max_shares in f64 (f64 actions)
    int32_t act_f64;       // this step's actions are f64 (TSE:298-302 in f64)
    int32_t has_stats;     // run_ret != nullptr (set per launch): the resident flag in front of the statistics pointers, which
                           // the step kernel reads from the kernarg segment at their use (fe_step_kernel.h, cold parameters)
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

struct Sleeve {
    float cash, lng, sht;
    double margin;
    double pos_obs;
    double rew;
    bool bankrupt;
};

// One (env, asset) account for one bar: TSE:298-421 (trade), TSE:428-431
// (position feature), TSE:447-475 (reward).  Pure register arithmetic.
// PROMO = the arithmetic of an env whose share tensors have been PROMOTED to f64 (fe_env_step_promoted): the reference's
// long_shares / short_shares are f32 until the first step() with f64 actions rebinds them to f64 results (TSE:353-361,
// 367-374) and stay f64 from then on.  Share counts are small integers -- the same values in either dtype -- so what
// changes is the precision of three kinds of products: shares x per_share_commission is an f64 product added once-
// rounded into the f32 commission accumulator (TSE:363-365) for the sell / buy-back legs always, and for the entry legs
// when this step's share CHANGES are f64 too (f64 actions; with f32 actions they are f32 tensors and stay f32 products);
// the short-entry commission likewise (TSE:401-421); and the share change itself is scaled, rounded and clamped in
// the actions' dtype (TSE:298-302).  The liquidation fee follows in account_core.
template <bool PROMO>
__device__ __forceinline__ void sleeve_step(const Params &p, typename std::conditional<PROMO, double, float>::type action,
                                            double O, double H, double Lo, double C, Sleeve &s) {
    float cash = s.cash, lng = s.lng, sht = s.sht;
    double margin = s.margin;
    float comm = 0.0f;  // TSE:305
    const bool a64 = PROMO && p.act_f64 != 0;  // this step's share changes are f64 (uniform)

    // TSE:298-302  round-half-even then clamp
    float sc;
    if constexpr (PROMO) {
        if (a64) {
            double s64 = rint(action * p.scale64);
            s64 = s64 < -p.ms64 ? -p.ms64 : s64;
            s64 = s64 > p.ms64 ? p.ms64 : s64;
            sc = (float)s64;  // an integer in [-max_shares, max_shares] (or NaN): exact
        } else {
            sc = rintf((float)action * p.scale32);  // (an f32 action, carried as a double)
            sc = sc < -p.ms32 ? -p.ms32 : sc;
            sc = sc > p.ms32 ? p.ms32 : sc;
        }
    } else {
        sc = rintf(action * p.scale32);
        sc = sc < -p.ms32 ? -p.ms32 : sc;
        sc = sc > p.ms32 ? p.ms32 : sc;
    }
    float pos = sc < 0.0f ? 0.0f : sc;  // TSE:344-351
    float neg = sc > 0.0f ? 0.0f : sc;
    // commissions += num_shares * per_share_commission, TSE:363-365: an f32 product and an f32 sum, or -- `wide`: the
    // share count is an f64 tensor -- an f64 product added to the f32 accumulator with ONE rounding
    auto add_commission = [&](float shares, bool wide) {
        if (wide) comm = (float)((double)comm + (double)shares * p.comm);
        else comm += shares * p.c32;
    };

    // sell long positions first, TSE:353-361
    float nl = relu32(lng + neg);
    float sell = lng - nl;
    neg += sell;
    add_commission(sell, PROMO);
    cash = (float)((double)cash + (double)sell * (O - p.comm));
    lng = nl;

    // buy back shorts and re-mark the margin account, TSE:367-383
    float ns = relu32(sht - pos);
    float bb = sht - ns;
    pos -= bb;
    add_commission(bb, PROMO);
    cash = (float)((double)cash - (double)bb * (O + p.comm));
    sht = ns;
    // new_margin = initial_margin_requirement * short_shares * open, TSE:376-379: the first product is in short_shares'
    // dtype -- f32 (imr rounded to f32) until the promotion, f64 with the full-precision imr after it
    double nm;
    if constexpr (PROMO) nm = (p.imr * (double)sht) * O;
    else nm = (double)(p.imr32 * sht) * O;
    cash = (float)((double)cash - (nm - margin));
    margin = nm;

    // long entry unless unaffordable, TSE:385-399
    if ((double)cash - (double)pos * (O + p.comm) < 0.0) pos = 0.0f;
    add_commission(pos, a64);
    cash = (float)((double)cash - (double)pos * (O + p.comm));
    lng += pos;

    // short entry unless the 150% margin is unaffordable, TSE:401-421
    float q = -neg;
    // short_commission = -negative_share_changes * per_share_commission: in the share changes' dtype
    if (((double)cash - p.imr * ((double)q * O)) - (a64 ? (double)q * p.comm : (double)(q * p.c32)) < 0.0) {
        neg = 0.0f;
        q = -neg;
    }
    add_commission(q, a64);
    double req = p.imr * ((double)q * O);
    cash = (float)((double)cash - (req + (a64 ? (double)q * p.comm : (double)(q * p.c32))));
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
// `per_sleeve_arrays`: the multi-asset tile loop's rew / shr / flg / any arrays (always for A > 1; promoted launches with f32
// observations run that loop for A = 1 as well)
__host__ __device__ inline size_t lds_bytes(int EB, int A, bool per_sleeve_arrays = false) {
    size_t S = (size_t)EB * A;
    size_t b = 4 * (size_t)kStageBytes + (size_t)EB * 8 + S * 8;
    if (A > 1 || per_sleeve_arrays) b += S * 8 + S * 4 + S * 4 + (size_t)EB * 4;
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
    if (active) {
        idx = p.env_idx[n];
        spot = p.spot0[n];
    }
}

// the part of the body that needs no index: account state of the sleeve (issued together with the head for a
// workgroup's first tile, so that only the L2-resident bar gather sits behind the index load)
__device__ __forceinline__ void load_state(const Params &p, bool active, int64_t sl, SleeveIn &in) {
    if (!active) return;
    in.cash = p.cash[sl];
    in.lng = p.lng[sl];
    in.sht = p.sht[sl];
    in.margin = p.margin[sl];
}

// the part that does: the bar at the window's last row and the NaN probe of the next row
__device__ __forceinline__ void load_bar(const Params &p, int A, int a, bool active, int64_t idx, int64_t spot,
                                         SleeveIn &in) {
    if (!active) return;
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

}  // namespace
