// fe_step_kernel.h -- part of fe_env.hip (one translation unit; see the overview there): the fused step kernel fe_env_kernel: phases 1 / 1b (account_core), phase 2 (stream_tile), the single-asset software pipeline.
#pragma once
#include "fe_device_common.h"

namespace {

// Phases 1 and 1b for one tile from preloaded inputs: every thread of the workgroup must call it
// (it holds barriers).  On return l.src / l.pos describe the observation of this step (terminal
// window on done steps, exactly what step() returns, TSE:321) and the state arrays hold the
// post-step (post-reset) state.
// FORM: kLean = launches that have none of the optional outputs but the action copy (no evaluate-mode bookkeeping, episode
// statistics or trajectory descriptors): their pointers never become live scalars; kFull = all of them;
// kNotify / kFullNotify = the lean / full form + the host flag of fe_env_step_notify / fe_env_step_traj_notify (and their
// last-tile-first walk).
constexpr int kLean = 0, kFull = 1, kNotify = 2, kFullNotify = 3;
constexpr bool form_is_full(int form) { return form == kFull || form == kFullNotify; }
constexpr bool form_notifies(int form) { return form == kNotify || form == kFullNotify; }
// COLD PARAMETERS.  Params is ~100 dwords and the compiler keeps all of it in SGPRs from kernel entry on: in the full forms
// it then runs out of the 106 it has, spills 16-register tuples into VGPR lanes and re-reads WHOLE tuples (v_readlane x 16) to
// use two registers of them -- the full + host-flag form (FORM 3, the step of redraw='torch' with trajectory outputs) carried
// 553 v_readlane on the accounting path and ran 29.9 us where forms 1 and 2 run 28.5 (profiles/r05_microbench/form_ab.txt).
// Everything only rare or mode-specific branches need (evaluate-mode bookkeeping, episode statistics, host flag, Philox
// redraw, the evaluate-mode ticket) is therefore read FROM THE KERNARG SEGMENT AT ITS USE, behind a resident scalar flag
// (p.evaluate, p.has_stats, n == p.eval_env):
// `ColdFor<FORM, SINGLE>::of(p)` is the kernel's own argument block (its only argument, at offset 0 of the segment) behind an opaque pointer --
// the empty asm keeps the compiler from hoisting the scalar loads back to the kernel's entry -- so each becomes an
// s_load next to its use and occupies registers only there.
#ifndef FE_COLD_PARAMS
#define FE_COLD_PARAMS 1
#endif
// Only the full + host-flag form takes its cold parameters this way (FE_COLD_PARAMS 1): it is the one that spilled, and the
// other forms keep the code they were measured with (with the cold reads in ALL forms: FORM 1 28.6 -> 28.85 us, forms 0 / 2
// unchanged, FORM 3 29.9 -> 28.95 us; profiles/r05_microbench/form_ab.txt).
template <bool LAUNDER>
struct Cold;
template <>
struct Cold<true> {
    typedef const __attribute__((address_space(4))) Params *Ptr;
    static __device__ __forceinline__ Ptr of(const Params &) {
        Ptr kp = (Ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return kp;
    }
};
template <>
struct Cold<false> {
    typedef const Params *Ptr;
    static __device__ __forceinline__ Ptr of(const Params &p) { return &p; }
};
// (... and only in the single-asset kernels: the multi-asset tile loop never re-read spilled tuples on a hot path, and with
// the cold reads three of its FORM 3 instantiations started to use 36 bytes of scratch.)
template <int FORM, bool SINGLE>
using ColdFor = Cold<(FE_COLD_PARAMS != 0) && FORM == 3 && SINGLE>;
template <bool PROMO>
using ActionT = typename std::conditional<PROMO, double, float>::type;
template <bool SINGLE, int FORM, bool PROMO = false>
__device__ __forceinline__ void account_core(const Params &p, const TileLds &l, int A, int e, int a, bool active,
                                             int64_t n, int64_t sl, const SleeveIn &in, ActionT<PROMO> action,
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
        sleeve_step<PROMO>(p, action, in.bar.x, in.bar.y, in.bar.z, in.bar.w, s);
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
        if constexpr (form_is_full(FORM)) {
            // (the trajectory outputs are written every step when they are on: their pointers stay resident -- read at their
            // use, the scalar loads' latency landed on every tile's accounting chain: FORM 1 28.6 -> 29.3 us)
            if (p.desc_src) {  // the returned observation as descriptors, 8 + 8A bytes per env (a trajectory's `states`)
                p.desc_pos[sl] = s.pos_obs;
                if (a == 0) p.desc_src[n] = l.src[e];
            }
        }
        // agent.store's `actions` field, no copy kernel (f32 actions only) -- in EVERY form: one resident pointer, and a rollout
        // loop that has the kernel write rewards / dones / its action copy into trajectory slots (what bench.py times) stays on the
        // lean forms: 28.26 instead of 28.57 us (plain) and 28.55 instead of 29.00 us (host flag) at config 2 against the round-4
        // dispatch that sent the action copy to the full forms (profiles/r05_microbench/form_ab.txt, 'leanact')
        if (p.act_store) p.act_store[sl] = (float)action;
    }
    // ---------------- phase 1b: one lane per env ----------------
    bool any = sdone;
    if constexpr (!SINGLE) lds_barrier();
    if (active && a == 0) {
        double rew;
        if constexpr (SINGLE && PROMO) {  // num_shares is an f64 tensor: the fee is an f64 product, TSE:288-289
            rew = s.rew - ((any ? 1.0 : 0.0) * (double)(s.sht + s.lng)) * p.comm;
        } else if constexpr (SINGLE) {
            float fee = ((any ? 1.0f : 0.0f) * (s.sht + s.lng)) * p.c32;  // TSE:288-289
            rew = s.rew - (double)fee;
        } else {
            any = false;
            for (int k = 0; k < A; ++k) any |= l.flg[e * A + k] != 0;
            rew = 0.0;
            for (int k = 0; k < A; ++k) {  // sleeve contract: sum in asset order
                double r;
                if constexpr (PROMO) {  // num_shares is an f64 tensor: dones * num_shares * commission in f64, TSE:288-289
                    r = l.rew[e * A + k] - ((any ? 1.0 : 0.0) * (double)l.shr[e * A + k]) * p.comm;
                } else {
                    float fee = ((any ? 1.0f : 0.0f) * l.shr[e * A + k]) * p.c32;
                    r = l.rew[e * A + k] - (double)fee;
                }
                rew = (k == 0) ? r : rew + r;
            }
            l.any[e] = any ? 1 : 0;
        }
        if (any) {
            s0 = 0;  // window rewinds to rows 0..W-1, TSE:514-521
            if (!p.evaluate && p.redraw_mode == 1 && n == p.eval_env) {  // TSE:504-513
                const typename ColdFor<FORM, SINGLE>::Ptr c = ColdFor<FORM, SINGLE>::of(p);
                unsigned long long *const counters = c->counters;
                unsigned long long ctr = counters[1];
                p.env_idx[n] = (int64_t)(((uint64_t)philox_u32(c->seed, ctr) * (uint64_t)c->D) >> 32);
                counters[1] = ctr + 1;
            }
        }
        p.spot0[n] = s0;
        if constexpr (form_is_full(FORM)) {
            if (p.evaluate) {  // TSE:523-536
                const typename ColdFor<FORM, SINGLE>::Ptr c = ColdFor<FORM, SINGLE>::of(p);
                uint8_t *const terminated = c->terminated;
                float *const ep_ret = c->ep_ret;
                const bool term = terminated[n] != 0;
                if (term) rew = 0.0;
                if (any && !term) {
                    terminated[n] = 1;
                    atomicAdd(&c->counters[0], 1ull);
                }
                ep_ret[n] = (float)((double)ep_ret[n] + rew);
            }
        }
        rew_out[n] = rew;
        done_out[n] = any ? 1 : 0;
        if constexpr (form_notifies(FORM)) {
            // fe_env_step_notify: the host polls this instead of copying dones back after the launch (TSE:510); a relaxed
            // system-scope store -- the host reads nothing else of this launch through it
            if (n == p.eval_env) {
                const typename ColdFor<FORM, SINGLE>::Ptr c = ColdFor<FORM, SINGLE>::of(p);
                __hip_atomic_store(c->host_flag, (c->flag_seq << 1) | (any ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if constexpr (form_is_full(FORM)) {
            if (p.has_stats) {  // PPO_agent.py:120-132 without its per-step host sync
                const typename ColdFor<FORM, SINGLE>::Ptr c = ColdFor<FORM, SINGLE>::of(p);
                float *const run_ret = c->run_ret;
                float cr = (float)((double)run_ret[n] + rew);
                if (any) {
                    if (n == p.eval_env) {
                        float *const stat_eval = c->stat_eval;
                        stat_eval[0] = cr;
                        stat_eval[1] += 1.0f;
                    } else {
                        // per-env partial sums: env n's slots have ONE writer (the lane that owns env n; launches are
                        // stream-ordered), so what they add up to does not depend on the tile walk, the launch geometry
                        // or the form of the kernel; fe_env_stats_reduce adds the envs up in a fixed order.  The adds
                        // are issued as no-return memory atomics only so that the old values never occupy registers
                        // (as plain read-modify-writes they cost the f32 notify form a VGPR spill).
                        double *acc = c->stat_acc + 3 * n;  // (N, 3): one address, three immediate offsets
                        atomicAdd(acc, 1.0);
                        atomicAdd(acc + 1, (double)cr);
                        atomicAdd(acc + 2, (double)cr * (double)cr);
                    }
                    cr = 0.0f;
                }
                run_ret[n] = cr;
            }
        }
    }
    if constexpr (!SINGLE) {
        lds_barrier();
        if (active) any = l.any[e] != 0;
    }
    if (active) {  // state write-back with the episodic reset folded in, TSE:498-502
        p.cash[sl] = any ? p.S32 : s.cash;
        p.lng[sl] = any ? 0.0f : s.lng;
        p.sht[sl] = any ? 0.0f : s.sht;
        p.margin[sl] = any ? 0.0 : s.margin;
    }
}

// unpipelined form: load, then account
template <bool SINGLE, int FORM, bool PROMO = false>
__device__ __forceinline__ void account_tile(const Params &p, const TileLds &l, int A, int e, int a, bool active,
                                             int64_t n, int64_t sl, ActionT<PROMO> action, double *rew_out,
                                             int32_t *done_out) {
    int64_t idx, spot;
    SleeveIn in;
    load_head(p, active, n, idx, spot);
    load_body(p, A, a, active, sl, idx, spot, in);
    account_core<SINGLE, FORM, PROMO>(p, l, A, e, a, active, n, sl, in, action, rew_out, done_out);
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
template <typename OT>
__device__ __forceinline__ void stream_load(const Params &p, const TileLds &l, int A, uint32_t tuples, uint32_t base,
                                            int lane, TupleOf<OT> (&v)[kTuplesPerIter<OT> / 64], bool skip = false) {
    constexpr int G = kTuplesPerIter<OT> / 64;
    const uint32_t WA = (uint32_t)p.W * (uint32_t)A;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) v[gi] = TupleOf<OT>{};
    if (skip) return;
    // f32 observations read a pre-cast f32 copy of the table when one is bound: half the L2 traffic,
    // same values ((float) of the f64 entry either way).  One uniform branch around the whole group of loads.
    uint32_t ee[G], r[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const uint32_t t = base + gi * 64 + lane;
        const uint32_t tc = t < tuples ? t : tuples - 1;  // tail lanes re-read the last tuple
        ee[gi] = fdiv(tc, p.div_WA);                      // env within the tile
        r[gi] = tc - ee[gi] * WA;                         // (row j, asset a) = r / A, r % A
    }
    if constexpr (sizeof(OT) == 4) {
        if (p.LR32 != nullptr) {
#pragma unroll
            for (int gi = 0; gi < G; ++gi) v[gi] = *reinterpret_cast<const float4 *>(p.LR32 + l.src[ee[gi]] + 4u * r[gi]);
        } else {
#pragma unroll
            for (int gi = 0; gi < G; ++gi) {
                const double4 d = *reinterpret_cast<const double4 *>(p.LR + l.src[ee[gi]] + 4u * r[gi]);
                v[gi] = make_float4((float)d.x, (float)d.y, (float)d.z, (float)d.w);
            }
        }
    } else {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) v[gi] = *reinterpret_cast<const double4 *>(p.LR + l.src[ee[gi]] + 4u * r[gi]);
    }
}

// Phase 2 for one tile: l.src / l.pos -> (ebt, W, 5A) observation at dst, through this wavefront's
// private LDS image.  No workgroup barrier inside.  `pre` (optional): the table tuples of this wavefront's FIRST
// iteration, loaded earlier by stream_load (the start-up chain of a workgroup's first tile).
template <typename OT, int VEC, bool SINGLE>
__device__ __forceinline__ void stream_tile(const Params &p, const TileLds &l, OT *stage, int A, int ebt, OT *dst,
                                            int lane, int wave, bool use_pre, PreTuples<OT> pre) {
    constexpr int TPI = kTuplesPerIter<OT>;  // tuples per wave iteration
    constexpr int G = TPI / 64;              // tuples per lane per iteration
    constexpr int nwaves = kBlock / 64;
    const uint32_t WA = (uint32_t)p.W * (uint32_t)A;           // 32-byte table tuples per env
    const uint32_t tuples = (uint32_t)ebt * WA;
    for (uint32_t base = wave * TPI; base < tuples; base += nwaves * TPI) {
        using TupleT = TupleOf<OT>;
        TupleT v[G];
        OT pz[G];
        stream_load<OT>(p, l, A, tuples, base, lane, v, /*skip=*/use_pre && base == (uint32_t)wave * TPI);
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
            pz[gi] = (OT)l.pos[ee * A + aa];
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            OT *w = stage + (gi * 64 + lane) * 5;
            w[0] = (OT)v[gi].x; w[1] = (OT)v[gi].y; w[2] = (OT)v[gi].z; w[3] = (OT)v[gi].w;
            w[4] = pz[gi];
        }
        // the image is private to this wavefront: order its LDS writes before the reads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t left = tuples - base;
        const uint32_t nvalid = (left < (uint32_t)TPI ? left : (uint32_t)TPI) * 5u / VEC;  // packs to store
        constexpr int kStores = TPI * 5 / VEC / 64;  // 5 full-width store instructions at 16 B/lane
        const Pack<OT, VEC> *rd = reinterpret_cast<const Pack<OT, VEC> *>(stage);
        Pack<OT, VEC> *o = reinterpret_cast<Pack<OT, VEC> *>(dst + (size_t)base * 5u);
        if constexpr (sizeof(OT) * VEC == 16) {
            // observation stores with explicit cache bits (kStoreAux*) through a buffer descriptor over this
            // wavefront's 5-KiB slab; the descriptor is wave-uniform
            using u4 = __attribute__((ext_vector_type(4))) unsigned int;
            constexpr int kAux = SINGLE ? kStoreAuxSingle : kStoreAuxMulti;
            // single-asset envs whose observation ring cannot live in the 256 MiB Infinity Cache stream past it
            // (sc1 | nt, p.obs_stream): one uniform branch around the whole store group, straight-line stores inside
            const bool stream_past_mall = SINGLE && p.obs_stream != 0;
            const uint64_t basep = reinterpret_cast<uint64_t>(o);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)basep);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(basep >> 32));
            const unsigned nb = __builtin_amdgcn_readfirstlane(nvalid * 16u);
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, nb,
                                                          0x00020000);
            // No per-lane predicate: the descriptor covers exactly the nvalid packs of this iteration (a multiple of
            // 16 bytes, like every pack offset), and the buffer range check drops the lanes past it -- straight-line
            // reads and stores, one LDS address and one store offset register instead of five each.
            const u4 *rd4 = reinterpret_cast<const u4 *>(stage) + lane;
            if (stream_past_mall) {
#pragma unroll
                for (int i = 0; i < kStores; ++i) {
                    __builtin_amdgcn_raw_buffer_store_b128(rd4[64 * i], rsrc, (uint32_t)lane * 16u + 1024u * i, 0, kStoreAuxSingleStream);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < kStores; ++i) {
                    __builtin_amdgcn_raw_buffer_store_b128(rd4[64 * i], rsrc, (uint32_t)lane * 16u + 1024u * i, 0, kAux);
                    // image read -> store, one pack at a time: with all five reads issued first the stores leave as one
                    // burst, measured 0 - 2 % slower at 64k envs (profiles/r03_microbench/ab_store_form.txt)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // observation sizes that are not a multiple of 16 bytes (odd W * A): 8- / 4-byte stores, plain cache policy;
            // a rolled loop -- unrolled, its 10 / 20 predicated stores cost more registers than the whole pipeline
#pragma unroll 1
            for (int i = 0; i < kStores; ++i) {
                const uint32_t c = (uint32_t)lane + 64u * i;
                if (c < nvalid) o[c] = rd[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // reads done before the next iteration overwrites the image
    }
}
template <typename OT, int VEC, bool SINGLE>
__device__ __forceinline__ void stream_tile(const Params &p, const TileLds &l, OT *stage, int A, int ebt, OT *dst,
                                            int lane, int wave) {
    stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt, dst, lane, wave, false, PreTuples<OT>{});
}

// The k-th tile of this workgroup, or p.num_tiles when it has none left: grid-strided (tile t -> workgroup t % grid;
// the grid is a multiple of 8, so t % 8 -- the XCD -- is stable per workgroup and envs mapped n mod D keep each XCD's
// L2 on the same days).  XCD-blocked and scrambled orders were measured slower (profiles/r02_microbench/ab_xcd_blocked.txt).
// `last_first` (fe_env_step_notify): the same walk from the other end, so that the tile holding the evaluation env --
// the last env -- is the first tile of workgroup 0.  (An ascending walk with only that tile moved to the front measures the
// same, and so does a flag word in device memory: the 0.2 us the notify forms cost is code shape, not the walk or the PCIe store;
// profiles/r05_microbench/form_ab.txt.)
__device__ __forceinline__ int64_t tile_at(const Params &p, int64_t k, bool last_first = false) {
    const int64_t t = blockIdx.x + k * (int64_t)gridDim.x;
    if (t >= p.num_tiles) return p.num_tiles;
    return last_first ? p.num_tiles - 1 - t : t;
}

// Software pipeline state of the single-asset step kernel: inputs of the current tile, prefetched inputs of the
// next one, indices of the one after.  Env numbers are recomputed from the (uniform) tile number where they are
// needed instead of being carried through phase 2.
template <bool PROMO>
struct PipeStateT {
    SleeveIn in_cur, in_nxt;
    ActionT<PROMO> action_cur, action_nxt, action_nn;  // actions run TWO tiles ahead: a caller's action buffer may be cold (a
                                                       // new trajectory slot every step costs a TLB walk + an HBM round trip, ~5 us)
    int64_t idx1, spot1, idx2, spot2;
};

// the action of sleeve sl: f32, or -- promoted launches -- a double that is either the caller's f64 action or its f32 action
// carried exactly (sleeve_step tells them apart by p.act_f64)
template <bool PROMO>
__device__ __forceinline__ ActionT<PROMO> load_action(const Params &p, int64_t sl) {
    if constexpr (PROMO) {
        return p.act_f64 ? reinterpret_cast<const double *>(p.actions)[sl] : (double)p.actions[sl];
    } else {
        return p.actions[sl];
    }
}

// env number of lane e in tile t, and whether that lane has an env there
__device__ __forceinline__ int64_t pipe_env_of(const Params &p, int EB, int e, int64_t t, bool &act) {
    const int64_t n0 = t * EB;
    const int64_t left = p.N - n0;
    act = t < p.num_tiles && (int64_t)e < (left < (int64_t)EB ? left : (int64_t)EB);
    return n0 + e;
}

// Actions are fetched two tiles ahead with f64 observations (measured on a shared ring, round 2: -2.6 % with hot
// action buffers, 41.1 -> 37.8 us/step with cold ones, tools/cold_slots.py); the f32-observation kernel runs more
// wavefronts per SIMD on fewer registers and keeps one tile ahead.
template <typename OT>
constexpr bool kActionsTwoAhead = sizeof(OT) == 8;

// One tile of the single-asset pipeline: account it (inputs already in registers), prefetch the next tile's body and
// the head of the one after, stream its observation.  FIRST: the workgroup's first tile, whose first phase-2
// iteration may use table tuples loaded before the accounting (`pre`).
template <typename OT, int VEC, bool FIRST, int FORM, bool PROMO>
__device__ __forceinline__ void single_tile(const Params &p, const TileLds &l, OT *stage, PipeStateT<PROMO> &ps, int64_t tile,
                                            int64_t k, int EB, int e, int lane, int wave, PreTuples<OT> pre) {
    const int64_t n0 = tile * EB;
    const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
    {
        bool act0;
        const int64_t n_cur = pipe_env_of(p, EB, e, tile, act0);
        account_core<true, FORM, PROMO>(p, l, 1, e, 0, act0, n_cur, n_cur, ps.in_cur, ps.action_cur, p.rew, p.done);
    }
    lds_barrier();
    // prefetch: body of the next tile (its head arrived during the previous phase 2), head of the one after
    bool act1, act2;
    constexpr bool rev = form_notifies(FORM);
    const int64_t n_nxt = pipe_env_of(p, EB, e, tile_at(p, k + 1, rev), act1);
    const int64_t n_nn = pipe_env_of(p, EB, e, tile_at(p, k + 2, rev), act2);
    load_body(p, 1, 0, act1, n_nxt, ps.idx1, ps.spot1, ps.in_nxt);
    if constexpr (!kActionsTwoAhead<OT>)
        if (act1) ps.action_nxt = load_action<PROMO>(p, n_nxt);
    load_head(p, act2, n_nn, ps.idx2, ps.spot2);
    if constexpr (kActionsTwoAhead<OT>)
        if (act2) ps.action_nn = load_action<PROMO>(p, n_nn);
    stream_tile<OT, VEC, true>(p, l, stage, 1, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave,
                               FIRST && kHoistFirst<OT>, pre);
    lds_barrier();  // LDS is reused by the next tile
    ps.in_cur = ps.in_nxt;
    ps.action_cur = ps.action_nxt;
    if constexpr (kActionsTwoAhead<OT>) ps.action_nxt = ps.action_nn;
    ps.idx1 = ps.idx2; ps.spot1 = ps.spot2;
}

// Wavefronts per SIMD the kernels are built for = workgroups per CU they are launched with (configure_launch): the
// single-asset step kernel runs 4 per CU with f64 observations (room for the hoisted first-iteration tuples) and
// FE_F32_WAVES with f32 (one less for the rare observation sizes that are not a multiple of 16 bytes: their rolled
// store loop holds a few more values); reset / render and the multi-asset kernels keep round 1's 7 and 6.
// tests/test_resource_usage.py asserts that none of them uses scratch memory.
template <typename OT, int VEC>
constexpr int kF32StepWaves = (sizeof(OT) * VEC == 16 || FE_F32_WAVES < 6) ? FE_F32_WAVES : FE_F32_WAVES - 1;
template <typename OT, int VEC, bool SINGLE, bool RESET_ONLY>
constexpr int kEnvKernelWaves = !SINGLE ? kMultiAssetWaves
                                : (RESET_ONLY ? kRenderWaves
                                              : (sizeof(OT) == 8 ? (kHoistFirst<OT> ? 4 : kRenderWaves) : kF32StepWaves<OT, VEC>));
// The body of fe_env_kernel and of fe_env_promoted_kernel (PROMO: the arithmetic of an env whose share tensors the reference
// has promoted to f64, see sleeve_step; full forms only).
template <typename OT, int VEC, bool SINGLE, bool RESET_ONLY, int FORM, bool PROMO>
__device__ __forceinline__ void env_kernel_body(const Params &p) {
    static_assert(!PROMO || (!RESET_ONLY && form_is_full(FORM)), "promoted arithmetic: full forms of the step");
    extern __shared__ __align__(16) unsigned char smem[];
    const int A = SINGLE ? 1 : p.A;
    const int EB = p.EB;
    const TileLds l = carve_lds(smem + 4 * kStageBytes, EB, EB * A);
    const int tid = threadIdx.x;
    const int e = SINGLE ? tid : (int)fdiv((uint32_t)tid, p.div_A);
    const int a = SINGLE ? 0 : tid - e * A;
    const int lane = tid & 63, wave = tid >> 6;
    OT *stage = reinterpret_cast<OT *>(smem + wave * kStageBytes);

    if constexpr (RESET_ONLY) {
        for (int64_t k = 0, tile; (tile = tile_at(p, k)) < p.num_tiles; ++k) {
            const int64_t n0 = tile * EB;
            const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
            const int64_t n = n0 + e;
            describe_tile(p, l, A, e, a, e < ebt, n, n * A + a);
            lds_barrier();
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt,
                                         reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            lds_barrier();  // LDS is reused by the next tile
        }
    } else if constexpr (!SINGLE) {
        // multi-asset tiles stream hundreds of KiB each: phase 1 is <1 % of a tile, no pipelining needed
        constexpr bool rev = form_notifies(FORM);
        for (int64_t k = 0, tile; (tile = tile_at(p, k, rev)) < p.num_tiles; ++k) {
            const int64_t n0 = tile * EB;
            const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
            const bool active = e < ebt;
            const int64_t n = n0 + e;
            const int64_t sl = n * A + a;
            ActionT<PROMO> act = 0;
            if (active) act = load_action<PROMO>(p, sl);
            account_tile<SINGLE, FORM, PROMO>(p, l, A, e, a, active, n, sl, act, p.rew, p.done);
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt,
                                         reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            lds_barrier();  // LDS is reused by the next tile
        }
    } else {
        // Software pipeline over this workgroup's tiles: while tile i streams its observation
        // (phase 2, the long part), the state + bar gathers of tile i+1 and the index loads of tile
        // i+2 are already in flight, so only the very first tile pays phase 1's two dependent
        // memory round trips.
        constexpr bool rev = form_notifies(FORM);
        int64_t tile = tile_at(p, 0, rev);
        PipeStateT<PROMO> ps;
        ps.action_cur = 0; ps.action_nxt = 0; ps.action_nn = 0;
        bool act0, act1;
        const int64_t n_cur = pipe_env_of(p, EB, e, tile, act0);
        const int64_t n_nxt = pipe_env_of(p, EB, e, tile_at(p, 1, rev), act1);
        // first tile: everything that needs no index goes out with the index loads (one round trip), only the
        // bar gather (an L2 hit) waits for them
        load_head(p, act0, n_cur, ps.idx1, ps.spot1);
        load_state(p, act0, n_cur, ps.in_cur);
        if (act0) ps.action_cur = load_action<PROMO>(p, n_cur);
        if constexpr (kActionsTwoAhead<OT>)
            if (act1) ps.action_nxt = load_action<PROMO>(p, n_nxt);  // the second tile's action leaves with the first one's
        load_bar(p, 1, 0, act0, ps.idx1, ps.spot1, ps.in_cur);
        PreTuples<OT> pre{};  // table tuples of this wavefront's first phase-2 iteration (FE_HOIST_FIRST)
        // Start-up chain of the first tile: the window descriptors need the index loads only, so they are published
        // now and every wavefront issues the table loads of its first phase-2 iteration BEFORE the accounting --
        // one L2/MALL round trip less between kernel entry and the first observation store.
        if constexpr (kHoistFirst<OT>) {
            if (act0) {
                const int64_t s0 = ps.spot1 + 1;
                const int64_t s0c = s0 + p.W <= p.L ? s0 : p.L - p.W;
                l.src[e] = (ps.idx1 * p.L + s0c) * 4;  // A == 1
            }
            lds_barrier();
            const int64_t left0 = p.N - tile * EB;
            const uint32_t tuples0 = (uint32_t)(left0 < (int64_t)EB ? left0 : (int64_t)EB) * (uint32_t)p.W;
            if ((uint32_t)wave * kTuplesPerIter<OT> < tuples0 && tile < p.num_tiles) {
                TupleOf<OT> v0[kTuplesPerIter<OT> / 64];
                stream_load<OT>(p, l, 1, tuples0, (uint32_t)wave * kTuplesPerIter<OT>, lane, v0);
                pre.v0 = v0[0];
                pre.v1 = v0[1];
                if constexpr (kTuplesPerIter<OT> / 64 == 4) {
                    pre.v2 = v0[2];
                    pre.v3 = v0[3];
                }
            }
        }
        load_head(p, act1, n_nxt, ps.idx1, ps.spot1);
        // one tile per call; the workgroup's first tile is peeled (FIRST) so that `pre` dies before the loop
        if (tile < p.num_tiles) {
            single_tile<OT, VEC, true, FORM, PROMO>(p, l, stage, ps, tile, 0, EB, e, lane, wave, pre);
            for (int64_t k = 1; (tile = tile_at(p, k, rev)) < p.num_tiles; ++k)
                single_tile<OT, VEC, false, FORM, PROMO>(p, l, stage, ps, tile, k, EB, e, lane, wave, PreTuples<OT>{});
        }
    }
    // Evaluate mode has no evaluation env; what its host reads every step is "have ALL envs terminated?" (TSE:531), a
    // fact of the whole launch.  In the notify form the LAST workgroup to finish reports the terminated-count:
    // (seq << 32) | count into the host flag -- the host polls that instead of copying the counter back.
    if constexpr (!RESET_ONLY && FORM == kFullNotify) {
        if (p.evaluate) {
            __threadfence();   // this thread's counter atomics are visible device-wide ...
            __syncthreads();   // ... and so are those of the whole workgroup
            if (tid == 0) {
                const typename ColdFor<FORM, SINGLE>::Ptr c = ColdFor<FORM, SINGLE>::of(p);
                unsigned int *const ticket = c->ticket;
                const unsigned done = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (done == gridDim.x - 1) {
                    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                    const unsigned long long cnt = __hip_atomic_load(&c->counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(c->host_flag, (c->flag_seq << 32) | (cnt & 0xffffffffull), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
}

// KERNARG LAYOUT CONTRACT (Cold<true> above): `Params` is the FIRST and ONLY explicit argument of both wrappers, so the kernarg
// segment pointer IS a `const Params *`.  A kernel with a leading argument must not instantiate FORM 3 with SINGLE (or must pass
// the block's offset to Cold).  tests/test_resource_usage.py reads the code object's metadata and fails if this stops being true.
static_assert(std::is_trivially_copyable<Params>::value && sizeof(Params) % 8 == 0,
              "Params is passed by value in the kernarg segment and re-read from there (Cold<true>)");
template <typename OT, int VEC, bool SINGLE, bool RESET_ONLY, int FORM>
__global__ __launch_bounds__(kBlock, (kEnvKernelWaves<OT, VEC, SINGLE, RESET_ONLY>)) void fe_env_kernel(const Params p) {
    env_kernel_body<OT, VEC, SINGLE, RESET_ONLY, FORM, false>(p);
}

// fe_env_step_promoted: the same kernel structure (single-asset pipeline / multi-asset tile loop) with the promoted
// arithmetic; FORM kFull or kFullNotify
template <typename OT, int VEC, bool SINGLE, int FORM>
__global__ __launch_bounds__(kBlock, (kEnvKernelWaves<OT, VEC, SINGLE, false>)) void fe_env_promoted_kernel(const Params p) {
    env_kernel_body<OT, VEC, SINGLE, false, FORM, true>(p);
}

}  // namespace
