// fe_step_kernel.h -- part of fe_env.hip (one translation unit; see the overview there): the fused step kernel fe_env_kernel: phases 1 / 1b (account_core), phase 2 (stream_tile), the single-asset software pipeline.
#pragma once
#include "fe_device_common.h"

namespace {

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

// The k-th tile of this workgroup, or p.num_tiles when it has none left.  Default: grid-strided (tile t -> workgroup
// t % grid, so t % 8 -- the XCD -- is stable per workgroup).  FE_XCD_BLOCKED (experiment builds): every XCD owns one
// contiguous eighth of the tiles, = 2: scrambled order (tools/placement_modes.hip: plain stores gain 2 - 4 % at 20 GB from
// either; the step kernel does not, profiles/r02_microbench/ab_xcd_blocked.txt).
__device__ __forceinline__ int64_t tile_at(const Params &p, int64_t k) {
#if FE_XCD_BLOCKED == 2
    {   // scrambled order: a permutation of the tiles (2654435761 is prime and larger than any tile count)
        const int64_t t = blockIdx.x + k * (int64_t)gridDim.x;
        return t < p.num_tiles ? (int64_t)(((unsigned long long)t * 2654435761ull) % (unsigned long long)p.num_tiles) : p.num_tiles;
    }
#elif FE_XCD_BLOCKED
    const int64_t G = gridDim.x;
    if ((G & 7) == 0) {
        const int64_t x = blockIdx.x & 7, j = blockIdx.x >> 3, gx = G >> 3, per = (p.num_tiles + 7) >> 3;
        const int64_t q = j + k * gx, t = x * per + q;
        return (q < per && t < p.num_tiles) ? t : p.num_tiles;
    }
#endif
    const int64_t t = blockIdx.x + k * (int64_t)gridDim.x;
    return t < p.num_tiles ? t : p.num_tiles;
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
// kHoistAll (experiment builds, FE_HOIST_ALL=1; measured 5 % SLOWER at config 2, profiles/r02_microbench/ab_xcd_blocked.txt:
// the other workgroups of the CU already hide a tile's table latency): EVERY tile's table tuples are loaded one tile ahead -- the next tile's window descriptors
// need only its head (idx, spot), which arrived during the previous phase 2, so they are published (src_next) under the
// barrier the accounting needs anyway, and each wavefront issues the next tile's table loads before it streams the
// current tile; single_tile returns them for the next call.
template <typename OT>
constexpr bool kHoistAll = kHoistFirst<OT> && FE_HOIST_ALL;

template <typename OT, int VEC, bool FIRST>
__device__ __forceinline__ PreTuples<OT> single_tile(const Params &p, const TileLds &l, int64_t *src_next, OT *stage,
                                                     PipeState &ps, int64_t tile, int64_t k, int EB, int e, int lane,
                                                     int wave, PreTuples<OT> pre) {
    const int64_t n0 = tile * EB;
    const int ebt = (p.N - n0) < (int64_t)EB ? (int)(p.N - n0) : EB;
    if constexpr (kHoistAll<OT>) {
        if (ps.act1) {  // (memory safety only, as in account_core: a well-formed state has s0 + W <= L)
            const int64_t s0 = ps.spot1 + 1;
            const int64_t s0c = s0 + p.W <= p.L ? s0 : p.L - p.W;
            src_next[e] = (ps.idx1 * p.L + s0c) * 4;  // A == 1
        }
    }
    account_core<true>(p, l, 1, e, 0, ps.act0, ps.n_cur, ps.n_cur, ps.in_cur, ps.action_cur, p.rew, p.done);
    tile_barrier();
#if FE_STAMP
    if (FIRST) ps.t_accounted = __builtin_amdgcn_s_memrealtime();
#endif
    // prefetch: body of the next tile (its head arrived during the previous phase 2), head of the one after
    load_body(p, 1, 0, ps.act1, ps.n_nxt, ps.idx1, ps.spot1, ps.in_nxt);
    if constexpr (!kActionsTwoAhead<OT>)
        if (ps.act1) ps.action_nxt = p.actions[ps.n_nxt];
    ps.n_nn = pipe_env_of(p, EB, e, tile_at(p, k + 2), ps.act2);
    load_head(p, ps.act2, ps.n_nn, ps.idx2, ps.spot2);
    if constexpr (kActionsTwoAhead<OT>)
        if (ps.act2) ps.action_nn = p.actions[ps.n_nn];
#if FE_STEP_PIN
    __builtin_amdgcn_sched_barrier(0);  // experiment: keep the prefetch loads issued BEFORE phase 2 (the scheduler may sink them)
#endif
    PreTuples<OT> pre_next{};
    if constexpr (kHoistAll<OT>) {
        const int64_t tn = tile_at(p, k + 1);
        if (tn < p.num_tiles) {
            const int64_t leftn = p.N - tn * EB;
            const uint32_t tuplesn = (uint32_t)(leftn < (int64_t)EB ? leftn : (int64_t)EB) * (uint32_t)p.W;
            if ((uint32_t)wave * kTuplesPerIter<OT> < tuplesn) {
                TileLds ln = l;
                ln.src = src_next;
                TupleOf<OT> vn[kTuplesPerIter<OT> / 64];
                stream_load<OT, true>(p, ln, 1, tuplesn, (uint32_t)wave * kTuplesPerIter<OT>, lane, vn);
                pre_next.v0 = vn[0];
                pre_next.v1 = vn[1];
                if constexpr (kTuplesPerIter<OT> / 64 == 4) {
                    pre_next.v2 = vn[2];
                    pre_next.v3 = vn[3];
                }
            }
        }
    }
    stream_tile<OT, VEC, true>(p, l, stage, 1, ebt, reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave,
                               kBlock / 64, (FIRST && kHoistFirst<OT>) || (kHoistAll<OT> && !FIRST), pre);
    tile_barrier();  // LDS is reused by the next tile
    ps.in_cur = ps.in_nxt;
    ps.action_cur = ps.action_nxt;
    if constexpr (kActionsTwoAhead<OT>) ps.action_nxt = ps.action_nn;
    ps.n_cur = ps.n_nxt; ps.act0 = ps.act1;
    ps.n_nxt = ps.n_nn; ps.act1 = ps.act2;
    ps.idx1 = ps.idx2; ps.spot1 = ps.spot2;
    return pre_next;
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
    // single asset: window descriptors of the NEXT tile (kHoistAll), behind src[EB] and pos[EB]
    int64_t *src_next = reinterpret_cast<int64_t *>(smem + 4 * kStageBytes) + 2 * EB;
    (void)src_next;
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
            tile_barrier();
            stream_tile<OT, VEC, SINGLE>(p, l, stage, A, ebt,
                                         reinterpret_cast<OT *>(p.obs) + n0 * (int64_t)p.env_elems, lane, wave);
            tile_barrier();  // LDS is reused by the next tile
        }
    } else if constexpr (!SINGLE) {
        // multi-asset tiles stream hundreds of KiB each: phase 1 is <1 % of a tile, no pipelining needed
        for (int64_t k = 0, tile; (tile = tile_at(p, k)) < p.num_tiles; ++k) {
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
        int64_t k = 0;
        int64_t tile = tile_at(p, 0);
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
        ps.n_nxt = pipe_env_of(p, EB, e, tile_at(p, 1), ps.act1);
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
            PreTuples<OT> carried = single_tile<OT, VEC, true>(p, l, src_next, stage, ps, tile, k, EB, e, lane, wave, pre);
#if FE_STAMP
            if (stamps && tid == 0) {
                stamps[blockIdx.x * 8 + 1] = ps.t_accounted;
                stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            }
#endif
            for (k = 1; (tile = tile_at(p, k)) < p.num_tiles; ++k)
                carried = single_tile<OT, VEC, false>(p, l, src_next, stage, ps, tile, k, EB, e, lane, wave, carried);
        }
#if FE_STAMP
        if (stamps && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);  // this wavefront's stores have left
            stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    }
}

}  // namespace
