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
// One translation unit, seven files:
//   fe_device_common.h    constants / build knobs, Params, Philox, sleeve accounting, LDS tile layout, input loads
//   fe_step_kernel.h      fe_env_kernel (the fused step and reset() rendering)
//   fe_rollout_kernels.h  K-step fused rollouts with an in-kernel policy: linear window / table form, MLP head (MFMA)
//   fe_activations.h      exact-operation sigmoid / tanh shared by the LSTM and MLP heads
//   fe_lstm_kernel.h      K-step fused rollout with the reference's LSTM actor (MFMA)
//   fe_aux_kernels.h      descriptor / render kernels, init kernels (log-returns, day tables), trajectory kernels
//   fe_env.hip            (this file) launch geometry, the env object, the C ABI of include/finenvs_amd.h
//
// Arithmetic contract: every (float)/(double) cast is a rounding point of the
// reference's mixed f32/f64 tensor arithmetic; this file
// must be compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>
#include <type_traits>

#include "finenvs_amd.h"
#include "finenvs_amd_ext.h"

#include "fe_device_common.h"
#include "fe_step_kernel.h"
#include "fe_rollout_kernels.h"
#include "fe_lstm_kernel.h"
#include "fe_aux_kernels.h"

namespace {

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
    size_t lds_promoted;  // dynamic LDS of the kernel fe_env_step_promoted dispatches to (== lds unless it takes the tile loop at A = 1)
    bool promoted_used;   // sticky: fe_env_step_promoted has been called (fe_env_launch_info then describes that kernel)
    // the observation buffer of the previous step launch: the store policy is decided per launch from how the buffers are
    // actually used (launch_env).  Relaxed: a stale value costs one launch the other policy, never correctness.
    mutable std::atomic<const void *> last_obs{nullptr};
    bool bound;
    int cus;              // compute units of that device
    int tile_override, grid_override, rollout_tile_override;  // fe_env_set_launch (tuning), 0 = automatic
    int device;           // HIP device the tables live on; every launch runs there
    double *owned_logret; // log-return table computed by fe_env_create(logret = NULL), else null
    unsigned int *ticket; // evaluate mode: 4 bytes of device memory for the notify form's last-workgroup detection
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

// Host-side preparation of kernels with more than the default 64 KiB of dynamic LDS, done once instead of per call:
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (device, function) and monotone here (only ever raised, so
// envs with different sizes cannot undercut each other), the occupancy query is cached per (device, function, block,
// LDS).  At small env counts the fused rollouts are host-bound: the two runtime calls cost as much as a launch
// (profiles/r03_microbench/host_prep_cache.txt).
struct LaunchPrep {
    int device;
    const void *kern;
    int block;
    size_t lds_max;      // largest dynamic LDS size set for (device, kern) so far
    size_t occ_lds;      // the LDS size the cached occupancy answer belongs to
    int occ_per_cu;      // 0 = not asked yet
};
static LaunchPrep g_prep[64];
static int g_nprep = 0;
static pthread_mutex_t g_prep_mu = PTHREAD_MUTEX_INITIALIZER;

// Makes `kern` launchable with `lds` bytes of dynamic LDS on `device` and, if per_cu != null, returns how many
// `block`-thread workgroups of it fit a CU.  The caller has made `device` current.
static hipError_t prepare_kernel(int device, const void *kern, int block, size_t lds, int *per_cu) {
    pthread_mutex_lock(&g_prep_mu);
    LaunchPrep *e = nullptr;
    for (int i = 0; i < g_nprep; ++i)
        if (g_prep[i].device == device && g_prep[i].kern == kern && g_prep[i].block == block) e = &g_prep[i];
    if (!e && g_nprep < (int)(sizeof(g_prep) / sizeof(g_prep[0]))) {
        e = &g_prep[g_nprep++];
        *e = LaunchPrep{device, kern, block, 0, 0, 0};
    }
    LaunchPrep local{device, kern, block, 0, 0, 0};
    if (!e) e = &local;  // table full: behave as before (per call)
    hipError_t he = hipSuccess;
    if (lds > e->lds_max) {
        he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (he == hipSuccess) e->lds_max = lds;
    }
    if (he == hipSuccess && per_cu) {
        if (e->occ_per_cu == 0 || e->occ_lds != lds) {
            int n = 0;
            he = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, block, lds);
            if (he == hipSuccess) {
                e->occ_per_cu = n < 1 ? 1 : n;
                e->occ_lds = lds;
            }
        }
        *per_cu = e->occ_per_cu;
    }
    pthread_mutex_unlock(&g_prep_mu);
    return he;
}

// The kernel instantiation a given env dispatches to (shared by launch and occupancy query).  FORM (fe_step_kernel.h):
// kFull = the launch has optional outputs (evaluate-mode bookkeeping, episode statistics, trajectory descriptors), kLean =
// none of them (the action copy of fe_env_step_traj is written by every form), kNotify = lean + the host flag of
// fe_env_step_notify; same launch bounds, same LDS.
template <bool RESET_ONLY, int FORM>
static const void *kernel_for(bool f32, int vec, bool single) {
#define FE_PICK(OT, VEC) \
    (single ? (const void *)fe_env_kernel<OT, VEC, true, RESET_ONLY, FORM> : (const void *)fe_env_kernel<OT, VEC, false, RESET_ONLY, FORM>)
    if (f32) return vec == 4 ? FE_PICK(float, 4) : (vec == 2 ? FE_PICK(float, 2) : FE_PICK(float, 1));
    return vec == 2 ? FE_PICK(double, 2) : FE_PICK(double, 1);
#undef FE_PICK
}

// fe_env_step_promoted: the step kernel with the promoted arithmetic, full forms only.  `pipelined`: the single-asset
// software pipeline (f64 observations, the reference's dtype); else the tile loop, which serves any A -- single-asset envs
// with f32 observations take it too (the pipeline's f32 instantiation has no registers to spare for f64 actions at its 6
// wavefronts per SIMD: it spilled).
template <int FORM>
static const void *promoted_kernel_for(bool f32, int vec, bool pipelined) {
#define FE_LOOP(OT, VEC) ((const void *)fe_env_promoted_kernel<OT, VEC, false, FORM>)
    if (f32) return vec == 4 ? FE_LOOP(float, 4) : (vec == 2 ? FE_LOOP(float, 2) : FE_LOOP(float, 1));
    if (pipelined)
        return vec == 2 ? (const void *)fe_env_promoted_kernel<double, 2, true, FORM> : (const void *)fe_env_promoted_kernel<double, 1, true, FORM>;
    return vec == 2 ? FE_LOOP(double, 2) : FE_LOOP(double, 1);
#undef FE_LOOP
}

// Per-call pointers go into a local copy of the parameter block: the env object itself is not
// modified by reset/step, so concurrent calls on different streams do not race on the host side.
template <bool RESET_ONLY>
static int launch_env(const fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                      hipStream_t st, int64_t *desc_src = nullptr, double *desc_pos = nullptr, float *act_store = nullptr,
                      uint64_t *host_flag = nullptr, uint64_t flag_seq = 0, int promoted = -1) {
    Params p = env->p;
    p.actions = actions;
    p.act_f64 = promoted == 1 ? 1 : 0;
    p.obs = obs;
    p.rew = rewards;
    p.done = dones;
    p.desc_src = desc_src;
    p.desc_pos = desc_pos;
    p.act_store = act_store;
    p.host_flag = reinterpret_cast<unsigned long long *>(host_flag);
    p.flag_seq = flag_seq;
    p.has_stats = p.run_ret != nullptr ? 1 : 0;
    void *args[] = {&p};
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    const bool f32 = env->cfg.obs_is_f32 != 0, single = p.A == 1;
    // (the action copy alone does not need the full form: every form writes it)
    const bool full = !RESET_ONLY && (p.evaluate || p.run_ret || desc_src);
    const void *kern = RESET_ONLY ? kernel_for<true, kLean>(f32, env->vec, single)
                       : (host_flag ? (full ? kernel_for<false, kFullNotify>(f32, env->vec, single) : kernel_for<false, kNotify>(f32, env->vec, single))
                                    : (full ? kernel_for<false, kFull>(f32, env->vec, single) : kernel_for<false, kLean>(f32, env->vec, single)));
    size_t lds = env->lds;
    if (!RESET_ONLY && promoted >= 0) {
        const bool pipelined = single && !f32;
        kern = host_flag ? promoted_kernel_for<kFullNotify>(f32, env->vec, pipelined) : promoted_kernel_for<kFull>(f32, env->vec, pipelined);
        lds = env->lds_promoted;
    }
    if (!RESET_ONLY) {
        // Store policy of a large single-asset observation (Params::obs_stream, fe_device_common.h): sc1 | nt keeps the
        // stream out of the 256 MiB Infinity Cache, which pays when the caller ALTERNATES over buffers that together
        // overflow it (a ring of two, fresh tensors per call) -- but a caller that rewrites ONE buffer of 128 - 256 MiB
        // (obs_buffers=1, a C host with one d_obs) is absorbed by the cache and runs 3 % faster with plain sc1
        // (profiles/r04_microbench/ring_alternation.txt: 27.5 vs 28.4 us).  So: stream unless this launch writes the very
        // buffer the previous one wrote.
        const void *prev = env->last_obs.exchange(obs, std::memory_order_relaxed);
        if (p.obs_stream && prev == obs && (size_t)p.N * p.env_elems * (f32 ? 4 : 8) <= (256ull << 20)) p.obs_stream = 0;
    }
    hipError_t he = hipLaunchKernel(kern, dim3(env->grid), dim3(kBlock), args, lds, st);
    if (he != hipSuccess) return hip_fail(he, RESET_ONLY ? "fe_env_reset_obs launch" : "fe_env_step launch");
    return FE_OK;
}

// Launch geometry of the step / reset kernels: tile size EB, tile count, grid, dynamic LDS.
static int configure_launch(fe_env *env) {
    const fe_config &cfg = env->cfg;
    const int A = cfg.A;
    const void *kern = kernel_for<false, kFull>(cfg.obs_is_f32 != 0, env->vec, A == 1);
    // How many workgroups the chip holds at once for this kernel variant (registers + LDS).
    int64_t cap = kBlock / A > 0 ? kBlock / A : 1;  // one sleeve per lane in phase 1
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
    if (A == 1) {
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
            wgs_per_cu = cfg.obs_is_f32 ? (env->vec == 4 ? kF32StepWaves<float, 4> : kF32StepWaves<float, 1>) : 4;
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
    if (A > 1) {
        // Multi-asset envs launch k x the resident workgroup count (k = 8, 4 or 2; every workgroup still walks >= 2 tiles, grid-strided):
        // the hardware hands the later workgroups out in order as earlier ones retire, so the launch is balanced by dispatch instead of
        // by a fixed share per persistent workgroup, whose shares of HBM differ by 10x (profiles/r06_microbench/config3_launch_size.md).
        // Measured (round 6, interleaved on one ring where it fits; table 11 there): k = 8 against k = 1 -- config 4 25.59 -> 24.00 ms
        // (6.29 -> 6.71 TB/s of observation: torch's fill rate on that buffer), config-5 shard 13.03 -> 12.59 ms, config 3 3.62 -> 3.53 ms;
        // multiples of the resident count only (4 096 / 8 192 of 1 536 resident: no gain at config 3), and not one tile per
        // workgroup (k = 64 at config 4: 2 % slower than k = 1).  The tile % 8 = XCD mapping is unchanged (k x resident is a multiple of 8).
        // Single-asset envs keep the resident grid: there every larger grid measured slower (4 - 57 %, their launch is 28 us).
        for (int k = 8; k > 1; k >>= 1) {
            if ((int64_t)k * resident <= num_tiles / 2) {
                grid = (int64_t)k * resident;
                break;
            }
        }
    }
    if (env->grid_override > 0) grid = env->grid_override;
    env->grid = (int)grid;
    env->lds = lds_bytes((int)EB, A);
    // (fe_env_step_promoted: single-asset envs with f32 observations take the tile loop, which keeps per-sleeve arrays in LDS)
    env->lds_promoted = (A == 1 && cfg.obs_is_f32 == 0) ? env->lds : lds_bytes((int)EB, A, /*per_sleeve_arrays=*/true);
    env->p.EB = (int)EB;
    env->p.num_tiles = num_tiles;
    return FE_OK;
}

extern "C" {

// shared with fe_csv.cpp: internal to the library (hidden visibility, not an exported symbol)
__attribute__((visibility("hidden"))) int fe_set_error(int code, const char *fmt, ...) {
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
    env->ticket = nullptr;
    if (cfg->evaluate) {
        if ((he = hipMalloc(&env->ticket, sizeof(unsigned int))) != hipSuccess || (he = hipMemset(env->ticket, 0, sizeof(unsigned int))) != hipSuccess) {
            if (env->ticket) (void)hipFree(env->ticket);
            delete env;
            return hip_fail(he, "fe_env_create: hipMalloc(ticket)");
        }
    }
    if (!logret) {
        // logret = NULL: compute the table from the prices (the one allocation this library owns)
        const int64_t tuples = cfg->D * cfg->L * (int64_t)cfg->A;
        if ((he = hipMalloc(&env->owned_logret, (size_t)tuples * 32)) != hipSuccess) {
            if (env->ticket) (void)hipFree(env->ticket);
            delete env;
            return hip_fail(he, "fe_env_create: hipMalloc(logret)");
        }
        hipLaunchKernelGGL(fe_logret_tables_kernel, dim3(grid_for(tuples)), dim3(kBlock), 0, (hipStream_t) nullptr,
                           prices, env->owned_logret, cfg->D, cfg->L, cfg->A);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
        if (he != hipSuccess) {
            (void)hipFree(env->owned_logret);
            if (env->ticket) (void)hipFree(env->ticket);
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
        if (env->ticket) (void)hipFree(env->ticket);
        delete env;
        return rc;
    }
    p.ticket = env->ticket;
    p.P = prices;
    p.LR = logret;
    p.N = cfg->N; p.D = cfg->D; p.L = cfg->L;
    p.W = cfg->W; p.A = A;
    p.eval_env = cfg->evaluate ? -1 : cfg->eval_env;
    p.seed = cfg->seed;
    p.evaluate = cfg->evaluate ? 1 : 0;
    p.redraw_mode = cfg->redraw_mode;
    p.env_elems = (uint32_t)env_elems;
    // One observation buffer of 128 MiB or more: a ring of two (or the allocator's recycled blocks behind fresh tensors)
    // overflows the 256 MiB Infinity Cache, and the store stream is better kept out of it (fe_device_common.h, store policy)
    p.obs_stream = cfg->A == 1 && cfg->N * env_elems * (cfg->obs_is_f32 ? 4 : 8) >= (128ll << 20) ? 1 : 0;
    p.div_WA = make_fastdiv((uint32_t)((int64_t)cfg->W * cfg->A));
    p.div_A = make_fastdiv((uint32_t)A);
    p.scale32 = (float)((double)cfg->max_shares + 0.5);
    p.scale64 = (double)cfg->max_shares + 0.5;
    p.ms64 = (double)cfg->max_shares;
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
    env->p.run_ret = running_returns;
    env->p.stat_acc = accumulators;
    env->p.stat_eval = eval_return;
    return FE_OK;
}

int fe_env_stats_reduce(fe_env *env, double *out, void *stream) {
    if (!env || !out) return fail(FE_ERR_ARG, "fe_env_stats_reduce: null argument");
    if (!env->p.stat_acc) return fail(FE_ERR_STATE, "fe_env_stats_reduce: no statistics bound (fe_env_bind_stats)");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipLaunchKernelGGL(fe_stats_reduce_kernel, dim3(1), dim3(kStatsLanes), 0, (hipStream_t)stream, env->p.stat_acc, env->p.N, out);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_stats_reduce launch");
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

int fe_env_step_notify(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                       uint64_t *host_flag, uint64_t seq, void *stream) {
    return fe_env_step_traj_notify(env, actions, obs, rewards, dones, nullptr, nullptr, nullptr, host_flag, seq, stream);
}

int fe_env_step_traj_notify(fe_env *env, const float *actions, void *obs, double *rewards, int32_t *dones,
                            float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out, uint64_t *host_flag,
                            uint64_t seq, void *stream) {
    if (!env || !actions || !obs || !rewards || !dones || !host_flag) return fail(FE_ERR_ARG, "fe_env_step_notify: null argument");
    if ((obs_src_out == nullptr) != (obs_pos_out == nullptr))
        return fail(FE_ERR_ARG, "fe_env_step_traj_notify: obs_src_out and obs_pos_out go together");
    if (actions_store_out == actions) actions_store_out = nullptr;
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_step_notify: state not bound");
    if (env->p.eval_env < 0 && !env->cfg.evaluate)
        return fail(FE_ERR_ARG, "fe_env_step_notify: this training-mode env has no evaluation env (a shard that does not own it)");
    return launch_env<false>(env, actions, obs, rewards, dones, (hipStream_t)stream, obs_src_out, obs_pos_out, actions_store_out,
                             host_flag, seq);
}

int fe_env_step_promoted(fe_env *env, const void *actions, int32_t actions_are_f64, void *obs, double *rewards,
                         int32_t *dones, float *actions_store_out, int64_t *obs_src_out, double *obs_pos_out,
                         uint64_t *host_flag, uint64_t seq, void *stream) {
    if (!env || !actions || !obs || !rewards || !dones) return fail(FE_ERR_ARG, "fe_env_step_promoted: null argument");
    if (actions_are_f64 != 0 && actions_are_f64 != 1) return fail(FE_ERR_ARG, "fe_env_step_promoted: actions_are_f64 must be 0 or 1");
    if ((obs_src_out == nullptr) != (obs_pos_out == nullptr))
        return fail(FE_ERR_ARG, "fe_env_step_promoted: obs_src_out and obs_pos_out go together");
    if (actions_store_out && actions_are_f64)
        return fail(FE_ERR_ARG, "fe_env_step_promoted: actions_store_out is an f32 copy; f64 actions have none");
    if (actions_store_out == actions) actions_store_out = nullptr;
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_step_promoted: state not bound");
    if (host_flag && env->p.eval_env < 0 && !env->cfg.evaluate)
        return fail(FE_ERR_ARG, "fe_env_step_promoted: this training-mode env has no evaluation env (a shard that does not own it)");
    env->promoted_used = true;
    return launch_env<false>(env, reinterpret_cast<const float *>(actions), obs, rewards, dones, (hipStream_t)stream, obs_src_out,
                             obs_pos_out, actions_store_out, host_flag, seq, actions_are_f64);
}

int fe_host_flag_create(uint64_t **host_flag) {
    if (!host_flag) return fail(FE_ERR_ARG, "fe_host_flag_create: null argument");
    void *ptr = nullptr;
    // mapped into the device's address space, coherent (fine-grained): a device store is visible to a polling host thread
    hipError_t he = hipHostMalloc(&ptr, sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable);
    if (he != hipSuccess) return hip_fail(he, "fe_host_flag_create: hipHostMalloc");
    *reinterpret_cast<volatile uint64_t *>(ptr) = 0;
    *host_flag = reinterpret_cast<uint64_t *>(ptr);
    return FE_OK;
}

int fe_host_flag_destroy(uint64_t *host_flag) {
    if (host_flag) (void)hipHostFree(host_flag);
    return FE_OK;
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

int fe_env_check_descriptors(fe_env *env, const int64_t *obs_src, int64_t count, int64_t *first_bad, void *stream) {
    if (!env || !obs_src || !first_bad || count < 0) return fail(FE_ERR_ARG, "fe_env_check_descriptors: bad argument");
    *first_bad = -1;
    if (count == 0) return FE_OK;
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d = nullptr, h[2] = {0ull, (unsigned long long)count};
    hipError_t he = hipMalloc(&d, sizeof(h));
    if (he != hipSuccess) return hip_fail(he, "fe_env_check_descriptors: hipMalloc");
    he = hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) {
        const Params &p = env->p;
        hipLaunchKernelGGL(fe_check_descriptors_kernel, dim3(grid_for(count)), dim3(kBlock), 0, st, obs_src, count,
                           4 * (int64_t)p.A, (int64_t)p.W, p.D * p.L, d);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    int64_t bad_value = 0;
    if (he == hipSuccess && h[0] != 0) {
        he = hipMemcpy(&bad_value, obs_src + h[1], sizeof(int64_t), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (he != hipSuccess) return hip_fail(he, "fe_env_check_descriptors");
    if (h[0] != 0) {
        *first_bad = (int64_t)h[1];
        return fail(FE_ERR_ARG, "fe_env_check_descriptors: %llu of %lld descriptors lie outside this env's log-return table "
                    "(first: obs_src[%lld] = %lld; valid: multiples of %d with offset / %d + W <= D*L = %lld, W = %d)",
                    h[0], (long long)count, (long long)h[1], (long long)bad_value, 4 * env->p.A, 4 * env->p.A,
                    (long long)(env->p.D * env->p.L), env->p.W);
    }
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
    hipError_t he = prepare_kernel(env->device, kern, block, lds, nullptr);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_mlp: hipFuncSetAttribute");
    void *args[] = {&p, &r};
    he = hipLaunchKernel(kern, dim3((unsigned)grid), dim3(block), args, lds, (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_mlp launch");
    return FE_OK;
}

// Shared by fe_env_rollout_lstm and fe_lstm_forward: geometry, kernel choice, launch.  `count` = envs (rollout) or
// descriptors (forward).
static int launch_lstm(fe_env *env, LstmArgs &r, int64_t count, const char *who, void *stream) {
    const int32_t H = r.H;
    const bool big = H == 256 || H == 512 || H == 1024;  // weights streamed from L2 (fragment-major whh)
    if (H != 32 && H != 64 && H != 128 && !big)
        return fail(FE_ERR_ARG, "%s: H must be 32, 64, 128, 256, 512 or 1024 (got %d)", who, (int)H);
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    p.N = count;
    // SP (env, asset) pairs per workgroup: 1 (H >= 256), 2 (H = 128) or 4 column tiles of 32; an env's sleeves stay together
    const int SP = big ? 32 : (H == 128 ? LstmGeom<4>::SP : LstmGeom<2>::SP);
    if (p.A > SP)
        return fail(FE_ERR_ARG, "%s: %d assets per env exceed the %d pairs of a workgroup tile (H = %d)", who, (int)p.A, SP, (int)H);
    int64_t eb = SP / p.A;
    // few envs: a tile lives on one CU for a whole step, so spread them over the CUs -- halve the tile (down to one
    // 32-pair column tile) while that fills otherwise idle CUs; a wavefront then runs fewer column tiles per time step
    const int64_t min_eb = 32 / p.A > 1 ? 32 / p.A : 1;
    while (!big && eb > min_eb && (p.N + eb - 1) / eb < env->cus) eb = eb / 2 > min_eb ? eb / 2 : min_eb;
    if (env->rollout_tile_override > 0 && env->rollout_tile_override < eb) eb = env->rollout_tile_override;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    const size_t lds = big ? lstm_big_lds_bytes(p.EB, p.A, H) : lstm_lds_bytes(p.EB, p.A, H, SP);
    const bool single = p.A == 1;
#define FE_LSTM(NT) (single ? (const void *)fe_rollout_lstm_kernel<true, NT> : (const void *)fe_rollout_lstm_kernel<false, NT>)
#define FE_LSTM_BIG(RTW) (single ? (const void *)fe_rollout_lstm_big_kernel<true, RTW> : (const void *)fe_rollout_lstm_big_kernel<false, RTW>)
    const void *kern = H == 32 ? FE_LSTM(1) : (H == 64 ? FE_LSTM(2) : (H == 128 ? FE_LSTM(4) :
                       (H == 256 ? FE_LSTM_BIG(4) : (H == 512 ? FE_LSTM_BIG(8) : FE_LSTM_BIG(16)))));
#undef FE_LSTM
#undef FE_LSTM_BIG
    int per_cu = 0;
    hipError_t he = prepare_kernel(env->device, kern, kLstmBlock, lds, &per_cu);
    if (he != hipSuccess) return hip_fail(he, "LSTM kernel: hipFuncSetAttribute / occupancy query");
    const int64_t resident = (int64_t)env->cus * per_cu;  // one pass of resident workgroups, each looping over its tiles
    const int64_t grid = p.num_tiles < resident ? p.num_tiles : resident;
    void *args[] = {&p, &r};
    he = hipLaunchKernel(kern, dim3((unsigned)grid), dim3(kLstmBlock), args, lds, (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "LSTM kernel launch");
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
    if (out_activation < 0 || out_activation > 1) return fail(FE_ERR_ARG, "fe_env_rollout_lstm: out_activation must be 0 (tanh) or 1 (clamp)");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_lstm: state not bound");
    LstmArgs r;
    r.lr32 = logret_f32; r.whh = whh; r.wx = wx; r.wout = wout; r.bout = bout; r.H = H; r.out_act = out_activation; r.K = K;
    r.obs_src = obs_src; r.obs_pos = obs_pos; r.actions_out = actions_out; r.rew_out = rewards_out; r.done_out = dones_out;
    r.noise = noise; r.std = std; r.means_out = means_out; r.traj_src = states_src_out; r.traj_pos = states_pos_out;
    r.forward_only = 0;
    return launch_lstm(env, r, env->cfg.N, "fe_env_rollout_lstm", stream);
}

int64_t fe_lstm_split_workspace_floats(int32_t H, int64_t pairs) {
    if (H < 8 || pairs < 1) return 0;
    return 3 * ((pairs + 31) / 32) * (int64_t)H * 32;  // h (two buffers) + c, fragment-major: [column tile][H/8][64][4]
}

int fe_env_rollout_lstm_split(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout,
                              float bout, int32_t H, int32_t out_activation, int32_t K, int64_t *obs_src, double *obs_pos,
                              const float *noise, float std, float *actions_out, float *means_out, double *rewards_out,
                              int32_t *dones_out, int64_t *states_src_out, double *states_pos_out, float *workspace,
                              void *stream) {
    if ((states_src_out == nullptr) != (states_pos_out == nullptr))
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: states_src_out and states_pos_out go together");
    if (noise && !(std >= 0.0f)) return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: std must be >= 0 when noise is given");
    if (!env || !logret_f32 || !whh || !wx || !wout || !obs_src || !obs_pos || !rewards_out || !dones_out || !workspace || K < 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: bad argument");
    if (H != 256 && H != 512 && H != 1024)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: H must be 256, 512 or 1024 (got %d)", (int)H);
    if (out_activation < 0 || out_activation > 1)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: out_activation must be 0 (tanh) or 1 (clamp)");
    if (!env->bound) return fail(FE_ERR_STATE, "fe_env_rollout_lstm_split: state not bound");
    DeviceGuard guard(env->device);
    if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
    Params p = env->p;
    const int64_t NA = p.N * p.A, CT = (NA + 31) / 32;
    // the accounting launch: few sleeves per workgroup, so that a handful of envs still spreads over the CUs
    int64_t eb = 16 / p.A > 1 ? 16 / p.A : 1;  // (and their h_W rows, 4 H bytes per pair, are staged in LDS)
    if (env->rollout_tile_override > 0 && env->rollout_tile_override < eb) eb = env->rollout_tile_override;
    p.EB = (int)eb;
    p.num_tiles = (p.N + eb - 1) / eb;
    const size_t lds = ((table_rollout_lds_bytes(p.EB, p.A) + 15) & ~(size_t)15) + (size_t)p.EB * p.A * H * 4;
    if (lds > 160 * 1024)
        return fail(FE_ERR_ARG, "fe_env_rollout_lstm_split: %d sleeves per env x H = %d do not fit the LDS of the accounting launch", (int)p.A, (int)H);
    {
        const void *fk = p.A == 1 ? (const void *)fe_lstm_split_finish_kernel<true> : (const void *)fe_lstm_split_finish_kernel<false>;
        hipError_t ha = prepare_kernel(env->device, fk, kBlock, lds, nullptr);
        if (ha != hipSuccess) return hip_fail(ha, "fe_env_rollout_lstm_split: hipFuncSetAttribute");
    }
    const int64_t fgrid = p.num_tiles < 8 * 256 ? p.num_tiles : 8 * 256;
    const bool single = p.A == 1;
    LstmSplitArgs s;
    s.a.lr32 = logret_f32; s.a.whh = whh; s.a.wx = wx; s.a.wout = wout; s.a.bout = bout; s.a.H = H; s.a.out_act = out_activation;
    s.a.K = 1; s.a.obs_src = obs_src; s.a.obs_pos = obs_pos; s.a.std = std; s.a.traj_src = states_src_out; s.a.traj_pos = states_pos_out;
    s.a.forward_only = 0;
    s.hbuf = workspace; s.cbuf = workspace + 2 * CT * (int64_t)H * 32; s.pairs = NA;
    const dim3 ggrid((unsigned)(H / 8), (unsigned)((CT + kBlock / 64 - 1) / (kBlock / 64)));
    const size_t glds = (size_t)(H / 8) * 64 * 16;  // one gate-row tile of weights: H / 8 KiB (128 KiB at H = 1024)
    {
        const void *gk = single ? (const void *)fe_lstm_split_gates_kernel<true> : (const void *)fe_lstm_split_gates_kernel<false>;
        hipError_t ha = prepare_kernel(env->device, gk, kBlock, glds, nullptr);
        if (ha != hipSuccess) return hip_fail(ha, "fe_env_rollout_lstm_split: hipFuncSetAttribute");
    }
    for (int k = 0; k < K; ++k) {
        s.k = k;
        s.a.noise = noise ? noise + (int64_t)k * NA : nullptr;
        s.a.actions_out = actions_out ? actions_out + (int64_t)k * NA : nullptr;
        s.a.means_out = means_out ? means_out + (int64_t)k * NA : nullptr;
        s.a.rew_out = rewards_out + (int64_t)k * p.N;
        s.a.done_out = dones_out + (int64_t)k * p.N;
        for (int t = 0; t < p.W; ++t) {
            s.t = t;
            if (single) hipLaunchKernelGGL(fe_lstm_split_gates_kernel<true>, ggrid, dim3(kBlock), glds, (hipStream_t)stream, p, s);
            else hipLaunchKernelGGL(fe_lstm_split_gates_kernel<false>, ggrid, dim3(kBlock), glds, (hipStream_t)stream, p, s);
            // a launch that fails (bad geometry, LDS) fails the first time: stop before queueing W * K launches on
            // half-written h / c state
            if (k == 0 && t == 0) {
                hipError_t hl = hipGetLastError();
                if (hl != hipSuccess) return hip_fail(hl, "fe_env_rollout_lstm_split: gates launch");
            }
        }
        if (single) hipLaunchKernelGGL(fe_lstm_split_finish_kernel<true>, dim3((unsigned)fgrid), dim3(kBlock), lds, (hipStream_t)stream, p, s);
        else hipLaunchKernelGGL(fe_lstm_split_finish_kernel<false>, dim3((unsigned)fgrid), dim3(kBlock), lds, (hipStream_t)stream, p, s);
        if (k == 0) {
            hipError_t hl = hipGetLastError();
            if (hl != hipSuccess) return hip_fail(hl, "fe_env_rollout_lstm_split: accounting launch");
        }
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_rollout_lstm_split launch");
    return FE_OK;
}

int fe_lstm_forward(fe_env *env, const float *logret_f32, const float *whh, const float *wx, const float *wout, float bout,
                    int32_t H, int32_t out_activation, const int64_t *obs_src, const double *obs_pos, int64_t count,
                    float *out, void *stream) {
    if (!env || !logret_f32 || !whh || !wx || !wout || !obs_src || !obs_pos || !out || count < 0)
        return fail(FE_ERR_ARG, "fe_lstm_forward: bad argument");
    if (out_activation < 0 || out_activation > 2)
        return fail(FE_ERR_ARG, "fe_lstm_forward: out_activation must be 0 (tanh), 1 (clamp) or 2 (none)");
    if (count == 0) return FE_OK;
    LstmArgs r;
    r.lr32 = logret_f32; r.whh = whh; r.wx = wx; r.wout = wout; r.bout = bout; r.H = H; r.out_act = out_activation; r.K = 1;
    r.obs_src = const_cast<int64_t *>(obs_src); r.obs_pos = const_cast<double *>(obs_pos);  // read only in this mode
    r.actions_out = out; r.rew_out = nullptr; r.done_out = nullptr;
    r.noise = nullptr; r.std = 0.0f; r.means_out = nullptr; r.traj_src = nullptr; r.traj_pos = nullptr;
    r.forward_only = 1;
    return launch_lstm(env, r, count, "fe_lstm_forward", stream);
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
    // a one-lane launch with the day as a kernel argument: stream-ordered behind the step that finished the episode and
    // ahead of the next one, WITHOUT a host synchronisation (round 3 copied a stack variable and had to wait for it)
    hipLaunchKernelGGL(fe_set_day_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, env->p.env_idx, env_index, day);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_env_set_day launch");
    return FE_OK;
}

int fe_env_launch_info(const fe_env *env, int32_t *grid, int32_t *block, int32_t *tile_envs, int32_t *lds) {
    if (!env) return fail(FE_ERR_ARG, "fe_env_launch_info: null env");
    if (grid) *grid = env->grid;
    if (block) *block = kBlock;
    if (tile_envs) *tile_envs = env->p.EB;
    // the kernel the NEXT step dispatches to: once the env has been stepped through fe_env_step_promoted, that one
    if (lds) *lds = (int32_t)(env->promoted_used ? env->lds_promoted : env->lds);
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
    if (env && (env->owned_logret || env->ticket)) {
        DeviceGuard guard(env->device);
        if (env->owned_logret) (void)hipFree(env->owned_logret);
        if (env->ticket) (void)hipFree(env->ticket);
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
    if (N >= (1 << 19))
        hipLaunchKernelGGL(fe_traj_returns_kernel<1>, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, rewards, dones,
                           values, last_values, T, N, (float)gamma, returns, advantages);
    else
        hipLaunchKernelGGL(fe_traj_returns_kernel<4>, dim3(grid_for(N)), dim3(kBlock), 0, (hipStream_t)stream, rewards, dones,
                           values, last_values, T, N, (float)gamma, returns, advantages);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) return hip_fail(he, "fe_traj_returns launch");
    return FE_OK;
}

}  // extern "C"
