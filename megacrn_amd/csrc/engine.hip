// engine.hip - host-side orchestration + C ABI of libmegacrn_hip.so  (see include/megacrn_hip.h)
//
// Data layout in HBM (all fp32):
//   activations are NODE-MAJOR planes  Z[g][n][b][c]  (g = Chebyshev plane, row r = n*B + b,
//   c < Cp = roundup(H + d_in, 4); channels [h | x | 0-pad]).  Plane g viewed as an
//   (N x B*Cp) matrix is the right operand of the propagation GEMM  S x Z[g]; the same memory
//   viewed as (N*B x G*Cp) is the left operand of the weight-pool GEMM.  Planes:
//     g = 0            x                       (identity terms of both supports share it)
//     g = 1 + s*(K-1)  S_s x                   (s = support 0/1)
//     g = 2 + s*(K-1)  2 S_s (S_s x) - x       (cheb_k = 3)
//   Per time step one Z (gate input [h | x]) and one Y (candidate input [z*h | x]) plane set is
//   kept for the backward pass, plus zr = sigmoid(gate) and hc = tanh(update).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <map>
#include <algorithm>
#include <stdlib.h>
#include <atomic>
#include <vector>

#include "../../include/megacrn_hip.h"
#include "kernels_api.h"
#include "gemm_bf16_api.h"
#include "wgrad_stream_api.h"
#include "wp_stream_api.h"
#include "ops.h"

namespace mcrn {
using namespace ext;   // host entry points of the kernel units (kernels_api.h)

GemmStats g_gemm_stats = {0, 0.0};
int g_force_cfg = -1;
int g_debug = 0;                        // ablation bits of the measurement builds (mcrn_set_debug), 0 in production
static int g_precision = MCRN_BF16X3;   // contraction arithmetic of every GEMM launch (MCRN_F32 or MCRN_BF16X3)
static bool g_prop_bf16 = false;        // MCRN_BF16: propagation / its transpose / adjacency gradient on bf16-resident operands
                                        // (gemm_bf16.h); every other contraction keeps the bf16x3 arithmetic
static bool g_x3r = false;              // ... with hi/lo operand PAIRS and three MFMAs per product (gemm_bf16.h, nterm = 3): what a bf16x3 session
                                        // runs on the large graphs since round 5 - the bf16 mode's data flow in the 1e-4 arithmetic (ModelPlan::x3r)
static char g_err[512] = "";
static int g_launches = 0;

#define CK(expr)                                                                              \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            snprintf(g_err, sizeof g_err, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,       \
                     hipGetErrorString(e__));                                                 \
            return (int)e__;                                                                  \
        }                                                                                     \
    } while (0)
#define CKI(expr)                                                                             \
    do {                                                                                      \
        int r__ = (expr);                                                                     \
        if (r__ != 0) return r__;                                                             \
    } while (0)
#define FAIL(...)                                                                             \
    do {                                                                                      \
        snprintf(g_err, sizeof g_err, __VA_ARGS__);                                           \
        return MCRN_EINVAL;                                                                   \
    } while (0)
#define LAUNCH(kern, grid, blk, shm, st, ...)                                                 \
    do {                                                                                      \
        (void)hipGetLastError(); /* drop stale sticky errors of other runtime users (e.g. torch) */ \
        hipLaunchKernelGGL(kern, grid, blk, shm, st, __VA_ARGS__);                            \
        ++g_launches;                                                                         \
        CK(hipGetLastError());                                                                \
    } while (0)

// ---- process-global state guard ------------------------------------------------------------------
// The library keeps ONE helper stream, tile cache, profiler and arithmetic mode per process:
// its launch model is one process per GPU with one host thread driving it.  A second host thread inside the
// library at the same time, or a call on a second device, would silently share that state - both are refused.
static std::atomic<int> g_busy{0};
static int g_device = -1;
struct CallGuard {
    bool entered = false;
    int enter() {
        if (g_busy.exchange(1, std::memory_order_acquire) != 0) {
            // (g_err is shared too: the message is best-effort, the return code is what counts)
            snprintf(g_err, sizeof g_err, "libmegacrn_hip: concurrent call from a second host thread (one host thread per process)");
            return MCRN_EINVAL;
        }
        entered = true;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof g_err, "hipGetDevice failed"); return MCRN_EINVAL; }
        if (g_device < 0) g_device = dev;
        else if (dev != g_device) {
            snprintf(g_err, sizeof g_err, "libmegacrn_hip: called on device %d after device %d (one device per process: the tile cache, "
                     "helper stream and tile cache belong to the first device)", dev, g_device);
            return MCRN_EINVAL;
        }
        return 0;
    }
    ~CallGuard() { if (entered) g_busy.store(0, std::memory_order_release); }
};
#define ENTER()                                                                               \
    CallGuard guard__;                                                                        \
    do { int r__ = guard__.enter(); if (r__) return r__; } while (0)

// ---- live profiler: HIP events around every launch of one role (bench.py roofline leg) ---------
struct Prof {
    int role = -1;
    int n = 0;
    double alg_flops = 0, exec_flops = 0;
    static const int MAXEV = 8192;
    hipEvent_t ev[2 * MAXEV];
    bool created = false;
    unsigned long long* clk = nullptr;      // device, [MAXEV][4]: in-kernel clock stamps of the profiled bf16-resident products (Bf16GemmP::clk)
    int nclk = 0;
    double last_mhz = 0; long long last_nclk = 0;
};
static Prof g_prof;

// ---- launch histogram: which kernel FAMILY every planned launch site chose --------------------------------------
// The plan (plan_model, the *_ok predicates) picks a family by SHAPE; a test case that names a family in its `why` string can ask
// which ones really launched (mcrn_launch_histogram): tests/test_gpu_parity.py::test_planned_kernel_families_launch fails when a
// plan edit silently moves a case off the path it was written for.  Host-side counters only (a pointer compare per launch).
static const int NFAM_MAX = 96;
static const char* g_fam_name[NFAM_MAX];
static long long g_fam_n[NFAM_MAX];
static int g_nfam = 0;
static inline void fam(const char* name) {
    for (int i = 0; i < g_nfam; ++i)
        if (g_fam_name[i] == name || !strcmp(g_fam_name[i], name)) { ++g_fam_n[i]; return; }
    if (g_nfam < NFAM_MAX) { g_fam_name[g_nfam] = name; g_fam_n[g_nfam++] = 1; }
}
static const char* const kFamTiledX3[8] = {"tiled_x3:misc", "tiled_x3:prop", "tiled_x3:wp", "tiled_x3:dgrad", "tiled_x3:propT", "tiled_x3:ds", "tiled_x3:wgrad", "tiled_x3:?"};
static const char* const kFamTiledF32[8] = {"tiled_f32:misc", "tiled_f32:prop", "tiled_f32:wp", "tiled_f32:dgrad", "tiled_f32:propT", "tiled_f32:ds", "tiled_f32:wgrad", "tiled_f32:?"};
static const char* const kFamBf16[8] = {"bf16_gemm:misc", "bf16_gemm:prop", "bf16_gemm:wp", "bf16_gemm:dgrad", "bf16_gemm:propT", "bf16_gemm:ds", "bf16_gemm:wgrad", "bf16_gemm:prop_in"};
static const char* const kFamBf16HiLo[8] = {"bf16_gemm_hilo:misc", "bf16_gemm_hilo:prop", "bf16_gemm_hilo:wp", "bf16_gemm_hilo:dgrad", "bf16_gemm_hilo:propT", "bf16_gemm_hilo:ds", "bf16_gemm_hilo:wgrad", "bf16_gemm_hilo:prop_in"};

// ---- tile-configuration autotuner --------------------------------------------------------------
// The analytic cost model in choose_cfg() is only a prior: on shapes this small the best tile is
// decided by latency / occupancy effects it cannot see (measured spread 2x).  In tuning mode
// (mcrn_model_autotune) every distinct GEMM signature is timed once per configuration with HIP
// events on scratch data and the winner is cached; later launches look it up.
struct TuneKey {
    int role, M, N, K, nbatch, akc, bkc, prec, split;
    bool operator<(const TuneKey& o) const {
        return memcmp(this, &o, sizeof(TuneKey)) < 0;
    }
};
static std::map<TuneKey, int> g_tuned;
static bool g_tuning = false;
static hipEvent_t g_tune_ev[2];
static bool g_tune_ev_ok = false;

static inline int launch_any(GemmP& p, bool akc, bool bkc, int max_split, int role, hipStream_t st) {
    if (g_precision == MCRN_BF16X3) CK(launch_gemm_x3(p, akc, bkc, max_split, role, st));
    else CK(launch_gemm_f32(p, akc, bkc, max_split, st));
    return 0;
}

static inline int gemm(GemmP& p, bool akc, bool bkc, int max_split, int role, hipStream_t st) {
    if (!p.nbatch) p.nbatch = 1;
    for (int b = 0; b < 2; ++b) {   // per-batch hi strides default to the Dim2 value
        if (!p.ak_hi[b]) p.ak_hi[b] = p.ak.hi;
        if (!p.bk_hi[b]) p.bk_hi[b] = p.bk.hi;
    }
    ++g_launches;
    fam((g_precision == MCRN_BF16X3 ? kFamTiledX3 : kFamTiledF32)[role & 7]);
    const TuneKey key{role, p.M, p.N, p.K, p.nbatch, (int)akc, (int)bkc, g_precision, max_split};
    const int saved_force = g_force_cfg;
    if (g_force_cfg < 0) {
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            g_force_cfg = it->second;
        } else if (g_tuning) {
            if (!g_tune_ev_ok) {
                CK(hipEventCreate(&g_tune_ev[0]));
                CK(hipEventCreate(&g_tune_ev[1]));
                g_tune_ev_ok = true;
            }
            int best = -1;
            float best_ms = 1e30f;
            float cfg_ms[NCFG];
            for (int c = 0; c < NCFG; ++c) cfg_ms[c] = 1e30f;
            for (int round = 0; round < 3; ++round) {          // min over 3 rounds of 6 launches: robust to noise
                for (int c = 0; c < NCFG; ++c) {
                    g_force_cfg = c;
                    CKI(launch_any(p, akc, bkc, max_split, role, st));            // warm-up
                    CK(hipEventRecord(g_tune_ev[0], st));
                    for (int r = 0; r < 6; ++r) CKI(launch_any(p, akc, bkc, max_split, role, st));
                    CK(hipEventRecord(g_tune_ev[1], st));
                    CK(hipEventSynchronize(g_tune_ev[1]));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, g_tune_ev[0], g_tune_ev[1]));
                    if (ms < cfg_ms[c]) cfg_ms[c] = ms;
                }
            }
            for (int c = 0; c < NCFG; ++c)
                if (cfg_ms[c] < best_ms) { best_ms = cfg_ms[c]; best = c; }
            g_tuned[key] = best;
            g_force_cfg = best;
        }
    }
    const bool prof = g_prof.role == role && g_prof.n < Prof::MAXEV;
    if (prof) CK(hipEventRecord(g_prof.ev[2 * g_prof.n], st));
    const int rc_ = launch_any(p, akc, bkc, max_split, role, st);
    g_force_cfg = saved_force;
    if (rc_) return rc_;
    if (prof) {
        CK(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st));
        const double ex = 2.0 * p.M * p.N * (double)p.K * p.nbatch;
        g_prof.exec_flops += ex;
        g_prof.alg_flops += p.alg_flops > 0 ? p.alg_flops : ex;
        ++g_prof.n;
    }
    return 0;
}

// adjacency-stationary propagation (prop_small.h) with the same profiling hooks as gemm()
#define MCRN_PROF_WRAP(ROLE, LAUNCH, EXEC, ALG)                                                \
    do {                                                                                      \
        ++g_launches;                                                                         \
        const bool prof__ = g_prof.role == (ROLE) && g_prof.n < Prof::MAXEV;                  \
        if (prof__) CK(hipEventRecord(g_prof.ev[2 * g_prof.n], st));                          \
        CK(LAUNCH);                                                                           \
        if (prof__) {                                                                         \
            CK(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st));                              \
            g_prof.exec_flops += (EXEC);                                                      \
            g_prof.alg_flops += (ALG);                                                        \
            ++g_prof.n;                                                                       \
        }                                                                                     \
    } while (0)
// the same for launches whose parameter block Q carries ev0 / ev1: the two events are attached to the dispatch itself
// (hipExtLaunchKernelGGL), so their elapsed time is the kernel's own begin -> end, as rocprofv3 reports it
#define MCRN_PROF_WRAP_EXT(ROLE, Q, LAUNCH, EXEC, ALG)                                         \
    do {                                                                                      \
        ++g_launches;                                                                         \
        const bool prof__ = g_prof.role == (ROLE) && g_prof.n < Prof::MAXEV;                  \
        if (prof__) { (Q).ev0 = g_prof.ev[2 * g_prof.n]; (Q).ev1 = g_prof.ev[2 * g_prof.n + 1]; } \
        CK(LAUNCH);                                                                           \
        if (prof__) {                                                                         \
            g_prof.exec_flops += (EXEC);                                                      \
            g_prof.alg_flops += (ALG);                                                        \
            ++g_prof.n;                                                                       \
        }                                                                                     \
    } while (0)
static inline int prop_small(const PropP& p, int nbatch, int role, double alg, hipStream_t st) {
    const double ex = 2.0 * p.N * (double)p.N * p.ncols * nbatch * p.nseg;
    fam(role == ROLE_PROPT ? "prop_small:bwd" : "prop_small:fwd");
    MCRN_PROF_WRAP(role, launch_prop_small(p, nbatch, st), ex, alg > 0 ? alg : ex);
    return 0;
}
// ---- helper stream for work that is off the critical path (adjacency-gradient GEMMs) -------------
// Fork/join with events inside one entry-point call: the caller's stream stays the only stream the
// caller has to reason about (every side launch is joined back before the call returns its last kernel).
struct Side {
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr;                   // second helper queue: the decoder's deferred weight gradients (round 4), so that the
    hipEvent_t fork2, join2;                     // encoder's adjacency-gradient launches are not queued behind ~0.5 ms of them
    bool any2 = false;
    static const int NSLOT = 48;                 // plane sets: (update, gate) x the cells of the BPTT loops in flight (ModelPlan::MAXPAIR pairs)
    hipEvent_t ready[NSLOT], done[NSLOT], join, fork, mid;
    bool ok = false, pending[NSLOT] = {}, any = false;
    bool paired[NSLOT] = {};                     // done[buf] and done[buf ^ 1] mark the SAME launch (merged per cell)
};
static Side g_side;
static bool g_use_side = true;
static int side_init() {
    if (g_side.ok) return 0;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&g_side.st, hipStreamNonBlocking, lo));     // lowest priority: fills idle CUs (same priority measured: no change)
    for (int i = 0; i < Side::NSLOT; ++i) {
        CK(hipEventCreateWithFlags(&g_side.ready[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&g_side.done[i], hipEventDisableTiming));
    }
    CK(hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&g_side.mid, hipEventDisableTiming));
    CK(hipStreamCreateWithPriority(&g_side.st2, hipStreamNonBlocking, lo));
    CK(hipEventCreateWithFlags(&g_side.fork2, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&g_side.join2, hipEventDisableTiming));
    g_side.ok = true;
    return 0;
}
// main stream must not overwrite scratch plane set `buf` while a side GEMM still reads it
static int side_guard(int buf, hipStream_t st) {
    if (g_side.ok && g_side.pending[buf]) {
        CK(hipStreamWaitEvent(st, g_side.done[buf], 0));
        g_side.pending[buf] = false;
        // one merged launch read both plane sets of its cell: its completion frees the sibling set as well, and every
        // event wait is a barrier packet of ~6 us in the caller's queue (profiles/r3/experiments.md)
        if (g_side.paired[buf]) g_side.pending[buf ^ 1] = false;
    }
    return 0;
}
static int side_join(hipStream_t st) {
    if (g_side.ok && g_side.any2) {
        CK(hipEventRecord(g_side.join2, g_side.st2));
        CK(hipStreamWaitEvent(st, g_side.join2, 0));
        g_side.any2 = false;
    }
    if (g_side.ok && g_side.any) {
        CK(hipEventRecord(g_side.join, g_side.st));
        CK(hipStreamWaitEvent(st, g_side.join, 0));
        g_side.any = false;
        for (int i = 0; i < Side::NSLOT; ++i) g_side.pending[i] = false;
    }
    return 0;
}

// error paths: forget pending fork/join state (the caller's next call starts clean)
static void side_reset() {
    g_side.any = false; g_side.any2 = false;
    for (int i = 0; i < Side::NSLOT; ++i) g_side.pending[i] = false;
}
// The library keeps ONE process-wide arithmetic mode, helper stream and tile cache: one device and one host
// thread per process (the launch model of bench.py / megacrn_amd.train: one process per GPU).
// a bf16x3 session planned onto the bf16-resident data flow (ModelPlan::x3r): the product sites switch on g_prop_bf16 / g_x3r
struct X3rScope {
    bool saved_bf, saved_x3, on;
    explicit X3rScope(bool enable) : saved_bf(g_prop_bf16), saved_x3(g_x3r), on(enable) { if (on) { g_prop_bf16 = true; g_x3r = true; } }
    ~X3rScope() { if (on) { g_prop_bf16 = saved_bf; g_x3r = saved_x3; } }
};
struct PrecisionScope {
    int saved; bool saved_bf;
    explicit PrecisionScope(int p) : saved(g_precision), saved_bf(g_prop_bf16) {
        g_prop_bf16 = p == MCRN_BF16;
        g_precision = p == MCRN_BF16 ? MCRN_BF16X3 : p;
    }
    ~PrecisionScope() { g_precision = saved; g_prop_bf16 = saved_bf; }
};

// ---- bump allocator over the caller's workspace -------------------------------------------
struct Bump {
    char* base;
    size_t off;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : (T*)nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct Shp {   // one AGCN / cell geometry
    int B, N, d, H, C, Cp, K, G;
    long long R, ld, PS, ZT;   // rows, plane row stride (per node), plane size, plane-set size
    int Kp; long long ldp, PSb;   // MCRN_BF16: bf16 planes are [Kp = roundup(N, 64)][ldp = roundup(ld, 64)], zero padded
    bool hoist; long long ldh;    // MCRN_BF16, H % 32 == 0: the per-step propagation covers the B*H state columns only (ldh);
                                  // the input channels of every step are propagated once per stack
    long long PSbh;               // Kp * ldh: one packed bf16 [Kp][B*H] matrix
    bool lite;                    // ... and it writes bf16-RESIDENT planes [nb][N*B][H] that the streaming weight pool
                                  // (wp_stream.h) and the weight gradient read directly: no fp32 plane round trip
    long long lo_x0b, lo_x0c, lo_dPb, lo_dPin;   // x3r: element offsets of the lo images behind this stack's bf16 operands (0 otherwise)
    bool hoist_fwd;               // bf16x3, fused two-hop path, H % 64 == 0 (decoder of the small graphs): forward steps whose input channels
                                  // are known before the stack starts propagate the B*H state columns only (128 units = one round of 256
                                  // workgroups at METR-LA instead of 88 units of 96 columns: 22.4 -> 17.7 us); their input / pad channels
                                  // are propagated once per stack (hoist_inputs_small).  state_only: THIS call runs that way.
    bool state_only;
    bool hoist_bwd;               // ... and the backward cells of such steps run their transposed chain on the state columns only
};
static Shp mk_shape(int B, int N, int d, int H, int K, bool bf16_rows = false) {
    Shp s;
    s.B = B; s.N = N; s.d = d; s.H = H; s.K = K;
    s.C = d + H;
    s.Cp = (s.C + 3) & ~3;
    if (bf16_rows && ((long long)B * s.Cp) % 8) s.Cp = (s.C + 7) & ~7;   // bf16 plane rows are fetched in 16-byte chunks
    s.G = 2 * K - 1;
    s.R = (long long)N * B;
    s.ld = (long long)B * s.Cp;
    s.PS = s.R * s.Cp;
    s.ZT = s.PS * s.G;
    s.Kp = (N + 63) & ~63; s.ldp = (s.ld + 63) & ~63LL; s.PSb = (long long)s.Kp * s.ldp;
    s.hoist = bf16_rows && (H % 32) == 0 && d > 0;
    s.ldh = (long long)B * H;
    s.PSbh = (long long)s.Kp * s.ldh;
    s.hoist_fwd = false; s.state_only = false; s.hoist_bwd = false;
    s.lo_x0b = s.lo_x0c = s.lo_dPb = s.lo_dPin = 0;
    s.lite = s.hoist && wp_stream_ok(H, d, 2 * (K - 1), H) && wp_stream_ok(H, d, 2 * (K - 1), 2 * H);
    return s;
}

struct Sup {   // the two supports, their transposes, and the slabbed gradient accumulators
    const float* S[2];
    const float* St[2];
    const uint4* Sf[2];     // fragment-ordered bf16 hi/lo split of S (prop_small.h), or nullptr
    const uint4* Stf[2];    // same for S^T
    const uint4* Simg[2];   // tile-ordered pre-split images of S / S^T for the tiled GEMM (N > 256), or nullptr
    const uint4* Stimg[2];
    int simg_n;
    long long ldS;
    float* dS;      // [2][nslab][N*ldS]
    int nslab;
    long long slab;
    long long sup_stride;   // floats between the slab sets of support 0 and 1
    bool ds4 = false;   // fused model path, cheb_k = 3 (round 5): the adjacency gradient is four blocks [d1_a, e2_a, d1_b, e2_b] x x0^T over the
                        // RAW d-grad planes (dS then holds 4 blocks per slab: [nslab][4][N*ldS], slab = 4*N*ldS), the chain rule of 2 S S runs
                        // once per step (model_backward), and the transposed chain never writes d1t back
    // MCRN_BF16 (gemm_bf16.h): stacked bf16 adjacency [S1; T2(S1); S2; T2(S2)] (rows padded to Kp) and its transpose
    const uint16_t* Sstk = nullptr;
    const uint16_t* STstk = nullptr;
    int Kp = 0, nb = 0;
    float *mu = nullptr, *mu_part = nullptr;    // column sums of a plane over its nodes (+ partials)
    long long lo_Sstk = 0, lo_STstk = 0, lo_xin_b = 0;   // x3r: offsets of the lo images behind Sstk / STstk / the hoisted input operand
};
static int nslab_S(int N) {
    // N <= 256: one ds_small workgroup per slab; its slab read + write is ~9 us whatever its K range, so fewer,
    // fatter workgroups cost less GPU time (measured METR-LA: 16 -> 7070, 24 -> 8090, 32 -> 8260, 48 -> 8150, 64 -> 8060 samples/s)
    // (256 < N <= 512, ds_wide_kernel at N <= 352: 12 -> 5 783, 16 -> 6 210, 24 -> 6 381, 32 -> 6 383, 48 -> 6 221 samples/s at PEMS-BAY)
    return N <= 512 ? 32 : (N <= 1024 ? 16 : (N <= 2048 ? 4 : 1));
}
// four-block form (Sup::ds4): 4 x nslab workgroups per launch, each with one K segment per call instead of two.  Fewer, longer workgroups
// leave CUs to the main queue (the launch runs on the helper stream) until the helper stream becomes the longer chain: same call,
// METR-LA 12 -> 11 521, 16 -> 11 423, 20 -> 11 332, 32 -> 11 108 samples/s; PEMS-BAY 12 -> 6 670, 16 -> 6 762, 20 -> 6 676, 32 -> 6 603
// (profiles/r5/experiments.md section 7)
static int nslab_S4(int N) { return N <= 256 ? 12 : 16; }
static const int NSLAB_W = 256;   // slab capacity of the deferred weight gradients (reduced by k_wunprep)
static const int NSLAB_W_GEMM = 64;   // ... slabs the tiled-GEMM fallback splits K into
static const int NSLAB_T = 256;   // split-K slabs of the tiny-output, very-long-K products (dWq, dMem, dWp): one tile, so K must fill the chip
static const int NSLAB_E = 16;    // split-K slabs of dE1 / dE2 (N x D outputs, K = N)

static const int MCRN_MAX_CHEB_K = 8;     // model/MegaCRN.py:21-22 recurses for any cheb_k >= 2; 2 and 3 have the fused fast paths
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool use_prop_small(const Sup& u, const Shp& s) {
    return g_precision == MCRN_BF16X3 && s.K <= 3 && u.Sf[0] && u.Stf[0] && prop_small_ok(s.N, s.ld, (int)s.ld);
}
// fused two-hop kernels (cheb_k = 3): also 256 < N <= 352, where the other adjacency-stationary kernels do not reach
static inline bool use_prop2(const Sup& u, const Shp& s) {
    return g_precision == MCRN_BF16X3 && u.Sf[0] && u.Stf[0] && s.K == 3 && prop2_ok(s.N, s.ld, (int)s.ld);
}

static GemmP gp() {
    GemmP p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f;
    p.nbatch = 1;
    p.nsplit = 1;
    return p;
}

// ---- bf16-resident GEMM (gemm_bf16.h) with the library's profiling hooks and an on-device tile choice -------
struct Bf16Key {
    int btr, M, N, K, nsplit, role, nterm;     // (nterm: a key word of its own since round 6 - folded into K, a 3k-deep plain product aliased a k-deep hi/lo one)
    bool operator<(const Bf16Key& o) const { return memcmp(this, &o, sizeof(Bf16Key)) < 0; }
};
static std::map<Bf16Key, int> g_tuned_bf16;
static float g_last_tune_ms = 0.f;                     // best time of the signature bf16_gemm tuned last
// Tuning under the cache conditions of the model: between two launches of the same product a step streams 100+ MB
// through the L2s, so its operands come from the memory side - back-to-back trial launches would rank the tiles by their
// L2-hot time (measured at N = 1843: 46 us hot vs 54 us in the step for the same tile).  mcrn_model_autotune provides a
// scratch buffer; every timed trial launch is preceded by a fill of it.
static char* g_flush = nullptr;
static size_t g_flush_bytes = 0;
static int g_force_cfg_bf16 = getenv("MCRN_BF16_CFG") ? atoi(getenv("MCRN_BF16_CFG")) : -1;
static int bf16_cfg_prior(const Bf16GemmP& p, int nsplit) {
    // cost ~ (rounds over the CUs at this tile's residency) x (tile work) / (measured efficiency of the tile shape)
    static const double eff[NCFG_BF16] = {0.65, 0.8, 0.85, 0.9, 1.0, 0.85, 0.8, 0.65, 0.9, 0.8, 0.8, 0.8, 0, 0.85, 0.85, 0.85};
    int best = 0; double bt = 1e300;
    for (int c = 0; c < NCFG_BF16; ++c) {
        if (!bf16_cfg_tuned(c, p.nterm == 3)) continue;
        const int per_cu = kCfgBf16[c][2];
        const long long tiles = (long long)cdiv(p.M, kCfgBf16[c][0]) * cdiv(p.N, kCfgBf16[c][1]) * nsplit;
        const double rounds = ceil((double)tiles / (256.0 * per_cu));
        const double t = rounds * per_cu * kCfgBf16[c][0] * kCfgBf16[c][1] / eff[c];
        if (t < bt) { bt = t; best = c; }
    }
    return best;
}
static int bf16_gemm(Bf16GemmP& p, bool btr, int nsplit, int role, double alg, hipStream_t st, int prof_role = -1) {
    ++g_launches;
    if (nsplit < 1) nsplit = 1;
    fam((p.nterm == 3 ? kFamBf16HiLo : kFamBf16)[(prof_role >= 0 ? prof_role : role) & 7]);
    const int nterm_ = p.nterm > 1 ? p.nterm : 1;
    const bool x3 = p.nterm == 3;
    const Bf16Key key{(int)btr, p.M, p.N, p.nseg * p.seg_len, nsplit, role, nterm_};
    int cfg = g_force_cfg_bf16;
    if (x3 && cfg >= 0 && cfg < NCFG_BF16 && !bf16_cfg_is_sk(cfg) && !bf16_cfg_ok(cfg, true)) cfg = -1;   // (a forced tile the hi/lo form lacks: tuned as usual)
    if (cfg < 0 || cfg >= NCFG_BF16) {
        auto it = g_tuned_bf16.find(key);
        if (it != g_tuned_bf16.end()) cfg = it->second;
        else if (g_tuning) {
            if (!g_tune_ev_ok) {
                CK(hipEventCreate(&g_tune_ev[0]));
                CK(hipEventCreate(&g_tune_ev[1]));
                g_tune_ev_ok = true;
            }
            float best_ms = 1e30f, cfg_ms[NCFG_BF16];
            for (int c = 0; c < NCFG_BF16; ++c) cfg_ms[c] = 1e30f;
            for (int round = 0; round < 3; ++round)              // min over 3 rounds of 5 launches: robust to clock / cache noise
                for (int c = 0; c < NCFG_BF16; ++c) {
                    if (!bf16_cfg_tuned(c, x3)) continue;
                    CK(launch_gemm_bf16(p, btr, c, nsplit, role, st));                   // warm-up
                    float ms = 0;
                    if (g_flush) {
                        for (int r = 0; r < 5; ++r) {
                            CK(hipMemsetAsync(g_flush, r, g_flush_bytes, st));           // operands leave the L2s, as in a step
                            CK(hipEventRecord(g_tune_ev[0], st));
                            CK(launch_gemm_bf16(p, btr, c, nsplit, role, st));
                            CK(hipEventRecord(g_tune_ev[1], st));
                            CK(hipEventSynchronize(g_tune_ev[1]));
                            float m1 = 0;
                            CK(hipEventElapsedTime(&m1, g_tune_ev[0], g_tune_ev[1]));
                            ms += m1;
                        }
                    } else {
                        CK(hipEventRecord(g_tune_ev[0], st));
                        for (int r = 0; r < 5; ++r) CK(launch_gemm_bf16(p, btr, c, nsplit, role, st));
                        CK(hipEventRecord(g_tune_ev[1], st));
                        CK(hipEventSynchronize(g_tune_ev[1]));
                        CK(hipEventElapsedTime(&ms, g_tune_ev[0], g_tune_ev[1]));
                    }
                    if (ms < cfg_ms[c]) cfg_ms[c] = ms;
                }
            for (int c = 0; c < NCFG_BF16; ++c)
                if (cfg_ms[c] < best_ms) { best_ms = cfg_ms[c]; cfg = c; }
            g_last_tune_ms = best_ms;
            if (getenv("MCRN_TUNE_LOG")) {
                fprintf(stderr, "[mcrn tune] bf16%s %s role %d M=%d N=%d K=%dx%d split=%d:", x3 ? " hi/lo" : "", btr ? "nn" : "nt", role, p.M, p.N, p.nseg,
                        p.seg_len, nsplit);
                for (int c = 0; c < NCFG_BF16; ++c) if (bf16_cfg_tuned(c, x3)) fprintf(stderr, " %d:%.1fus", c, 1e3f * cfg_ms[c] / 5.f);
                fprintf(stderr, " -> %d\n", cfg);
            }
            g_tuned_bf16[key] = cfg;
        } else cfg = bf16_cfg_prior(p, nsplit);
    }
    const bool prof = g_prof.role == (prof_role >= 0 ? prof_role : role) && g_prof.n < Prof::MAXEV;
    const bool ext = prof;                                  // events attached to the dispatch
    if (ext) { p.ev0 = g_prof.ev[2 * g_prof.n]; p.ev1 = g_prof.ev[2 * g_prof.n + 1]; }
    if (prof && g_prof.clk && g_prof.nclk < Prof::MAXEV) p.clk = g_prof.clk + 4 * (size_t)g_prof.nclk++;
    else if (prof) CK(hipEventRecord(g_prof.ev[2 * g_prof.n], st));
    CK(launch_gemm_bf16(p, btr, cfg, nsplit, role, st));
    p.ev0 = p.ev1 = nullptr; p.clk = nullptr;
    if (prof) {
        if (!ext) CK(hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st));
        const double ex = 2.0 * p.M * (double)p.N * (double)p.nseg * p.seg_len * nterm_;
        g_prof.exec_flops += ex;
        g_prof.alg_flops += alg > 0 ? alg : ex;
        ++g_prof.n;
    }
    return 0;
}
static Bf16GemmP bgp(const Sup& u) {
    Bf16GemmP p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.nsplit = 1; p.xcd = 1;
    return p;
}
// hi/lo operand pairs of a bf16x3 session on the resident path (g_x3r): A_hi B_hi + A_hi B_lo + A_lo B_hi in one K loop
static inline void x3_terms(Bf16GemmP& p, long long a_lo, long long b_lo) {
    if (g_x3r) { p.nterm = 3; p.a_lo = a_lo; p.b_lo = b_lo; }
}
// plane (N x ld fp32) -> bf16 copy (propagation operand) and node-centred bf16 copy (adjacency-gradient operand)
static int plane_to_bf16(const Shp& s, const Sup& u, const float* X, uint16_t* xb, uint16_t* xc, hipStream_t st);
static int planes_to_bf16(const Shp& s, const float* X, int np, uint16_t* xb, uint16_t* xc, const float* mu, hipStream_t st,
                          int nvalid, long long src_ps, long long dst_ps, long long mu_stride, long long lo = 0);

// MCRN_BF16 forward propagation: ALL Chebyshev terms of both supports as ONE product
//   [S1; T2(S1); S2; T2(S2)] (nb*N x N, bf16)  x  plane 0 (N x B*Cp, bf16)  ->  planes 1 .. nb (fp32)
// T2(S) = 2 S S - I is the reference's own matrix form (model/MegaCRN.py:20-22), built once per step.
struct PackX {                    // optional arguments of k_pack_cols_bf16 (see ops.h)
    int ny = 1; long long src_y = 0, dst_y = 0;                    // planes (grid.y), strides: floats / uint4
    const float* mu = nullptr; long long mu_t = 0, mu_y = 0; float inv_rows = 0.f;
    int coff = 0, ncw = 0, tmul = 1, toff = 0, rows = -1;          // rows: destination rows (default Kp, zero beyond N)
    long long lo = 0;                                              // x3r: elements between the hi image and the lo image it also writes
};
static int pack_cols_bf16(const float* src, long long src_t, const Shp& s, int col0, int w, int T, int ldo, uint16_t* dst, hipStream_t st,
                          const PackX& x = PackX()) {
    const int rows = x.rows < 0 ? s.Kp : x.rows;
    const long long n = (long long)rows * ((x.ncw > 0 ? x.ncw : ldo) / 8);
    LAUNCH(k_pack_cols_bf16, dim3(cdiv(n, 256), x.ny), dim3(256), 0, st, src, src_t, s.N, s.ld, s.Cp, col0, w, s.B, T, rows, ldo,
           reinterpret_cast<uint4*>(dst), x.src_y, x.dst_y, x.mu, x.mu_t, x.mu_y, x.inv_rows, x.coff, x.ncw, x.tmul, x.toff, x.lo / 8);
    return 0;
}
static int prop_fwd_bf16(const Shp& s, const Sup& u, float* Z, uint16_t* x0b, uint16_t* x0c, hipStream_t st,
                         uint16_t* Pb = nullptr, bool packed = false) {
    Bf16GemmP p = bgp(u);
    p.A = u.Sstk; p.am = rm_plain(u.Kp); p.M = u.nb * s.N;
    p.nseg = 1; p.seg_len = s.N; p.a_seg = 0; p.b_seg = 0;
    p.C = Z + s.PS; p.cm = rm_two(s.N, s.PS, s.ld);
    if (s.lite && Pb) {
        // bf16-resident planes: the product is written ONCE, as bf16 [nb][N][B*H]; its B operand x0b was emitted by the
        // epilogue of the weight pool that produced the state (packed), or is packed here (first cell of a stack)
        if (!packed) CKI(pack_cols_bf16(Z, 0, s, 0, s.H, 1, (int)s.ldh, x0b, st));
        p.B = x0b; p.ldb = s.ldh; p.N = (int)s.ldh;
        p.C = nullptr; p.Cb = Pb; p.cbm = rm_plain(s.ldh);
        return bf16_gemm(p, true, 1, ROLE_PROP, (double)u.nb * 2.0 * (double)s.N * s.N * (double)s.B * s.H, st);
    }
    if (s.hoist) {
        // only the state channels change from step to step: B = their packed bf16 copy (N x B*H), the result lands in the
        // h-channel block of every (node, sample) row of planes 1 .. nb; the input-channel blocks were filled by hoist_inputs
        // (packed: the weight-pool epilogue that produced the state already wrote the operand, hi and lo image - round 6, x3r)
        if (!packed) { PackX xs_; xs_.lo = s.lo_x0b; CKI(pack_cols_bf16(Z, 0, s, 0, s.H, 1, (int)s.ldh, x0b, st, xs_)); }
        p.B = x0b; p.ldb = s.ldh; p.N = (int)s.ldh;
        p.cn_inner = s.H; p.cn_hi = s.Cp;
        x3_terms(p, u.lo_Sstk, s.lo_x0b);
        return bf16_gemm(p, true, 1, ROLE_PROP, (double)u.nb * 2.0 * (double)s.N * s.N * (double)s.B * s.H, st);
    }
    CKI(plane_to_bf16(s, u, Z, x0b, x0c, st));
    p.B = x0b; p.ldb = s.ldp; p.N = (int)s.ld;
    return bf16_gemm(p, true, 1, ROLE_PROP, (double)u.nb * 2.0 * (double)s.N * s.N * (double)s.B * s.C, st);
}
// MCRN_BF16, hoisted input channels: columns [col0, col0 + w) of plane 0 of the T plane sets Z[t] (stride s.ZT) are
// propagated in ONE product  [S1; T2(S1); S2; T2(S2)] x [N x T*B*w]  and written to the same columns of planes 1 .. nb of
// Z[t] and Y[t] (gate and candidate input share their input channels, model/MegaCRN.py:42,45).
static const int HOIST_MAX_SPLIT = 8;
static int hoist_inputs(const Shp& s, const Sup& u, float* Z, float* Y, int T, int col0, int w, uint16_t* xin_b, float* xin_t,
                        hipStream_t st, float* Xp = nullptr /* compact destination [T][nb][R][4] (k_scatter_compact) instead of the planes */) {
    if (!s.hoist || w <= 0 || T <= 0) return 0;
    const int ncols = T * s.B * w;
    const int ncp = ncols < 8 ? 8 : (ncols + 7) & ~7;
    PackX xs_; xs_.lo = u.lo_xin_b;
    CKI(pack_cols_bf16(Z, s.ZT, s, col0, w, T, ncp, xin_b, st, xs_));
    Bf16GemmP p = bgp(u);
    x3_terms(p, u.lo_Sstk, u.lo_xin_b);
    p.A = u.Sstk; p.am = rm_plain(u.Kp); p.M = u.nb * s.N;
    p.B = xin_b; p.ldb = ncp; p.N = ncp;
    p.nseg = 1; p.seg_len = s.N;
    p.C = xin_t; p.cm = rm_plain(ncp);
    // a narrow product (T*B*w = 32 .. 400 columns against 7372 rows at N = 1843): split K so that it still fills the chip
    const long long tiles = (long long)cdiv(p.M, 256) * cdiv(ncp, 128);
    int nsplit = (int)((240 + tiles / 2) / tiles);
    nsplit = nsplit < 1 ? 1 : (nsplit > HOIST_MAX_SPLIT ? HOIST_MAX_SPLIT : nsplit);
    // splits the launcher really makes (bf16_split_plan) for a K tile of 64 and of 32: the consumer below must know the count
    // whatever tile configuration the tuner picks, so a request that the two depths would round differently is not split
    auto eff = [&](int bk) { const int kt = cdiv(s.N, bk), ns = nsplit > kt ? kt : nsplit; return cdiv(kt, cdiv(kt, ns)); };
    // (K tiles are 64 or 32 deep in the plain form, 32 or 16 in the hi/lo form)
    if (eff(64) != eff(32) || (g_x3r && eff(16) != eff(32)) || bf16_cfg_is_sk(g_force_cfg_bf16)) nsplit = 1;    // (a forced stream-K configuration ignores the split)
    const int nsp = eff(64);
    p.slab = (long long)p.M * ncp;
    CKI(bf16_gemm(p, true, nsplit, ROLE_PROP, (double)u.nb * 2.0 * (double)s.N * s.N * (double)ncols, st, PROF_ROLE_PROP_IN));
    if (Xp) {
        const long long rows = (long long)T * u.nb * s.R;
        fam("hoisted_inputs:compact");
        LAUNCH(k_scatter_compact, dim3(cdiv(rows, 256)), dim3(256), 0, st, (const float*)xin_t, ncp, nsp, p.slab, u.nb, s.N, s.B, w, T, Xp,
               (long long)u.nb * s.R * 4, col0 - s.H, w == s.d ? 1 : 0);
        return 0;
    }
    const long long tot = (long long)u.nb * s.N * ncols;
    LAUNCH(k_scatter_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, (const float*)xin_t, ncp, nsp, p.slab, u.nb, s.N, s.B, w, T, Z, Y,
           s.ZT, s.PS, s.ld, s.Cp, col0);
    return 0;
}
// Small graphs (Shp::hoist_fwd): the same hoisting in fp32 / bf16x3.  Columns [col0, col0 + w) of plane 0 of T plane
// sets are packed into an N x (T*B*w) matrix, propagated by the fused two-hop kernel (planes 1 .. 4 of the scratch set) and
// scattered into planes 1 .. 4 of Z[t] and Y[t].  The pad channels ride along (w reaches Cp): S x 0 = 0 is what the
// full-width propagation writes there, and the adjacency gradient reads those columns.
static int prop_fwd(const Shp& s, const Sup& u, float* Z, hipStream_t st, uint16_t* x0b, uint16_t* x0c, uint16_t* Pb, bool packed);
static int hoist_inputs_small(const Shp& s, const Sup& u, float* Z, float* Y, int T, int col0, int w, float* xin_f, hipStream_t st) {
    if (!s.hoist_fwd || w <= 0 || T <= 0) return 0;
    const int ncols = T * s.B * w;
    const int ncp = (ncols + 3) & ~3;
    const long long tot0 = (long long)s.N * ncp;
    LAUNCH(k_pack_cols_f32, dim3(cdiv(tot0, 256)), dim3(256), 0, st, (const float*)Z, s.ZT, s.N, s.ld, s.Cp, col0, w, s.B, T, ncp, xin_f);
    Shp t = s;
    t.ld = ncp; t.PS = (long long)s.N * ncp; t.hoist = false; t.lite = false; t.hoist_fwd = false; t.state_only = false;
    Prop2P q;
    q.ev0 = q.ev1 = nullptr; q.nunits = q.cps = q.cstride = 0; q.no_d1 = 0;
    q.Sf[0] = u.Sf[0]; q.Sf[1] = u.Sf[1]; q.base = xin_f; q.extra = nullptr; q.PS = t.PS; q.ld = ncp; q.N = s.N; q.ncols = ncp;
    const double fl = 2.0 * 2.0 * 2.0 * (double)s.N * s.N * (double)ncols;
    fam("prop2_fwd:hoisted_inputs");
    MCRN_PROF_WRAP(ROLE_PROP, launch_prop2_fwd(q, st), fl, fl);
    const long long tot = (long long)4 * s.N * ncols;
    LAUNCH(k_scatter_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, (const float*)(xin_f + t.PS), ncp, 1, 0LL, 4, s.N, s.B, w, T, Z, Y,
           s.ZT, s.PS, s.ld, s.Cp, col0);
    return 0;
}
// MCRN_BF16 backward propagation: dP[0] += [S1^T | T2(S1)^T | S2^T | T2(S2)^T] x [dP[1]; ..; dP[nb]]  (K = nb*N)
// The output is only N x B*Cp (135 / 255 tiles of 128 x 128 at EXPY-TKY) while K is nb*N deep: K is split in two, the
// second half lands in the extra plane dT that the element-wise consumers of dP[0] add (same mechanism as the fused
// S^T chain of the small-graph path), so no reduction pass and still one writer per element.
static const int PROPT_MAX_X = 3;                       // extra partial planes behind dT (K splits 1 .. 3)
// K splits the launcher REALLY makes (bf16_split_plan) for a request `want` over nseg segments of seg_len, at a K tile of 64
// and of 32: the consumers of the partial planes must know the count whatever tile the tuner picks, so a request is only
// usable when both depths round it the same way and to itself (-1 otherwise).  want = 2 is exact for every K >= 2 tiles.
static int bf16_eff_splits(int nseg, int seg_len, int want) {
    auto eff = [&](int bk) { const int kt = nseg * cdiv(seg_len, bk), ns = want > kt ? kt : want; return cdiv(kt, cdiv(kt, ns)); };
    return eff(64) == eff(32) && eff(64) == want && (!g_x3r || eff(16) == want) ? want : -1;
}
struct SplitKey { int role, M, N, K, nterm; bool operator<(const SplitKey& o) const { return memcmp(this, &o, sizeof(SplitKey)) < 0; } };
static std::map<SplitKey, int> g_tuned_split;          // K splits of the transposed propagation, chosen with the tiles
// hoisted (the packed operand dPb = [nb][Kp][B*H]): only the state channels of plane 0 receive a propagated gradient here;
// the input channels' share (needed for the go symbol of a step that was not teacher-forced) is go_grad_bf16 below
// (Round 5 measured bf16 / packed partial planes behind this split product: the read-modify-write epilogue of split 0 is what makes it
//  slower than the forward product - 52 vs 45 us without it - but every byte-saving form rounds the partial sums to bf16 and costs the
//  mode 25 % of its accuracy and its batch additivity; the exact fp32 form below stays.  profiles/r5/experiments.md section 2.)
static int prop_bwd_bf16(const Shp& s, const Sup& u, float* dP, const uint16_t* dPb, float* dT, int* used_dT, hipStream_t st,
                         bool hoisted = false) {
    Bf16GemmP p = bgp(u);
    p.A = u.STstk; p.am = rm_plain((long long)u.nb * u.Kp); p.M = s.N;
    p.B = dPb; p.ldb = s.ldp; p.N = (int)s.ld;
    p.nseg = u.nb; p.seg_len = s.N; p.a_seg = u.Kp; p.b_seg = s.PSb;
    p.C = dP; p.Cin = dP; p.beta = 1.f; p.cm = rm_plain(s.ld);
    p.cin_pre = 1;                               // asks for the accumulator preload; honoured only by -DMCRN_BF16_PRELOAD=1 builds (round 6: the shipped
                                                 // kernels add plane 0 in the epilogue again - the preload's SGPRs cost more than its loads saved, gemm_bf16.h)
    if (hoisted) { p.ldb = s.ldh; p.N = (int)s.ldh; p.b_seg = s.PSbh; p.cn_inner = s.H; p.cn_hi = s.Cp; }
    x3_terms(p, u.lo_STstk, s.lo_dPb);
    int nsplit = 1;
    const int split_env = 0;                                 // K splits: tuned (2 .. 4) with the tile
    const double alg = (double)u.nb * 2.0 * (double)s.N * s.N * (double)s.B * (hoisted ? s.H : s.C);
    if (dT && used_dT && split_env != 1 && !bf16_cfg_is_sk(g_force_cfg_bf16) && (long long)cdiv(s.N, 128) * cdiv(p.N, 128) < 512) {
        // The output is only N x B*Cp while K is nb*N deep: K is split, split 0 accumulates into plane 0, the others land in
        // the extra planes dT .. that the element-wise consumers of plane 0 add in a fixed order (no reduction pass, one
        // writer per element).  How many splits fill the chip best depends on the tile the tuner picks: chosen with it.
        p.slab = dT - dP; p.slab2 = s.PS; p.cin_first_only = 1;
        const SplitKey key{ROLE_PROPT, p.M, p.N, u.nb * s.N, p.nterm > 1 ? p.nterm : 1};
        // (a request the launcher would round to fewer splits - short K: N <= 64 at cheb_k = 3 - is not taken: the consumers
        //  would add partial planes that nothing wrote; 2 is exact for any K of two tiles or more)
        nsplit = split_env >= 2 && split_env <= 1 + PROPT_MAX_X && bf16_eff_splits(u.nb, s.N, split_env) == split_env ? split_env : 2;
        if (split_env == 0) {
            auto it = g_tuned_split.find(key);
            if (it != g_tuned_split.end() && bf16_eff_splits(u.nb, s.N, it->second) == it->second) nsplit = it->second;
            else if (g_tuning) {
                float best = 1e30f;
                for (int ns = 2; ns <= 1 + PROPT_MAX_X; ++ns) {
                    if (bf16_eff_splits(u.nb, s.N, ns) != ns) continue;
                    CKI(bf16_gemm(p, true, ns, ROLE_PROPT, alg, st));        // tunes the tile for this split count
                    if (g_last_tune_ms < best) { best = g_last_tune_ms; nsplit = ns; }
                }
                g_tuned_split[key] = nsplit;
                if (getenv("MCRN_TUNE_LOG")) fprintf(stderr, "[mcrn tune] role 4 M=%d N=%d: %d K splits\n", p.M, p.N, nsplit);
            }
        }
        if (bf16_eff_splits(u.nb, s.N, nsplit) != nsplit) nsplit = 1;          // (a single K tile: nothing to split)
        *used_dT = nsplit - 1;
        if (nsplit == 1) { p.slab = 0; p.slab2 = 0; p.cin_first_only = 0; }
    }
    return bf16_gemm(p, true, nsplit, ROLE_PROPT, alg, st);
}
// MCRN_BF16 adjacency gradient of a whole cell stack, ONE launch: K runs over every AGCN call of the stack
//   dA[b] (N x N) (+)= sum_calls dP_call[1 + b] (N x B*Cp) x (X0_call - mean)^T        b < nb
static int ds_bf16(const Shp& s, const Sup& u, const uint16_t* dPb_all, const uint16_t* x0c_all, int ncalls, float* dA,
                   long long ldS, bool accumulate, hipStream_t st) {
    Bf16GemmP p = bgp(u);
    p.A = dPb_all; p.am = rm_two(s.N, s.PSb, s.ldp); p.M = u.nb * s.N;
    p.B = x0c_all; p.bm = rm_plain(s.ldp); p.N = s.N;
    p.nseg = ncalls; p.seg_len = (int)s.ld; p.a_seg = (long long)u.nb * s.PSb; p.b_seg = s.PSb;
    p.C = dA; p.cm = rm_two(s.N, (long long)s.N * ldS, ldS);
    if (accumulate) { p.Cin = dA; p.beta = 1.f; }
    return bf16_gemm(p, false, 1, ROLE_DS, (double)ncalls * u.nb * 2.0 * (double)s.N * s.N * (double)s.B * s.C, st);
}

// hoisted backward: the same product on the packed state-channel operands (K = B*H per call) ...
static int ds_bf16_h(const Shp& s, const Sup& u, const uint16_t* dPbh_all, const uint16_t* x0ch_all, int ncalls, float* dA,
                     long long ldS, bool accumulate, hipStream_t st) {
    Bf16GemmP p = bgp(u);
    p.A = dPbh_all; p.am = rm_two(s.N, s.PSbh, s.ldh); p.M = u.nb * s.N;
    p.B = x0ch_all; p.bm = rm_plain(s.ldh); p.N = s.N;
    p.nseg = ncalls; p.seg_len = (int)s.ldh; p.a_seg = (long long)u.nb * s.PSbh; p.b_seg = s.PSbh;
    p.C = dA; p.cm = rm_two(s.N, (long long)s.N * ldS, ldS);
    if (accumulate) { p.Cin = dA; p.beta = 1.f; }
    x3_terms(p, s.lo_dPb, s.lo_x0c);
    return bf16_gemm(p, false, 1, ROLE_DS, (double)ncalls * u.nb * 2.0 * (double)s.N * s.N * (double)s.B * s.H, st);
}
// ... plus the input channels of every call as ONE narrow product:  dA[b] += dPin[b] (N x kcols) x xin_c^T (N x kcols)
static int ds_bf16_in(const Shp& s, const Sup& u, const uint16_t* dPin, const uint16_t* xin_c, long long kin, int kcols, float* dA,
                      long long ldS, hipStream_t st, long long lo_xin_c = 0) {
    Bf16GemmP p = bgp(u);
    p.A = dPin; p.am = rm_plain(kin); p.M = u.nb * s.N;
    p.B = xin_c; p.bm = rm_plain(kin); p.N = s.N;
    p.nseg = 1; p.seg_len = kcols;
    p.C = dA; p.cm = rm_two(s.N, (long long)s.N * ldS, ldS); p.Cin = dA; p.beta = 1.f;
    x3_terms(p, s.lo_dPin, lo_xin_c);
    return bf16_gemm(p, false, 1, ROLE_DS, (double)u.nb * 2.0 * (double)s.N * s.N * (double)kcols, st);
}
// hoisted backward, a decoder cell whose go symbol was the projection of the previous step (not teacher-forced): the input
// channels of its planes 1 .. nb carry gradient for that symbol,  d go += sum_g T_g^T dP_g[:, input columns] ; both calls of the
// cell (adjacent column blocks of the stack-wide operand dPin) in one product, added to the go columns of the two planes 0
static const int GO_MAX_SPLIT = 16;
static int go_grad_bf16(const Shp& s, const Sup& u, const uint16_t* dPin, long long kin, int call0, float* tmp, float* dQ0, float* dP0,
                        int od, hipStream_t st) {
    const int bw = s.B * s.d;                                       // columns of one call
    Bf16GemmP p = bgp(u);
    p.A = u.STstk; p.am = rm_plain((long long)u.nb * u.Kp); p.M = s.N;
    p.B = dPin + (long long)call0 * bw; p.ldb = kin; p.N = 2 * bw;
    p.nseg = u.nb; p.seg_len = s.N; p.a_seg = u.Kp; p.b_seg = (long long)s.N * kin;
    p.C = tmp; p.cm = rm_plain(2 * bw);
    x3_terms(p, u.lo_STstk, s.lo_dPin);
    // N x 2*B*d output (8 tiles of 256 x 128 at N = 1843) over K = nb*N: split K so that the launch fills the chip (88 us unsplit)
    const long long tiles = (long long)cdiv(p.M, 256) * cdiv(p.N, 128);
    int nsplit = (int)((240 + tiles / 2) / tiles);
    nsplit = nsplit < 1 ? 1 : (nsplit > GO_MAX_SPLIT ? GO_MAX_SPLIT : nsplit);
    // splits the launcher really makes for a K tile of 64 and of 32 (bf16_split_plan): the consumer below must know the count
    // whatever tile the tuner picks, so the request is lowered until both depths round it the same way
    auto eff = [&](int bk, int want) { const int kt = u.nb * cdiv(s.N, bk), ns = want > kt ? kt : want; return cdiv(kt, cdiv(kt, ns)); };
    while (nsplit > 1 && (eff(64, nsplit) != eff(32, nsplit) || (g_x3r && eff(16, nsplit) != eff(32, nsplit)))) --nsplit;
    if (bf16_cfg_is_sk(g_force_cfg_bf16)) nsplit = 1;                             // (a forced stream-K configuration ignores the split)
    const int nsp = eff(64, nsplit);
    p.slab = (long long)s.N * 2 * bw;
    CKI(bf16_gemm(p, true, nsplit, ROLE_PROPT, (double)u.nb * 2.0 * (double)s.N * s.N * 2.0 * bw, st));
    const long long tot = (long long)s.N * s.B * od;
    LAUNCH(k_scatter_add_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, (const float*)tmp, 2 * bw, 0, nsp, p.slab, s.N, s.B, s.d, od, dQ0, s.ld, s.Cp, s.H);
    LAUNCH(k_scatter_add_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, (const float*)tmp, 2 * bw, bw, nsp, p.slab, s.N, s.B, s.d, od, dP0, s.ld, s.Cp, s.H);
    return 0;
}

// ---- K-hop propagation, forward:  planes[1..] from plane 0   (model/MegaCRN.py:19-25) --------
static int prop_fwd(const Shp& s, const Sup& u, float* Z, hipStream_t st, uint16_t* x0b = nullptr, uint16_t* x0c = nullptr,
                    uint16_t* Pb = nullptr, bool packed = false) {
    if (g_prop_bf16 && u.Sstk && x0b) return prop_fwd_bf16(s, u, Z, x0b, x0c, st, Pb, packed);
    if (use_prop2(u, s) && aligned16(Z)) {   // both hops, one launch
        Prop2P q;
        q.Sf[0] = u.Sf[0]; q.Sf[1] = u.Sf[1]; q.base = Z; q.extra = nullptr; q.PS = s.PS; q.ld = s.ld; q.N = s.N; q.ncols = (int)s.ld;
        q.ev0 = q.ev1 = nullptr; q.nunits = q.cps = q.cstride = 0; q.no_d1 = 0;
        double alg = 2.0 * 2.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.C;   // 2 hops x 2 supports
        double ex = 2.0 * 2.0 * 2.0 * (double)s.N * s.N * (double)s.ld;
        if (s.state_only) {          // the input / pad channels of this step's planes were propagated once per stack
            q.cps = s.H / 64; q.nunits = s.B * q.cps; q.cstride = s.Cp;
            alg = 2.0 * 2.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.H;
            ex = alg;
        }
        fam(s.state_only ? "prop2_fwd:state_only" : (s.N > 256 ? "prop2_fwd:streamed_S" : "prop2_fwd"));
        MCRN_PROF_WRAP_EXT(ROLE_PROP, q, launch_prop2_fwd(q, st), ex, alg);
        return 0;
    }
    if (use_prop_small(u, s) && aligned16(Z)) {
        PropP q;
        memset(&q, 0, sizeof q);
        q.nseg = 1; q.N = s.N; q.ncols = (int)s.ld; q.ld = s.ld; q.alpha = 1.f;
        for (int b = 0; b < 2; ++b) { q.Sf[b][0] = u.Sf[b]; q.X[b][0] = Z; q.C[b] = Z + (1 + b * (s.K - 1)) * s.PS; }
        CKI(prop_small(q, 2, ROLE_PROP, 2.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.C, st));
        if (s.K == 3) {   // x2 = 2 S x1 - x0
            for (int b = 0; b < 2; ++b) { q.X[b][0] = Z + (1 + 2 * b) * s.PS; q.C[b] = Z + (2 + 2 * b) * s.PS; q.Cin[b] = Z; }
            q.alpha = 2.f; q.beta = -1.f;
            CKI(prop_small(q, 2, ROLE_PROP, 2.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.C, st));
        }
        return 0;
    }
    GemmP p = gp();
    p.M = s.N; p.N = (int)s.ld; p.K = s.N;
    p.am = plain(u.ldS); p.ak = plain(1);
    p.bk = plain(s.ld);  p.bn = plain(1);
    p.cm = plain(s.ld);  p.cn = plain(1);
    p.nbatch = 2;
    const bool simg = g_precision == MCRN_BF16X3 && u.Simg[0] != nullptr;
    if (simg) { p.Aimg[0] = u.Simg[0]; p.Aimg[1] = u.Simg[1]; p.aimg_n = u.simg_n; }
    for (int b = 0; b < 2; ++b) {
        p.A[b] = u.S[b];
        p.B[b] = Z;
        p.C[b] = Z + (1 + b * (s.K - 1)) * s.PS;
        p.Cin[b] = nullptr;
    }
    // algorithmic flops per launch: 2 supports x 2*N^2*B*C with the TRUE channel count (no pad)
    p.alg_flops = 2.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.C;
    CKI(gemm(p, true, false, 0, ROLE_PROP, st));
    for (int k = 2; k < s.K; ++k) {   // x_k = 2 S x_{k-1} - x_{k-2}   (model/MegaCRN.py:21-22 as a feature recursion)
        for (int b = 0; b < 2; ++b) {
            const long long g1 = 1 + (long long)b * (s.K - 1);          // plane of x_1 of support b; x_k sits at g1 + k - 1
            p.B[b] = Z + (g1 + k - 2) * s.PS;
            p.C[b] = Z + (g1 + k - 1) * s.PS;
            p.Cin[b] = k == 2 ? Z : Z + (g1 + k - 3) * s.PS;
        }
        p.alpha = 2.f; p.beta = -1.f;
        CKI(gemm(p, true, false, 0, ROLE_PROP, st));
    }
    return 0;
}

// ---- weight pool with fused epilogue:  out = epi([Z planes] @ Wf + b)   (MegaCRN.py:26-27) ----
static int wp_fwd(const Shp& s, const float* Z, const float* Wf, int O, GemmP epi, hipStream_t st,
                  const uint4* img = nullptr) {
    GemmP p = epi;
    if (img && g_precision == MCRN_BF16X3) { p.Bimg = img; p.bimg_n = (O + 3) & ~3; }
    p.M = (int)s.R; p.N = O; p.K = s.G * s.Cp;
    p.A[0] = Z; p.B[0] = Wf;
    p.am = plain(s.Cp); p.ak = two(s.Cp, s.PS, 1); p.ak_hi[0] = s.PS;
    p.bk = plain(O); p.bn = plain(1);
    p.alg_flops = 2.0 * (double)s.R * (2.0 * s.K * s.C) * O;   // reference K = 2*cheb_k*C
    return gemm(p, true, false, 0, ROLE_WP, st);
}

// ---- AGCN backward for cheb_k > 3: the Chebyshev recursion differentiated step by step (oracle: agcn_bwd) on the tiled GEMM.
//   raw plane gradients d_k = dY W_k^T (Wd unfolded), then per support, k = K-1 .. 2:
//     dS += 2 d_k x_{k-1}^T ; d_{k-1} += 2 S^T d_k ; d_{k-2} -= d_k        and finally  dS += d_1 x_0^T ; d_0 += S^T d_1
static int agcn_bwd_general(const Shp& s, const Sup& u, const float* dY, int O, const float* Wd, const float* X, float* dP,
                            hipStream_t st, const uint4* imgd) {
    {   // d-grad: dP[g][r][c'] = sum_o dY[r][o] Wd[(g,c')][o]
        GemmP p = gp();
        p.M = (int)s.R; p.N = s.G * s.Cp; p.K = O;
        p.A[0] = dY; p.am = plain(O); p.ak = plain(1);
        p.B[0] = Wd; p.bk = plain(1); p.bn = plain(O);
        p.C[0] = dP; p.cm = plain(s.Cp); p.cn = two(s.Cp, s.PS, 1);
        (void)imgd;
        CKI(gemm(p, true, true, 0, ROLE_DGRAD, st));
    }
    auto ds_acc = [&](int b, const float* dk, const float* xk, float alpha) -> int {   // dS_b += alpha dk xk^T, split-K slabs
        GemmP p = gp();
        p.M = s.N; p.N = s.N; p.K = (int)s.ld;
        p.A[0] = dk; p.am = plain(s.ld); p.ak = plain(1);
        p.B[0] = xk; p.bn = plain(s.ld); p.bk = plain(1);
        p.C[0] = u.dS + (long long)b * u.sup_stride; p.Cin[0] = p.C[0]; p.cm = plain(u.ldS); p.cn = plain(1);
        p.alpha = alpha; p.beta = 1.f; p.slab = u.slab;
        return gemm(p, true, true, u.nslab, ROLE_DS, st);
    };
    auto st_acc = [&](int b, const float* src, float* dst, float alpha) -> int {       // dst += alpha S_b^T src
        GemmP p = gp();
        p.M = s.N; p.N = (int)s.ld; p.K = s.N;
        p.A[0] = u.St[b]; p.am = plain(u.ldS); p.ak = plain(1);
        p.B[0] = src; p.bk = plain(s.ld); p.bn = plain(1);
        p.C[0] = dst; p.Cin[0] = dst; p.cm = plain(s.ld); p.cn = plain(1);
        p.alpha = alpha; p.beta = 1.f;
        return gemm(p, true, false, 0, ROLE_PROPT, st);
    };
    for (int b = 0; b < 2; ++b) {
        const long long g1 = 1 + (long long)b * (s.K - 1);
        auto plane = [&](float* base, int k) { return k == 0 ? base : base + (g1 + k - 1) * s.PS; };
        for (int k = s.K - 1; k >= 2; --k) {
            float* dk = plane(dP, k);
            CKI(ds_acc(b, dk, plane(const_cast<float*>(X), k - 1), 2.f));
            CKI(st_acc(b, dk, plane(dP, k - 1), 2.f));
            LAUNCH(k_axpy, dim3(cdiv(s.PS, 256)), dim3(256), 0, st, plane(dP, k - 2), (const float*)dk, -1.f, s.PS);
        }
        CKI(ds_acc(b, plane(dP, 1), X, 1.f));
        CKI(st_acc(b, plane(dP, 1), dP, 1.f));
    }
    return 0;
}

// the d-grad of this call runs on the streaming kernel (dgrad_stream.h; imgd is then the B-fragment image of Wd, built by wprep
// under the same test)
static inline bool dgrad_streams(const Shp& s, int O, const float* dY, const uint4* imgd) {
    return s.K <= 3 && imgd && g_precision == MCRN_BF16X3 && dgrad_stream_ok(O) && aligned16(dY);
}
// ---- AGCN backward core: dY (R x O) -> dP planes; plane 0 of dP ends as d(input); dS slabs += ----
static int agcn_bwd_core(const Shp& s, const Sup& u, const float* dY, int O, const float* Wd,
                         const float* X, float* dP, hipStream_t st, int buf = 0, const uint4* imgd = nullptr,
                         float* dT = nullptr, int* used_dT = nullptr, uint16_t* dPb = nullptr, DsP* cell_ds = nullptr,
                         bool cell_ds_last = true, uint16_t* dPin = nullptr /* hoisted backward: stack-wide input operand */,
                         long long kin = 0, int in_col0 = 0) {
    if (used_dT) *used_dT = 0;
    if (s.K > 3) return agcn_bwd_general(s, u, dY, O, Wd, X, dP, st, imgd);
    bool dgrad_wrote_bf16 = false;
    const bool side = g_use_side && !g_tuning && g_prof.role < 0;   // tuning / profiling time kernels in-line
    if (side) { CKI(side_init()); CKI(side_guard(buf, st)); }
    if (dgrad_streams(s, O, dY, imgd)) {
        // d-grad, streaming form (dgrad_stream.h): imgd is the B-fragment image of Wd (built by wprep under the same test)
        DgradP q;
        memset(&q, 0, sizeof q);
        q.dY = dY; q.Wfrag = imgd; q.dP = dP; q.R = s.R; q.PS = s.PS; q.O = O; q.ncols = s.G * s.Cp; q.Cp = s.Cp; q.dbg = g_debug;
        q.dPb = nullptr; q.PSb = 0; q.H = 0; q.d = s.d; q.B = s.B; q.dPin = nullptr; q.kin = 0; q.in_plane = 0; q.in_col0 = 0;
        if (g_prop_bf16 && u.STstk && dPb && dPin) {            // hoisted backward: packed state channels + the input operand
            q.dPb = dPb; q.PSb = s.PSbh; q.H = s.H; q.dPin = dPin; q.kin = kin; q.in_plane = (long long)s.N * kin; q.in_col0 = in_col0;
            q.dPb_lo = s.lo_dPb; q.dPin_lo = s.lo_dPin;
            dgrad_wrote_bf16 = true;
        } else if (g_prop_bf16 && u.STstk && dPb && s.ldp == s.ld) {   // bf16 plane rows then coincide with the fp32 rows: write them here
            q.dPb = dPb; q.PSb = s.PSb; dgrad_wrote_bf16 = true;
        }
        const double fl = 2.0 * (double)s.R * O * (double)(s.G * s.Cp);
        fam(O > 128 ? "dgrad_stream:two_half" : "dgrad_stream");
        if (q.dPin) fam(q.dPb_lo > 0 ? "dgrad_stream:writes_hilo_operands" : "dgrad_stream:writes_bf16_operands");
        MCRN_PROF_WRAP(ROLE_DGRAD, launch_dgrad_stream(q, st), fl, 2.0 * (double)s.R * O * (double)(s.G * s.C));
    } else {   // d-grad: dP[g][r][c'] = sum_o dY[r][o] Wd[(g,c')][o]
        GemmP p = gp();
        p.M = (int)s.R; p.N = s.G * s.Cp; p.K = O;
        p.A[0] = dY; p.am = plain(O); p.ak = plain(1);
        p.B[0] = Wd; p.bk = plain(1); p.bn = plain(O);
        p.C[0] = dP; p.cm = plain(s.Cp); p.cn = two(s.Cp, s.PS, 1);
        if (imgd && g_precision == MCRN_BF16X3) { p.Bimg = imgd; p.bimg_n = (s.G * s.Cp + 3) & ~3; }
        CKI(gemm(p, true, true, 0, ROLE_DGRAD, st));
    }
    if (g_prop_bf16 && u.STstk && dPb && dPin) {
        if (!dgrad_wrote_bf16) {   // tiled d-grad (O > 128) wrote fp32 planes: pack the state channels and the input channels from them
            PackX xh; xh.ny = u.nb; xh.src_y = s.PS; xh.dst_y = s.PSbh / 8; xh.lo = s.lo_dPb;
            CKI(pack_cols_bf16(dP + s.PS, 0, s, 0, s.H, 1, (int)s.ldh, dPb, st, xh));
            PackX xi; xi.ny = u.nb; xi.src_y = s.PS; xi.dst_y = (long long)s.N * kin / 8; xi.coff = in_col0; xi.ncw = (s.B * s.d + 7) & ~7; xi.rows = s.N;
            xi.lo = s.lo_dPin;
            CKI(pack_cols_bf16(dP + s.PS, 0, s, s.H, s.d, 1, (int)kin, dPin, st, xi));
        }
        return prop_bwd_bf16(s, u, dP, dPb, dT, used_dT, st, true);
    }
    if (g_prop_bf16 && u.STstk && dPb) {
        // planes 1.. of dP as bf16 (operand of the S^T product now, of the stack's adjacency gradient later)
        if (!dgrad_wrote_bf16) CKI(planes_to_bf16(s, dP + s.PS, u.nb, dPb, nullptr, nullptr, st, -1, -1, -1, 0));
        return prop_bwd_bf16(s, u, dP, dPb, dT, used_dT, st);
    }
    const bool small = use_prop_small(u, s) && aligned16(dP);
    const bool fused_bwd = use_prop2(u, s) && aligned16(dP) && dT != nullptr;
    // The adjacency-gradient launch of this cell goes to the helper stream behind the fused chain below.  Its `ready` event is ATTACHED to
    // that chain's dispatch (completion signal of the kernel itself) instead of recorded behind it: a recorded event is a marker packet
    // in the caller's queue, and the next kernel there starts ~7.5 us later; behind the attached signal it is ~5 us (round 5: one such
    // bubble per cell; with the per-cell plane sets of ModelPlan::flat METR-LA +1.2 %, PEMS-BAY +0.8 %: profiles/r5/experiments.md 12).
    const bool ds_fused_path = (small || (s.N > 256 && s.N <= PROP2_MAX_N && fused_bwd && (s.ld % 4) == 0)) && aligned16(X);
    // MCRN_READY_EVENT=record (read once at load): the round-4 form, hipEventRecord behind the chain.  The attached form relies on the stop
    // event of hipExtLaunchKernelGGL behaving as a recorded event for a cross-stream wait; the GPU suite runs both forms over many back-to-back
    // steps and compares every gradient bit for bit (test_attached_ready_event_orders_the_helper_stream), so a runtime that changes that is
    // caught, and this switch is the way out.
    static const bool ready_record = getenv("MCRN_READY_EVENT") && !strcmp(getenv("MCRN_READY_EVENT"), "record");
    const bool ready_attached = side && fused_bwd && ds_fused_path && (!cell_ds || cell_ds_last) && !ready_record;
    if (fused_bwd) {
        // whole S^T chain for both supports in one launch: d1t_s = d1_s + S_s^T e2_s (written back),
        // dP[0] += S_1^T d1t_1 + S_2^T d1t_2.  The adjacency-gradient GEMM below then reads d1t / e2.
        Prop2P q;
        q.ev0 = q.ev1 = nullptr; q.nunits = q.cps = q.cstride = 0; q.no_d1 = u.ds4 ? 1 : 0;
        q.Sf[0] = u.Stf[0]; q.Sf[1] = u.Stf[1]; q.base = dP; q.extra = dT; q.PS = s.PS; q.ld = s.ld; q.N = s.N; q.ncols = (int)s.ld;
        if (used_dT) *used_dT = 1;
        double ex = 4.0 * 2.0 * (double)s.N * s.N * (double)s.ld, alg = 4.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.C;
        if (s.state_only) {
            // hoisted backward (small graphs): the chain runs on the B*H state columns only - nothing consumes the propagated gradient
            // of this cell's input channels (its go symbol was known in advance), and the adjacency gradient reads the raw d-grad
            // planes on every column (Sup::ds4; rounds 4's gathered first hop on the input channels is gone with d1t)
            q.cps = s.H / 64; q.nunits = s.B * q.cps; q.cstride = s.Cp;
            ex = alg = 4.0 * 2.0 * (double)s.N * s.N * (double)s.B * s.H;
        }
        if (ready_attached) q.ev1 = g_side.ready[buf];
        fam(s.state_only ? "prop2_bwd:state_only" : (s.N > 256 ? "prop2_bwd:streamed_S" : "prop2_bwd"));
        MCRN_PROF_WRAP(ROLE_PROPT, launch_prop2_bwd(q, st), ex, alg);
    } else if (s.K == 3 && small) {
        PropP q;
        memset(&q, 0, sizeof q);
        q.nseg = 1; q.N = s.N; q.ncols = (int)s.ld; q.ld = s.ld; q.alpha = 1.f; q.beta = 1.f;
        for (int b = 0; b < 2; ++b) {
            q.Sf[b][0] = u.Stf[b]; q.X[b][0] = dP + (2 + 2 * b) * s.PS;
            q.C[b] = dP + (1 + 2 * b) * s.PS; q.Cin[b] = q.C[b];
        }
        CKI(prop_small(q, 2, ROLE_PROPT, 0, st));
    } else if (s.K == 3) {   // d1 += S^T e2     (e2 = 2 d2, folded into Wd)
        GemmP p = gp();
        p.M = s.N; p.N = (int)s.ld; p.K = s.N;
        p.am = plain(u.ldS); p.ak = plain(1);
        p.bk = plain(s.ld); p.bn = plain(1);
        p.cm = plain(s.ld); p.cn = plain(1);
        p.nbatch = 2; p.beta = 1.f;
        if (g_precision == MCRN_BF16X3 && u.Stimg[0]) { p.Aimg[0] = u.Stimg[0]; p.Aimg[1] = u.Stimg[1]; p.aimg_n = u.simg_n; }
        for (int b = 0; b < 2; ++b) {
            p.A[b] = u.St[b];
            p.B[b] = dP + (2 + 2 * b) * s.PS;
            p.C[b] = dP + (1 + 2 * b) * s.PS;
            p.Cin[b] = p.C[b];
        }
        CKI(gemm(p, true, false, 0, ROLE_PROPT, st));
    }
    // 256 < N <= 352 (PEMS-BAY): the output-stationary kernel with column groups (ds_wide_kernel) behind the fused two-hop
    // chain, merged per cell like the N <= 256 one, instead of one tiled split-K GEMM per call
    const bool wide_ds = s.N > 256 && s.N <= PROP2_MAX_N && fused_bwd && (s.ld % 4) == 0;
    const bool ds_small_path = (small || wide_ds) && aligned16(X);
    if (cell_ds && cell_ds->nseg > 0 && !ds_small_path)
        // the update call of this cell queued its d1t x0^T / e2 x1^T segments for a merged launch; the gate call must
        // take the same branch, or those contributions to dS would be silently dropped
        FAIL("adjacency gradient: the two AGCN calls of a cell chose different paths (%d segments queued)", cell_ds->nseg);
    if (u.ds4 && !(fused_bwd && ds_small_path && s.K == 3))
        FAIL("adjacency gradient: the four-block form was planned but this call does not run the fused two-hop chain + ds_small / ds_wide");
    if (ds_small_path) {   // output-stationary adjacency-gradient kernel (prop_small.h)
        // The two AGCN calls of a cell (update first, gate second) share ONE launch: every workgroup adds both calls'
        // products to its slab partial while it sits in the accumulators, so the slab is read and written once per cell
        // instead of once per call (the slab read-modify-write was ~1/3 of this kernel's HBM bytes, profiles/r1).
        DsP q_local;
        DsP& q = cell_ds ? *cell_ds : q_local;
        if (!cell_ds || cell_ds->nseg == 0) memset(&q, 0, sizeof q);
        const int s0 = q.nseg, own = u.ds4 ? 1 : (s.K == 3 ? 2 : 1);
        const int nblk = u.ds4 ? 4 : 2;
        q.nseg = s0 + own; q.N = s.N; q.ncols = (int)s.ld; q.ld = s.ld; q.ldc = u.ldS; q.slab = u.slab;
        if (u.ds4) {
            // four blocks over the RAW d-grad planes, plane 0 of the call's input the only right operand:
            //   dA[2b] += d1_b x0^T ; dA[2b + 1] += e2_b x0^T      (dS_b = dA[2b] + dA[2b+1] S_b^T + S_b^T dA[2b+1]: model_backward)
            for (int k = 0; k < 4; ++k) {
                q.A[k][s0] = dP + (long long)(1 + k) * s.PS; q.B[k][s0] = X;
                q.C[k] = u.dS + (long long)k * s.N * u.ldS;
            }
        } else
        for (int b = 0; b < 2; ++b) {
            const long long g1 = 1 + b * (s.K - 1);
            q.A[b][s0] = dP + g1 * s.PS;       q.B[b][s0] = X;                      // d1t x0^T
            if (own == 2) { q.A[b][s0 + 1] = dP + (g1 + 1) * s.PS; q.B[b][s0 + 1] = X + g1 * s.PS; }   // e2  x1^T
            q.C[b] = u.dS + (long long)b * u.sup_stride;
        }
        if (!cell_ds || cell_ds_last) {
            const double ex = nblk * q.nseg * 2.0 * (double)s.N * s.N * (double)s.ld;
            hipStream_t ds_st = st;
            if (side) {
                if (!ready_attached) CK(hipEventRecord(g_side.ready[buf], st));
                CK(hipStreamWaitEvent(g_side.st, g_side.ready[buf], 0));
                ds_st = g_side.st;
            }
            {
                hipStream_t st = ds_st;   // MCRN_PROF_WRAP records on `st`
                fam(s.N > 256 ? (u.ds4 ? "ds_wide:four_block" : "ds_wide") : (u.ds4 ? "ds_small:four_block" : (cell_ds ? "ds_small:merged_cell" : "ds_small:per_call")));
                MCRN_PROF_WRAP(ROLE_DS, launch_ds_small(q, u.nslab, st, nblk), ex, nblk * q.nseg * 2.0 * (double)s.N * s.N * (double)s.B * s.C);
            }
            if (side) {
                const int nb_ = cell_ds ? 2 : 1;          // a merged launch reads the plane sets of both calls
                for (int i = 0; i < nb_; ++i) {
                    const int bi = cell_ds ? (buf & ~1) + i : buf;
                    CK(hipEventRecord(g_side.done[bi], g_side.st));
                    g_side.pending[bi] = true;
                    g_side.paired[bi] = cell_ds != nullptr;
                }
                g_side.any = true;
            }
        }
    } else
    {   // dS_s += d1t x0^T (+ e2 x1^T)      N x N, K = (1|2) * B*Cp, split-K into slabs
        GemmP p = gp();
        const int nk = s.K == 3 ? 2 : 1;
        p.M = s.N; p.N = s.N; p.K = (int)(nk * s.ld);
        p.am = plain(s.ld); p.ak = two((int)s.ld, s.PS, 1);
        p.bn = plain(s.ld); p.bk = two((int)s.ld, 0, 1);
        p.cm = plain(u.ldS); p.cn = plain(1);
        p.nbatch = 2; p.beta = 1.f; p.slab = u.slab;
        for (int b = 0; b < 2; ++b) {
            const long long g1 = 1 + b * (s.K - 1);
            p.A[b] = dP + g1 * s.PS; p.ak_hi[b] = s.PS;
            p.B[b] = X;              p.bk_hi[b] = g1 * s.PS;
            p.C[b] = u.dS + (long long)b * u.sup_stride;
            p.Cin[b] = p.C[b];
        }
        if (side) {   // off the critical path: only sup_bwd_core (after all BPTT) consumes the slabs
            CK(hipEventRecord(g_side.ready[buf], st));
            CK(hipStreamWaitEvent(g_side.st, g_side.ready[buf], 0));
            CKI(gemm(p, true, true, u.nslab, ROLE_DS, g_side.st));
            CK(hipEventRecord(g_side.done[buf], g_side.st));
            g_side.pending[buf] = true; g_side.paired[buf] = false; g_side.any = true;
        } else {
            CKI(gemm(p, true, true, u.nslab, ROLE_DS, st));
        }
    }
    if (fused_bwd) {
        // done above
    } else if (small) {   // dx0 = dP[0] + S1^T d1t_a + S2^T d1t_b : two K-segments into one accumulator
        PropP q;
        memset(&q, 0, sizeof q);
        q.nseg = 2; q.N = s.N; q.ncols = (int)s.ld; q.ld = s.ld; q.alpha = 1.f; q.beta = 1.f;
        q.Sf[0][0] = u.Stf[0]; q.Sf[0][1] = u.Stf[1];
        q.X[0][0] = dP + s.PS; q.X[0][1] = dP + (long long)(1 + (s.K - 1)) * s.PS;
        q.C[0] = dP; q.Cin[0] = dP;
        CKI(prop_small(q, 1, ROLE_PROPT, 0, st));
    } else if ((s.N & 3) == 0 && !(g_precision == MCRN_BF16X3 && u.Stimg[0])) {   // one K-concatenated GEMM
        GemmP p = gp();
        p.M = s.N; p.N = (int)s.ld; p.K = 2 * s.N;
        p.A[0] = u.St[0]; p.am = plain(u.ldS);
        p.ak = two(s.N, 0, 1); p.ak_hi[0] = (long long)(u.St[1] - u.St[0]);
        p.B[0] = dP + s.PS; p.bk = two(s.N, 0, s.ld); p.bk_hi[0] = (long long)(s.K - 1) * s.PS;
        p.bn = plain(1);
        p.C[0] = dP; p.Cin[0] = dP; p.cm = plain(s.ld); p.cn = plain(1);
        p.beta = 1.f;
        CKI(gemm(p, true, false, 0, ROLE_PROPT, st));
    } else {   // N % 4 != 0: the concatenated K map would leave the float4 fast path -> two accumulating launches
        for (int b = 0; b < 2; ++b) {
            GemmP p = gp();
            p.M = s.N; p.N = (int)s.ld; p.K = s.N;
            p.A[0] = u.St[b]; p.am = plain(u.ldS); p.ak = plain(1);
            if (g_precision == MCRN_BF16X3 && u.Stimg[b]) { p.Aimg[0] = u.Stimg[b]; p.aimg_n = u.simg_n; }
            p.B[0] = dP + (long long)(1 + b * (s.K - 1)) * s.PS; p.bk = plain(s.ld); p.bn = plain(1);
            p.C[0] = dP; p.Cin[0] = dP; p.cm = plain(s.ld); p.cn = plain(1);
            p.beta = 1.f;
            CKI(gemm(p, true, false, 0, ROLE_PROPT, st));
        }
    }
    return 0;
}

// ---- deferred weight gradient: slabs += X_all^T dY_all over T steps -------------------------------
// The streaming kernel (wgrad_stream.h) takes every shape of the library's bf16x3 sessions; exact-fp32 sessions and odd
// widths keep the tiled GEMM.  Returns the number of slabs written through *nslab.
static bool wgrad_streams(const Shp& s, int O, const float* Xall, const float* dYall, long long step_stride, int T) {
    // T > NSLAB_W (one slab per step at least): the tiled split-K GEMM has no such limit
    return T <= NSLAB_W && g_precision == MCRN_BF16X3 && wgrad_stream_ok(s.G, s.Cp, O) && aligned16(Xall) && aligned16(dYall) &&
           ((step_stride | s.PS) & 3) == 0;
}
static int agcn_wgrad(const Shp& s, const float* Xall, long long step_stride, int T, const float* dYall,
                      int O, float* slabs, hipStream_t st, int* nslab, bool* ones = nullptr,
                      const uint16_t* Xb = nullptr /* bf16-resident planes of these calls */, long long xb_step = 0, long long xb_plane = 0,
                      const float* Xc = nullptr /* compact input channels of planes 1 .. nb, [T][nb][R][4] */,
                      int slab_budget = NSLAB_W /* workgroups of the streaming kernel (= slabs): fewer leave CUs to a concurrent queue */) {
    // ones != null: the caller wants the column sums of dY (bias gradient) as row G*Cp of the slabs when the streaming
    // kernel runs (*ones = true), and computes them itself otherwise
    if (ones) *ones = false;
    const bool streams = wgrad_streams(s, O, Xall, dYall, step_stride, T);
    if (Xb && !streams) FAIL("weight gradient: bf16-resident planes need the streaming kernel (shape / alignment)");
    if (streams) {
        WgradP q;
        memset(&q, 0, sizeof q);
        q.Xb = Xb; q.xb_step = xb_step; q.xb_plane = xb_plane; q.H = s.H;
        q.Xc = Xb ? Xc : nullptr; q.xc_plane = s.R * 4; q.xc_step = (long long)(s.G - 1) * s.R * 4;
        q.X = Xall; q.step_stride = step_stride; q.PS = s.PS; q.Cp = s.Cp; q.G = s.G; q.T = T; q.R = s.R;
        q.dY = dYall; q.O = O; q.slabs = slabs; q.ones = ones ? 1 : 0;
        if (ones) *ones = true;
        if (slab_budget > NSLAB_W || slab_budget < T) slab_budget = NSLAB_W;
        q.cpt = slab_budget / T < 1 ? 1 : slab_budget / T;
        if (q.cpt > cdiv(s.R, 32)) q.cpt = (int)cdiv(s.R, 32);
        q.kch = (int)(cdiv(cdiv(s.R, q.cpt), 32) * 32);
        if (T * q.cpt > NSLAB_W) FAIL("weight gradient: %d steps exceed the slab capacity", T);
        *nslab = T * q.cpt;
        const double fl = 2.0 * (double)s.G * s.Cp * O * (double)T * (double)s.R;
        fam(Xb ? "wgrad_stream:bf16_planes" : "wgrad_stream");
        MCRN_PROF_WRAP(ROLE_WGRAD, launch_wgrad_stream(q, st), fl, 2.0 * (double)s.G * s.C * O * (double)T * (double)s.R);
        return 0;
    }
    *nslab = NSLAB_W_GEMM;
    CK(hipMemsetAsync(slabs, 0, (size_t)s.G * s.Cp * O * NSLAB_W_GEMM * sizeof(float), st));
    GemmP p = gp();
    p.M = s.G * s.Cp; p.N = O; p.K = (int)(T * s.R);
    p.A[0] = Xall; p.am = two(s.Cp, s.PS, 1);
    p.ak = two((int)s.R, step_stride, s.Cp); p.ak_hi[0] = step_stride;
    p.B[0] = dYall; p.bk = plain(O); p.bn = plain(1);
    p.C[0] = slabs; p.Cin[0] = slabs; p.cm = plain(O); p.cn = plain(1);
    p.beta = 1.f; p.slab = (long long)s.G * s.Cp * O;
    return gemm(p, false, false, NSLAB_W_GEMM, ROLE_WGRAD, st);
}

static const int COLSUM_CHUNK = 512;
static int colsum(const float* X, long long ld, long long rows, int C, float* part, float* out,
                  int accumulate, hipStream_t st, int chunk = COLSUM_CHUNK) {
    const int nchunk = cdiv(rows, chunk);
    LAUNCH(k_colsum_stage1, dim3(cdiv(C, 64), nchunk), dim3(256), 0, st, X, ld, rows, C, chunk, part);
    LAUNCH(k_colsum_stage2, dim3(cdiv(C, 64)), dim3(1024), 0, st, (const float*)part, nchunk, C, out, accumulate);
    return 0;
}
static size_t colsum_part_floats(long long rows, int C, int chunk = COLSUM_CHUNK) { return (size_t)cdiv(rows, chunk) * C; }
static const int MU_CHUNK = 64;       // rows per partial of the node-mean of a plane (wide grid: N / 64 x ld / 64 workgroups)
static int planes_to_bf16(const Shp& s, const float* X, int np, uint16_t* xb, uint16_t* xc, const float* mu, hipStream_t st,
                          int nvalid = -1, long long src_ps = -1, long long dst_ps = -1, long long mu_stride = 0, long long lo) {
    const long long n = (long long)np * s.Kp * (s.ldp / 8);
    LAUNCH(k_plane_to_bf16, dim3(cdiv(n, 256)), dim3(256), 0, st, X, src_ps < 0 ? s.PS : src_ps, s.N, (int)s.ld,
           nvalid < 0 ? (int)s.ld : nvalid, s.Kp, (int)s.ldp, np, reinterpret_cast<uint4*>(xb), reinterpret_cast<uint4*>(xc), mu,
           1.f / (float)s.N, mu_stride, (dst_ps < 0 ? s.PSb : dst_ps) / 8, lo / 8);
    return 0;
}
// forward: only the plain bf16 copy (operand of the propagation); the centred copies are made in the backward pass
static int plane_to_bf16(const Shp& s, const Sup& u, const float* X, uint16_t* xb, uint16_t* xc, hipStream_t st) {
    (void)u; (void)xc;
    return planes_to_bf16(s, X, 1, xb, nullptr, nullptr, st, -1);
}
// backward, once per cell stack: node-centred bf16 copies of plane 0 of EVERY AGCN call (operand of the stack's
// adjacency-gradient product): the gate calls read plane 0 of Z_t, the update calls plane 0 of Y_t; slot 2t + a.
static int centre_planes(const Shp& s, const Sup& u, const float* Zall, const float* Yall, int T, uint16_t* x0c_all, hipStream_t st) {
    const int nsamp = s.N < 64 ? s.N : 64;
    for (int a = 0; a < 2; ++a) {
        const float* X = a ? Yall : Zall;
        float* mu = u.mu + (long long)a * T * s.ld;
        LAUNCH(k_colsum_sample, dim3(cdiv(s.ld, 64), T), dim3(256), 0, st, X, s.ld, s.N, (int)s.ld, nsamp, mu, s.ZT, s.ld);
        CKI(planes_to_bf16(s, X, T, nullptr, x0c_all + (long long)a * s.PSb, mu, st, -1, s.ZT, 2 * s.PSb, s.ld));
    }
    return 0;
}

// ---- cell forward / backward cores (model/MegaCRN.py:38-48) ----------------------------------------
struct CellW { const float *Wf_g, *Wd_g, *bg, *Wf_u, *Wd_u, *bu; const uint4 *if_g = nullptr, *id_g = nullptr, *if_u = nullptr, *id_u = nullptr;
               const uint4 *wp_g = nullptr, *wp_u = nullptr; /* weight images of the streaming weight pool (wp_stream.h) */ };
// streaming weight pool on bf16-resident planes (wp_stream.h) with the library's profiling hooks
static int wp_stream(const Shp& s, const Sup& u, const float* Z, const uint16_t* Pb, const uint4* img, const float* bias, int epi,
                     float* out, float* out2, long long out2_ld, uint16_t* out2b, const float* hsrc, long long hsrc_ld,
                     const float* zr, hipStream_t st, const float* xc = nullptr, long long out2b_lo = 0) {
    WpP q;
    memset(&q, 0, sizeof q);
    q.out2b_lo = out2b ? out2b_lo : 0;
    q.Xc = xc; q.xc_plane = s.R * 4;
    q.Z = Z; q.Pb = Pb; q.PS = s.PS; q.PSh = s.R * s.H; q.R = s.R; q.Cp = s.Cp; q.H = s.H; q.d = s.d; q.nbp = u.nb;
    q.O = epi == WP_GATE ? 2 * s.H : s.H; q.Wimg = img; q.bias = bias; q.epi = epi;
    q.out = out; q.out2 = out2; q.out2_ld = out2_ld; q.out2b = out2b; q.hsrc = hsrc; q.hsrc_ld = hsrc_ld; q.zr = zr;
    const double alg = 2.0 * (double)s.R * (2.0 * s.K * s.C) * q.O;
    fam(Pb ? "wp_stream:bf16_planes" : "wp_stream");
    MCRN_PROF_WRAP(ROLE_WP, launch_wp_stream(q, st), 2.0 * (double)s.R * ((double)s.G * s.H + 16.0) * q.O, alg);
    return 0;
}

// x0b / x0c: bf16 slots of this cell's two AGCN calls (gate first, update second), or nullptr
static int cell_fwd_core(const Shp& s, const Sup& u, float* Z, float* Y, float* zr, float* hc,
                         const CellW& w, float* hnext, long long hnext_ld, hipStream_t st, uint16_t* x0b = nullptr,
                         uint16_t* x0c = nullptr, uint16_t* Pb = nullptr /* this cell's bf16 planes: gate call, update call */,
                         bool packed = false /* x0b of the gate call was emitted by the previous cell's epilogue */,
                         uint16_t* x0b_next = nullptr /* where this cell's new state goes as the next gate call's operand */,
                         const float* xc = nullptr /* this step's compact propagated input channels (both calls share them) */) {
    if (g_prop_bf16 && s.lite && Pb && x0b && w.wp_g && w.wp_u) {
        const long long PbS = (long long)u.nb * s.N * s.ldh;
        CKI(prop_fwd(s, u, Z, st, x0b, x0c, Pb, packed));
        CKI(wp_stream(s, u, Z, Pb, w.wp_g, w.bg, WP_GATE, zr, Y, s.Cp, x0b + s.PSb, nullptr, 0, nullptr, st, xc));
        CKI(prop_fwd(s, u, Y, st, x0b + s.PSb, x0c ? x0c + s.PSb : nullptr, Pb + PbS, true));
        CKI(wp_stream(s, u, Y, Pb + PbS, w.wp_u, w.bu, WP_UPDATE, hc, hnext, hnext_ld, x0b_next, Z, s.Cp, zr, st, xc));
        return 0;
    }
    if (w.wp_g && w.wp_u && g_precision == MCRN_BF16X3 && aligned16(Z) && aligned16(Y)) {
        // streaming weight pool on fp32 planes (bf16x3 arithmetic: the 1e-4 parity mode of the small graphs)
        // x3r (N > 352, hi/lo operand pairs): the GRU epilogues also emit the packed hi/lo operand of the NEXT propagation product
        // (gate: z*h -> the update call; update: h' -> the next cell's gate call), so only the first cell of a stack runs the pack pass
        const bool emit = g_x3r && g_prop_bf16 && s.hoist && x0b != nullptr;
        CKI(prop_fwd(s, u, Z, st, x0b, x0c, nullptr, emit && packed));
        CKI(wp_stream(s, u, Z, nullptr, w.wp_g, w.bg, WP_GATE, zr, Y, s.Cp, emit ? x0b + s.PSb : nullptr, nullptr, 0, nullptr, st, nullptr, s.lo_x0b));
        CKI(prop_fwd(s, u, Y, st, x0b ? x0b + s.PSb : nullptr, x0c ? x0c + s.PSb : nullptr, nullptr, emit));
        CKI(wp_stream(s, u, Y, nullptr, w.wp_u, w.bu, WP_UPDATE, hc, hnext, hnext_ld, emit ? x0b_next : nullptr, Z, s.Cp, zr, st, nullptr, s.lo_x0b));
        return 0;
    }
    CKI(prop_fwd(s, u, Z, st, x0b, x0c));
    GemmP e = gp();
    e.epi = EPI_GATE; e.C[0] = zr; e.bias = w.bg; e.hsrc = Z; e.hsrc_ld = s.Cp;
    e.out2 = Y; e.out2_ld = s.Cp; e.H = s.H;
    CKI(wp_fwd(s, Z, w.Wf_g, 2 * s.H, e, st, w.if_g));
    CKI(prop_fwd(s, u, Y, st, x0b ? x0b + s.PSb : nullptr, x0c ? x0c + s.PSb : nullptr));
    e = gp();
    e.epi = EPI_UPDATE; e.C[0] = hc; e.bias = w.bu; e.hsrc = Z; e.hsrc_ld = s.Cp; e.zr = zr;
    e.out2 = hnext; e.out2_ld = hnext_ld; e.H = s.H;
    CKI(wp_fwd(s, Y, w.Wf_u, s.H, e, st, w.if_u));
    return 0;
}

// float4 forms of the element-wise GRU backward kernels: H % 4 == 0 and every pointer 16-byte aligned (the plane row stride Cp
// and the plane stride PS are multiples of 4 floats by construction)
template <class... Ps>
static inline bool cell_bwd_vec(const Shp& s, Ps... ptrs) {
    if ((s.H & 3) || (s.Cp & 3) || (s.PS & 3)) return false;
    const void* a[] = {ptrs...};
    for (const void* p : a) if (p && !aligned16(p)) return false;
    return true;
}
// do_a / do_c: the model's BPTT loops fuse C of step t+1 with A of step t (cell_bwd_ca below) and skip them here
static int cell_bwd_core(const Shp& s, const Sup& u, const float* Z, const float* Y, const float* zr,
                         const float* hc, const CellW& w, const float* dhn, float* dU, float* dG,
                         float* dP, float* dQ, float* dacc, float* dxin, hipStream_t st, float* dTu = nullptr,
                         float* dTg = nullptr, bool do_a = true, bool do_c = true, int* xu_out = nullptr,
                         int* xg_out = nullptr, uint16_t* dPb = nullptr /* bf16 slots: gate call first, update second */,
                         int pair = 0 /* which (dP, dQ) plane-set pair the caller handed in: event slots 2*pair, 2*pair+1 */,
                         uint16_t* dPin = nullptr, long long kin = 0, int call0 = 0 /* hoisted backward: input operand, first column block of this cell */) {
    const long long RH = s.R * s.H;
    int xu = 0, xg = 0;                       // extra partial planes of plane 0 behind dTu / dTg (stride s.PS)
    if (do_a) LAUNCH(k_cell_bwd_a, dim3(cdiv(RH, 256)), dim3(256), 0, st, dhn, Z, (long long)s.Cp, zr, hc, s.H, s.R, dU, dG, dacc);
    DsP cell_ds;
    cell_ds.nseg = 0;
    // The two AGCN calls of a cell share one adjacency-gradient launch (slab read and written once per cell;
    // a launch per call was the earlier form).  With a single plane-set pair this was slower (8.82 vs 7.58 ms at METR-LA: the
    // merged launch holds both sets and the main queue waited for it); cells now alternate between two pairs
    // (7.29 vs 7.39 ms).
    DsP* cds = &cell_ds;
    const long long dPbS = dPin ? (long long)u.nb * s.PSbh : (long long)u.nb * s.PSb;     // slot of one call
    CKI(agcn_bwd_core(s, u, dU, s.H, w.Wd_u, Y, dP, st, 2 * pair, w.id_u, dTu, &xu, dPb ? dPb + dPbS : nullptr, cds, false, dPin, kin,
                      (call0 + 1) * s.B * s.d));
    // (Step B was also built INTO the loader of the gate call's streaming d-grad - one launch less per cell - and measured neutral:
    //  the kernel's 7 us of loads move into the d-grad's serial prologue, six times over; profiles/r5/experiments.md section 5.)
    const bool vec_b = cell_bwd_vec(s, dP, dTu, Z, zr, dG, dacc);
    fam(vec_b ? "k_cell_bwd:float4" : "k_cell_bwd:scalar");
    if (vec_b)
        LAUNCH(k_cell_bwd_b4, dim3(cdiv(RH / 4, 256)), dim3(256), 0, st, (const float*)dP, (const float*)dTu, xu, s.PS, (long long)s.Cp, Z, (long long)s.Cp, zr, s.H, s.R, dG, dacc);
    else
    LAUNCH(k_cell_bwd_b, dim3(cdiv(RH, 256)), dim3(256), 0, st, (const float*)dP, (const float*)dTu, xu, s.PS, (long long)s.Cp, Z, (long long)s.Cp, zr, s.H, s.R, dG, dacc);
    CKI(agcn_bwd_core(s, u, dG, 2 * s.H, w.Wd_g, Z, dQ, st, 2 * pair + 1, w.id_g, dTg, &xg, dPb, cds, true, dPin, kin, call0 * s.B * s.d));
    if (do_c) LAUNCH(k_cell_bwd_c, dim3(cdiv(s.R * s.C, 256)), dim3(256), 0, st, (const float*)dQ, (const float*)dTg, xg, (const float*)dP, (const float*)dTu, xu, s.PS, (dPin || s.state_only) ? s.H : s.Cp, (long long)s.Cp, s.H, s.d, s.R, dacc, dxin);
    if (xu_out) *xu_out = xu;
    if (xg_out) *xg_out = xg;
    return 0;
}
// C of the step just finished (its planes dP / dQ, extra planes dTu / dTg when xu / xg) + projection backward
// (decoder: Wp != null) + A of the next step to process (saved Z / zr / hc of that step)
static int cell_bwd_ca(const Shp& s, int xcols, const float* dP, const float* dQ, const float* dTu, const float* dTg, int xu, int xg,
                       const float* dout_bt, long long out_sb, long long out_sn, int use_next, const float* Wp, int od,
                       float* dgo_rows, const float* Z, const float* zr, const float* hc, float* dU, float* dG,
                       float* dacc, hipStream_t st) {
    if (cell_bwd_vec(s, dQ, dTg, dP, dTu, Z, zr, hc, dU, dG, dacc))
        LAUNCH(k_cell_bwd_ca4, dim3(cdiv(s.R * s.H / 4, 256)), dim3(256), 0, st, dQ, dTg, xg, dP,
               dTu, xu, s.PS, xcols, (long long)s.Cp, dout_bt, out_sb, out_sn, use_next, Wp, od, dgo_rows, s.B,
               Z, (long long)s.Cp, zr, hc, s.H, s.R, dU, dG, dacc);
    else
    LAUNCH(k_cell_bwd_ca, dim3(cdiv(s.R * s.H, 256)), dim3(256), 0, st, dQ, dTg, xg, dP,
           dTu, xu, s.PS, xcols, (long long)s.Cp, dout_bt, out_sb, out_sn, use_next, Wp, od, dgo_rows, s.B,
           Z, (long long)s.Cp, zr, hc, s.H, s.R, dU, dG, dacc);
    return 0;
}

// row softmax of the adaptive adjacency and its backward: one wave per row on small graphs, one workgroup per row from N = 512
static int relu_softmax_rows(const float* L, long long ldl, float* G, long long ldg, int N, hipStream_t st) {
    if (N >= 512) LAUNCH(k_relu_softmax_rows<4>, dim3(N), dim3(256), 0, st, L, ldl, G, ldg, N);
    else LAUNCH(k_relu_softmax_rows<1>, dim3(cdiv(N, 4)), dim3(256), 0, st, L, ldl, G, ldg, N);
    return 0;
}
static int relu_softmax_rows_bwd(const float* L, long long ldl, const float* G, long long ldg, const float* dS, long long ldd,
                                 int nslab, long long slab, float* dL, long long ldo, int N, hipStream_t st) {
    if (N >= 512) LAUNCH(k_relu_softmax_rows_bwd<4>, dim3(N), dim3(256), 0, st, L, ldl, G, ldg, dS, ldd, nslab, slab, dL, ldo, N);
    else LAUNCH(k_relu_softmax_rows_bwd<1>, dim3(cdiv(N, 4)), dim3(256), 0, st, L, ldl, G, ldg, dS, ldd, nslab, slab, dL, ldo, N);
    return 0;
}
// ---- supports (model/MegaCRN.py:169-172) ----------------------------------------------------------
struct SupBufs { float *E1, *E2, *L1, *L2, *g1, *g2, *St1, *St2, *dLa, *dLb, *dLs, *dE1, *dE2, *dE_s; long long ldS; uint4* frag[4]; uint4* simg[4]; int simg_n; };
static void plan_sup(Bump& b, int N, int M, int D, long long ldS, SupBufs& o) {
    size_t nn = (size_t)N * ldS, nd = (size_t)N * D;
    o.ldS = ldS;
    o.E1 = b.take<float>(nd); o.E2 = b.take<float>(nd);
    o.L1 = b.take<float>(nn); o.L2 = b.take<float>(nn);
    o.g1 = b.take<float>(nn); o.g2 = b.take<float>(nn);
    o.St1 = b.take<float>(nn); o.St2 = b.take<float>(nn);
    o.dLa = b.take<float>(nn); o.dLb = b.take<float>(nn); o.dLs = b.take<float>(nn);
    o.dE1 = b.take<float>(nd); o.dE2 = b.take<float>(nd);
    o.dE_s = b.take<float>(nd * NSLAB_E);
    for (int i = 0; i < 4; ++i) o.frag[i] = N <= PROP2_MAX_N ? b.take<uint4>(sfrag_uint4(N)) : nullptr;
    o.simg_n = (N + 3) & ~3;
    for (int i = 0; i < 4; ++i) o.simg[i] = N > 256 ? b.take<uint4>(bimg_uint4(N, N)) : nullptr;
}
static int transpose(float* dst, long long ldd, const float* src, long long lds_, const float* add,
                     long long lda, int N, hipStream_t st) {
    dim3 g(cdiv(N, 32), cdiv(N, 32));
    LAUNCH(k_transpose_add, g, dim3(256), 0, st, dst, ldd, src, lds_, add, lda, N);
    return 0;
}
// fragment-ordered bf16 hi/lo images of S1, S2, S1^T, S2^T for the adjacency-stationary kernels
static int build_frags(const float* s1, const float* s2, long long ld, int N, uint4* const* frag, hipStream_t st) {
    if (!frag[0] || N > PROP2_MAX_N) return 0;
    // all four images in one launch (round 4: four dependent 5 us launches in the step's opening chain before)
    const float* src[4] = {s1, s2, s1, s2};
    const int tr[4] = {0, 0, 1, 1};
    ++g_launches; CK(launch_sfrag_multi(src, tr, frag, 4, ld, N, st));
    return 0;
}
// scope guard: contractions issued while it lives use exact fp32 MFMA whatever the session precision
struct ExactFp32 {
    int saved;
    ExactFp32() : saved(g_precision) { g_precision = MCRN_F32; }
    ~ExactFp32() { g_precision = saved; }
};
static int sup_fwd_core(int N, int M, int D, const float* We1, const float* We2, const float* Mem,
                        const SupBufs& o, float* g1, long long ldg, float* g2, bool want_T, hipStream_t st) {
    // The logits go through relu: a logit within the bf16x3 error (1e-5) of zero would get the wrong SIGN and flip
    // its mask in the backward pass (one whole row of dWe1 / dWe2 off by 1e-2: tests/test_gpu_parity.py, H = 8,
    // mem_num = 4).  These three GEMMs are tiny and run once per step, so they are evaluated in exact fp32, like
    // the discrete top-2 choice of the memory head.
    {
    ExactFp32 exact_logits;
    for (int i = 0; i < 2; ++i) {   // E = We Mem
        GemmP p = gp();
        p.M = N; p.N = D; p.K = M;
        p.A[0] = i ? We2 : We1; p.am = plain(M); p.ak = plain(1);
        p.B[0] = Mem; p.bk = plain(D); p.bn = plain(1);
        p.C[0] = i ? o.E2 : o.E1; p.cm = plain(D); p.cn = plain(1);
        CKI(gemm(p, true, false, 0, ROLE_MISC, st));
    }
    {   // L1 = E1 E2^T
        GemmP p = gp();
        p.M = N; p.N = N; p.K = D;
        p.A[0] = o.E1; p.am = plain(D); p.ak = plain(1);
        p.B[0] = o.E2; p.bk = plain(1); p.bn = plain(D);
        p.C[0] = o.L1; p.cm = plain(o.ldS); p.cn = plain(1);
        CKI(gemm(p, true, true, 0, ROLE_MISC, st));
    }
    }   // exact_logits
    CKI(transpose(o.L2, o.ldS, o.L1, o.ldS, nullptr, 0, N, st));
    CKI(relu_softmax_rows(o.L1, o.ldS, g1, ldg, N, st));
    CKI(relu_softmax_rows(o.L2, o.ldS, g2, ldg, N, st));
    if (want_T) {
        CKI(transpose(o.St1, o.ldS, g1, ldg, nullptr, 0, N, st));
        CKI(transpose(o.St2, o.ldS, g2, ldg, nullptr, 0, N, st));
        CKI(build_frags(g1, g2, ldg, N, o.frag, st));
        if (o.simg[0] && g_precision == MCRN_BF16X3) {   // A[m][k] images: element(k, n=m) = S[m*ld + k]
            const float* src[4] = {g1, g2, o.St1, o.St2};
            const long long lds_[4] = {ldg, ldg, o.ldS, o.ldS};
            for (int i = 0; i < 4; ++i) { ++g_launches; CK(launch_bimg_build(src[i], 1LL, lds_[i], N, N, o.simg_n, 1, o.simg[i], st)); }
        }
    }
    return 0;
}
// dS given as slabs; g1,g2 as saved; writes dWe1,dWe2 directly and accumulates dMem into dMem_slab0 (+=)
static int sup_bwd_core(int N, int M, int D, const float* We1, const float* We2, const float* Mem,
                        const SupBufs& o, const float* g1, const float* g2, long long ldg,
                        const float* dS1, const float* dS2, long long ldd, int nslab, long long slab,
                        float* dWe1, float* dWe2, float* dMem_acc, hipStream_t st, int dmem_slabs = 0) {
    if (nslab > 16) {   // 64 per-workgroup slabs of ds_small: fold them 8-wide at full-chip parallelism first (11 MB per support)
        const int groups = 8;
        const long long n = (long long)N * ldd;
        LAUNCH(k_reduce_slabs_groups, dim3(cdiv(n, 256), groups), dim3(256), 0, st, const_cast<float*>(dS1), nslab, slab, n, groups);
        LAUNCH(k_reduce_slabs_groups, dim3(cdiv(n, 256), groups), dim3(256), 0, st, const_cast<float*>(dS2), nslab, slab, n, groups);
        nslab = groups;
    }
    CKI(relu_softmax_rows_bwd(o.L1, o.ldS, g1, ldg, dS1, ldd, nslab, slab, o.dLa, o.ldS, N, st));
    CKI(relu_softmax_rows_bwd(o.L2, o.ldS, g2, ldg, dS2, ldd, nslab, slab, o.dLb, o.ldS, N, st));
    CKI(transpose(o.dLs, o.ldS, o.dLb, o.ldS, o.dLa, o.ldS, N, st));   // dLs = dL1 + dL2^T
    // dE1 = dLs E2 and dE2 = dLs^T E1: N x D outputs (a few dozen tiles) over K = N: split K into NSLAB_E slabs so that the
    // launch fills the chip (at N = 1843 a single pass over K took ~80 us per product on 29 workgroups)
    for (int which = 0; which < 2; ++which) {
        CK(hipMemsetAsync(o.dE_s, 0, (size_t)NSLAB_E * N * D * sizeof(float), st));
        GemmP p = gp();
        p.M = N; p.N = D; p.K = N;
        p.A[0] = o.dLs;
        if (which == 0) { p.am = plain(o.ldS); p.ak = plain(1); } else { p.am = plain(1); p.ak = plain(o.ldS); }
        p.B[0] = which ? o.E1 : o.E2; p.bk = plain(D); p.bn = plain(1);
        p.C[0] = o.dE_s; p.Cin[0] = o.dE_s; p.cm = plain(D); p.cn = plain(1);
        p.beta = 1.f; p.slab = (long long)N * D;
        CKI(gemm(p, which == 0, false, NSLAB_E, ROLE_MISC, st));
        LAUNCH(k_reduce_slabs, dim3(cdiv((long long)N * D, 64)), dim3(1024), 0, st, which ? o.dE2 : o.dE1, (const float*)o.dE_s, NSLAB_E,
               (long long)N * D, (long long)N * D, 0);
    }
    for (int i = 0; i < 2; ++i) {
        {   // dWe = dE Mem^T
            GemmP p = gp();
            p.M = N; p.N = M; p.K = D;
            p.A[0] = i ? o.dE2 : o.dE1; p.am = plain(D); p.ak = plain(1);
            p.B[0] = Mem; p.bk = plain(1); p.bn = plain(D);
            p.C[0] = i ? dWe2 : dWe1; p.cm = plain(M); p.cn = plain(1);
            CKI(gemm(p, true, true, 0, ROLE_MISC, st));
        }
        {   // dMem += We^T dE
            GemmP p = gp();
            p.M = M; p.N = D; p.K = N;
            p.A[0] = i ? We2 : We1; p.am = plain(1); p.ak = plain(M);
            p.B[0] = i ? o.dE2 : o.dE1; p.bk = plain(D); p.bn = plain(1);
            p.C[0] = dMem_acc; p.Cin[0] = dMem_acc; p.cm = plain(D); p.cn = plain(1);
            p.beta = 1.f; p.slab = (long long)M * D;
            CKI(gemm(p, false, false, dmem_slabs, ROLE_MISC, st));   // one tile, K = N: spread over the slabs the caller reduces
        }
    }
    return 0;
}

// ---- plane-0 packing helpers ------------------------------------------------------------------
static int fill_cols(float* dst, long long dst_t, int Cp, int col0, int w, const float* src, long long sb,
                     long long stt, long long sn, int B, int N, int T, hipStream_t st) {
    if (w <= 0 || T <= 0) return 0;
    long long tot = (long long)T * N * B * w;
    LAUNCH(k_fill_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, dst, dst_t, Cp, col0, w, src, sb, stt, sn, B, N, T);
    return 0;
}
static int zero_cols(float* dst, long long dst_t, int Cp, int c0, int c1, long long R, int T, hipStream_t st) {
    if (c1 <= c0 || T <= 0) return 0;
    long long tot = (long long)T * R * (c1 - c0);
    LAUNCH(k_zero_cols, dim3(cdiv(tot, 256)), dim3(256), 0, st, dst, dst_t, Cp, c0, c1, R, T);
    return 0;
}
static int wprep(const float* W, float* Wf, float* Wd, const Shp& s, int O, hipStream_t st, uint4* imgf = nullptr,
                 uint4* imgd = nullptr) {
    long long tot = (long long)s.G * s.Cp * O;
    LAUNCH(k_wprep, dim3(cdiv(tot, 256)), dim3(256), 0, st, W, Wf, Wd, s.d, s.H, s.Cp, s.K, O, (g_prop_bf16 || s.K != 3) ? 0 : 1);
    if (imgf && g_precision == MCRN_BF16X3) {
        const int Kp = s.G * s.Cp;
        {   // weight pool: B[k = k'][n = o] = Wf[k'*O + o]
            const int npad = (O + 3) & ~3;
            ++g_launches; CK(launch_bimg_build((const float*)Wf, (long long)O, 1LL, Kp, O, npad, 0, imgf, st));
        }
        if (dgrad_stream_ok(O)) {   // d-grad, streaming kernel: Wd [(g,c')][o] in MFMA B-fragment order
            const int KS = O / 16;
            const long long tot = (long long)((Kp + 31) / 32) * KS * 64;
            ++g_launches; CK(launch_wfrag_build((const float*)Wd, (long long)O, Kp, O, KS, imgd, tot, st));
        } else {   // d-grad, tiled GEMM: B[k = o][n = k'] = Wd[k'*O + o]
            const int npad = (Kp + 3) & ~3;
            ++g_launches; CK(launch_bimg_build((const float*)Wd, 1LL, (long long)O, O, Kp, npad, 1, imgd, st));
        }
    }
    return 0;
}
static int wunprep(float* dW, const float* slabs, const Shp& s, int O, hipStream_t st, int nslab, float* dbias = nullptr) {
    long long tot = (long long)2 * s.K * s.C * O + (dbias ? O : 0);
    const long long slab = (long long)(s.G * s.Cp + (dbias ? 1 : 0)) * O;
    // four outputs per thread when every 4-group stays inside one row of both layouts and every address is 16-byte aligned
    if ((O & 3) == 0 && (slab & 3) == 0 && (((uintptr_t)dW | (uintptr_t)slabs | (uintptr_t)dbias) & 15) == 0) {
        fam("k_wunprep:float4");
        LAUNCH(k_wunprep4, dim3(cdiv(tot, 256)), dim3(1024), 0, st, dW, slabs, nslab, slab, s.d, s.H, s.Cp, s.K, O, dbias);
        return 0;
    }
    fam("k_wunprep:scalar");
    LAUNCH(k_wunprep, dim3(cdiv(tot, 64)), dim3(1024), 0, st, dW, slabs, nslab, slab, s.d, s.H, s.Cp, s.K, O, dbias);
    return 0;
}

// =================================================================================================
// whole model
// =================================================================================================
struct ModelPlan {
    Shp se, sd;
    long long ldS;
    int nslabS;
    SupBufs sup;
    float *dS;                               // [2][nslabS][N*ldS]   (ds4: [nslabS][4][N*ldS])
    bool ds4; float* dAm;                    // four-block adjacency gradient of the fused small-graph path (Sup::ds4), its reduced blocks
    float *Wf[4], *Wd[4], *dWs[4];           // enc gate, enc update, dec gate, dec update
    uint4 *imgf[4], *imgd[4];                // their pre-split tile images (weight pool / d-grad B operands)
    float *Zenc, *Yenc, *zr_e, *hc_e;
    float *Zdec, *Ydec, *zr_d, *hc_d;
    float *q_rows, *att_rows; int* ind_rows;
    float *dP, *dQ, *dTu, *dTg;
    // NPAIR plane-set pairs (pair 0 = dP, dQ): the cells of a BPTT loop rotate through them, so the helper stream's
    // adjacency-gradient launch may lag NPAIR - 1 cells behind the main stream before the main stream has to wait
    static const int NPAIR = 3, MAXPAIR = 24;
    // flat (round 5, small graphs): EVERY cell of both BPTT loops has its own pair (decoder cell t: pair t, encoder cell t: pair T_out + t),
    // so the caller's stream never waits for the helper stream's adjacency gradient before it writes a plane set - a wait on an event
    // that fired long ago still costs a barrier packet in the caller's queue (~6 us per cell).  288 GB of HBM: 1.7 GB at METR-LA.
    int npair; bool flat;
    float *dPp[MAXPAIR], *dQp[MAXPAIR];
    float *dU_e, *dG_e, *dU_d, *dG_d;
    float *dacc_e, *dacc_d, *dhn_d, *dxin_e, *dxin_d, *dgo;
    float *dval, *dsc, *dq;
    float *dWq_s, *dMem_s, *dWp_s;
    float *part, *part2, *part3;
    // MCRN_BF16: stacked bf16 adjacency and its transpose, T2 matrices, per-call bf16 operands, adjacency-gradient blocks
    bool bf16;
    bool x3r;                         // a bf16x3 session on the bf16-resident data flow: hi/lo operand pairs, three MFMAs per product (g_x3r)
    long long lo_Sstk, lo_STstk, lo_sqb, lo_xin_b, lo_xin_c;      // ... element offsets of the lo images behind the shared operands (0 otherwise)
    int nb, Kp;
    uint16_t *Sstk, *STstk, *sqb;   // sqb: one zero-padded bf16 [Kp][Kp] matrix (S for the T2 product, dT in the backward pass)
    float *T2[2], *t2part[2], *dA, *mu, *mu_part;
    uint16_t *x0b_e, *x0c_e, *x0b_d, *x0c_d, *dPb_e, *dPb_d;
    uint16_t* xin_b; float* xin_t;   // hoisted input channels: packed bf16 operand [Kp][ncp] and its fp32 product [nb*N][ncp]
    // hoisted backward of the bf16 mode: stack-wide bf16 operands of the adjacency gradient's input part, go-gradient scratch
    float *Xp_e, *Xp_d;              // compact propagated input channels of the stacks, [T][nb][R][4] (d <= 4; k_scatter_compact)
    bool bwd_hoist; long long kin_e, kin_d;
    uint16_t *dPin_e, *dPin_d, *xin_c; float* go_tmp;
    float* xin_f;                    // small graphs, Shp::hoist_fwd: scratch plane set [5][N][ncp] of the hoisted input channels
    uint16_t *Pb_e, *Pb_d;           // bf16-resident propagated planes of every AGCN call: [T][gate, update][nb][N][B*H]
    uint4* wpimg[4];                 // weight images of the streaming weight pool (enc gate, enc update, dec gate, dec update)
    size_t total;
};

static int check_dims(const mcrn_dims_t* d) {
    if (!d) FAIL("dims is NULL");
    if (d->B < 1 || d->N < 1 || d->T_in < 1 || d->T_out < 1 || d->input_dim < 1 || d->output_dim < 1 ||
        d->ycov_dim < 0 || d->H < 1 || d->mem_num < 1 || d->mem_dim < 1)
        FAIL("mcrn_dims: all sizes must be >= 1");
    if (d->cheb_k < 2 || d->cheb_k > MCRN_MAX_CHEB_K) FAIL("cheb_k must be in 2 .. %d (got %d; cheb_k = 1 is broken in the reference too)", MCRN_MAX_CHEB_K, d->cheb_k);
    if (d->precision == MCRN_BF16 && d->cheb_k > 3) FAIL("the bf16 mode builds the Chebyshev matrices [S, 2SS - I]: cheb_k 2 or 3 (got %d)", d->cheb_k);
    if (d->precision != MCRN_F32 && d->precision != MCRN_BF16X3 && d->precision != MCRN_BF16)
        FAIL("unsupported precision %d", d->precision);
    return 0;
}

static void plan_model(const mcrn_dims_t* d, char* base, ModelPlan& P) {
    Bump b{base, 0};
    const int B = d->B, N = d->N, H = d->H, D = d->mem_dim, M = d->mem_num, K = d->cheb_k;
    const int Hd = H + D, od = d->output_dim, yd = d->ycov_dim;
    // bf16x3 sessions on the large graphs (no fused two-hop kernels beyond N = 352) take the bf16 mode's data flow - stacked adjacency
    // operands, hoisted state-channel products, one adjacency-gradient product per stack - with hi/lo operand PAIRS and three MFMAs per
    // product (round 5, "x3r"): the same ~1e-5 arithmetic as the tiled bf16x3 GEMM it replaces there, on the LDS-DMA kernels of
    // gemm_bf16.h.  Needs the hoisted forward and backward (H % 32 == 0, B * input channels % 8 == 0); other shapes keep the tiled path.
    const bool x3r_try = d->precision == MCRN_BF16X3 && N > PROP2_MAX_N && K <= 3;
    P.bf16 = d->precision == MCRN_BF16 || x3r_try;
    P.se = mk_shape(B, N, d->input_dim, H, K, P.bf16);
    P.sd = mk_shape(B, N, od + yd, Hd, K, P.bf16);
    P.x3r = x3r_try && P.se.hoist && P.sd.hoist && (B * d->input_dim) % 8 == 0 && (B * (od + yd)) % 8 == 0;
    if (x3r_try && !P.x3r) {
        P.bf16 = false;
        P.se = mk_shape(B, N, d->input_dim, H, K, false);
        P.sd = mk_shape(B, N, od + yd, Hd, K, false);
    }
    if (P.x3r) P.se.lite = P.sd.lite = false;      // the propagated planes stay fp32 (bf16-resident planes would cost the 1e-4)
    P.lo_Sstk = P.lo_STstk = P.lo_sqb = P.lo_xin_b = P.lo_xin_c = 0;
    P.ldS = P.bf16 ? (N + 7) & ~7 : (N + 3) & ~3;
    P.nslabS = nslab_S(N);
    plan_sup(b, N, M, D, P.ldS, P.sup);
    // four-block adjacency gradient (Sup::ds4) wherever the fused two-hop chain and the output-stationary kernels take both cell shapes
    P.ds4 = d->precision == MCRN_BF16X3 && K == 3 && prop2_ok(N, P.se.ld, (int)P.se.ld) && prop2_ok(N, P.sd.ld, (int)P.sd.ld);
    if (P.ds4) P.nslabS = nslab_S4(N);
    P.dS = b.take<float>((size_t)(P.ds4 ? 4 : 2) * P.nslabS * N * P.ldS);   // split-K slabs of the adjacency gradient (per support / per block)
    P.dAm = P.ds4 ? b.take<float>((size_t)4 * N * P.ldS) : nullptr;         // ... the four reduced blocks
    const Shp* sh[4] = {&P.se, &P.se, &P.sd, &P.sd};
    const int Os[4] = {2 * H, H, 2 * Hd, Hd};
    for (int i = 0; i < 4; ++i) {
        size_t n = (size_t)sh[i]->G * sh[i]->Cp * Os[i];
        P.Wf[i] = b.take<float>(n);
        P.Wd[i] = b.take<float>(n);
        P.dWs[i] = b.take<float>((n + Os[i]) * NSLAB_W);   // + the column-sum row of the streaming kernel
        P.imgf[i] = b.take<uint4>(bimg_uint4(sh[i]->G * sh[i]->Cp, Os[i]));
        P.imgd[i] = b.take<uint4>(std::max(bimg_uint4(Os[i], sh[i]->G * sh[i]->Cp), wfrag_uint4(sh[i]->G * sh[i]->Cp, Os[i])));
    }
    const long long R = P.se.R;
    P.Zenc = b.take<float>((size_t)(d->T_in + 1) * P.se.ZT);
    P.Yenc = b.take<float>((size_t)d->T_in * P.se.ZT);
    P.zr_e = b.take<float>((size_t)d->T_in * R * 2 * H);
    P.hc_e = b.take<float>((size_t)d->T_in * R * H);
    P.Zdec = b.take<float>((size_t)(d->T_out + 1) * P.sd.ZT);
    P.Ydec = b.take<float>((size_t)d->T_out * P.sd.ZT);
    P.zr_d = b.take<float>((size_t)d->T_out * R * 2 * Hd);
    P.hc_d = b.take<float>((size_t)d->T_out * R * Hd);
    P.q_rows = b.take<float>((size_t)R * D);
    P.att_rows = b.take<float>((size_t)R * M);
    P.ind_rows = b.take<int>((size_t)R * 2);
    size_t zmax = (size_t)(P.se.ZT > P.sd.ZT ? P.se.ZT : P.sd.ZT);
    P.dP = b.take<float>(zmax);
    P.dQ = b.take<float>(zmax);
    P.dPp[0] = P.dP; P.dQp[0] = P.dQ;
    {
        const int want = d->T_in + d->T_out;
        // MCRN_FLAT_SETS=0 (read once): keep the rotating three-pair form - up to 8 GB less workspace on a smaller part, +1 % step time
        // (one guard wait per BPTT cell; profiles/r5/experiments.md section 12).  Both forms are parity-tested (T_in + T_out = 26 case).
        static const bool flat_off = getenv("MCRN_FLAT_SETS") && atoi(getenv("MCRN_FLAT_SETS")) == 0;
        P.flat = !flat_off && !P.bf16 && d->precision == MCRN_BF16X3 && want <= ModelPlan::MAXPAIR &&
                 (double)want * 2.0 * (double)zmax * sizeof(float) <= 8e9;
        P.npair = P.flat ? want : ModelPlan::NPAIR;
    }
    for (int i = 1; i < P.npair; ++i) { P.dPp[i] = b.take<float>(zmax); P.dQp[i] = b.take<float>(zmax); }
    {
        size_t pmax = (size_t)(P.se.PS > P.sd.PS ? P.se.PS : P.sd.PS);
        P.dTu = b.take<float>(pmax * PROPT_MAX_X);        // extra partial planes of plane 0 (K splits 1 .. 3 / second support)
        P.dTg = b.take<float>(pmax * PROPT_MAX_X);
    }
    P.dU_e = b.take<float>((size_t)d->T_in * R * H);
    P.dG_e = b.take<float>((size_t)d->T_in * R * 2 * H);
    P.dU_d = b.take<float>((size_t)d->T_out * R * Hd);
    P.dG_d = b.take<float>((size_t)d->T_out * R * 2 * Hd);
    P.dacc_e = b.take<float>((size_t)R * H);
    P.dacc_d = b.take<float>((size_t)R * Hd);
    P.dhn_d = b.take<float>((size_t)R * Hd);
    P.dxin_e = b.take<float>((size_t)R * d->input_dim);
    P.dxin_d = b.take<float>((size_t)R * (od + yd));
    P.dgo = b.take<float>((size_t)d->T_out * R * od);
    P.dval = b.take<float>((size_t)R * D);
    P.dsc = b.take<float>((size_t)R * M);
    P.dq = b.take<float>((size_t)R * D);
    P.dWq_s = b.take<float>((size_t)NSLAB_T * H * D);
    P.dMem_s = b.take<float>((size_t)NSLAB_T * M * D);
    P.dWp_s = b.take<float>((size_t)NSLAB_T * od * Hd);
    int Tm = d->T_in > d->T_out ? d->T_in : d->T_out;
    P.part = b.take<float>(colsum_part_floats((long long)Tm * R, 2 * Hd) + 1024);
    P.part2 = b.take<float>(colsum_part_floats((long long)Tm * R, 2 * Hd) + 1024);
    P.part3 = b.take<float>(colsum_part_floats((long long)Tm * R, 2 * Hd) + 1024);   // (second helper queue)
    P.nb = 2 * (K - 1); P.Kp = (N + 63) & ~63;
    P.Sstk = P.STstk = P.sqb = nullptr; P.T2[0] = P.T2[1] = P.t2part[0] = P.t2part[1] = P.dA = P.mu = P.mu_part = nullptr;
    P.x0b_e = P.x0c_e = P.x0b_d = P.x0c_d = P.dPb_e = P.dPb_d = nullptr;
    P.xin_b = nullptr; P.xin_t = nullptr;
    P.Pb_e = P.Pb_d = nullptr;
    P.Xp_e = P.Xp_d = nullptr;
    P.bwd_hoist = false; P.kin_e = P.kin_d = 0; P.dPin_e = P.dPin_d = P.xin_c = nullptr; P.go_tmp = nullptr;
    for (int i = 0; i < 4; ++i) P.wpimg[i] = nullptr;
    if (P.bf16) {
        // x3r: every bf16 operand is a hi image followed by its lo image (lo offset = the hi image's size, multiples of 8 elements)
        const size_t lm = P.x3r ? 2 : 1;
        auto take16 = [&](size_t n, long long* lo) { n = (n + 7) & ~(size_t)7; uint16_t* q = b.take<uint16_t>(lm * n); if (lo) *lo = P.x3r ? (long long)n : 0; return q; };
        if (P.se.lite && P.sd.lite) {
            P.Pb_e = b.take<uint16_t>((size_t)2 * d->T_in * P.nb * N * P.se.ldh + 64);     // (+ slack: 16-byte reads of 8-byte quads)
            P.Pb_d = b.take<uint16_t>((size_t)2 * d->T_out * P.nb * N * P.sd.ldh + 64);
            if (d->input_dim <= 4 && od + yd <= 4) {
                P.Xp_e = b.take<float>((size_t)d->T_in * P.nb * R * 4 + 64);
                P.Xp_d = b.take<float>((size_t)d->T_out * P.nb * R * 4 + 64);
            }
        }
        if ((P.se.lite && P.sd.lite) || P.x3r) {
            const int bwe = B * d->input_dim, bwd_ = B * (od + yd);
            P.bwd_hoist = (bwe % 8) == 0 && (bwd_ % 8) == 0;
            if (P.bwd_hoist) {
                P.kin_e = ((long long)2 * d->T_in * bwe + 63) & ~63LL;
                P.kin_d = ((long long)2 * d->T_out * bwd_ + 63) & ~63LL;
                P.dPin_e = take16((size_t)(P.nb * N + 64) * P.kin_e, &P.se.lo_dPin);
                P.dPin_d = take16((size_t)(P.nb * N + 64) * P.kin_d, &P.sd.lo_dPin);
                P.xin_c = take16((size_t)(N + 64) * (P.kin_e > P.kin_d ? P.kin_e : P.kin_d), &P.lo_xin_c);
                P.go_tmp = b.take<float>((size_t)GO_MAX_SPLIT * N * 2 * bwd_ + 64);
            }
        }
        {
            const long long ce = (long long)d->T_in * B * d->input_dim, cd = (long long)d->T_out * B * (od + yd);
            const size_t ncp = (size_t)(((ce > cd ? ce : cd) + 7) & ~7LL) + 8;
            P.xin_b = take16((size_t)P.Kp * ncp, &P.lo_xin_b);
            P.xin_t = b.take<float>((size_t)8 /* HOIST_MAX_SPLIT */ * P.nb * N * ncp);
        }
        P.Sstk = take16((size_t)P.nb * N * P.Kp, &P.lo_Sstk);
        P.STstk = take16((size_t)N * P.nb * P.Kp, &P.lo_STstk);
        P.sqb = take16((size_t)P.Kp * P.Kp, &P.lo_sqb);
        for (int i = 0; i < 2; ++i) P.T2[i] = K == 3 ? b.take<float>((size_t)N * P.ldS) : nullptr;
        for (int i = 0; i < 2; ++i) P.t2part[i] = K == 3 ? b.take<float>((size_t)N * P.ldS) : nullptr;   // second K split of the N^3 products
        P.dA = b.take<float>((size_t)P.nb * N * P.ldS);
        const long long ldm = P.se.ld > P.sd.ld ? P.se.ld : P.sd.ld;
        P.mu = b.take<float>((size_t)2 * (d->T_in > d->T_out ? d->T_in : d->T_out) * ldm);
        P.mu_part = b.take<float>(colsum_part_floats(N, (int)ldm, MU_CHUNK) + 1024);
        P.x0b_e = take16((size_t)2 * d->T_in * P.se.PSb, &P.se.lo_x0b);
        P.x0c_e = take16((size_t)2 * d->T_in * P.se.PSb, &P.se.lo_x0c);
        P.x0b_d = take16((size_t)2 * d->T_out * P.sd.PSb, &P.sd.lo_x0b);
        P.x0c_d = take16((size_t)2 * d->T_out * P.sd.PSb, &P.sd.lo_x0c);
        P.dPb_e = take16((size_t)2 * d->T_in * P.nb * P.se.PSb, &P.se.lo_dPb);
        P.dPb_d = take16((size_t)2 * d->T_out * P.nb * P.sd.PSb, &P.sd.lo_dPb);
    }
    P.xin_f = nullptr;
    {
        // forward hoisting of the decoder's input channels on the fused two-hop path (see Shp::hoist_fwd); the encoder gains nothing
        // from it (4352 and 4096 columns are both one round of 128 - 136 workgroups: profiles/r4/experiments.md)
        static const bool hf_off = getenv("MCRN_HOIST_FWD") && atoi(getenv("MCRN_HOIST_FWD")) == 0;
        // ... where it saves a PASS of the fused kernels: the wide variant (N > 256: 64-column units only) walks the decoder's
        // 132 units of PEMS-BAY in two passes of 66 workgroups per support and its 128 state-only units in one (54.6 -> 29 us
        // forward, 65.5 -> 31 us backward in tools/kbench/prop1_test).  At N <= 256 the full width already fits one pass of
        // 96-column units (METR-LA: 88 units) and the once-per-stack product + the gathered backward hop cost what the shorter
        // launches return (measured 10 735 samples/s without, 10 659 forward only, 10 326 both: profiles/r4/experiments.md);
        // MCRN_HOIST_FWD=2 forces it there (tests).
        static const int hf_env = getenv("MCRN_HOIST_FWD") ? atoi(getenv("MCRN_HOIST_FWD")) : 1;
        const int NFp = (N + 31) / 32, u2f = cdiv(P.sd.ld, 64), u3f = cdiv(P.sd.ld, 96), ust = B * (Hd / 64);
        const int passes_full = (NFp <= 8 && u3f <= 128) ? 1 : cdiv(u2f, 128), passes_state = cdiv(ust > 0 ? ust : 1, 128);
        P.sd.hoist_fwd = !hf_off && d->precision == MCRN_BF16X3 && K == 3 && prop2_ok(N, P.sd.ld, (int)P.sd.ld) &&
                         (Hd % 64) == 0 && P.sd.Cp > Hd && (passes_state < passes_full || hf_env == 2);
        P.sd.hoist_bwd = P.sd.hoist_fwd;
        if (P.sd.hoist_fwd) {
            const long long cd = (long long)d->T_out * B * (P.sd.Cp - Hd);
            const size_t ncp = (size_t)((cd + 3) & ~3LL) + 4;
            P.xin_f = b.take<float>((size_t)5 * N * ncp);
        }
    }
    {
        // streaming weight pool (wp_stream.h): bf16x3 sessions (fp32 planes) and the bf16 mode (bf16-resident planes when lite)
        const bool ok = d->precision != MCRN_F32 && wp_stream_ok(H, d->input_dim, P.nb, H) && wp_stream_ok(H, d->input_dim, P.nb, 2 * H) &&
                        wp_stream_ok(Hd, od + yd, P.nb, Hd) && wp_stream_ok(Hd, od + yd, P.nb, 2 * Hd);
        if (ok)
            for (int i = 0; i < 4; ++i) P.wpimg[i] = b.take<uint4>(wp_img_uint4(i < 2 ? H : Hd, P.nb, Os[i]));
    }
    P.total = (b.off + 255) & ~(size_t)255;
}

static Sup model_sup(const ModelPlan& P, int N) {
    Sup u;
    u.S[0] = P.sup.g1; u.S[1] = P.sup.g2;
    u.St[0] = P.sup.St1; u.St[1] = P.sup.St2;
    u.Sf[0] = P.sup.frag[0]; u.Sf[1] = P.sup.frag[1]; u.Stf[0] = P.sup.frag[2]; u.Stf[1] = P.sup.frag[3];
    u.Simg[0] = P.sup.simg[0]; u.Simg[1] = P.sup.simg[1]; u.Stimg[0] = P.sup.simg[2]; u.Stimg[1] = P.sup.simg[3];
    u.simg_n = P.sup.simg_n;
    u.ldS = P.ldS;
    u.dS = P.dS; u.nslab = P.nslabS; u.slab = (long long)N * P.ldS;
    u.sup_stride = (long long)P.nslabS * u.slab;
    if (P.ds4) { u.ds4 = true; u.slab = (long long)4 * N * P.ldS; u.sup_stride = 0; }   // one slab = the four blocks side by side
    u.Sstk = P.Sstk; u.STstk = P.STstk; u.Kp = P.Kp; u.nb = P.nb; u.mu = P.mu; u.mu_part = P.mu_part;
    u.lo_Sstk = P.lo_Sstk; u.lo_STstk = P.lo_STstk; u.lo_xin_b = P.lo_xin_b;
    return u;
}

// K splits of the N x N x N products of the T2 matrices and of their chain rule: 2 when the unsplit product leaves more than
// a third of the CUs without a tile (and K is long enough for the launcher to make exactly 2, see bf16_eff_splits)
static int t2_splits(int N) {
    const long long tiles = (long long)cdiv(N, 256) * cdiv(N, 128);
    return !bf16_cfg_is_sk(g_force_cfg_bf16) && tiles <= 170 && N >= 512 && bf16_eff_splits(1, N, 2) == 2 ? 2 : 1;
}
// MCRN_BF16, once per forward: the stacked bf16 operands.  T2(S) = 2 S S - I (model/MegaCRN.py:20-22) is itself a
// bf16-resident product: A = the S block of the stack just built (rows, K-contiguous), B = the same block read as [k][n].
static int build_stacks(const ModelPlan& P, const Sup& u, int N, int K, hipStream_t st) {
    const int nb = P.nb;
    const dim3 g(cdiv(P.Kp, 32), cdiv(N, 32));
    auto put = [&](const float* M_, int blk) -> int {
        LAUNCH(k_stack_build, g, dim3(256), 0, st, M_, P.ldS, N, P.Kp, 0, P.Sstk, (long long)P.Kp, (long long)blk * N, 0LL, P.lo_Sstk);
        LAUNCH(k_stack_build, g, dim3(256), 0, st, M_, P.ldS, N, P.Kp, 1, P.STstk, (long long)nb * P.Kp, 0LL, (long long)blk * P.Kp, P.lo_STstk);
        return 0;
    };
    for (int sidx = 0; sidx < 2; ++sidx) {
        const float* S = sidx ? P.sup.g2 : P.sup.g1;
        const int blk = sidx * (K - 1);
        CKI(put(S, blk));
        if (K == 3) {
            // S as a zero-padded [Kp][Kp] bf16 matrix: the [k][n] operand must be finite for k up to Kp
            Shp t; t.N = N; t.ld = P.ldS; t.PS = (long long)N * P.ldS; t.Kp = P.Kp; t.ldp = P.Kp; t.PSb = (long long)P.Kp * P.Kp;
            CKI(planes_to_bf16(t, S, 1, P.sqb, nullptr, nullptr, st, N, -1, -1, 0, P.lo_sqb));
            Bf16GemmP q = bgp(u);
            q.A = P.sqb; q.am = rm_plain(P.Kp); q.M = N;
            q.B = P.sqb; q.ldb = P.Kp; q.N = (N + 7) & ~7;        // columns N .. are zero padding
            q.nseg = 1; q.seg_len = N;
            x3_terms(q, P.lo_sqb, P.lo_sqb);
            q.C = P.T2[sidx]; q.cm = rm_plain(P.ldS); q.alpha = 2.f;
            // N x N output = 120 tiles of 256 x 128 at N = 1843 (half a chip, 40 - 46 us per product in round 3): K is split in
            // two, the second half lands in a scratch matrix that the "- I" pass folds in
            const int ns = t2_splits(N);
            if (ns == 2) {
                q.slab = P.t2part[0] - P.T2[sidx];
                CKI(bf16_gemm(q, true, 2, ROLE_MISC, 0, st));
                LAUNCH(k_fold_splits, dim3(cdiv((long long)N * P.ldS, 256)), dim3(256), 0, st, P.T2[sidx], (const float*)P.t2part[0],
                       (const float*)nullptr, P.ldS, N, 1);
            } else {
            CKI(bf16_gemm(q, true, 1, ROLE_MISC, 0, st));
            LAUNCH(k_sub_eye, dim3(cdiv(N, 256)), dim3(256), 0, st, P.T2[sidx], P.ldS, N);
            }
            CKI(put(P.T2[sidx], blk + 1));
        }
    }
    return 0;
}
// MCRN_BF16, once per backward: chain rule of T2 = 2 S S - I onto S, in place in the S blocks of dA
//   dS = dA[S] + 2 (dT S^T + S^T dT),  dT = dA[T2]   - two bf16-resident products per support
static int t2_backward(const ModelPlan& P, const Sup& u, int N, int K, hipStream_t st) {
    if (K != 3) return 0;
    for (int sidx = 0; sidx < 2; ++sidx) {
        float* dS = P.dA + (long long)(2 * sidx) * N * P.ldS;
        const float* dT = P.dA + (long long)(2 * sidx + 1) * N * P.ldS;
        uint16_t* dTb = P.sqb;              // dT as a zero-padded bf16 [Kp][Kp] matrix (rows K-contiguous / [k][n])
        {
            Shp t; t.N = N; t.ld = P.ldS; t.PS = (long long)N * P.ldS; t.Kp = P.Kp; t.ldp = P.Kp; t.PSb = (long long)P.Kp * P.Kp;
            CKI(planes_to_bf16(t, dT, 1, dTb, nullptr, nullptr, st, N, -1, -1, 0, P.lo_sqb));
        }
        const uint16_t* Sb = P.Sstk + (long long)(2 * sidx) * N * P.Kp;          // S rows
        const uint16_t* STb = P.STstk + (long long)(2 * sidx) * P.Kp;            // S^T rows (row stride nb*Kp)
        const int ns = t2_splits(N);      // 2: the second K half of each product lands in a scratch matrix, folded in below
        {   // dS += 2 dT S^T :  B(k, n) = S[n][k]  ->  NT with B = S rows
            Bf16GemmP q = bgp(u);
            q.A = dTb; q.am = rm_plain(P.Kp); q.M = N;
            q.B = Sb; q.bm = rm_plain(P.Kp); q.N = N;
            q.nseg = 1; q.seg_len = N;
            q.C = dS; q.Cin = dS; q.cm = rm_plain(P.ldS); q.alpha = 2.f; q.beta = 1.f;
            x3_terms(q, P.lo_sqb, P.lo_Sstk);
            if (ns == 2) { q.slab = P.t2part[0] - dS; q.cin_first_only = 1; }
            CKI(bf16_gemm(q, false, ns, ROLE_MISC, 0, st));
        }
        {   // dS += 2 S^T dT :  A = S^T rows, B = dT as [k][n]
            Bf16GemmP q = bgp(u);
            q.A = STb; q.am = rm_plain((long long)P.nb * P.Kp); q.M = N;
            q.B = dTb; q.ldb = P.Kp; q.N = (N + 7) & ~7;
            q.nseg = 1; q.seg_len = N;
            q.C = dS; q.Cin = dS; q.cm = rm_plain(P.ldS); q.alpha = 2.f; q.beta = 1.f;
            x3_terms(q, P.lo_STstk, P.lo_sqb);
            if (ns == 2) { q.slab = P.t2part[1] - dS; q.cin_first_only = 1; }
            CKI(bf16_gemm(q, true, ns, ROLE_MISC, 0, st));
        }
        if (ns == 2)
            LAUNCH(k_fold_splits, dim3(cdiv((long long)N * P.ldS, 256)), dim3(256), 0, st, dS, (const float*)P.t2part[0],
                   (const float*)P.t2part[1], P.ldS, N, 0);
    }
    return 0;
}

// q = h Wq and sc = q Mem^T on the MFMA GEMM, then one light row kernel (softmax, top-2, value, outputs)
static int memory_fwd_launch(const float* h, long long ldh, const float* Wq, const float* Mem, int B, int N,
                             int H, int M, int D, float* q_rows, float* att_rows, float* sc_rows, int* ind_rows,
                             float* s0, long long lds0, float* val, float* q, float* pos, float* neg, int* ind_bnc,
                             hipStream_t st) {
    const long long R = (long long)N * B;
    // the top-2 prototype choice is discrete: evaluate query and scores in exact fp32 so that near-ties
    // resolve like the reference's fp32 path (the bf16x3 error of 1e-5 would flip some of them)
    const int saved_prec = g_precision;
    g_precision = MCRN_F32;
    {
        GemmP p = gp();
        p.M = (int)R; p.N = D; p.K = H;
        p.A[0] = h; p.am = plain(ldh); p.ak = plain(1);
        p.B[0] = Wq; p.bk = plain(D); p.bn = plain(1);
        p.C[0] = q_rows; p.cm = plain(D); p.cn = plain(1);
        const int rc_ = gemm(p, true, false, 0, ROLE_MISC, st);
        if (rc_) { g_precision = saved_prec; return rc_; }
    }
    {   // raw scores (sc_rows is scratch; the backward pass reuses it for d(score))
        GemmP p = gp();
        p.M = (int)R; p.N = M; p.K = D;
        p.A[0] = q_rows; p.am = plain(D); p.ak = plain(1);
        p.B[0] = Mem; p.bk = plain(1); p.bn = plain(D);
        p.C[0] = sc_rows; p.cm = plain(M); p.cn = plain(1);
        const int rc_ = gemm(p, true, true, 0, ROLE_MISC, st);
        g_precision = saved_prec;
        CKI(rc_);
    }
    int nb = cdiv(R, 4);
    if (nb > 4096) nb = 4096;
    LAUNCH(k_memory_rows, dim3(nb), dim3(256), 0, st, (const float*)q_rows, (const float*)sc_rows, Mem, h, ldh, B, N, H,
           M, D, att_rows, ind_rows, s0, lds0, val, q, pos, neg, ind_bnc);
    return 0;
}
static int memory_bwd_rows_launch(const float* dval_rows, long long ldv, int c0, const float* dval_bnc,
                                  const float* dq_bnc, const float* att_rows, const float* Mem, int B, int N,
                                  int M, int D, float* dval_out, float* dsc, float* dq, hipStream_t st) {
    if (M <= 64 && D <= 256 && (size_t)M * D * sizeof(float) <= 64 * 1024) {   // one wave per row
        const long long R = (long long)N * B;
        const int blocks = (int)std::min<long long>(cdiv(R, 4), 2048);
        LAUNCH(k_memory_bwd_rows_w, dim3(blocks), dim3(256), (size_t)M * D * sizeof(float), st, dval_rows, ldv, c0, dval_bnc,
               dq_bnc, att_rows, Mem, B, N, M, D, dval_out, dsc, dq);
        return 0;
    }
    const int nt = 64;
    size_t shm = ((size_t)M * D + (size_t)(D + M) * nt) * sizeof(float);
    if (shm > 160 * 1024) FAIL("memory head too large for LDS");
    if (shm > 64 * 1024)
        CK(hipFuncSetAttribute((const void*)k_memory_bwd_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    LAUNCH(k_memory_bwd_rows, dim3(cdiv((long long)N * B, nt)), dim3(nt), shm, st, dval_rows, ldv, c0, dval_bnc,
           dq_bnc, att_rows, Mem, B, N, M, D, dval_out, dsc, dq);
    return 0;
}
// dh (+)= dq Wq^T ; dWq slabs += h^T dq ; dMem slabs += att^T dval + dsc^T q
static int memory_bwd_gemms(const float* h, long long ldh, const float* Wq, long long R, int H, int M, int D,
                            const float* att, const float* q_rows, const float* dval, const float* dsc,
                            const float* dq, float* dh, long long lddh, bool dh_accumulate, float* dWq_s,
                            float* dMem_s, hipStream_t st) {
    {
        GemmP p = gp();
        p.M = (int)R; p.N = H; p.K = D;
        p.A[0] = dq; p.am = plain(D); p.ak = plain(1);
        p.B[0] = Wq; p.bk = plain(1); p.bn = plain(D);
        p.C[0] = dh; p.cm = plain(lddh); p.cn = plain(1);
        if (dh_accumulate) { p.Cin[0] = dh; p.beta = 1.f; }
        CKI(gemm(p, true, true, 0, ROLE_MISC, st));
    }
    {
        GemmP p = gp();
        p.M = H; p.N = D; p.K = (int)R;
        p.A[0] = h; p.am = plain(1); p.ak = plain(ldh);
        p.B[0] = dq; p.bk = plain(D); p.bn = plain(1);
        p.C[0] = dWq_s; p.Cin[0] = dWq_s; p.cm = plain(D); p.cn = plain(1);
        p.beta = 1.f; p.slab = (long long)H * D;
        CKI(gemm(p, false, false, NSLAB_T, ROLE_MISC, st));
    }
    for (int i = 0; i < 2; ++i) {
        GemmP p = gp();
        p.M = M; p.N = D; p.K = (int)R;
        p.A[0] = i ? dsc : att; p.am = plain(1); p.ak = plain(M);
        p.B[0] = i ? q_rows : dval; p.bk = plain(D); p.bn = plain(1);
        p.C[0] = dMem_s; p.Cin[0] = dMem_s; p.cm = plain(D); p.cn = plain(1);
        p.beta = 1.f; p.slab = (long long)M * D;
        CKI(gemm(p, false, false, NSLAB_T, ROLE_MISC, st));
    }
    return 0;
}

static int model_forward(const mcrn_dims_t* d, const mcrn_params_t* p, const float* x, const float* ycov,
                         const float* labels, const int* teacher, char* ws, float* output, float* h_att,
                         float* query, float* pos, float* neg, hipStream_t st) {
    ModelPlan P;
    plan_model(d, ws, P);
    X3rScope x3scope(P.x3r);
    const int B = d->B, N = d->N, H = d->H, D = d->mem_dim, M = d->mem_num;
    const int Hd = H + D, od = d->output_dim, yd = d->ycov_dim, din = d->input_dim;
    const int Ti = d->T_in, To = d->T_out;
    const Shp &se = P.se, &sd = P.sd;
    const long long R = se.R;
    // forward hoisting of the decoder's input channels (Shp::hoist_fwd) only pays when steps beyond the first are teacher-forced
    // (evaluation and the late curriculum run every step at full width, with no once-per-stack product)
    // teacher flags without labels: the forward pass would feed the projection back while the backward pass (which never sees
    // `labels`) would treat the step as teacher-forced - refused here, so the two passes always agree on every step's go symbol
    if (teacher && !labels)
        for (int t = 0; t < To; ++t)
            if (teacher[t]) FAIL("model_forward: teacher[%d] is set but labels is NULL (model/MegaCRN.py:188-191 needs labels[:, t])", t);
    bool any_teacher = false;
    for (int t = 0; t + 1 < To; ++t) any_teacher |= teacher && teacher[t];
    // The step opens with ~40 tiny, mutually independent preparation launches (supports, weight images, input packing:
    // ~0.45 ms of the 6.7 ms METR-LA step, each 5 us of work behind 7 us of launch gap).  Everything that depends on
    // the WEIGHTS and the decoder's inputs only goes to the helper stream and is joined before the first cell.
    hipStream_t ps = st;
    const bool forked = g_use_side && !g_tuning && g_prof.role < 0;
    if (forked) {
        CKI(side_init());
        CK(hipEventRecord(g_side.fork, st));
        CK(hipStreamWaitEvent(g_side.st, g_side.fork, 0));
        ps = g_side.st;
    }
    const float* Wsrc[4] = {p->enc_gate_w, p->enc_update_w, p->dec_gate_w, p->dec_update_w};
    const int Os[4] = {2 * H, H, 2 * Hd, Hd};
    for (int i = 0; i < 4; ++i) CKI(wprep(Wsrc[i], P.Wf[i], P.Wd[i], i < 2 ? se : sd, Os[i], ps, P.imgf[i], P.imgd[i]));
    const bool lite = P.bf16 && P.Pb_e != nullptr;
    if ((lite || P.x3r) && P.Kp > N) {
        // the propagation operands [Kp][B*H] that the weight-pool epilogues emit cover the N data rows only: their pad
        // rows must be finite (gemm_bf16.h contract: they meet the zero K-padding of the stacked adjacency); x3r: the lo images too
        for (int e_ = 0; e_ < 2; ++e_) {
            const Shp& s_ = e_ ? sd : se;
            const long long np_ = (long long)2 * (e_ ? To : Ti), per = (long long)(s_.Kp - N) * (s_.ldh / 8);
            LAUNCH(k_zero_pad_rows, dim3(cdiv(per * np_, 256)), dim3(256), 0, ps, reinterpret_cast<uint4*>(e_ ? P.x0b_d : P.x0b_e),
                   s_.PSb / 8, N, s_.Kp, (int)(s_.ldh / 8), np_);
            if (s_.lo_x0b > 0)
                LAUNCH(k_zero_pad_rows, dim3(cdiv(per * np_, 256)), dim3(256), 0, ps, reinterpret_cast<uint4*>((e_ ? P.x0b_d : P.x0b_e) + s_.lo_x0b),
                       s_.PSb / 8, N, s_.Kp, (int)(s_.ldh / 8), np_);
        }
    }
    const bool wps = P.wpimg[0] != nullptr;
    if (lite && !wps) FAIL("bf16-resident planes need the streaming weight pool");
    if (wps)
        for (int i = 0; i < 4; ++i) {
            const Shp& sh_ = i < 2 ? se : sd;
            ++g_launches;
            CK(launch_wp_img_build(P.Wf[i], sh_.Cp, sh_.H, sh_.d, P.nb, Os[i], sh_.R, P.wpimg[i], ps, 0));
        }
    // decoder input columns (:181-183): covariates, zero pad columns, go symbol = 0; the state columns of Zdec[0] are
    // written by the memory head after the encoder
    if (yd > 0) {
        CKI(fill_cols(P.Zdec, sd.ZT, sd.Cp, Hd + od, yd, ycov, (long long)To * N * yd, (long long)N * yd, yd, B, N, To, ps));
        CKI(fill_cols(P.Ydec, sd.ZT, sd.Cp, Hd + od, yd, ycov, (long long)To * N * yd, (long long)N * yd, yd, B, N, To, ps));
    }
    CKI(zero_cols(P.Zdec, sd.ZT, sd.Cp, sd.C, sd.Cp, R, To, ps));
    CKI(zero_cols(P.Ydec, sd.ZT, sd.Cp, sd.C, sd.Cp, R, To, ps));
    CKI(zero_cols(P.Zdec, sd.ZT, sd.Cp, Hd, Hd + od, R, 1, ps));   // go = 0 (:182)
    CKI(zero_cols(P.Ydec, sd.ZT, sd.Cp, Hd, Hd + od, R, 1, ps));
    if ((sd.hoist || sd.hoist_fwd) && To > 1) {
        // the go symbol of step t+1 is labels[:, t] wherever step t is teacher-forced (:188-191): known now, so it takes
        // part in the hoisted propagation of the decoder's input channels; the other steps' go columns are zero until
        // their projection writes them (their planes are then propagated per step, below)
        if (labels) {
            CKI(fill_cols(P.Zdec + sd.ZT, sd.ZT, sd.Cp, Hd, od, labels, (long long)To * N * od, (long long)N * od, od, B, N, To - 1, ps));
            CKI(fill_cols(P.Ydec + sd.ZT, sd.ZT, sd.Cp, Hd, od, labels, (long long)To * N * od, (long long)N * od, od, B, N, To - 1, ps));
        } else {
            CKI(zero_cols(P.Zdec + sd.ZT, sd.ZT, sd.Cp, Hd, Hd + od, R, To - 1, ps));
            CKI(zero_cols(P.Ydec + sd.ZT, sd.ZT, sd.Cp, Hd, Hd + od, R, To - 1, ps));
        }
    }
    CKI(sup_fwd_core(N, M, D, p->We1, p->We2, p->Memory, P.sup, P.sup.g1, P.ldS, P.sup.g2, !P.bf16, st));
    Sup u = model_sup(P, N);
    if (P.bf16) CKI(build_stacks(P, u, N, d->cheb_k, st));
    // ---- encoder (MegaCRN.py:65-83): inputs for all t packed once
    CKI(fill_cols(P.Zenc, se.ZT, se.Cp, H, din, x, (long long)Ti * N * din, (long long)N * din, din, B, N, Ti, st));
    CKI(fill_cols(P.Yenc, se.ZT, se.Cp, H, din, x, (long long)Ti * N * din, (long long)N * din, din, B, N, Ti, st));
    CKI(zero_cols(P.Zenc, se.ZT, se.Cp, se.C, se.Cp, R, Ti, st));
    CKI(zero_cols(P.Yenc, se.ZT, se.Cp, se.C, se.Cp, R, Ti, st));
    CKI(zero_cols(P.Zenc, se.ZT, se.Cp, 0, H, R, 1, st));   // init_hidden = 0 (:50-51)
    if (forked) {
        CK(hipEventRecord(g_side.join, g_side.st));
        CK(hipStreamWaitEvent(st, g_side.join, 0));
    }
    if (P.bf16) {   // t-invariant part of the propagation: the input channels of every step, once per stack
        CKI(hoist_inputs(se, u, P.Zenc, P.Yenc, Ti, H, din, P.xin_b, P.xin_t, st, P.Xp_e));
        CKI(hoist_inputs(sd, u, P.Zdec, P.Ydec, To, Hd, od + yd, P.xin_b, P.xin_t, st, P.Xp_d));
    } else if (sd.hoist_fwd && any_teacher) {
        // decoder input / pad channels of every step in one product (steps whose go symbol is not known yet contribute zeros here
        // and are propagated at full width below, which rewrites their columns)
        CKI(hoist_inputs_small(sd, u, P.Zdec, P.Ydec, To, Hd, sd.Cp - Hd, P.xin_f, st));
    }
    CellW we{P.Wf[0], P.Wd[0], p->enc_gate_b, P.Wf[1], P.Wd[1], p->enc_update_b, P.imgf[0], P.imgd[0], P.imgf[1], P.imgd[1]};
    if (wps) { we.wp_g = P.wpimg[0]; we.wp_u = P.wpimg[1]; }
    const long long PbS_e = (long long)P.nb * N * se.ldh, PbS_d = (long long)P.nb * N * sd.ldh;
    for (int t = 0; t < Ti; ++t)
        CKI(cell_fwd_core(se, u, P.Zenc + t * se.ZT, P.Yenc + t * se.ZT, P.zr_e + t * R * 2 * H, P.hc_e + t * R * H,
                          we, P.Zenc + (t + 1) * se.ZT, se.Cp, st, P.bf16 ? P.x0b_e + (long long)2 * t * se.PSb : nullptr,
                          P.bf16 ? P.x0c_e + (long long)2 * t * se.PSb : nullptr,
                          lite ? P.Pb_e + (long long)2 * t * PbS_e : nullptr, t > 0,
                          (lite || P.x3r) && t + 1 < Ti ? P.x0b_e + (long long)2 * (t + 1) * se.PSb : nullptr,
                          P.Xp_e ? P.Xp_e + (long long)t * P.nb * R * 4 : nullptr));
    // ---- memory head (:159-166, :178-179): writes decoder state [h_t | value] into Zdec[0]
    CKI(memory_fwd_launch(P.Zenc + Ti * se.ZT, se.Cp, p->Wq, p->Memory, B, N, H, M, D, P.q_rows, P.att_rows,
                          P.dsc, P.ind_rows, P.Zdec, sd.Cp, h_att, query, pos, neg, nullptr, st));
    // ---- decoder (:181-192): its input columns were packed at the top of the step
    CellW wd{P.Wf[2], P.Wd[2], p->dec_gate_b, P.Wf[3], P.Wd[3], p->dec_update_b, P.imgf[2], P.imgd[2], P.imgf[3], P.imgd[3]};
    if (wps) { wd.wp_g = P.wpimg[2]; wd.wp_u = P.wpimg[3]; }
    for (int t = 0; t < To; ++t) {
        float* Zn = P.Zdec + (t + 1) * sd.ZT;
        // forward hoisting (bf16x3 small graphs): this step's input channels were known when the stack started iff its go symbol
        // is zero (t = 0) or the label of a teacher-forced step (model/MegaCRN.py:182,188-191)
        Shp sdt = sd;
        sdt.state_only = sd.hoist_fwd && any_teacher && (t == 0 || teacher[t - 1]);
        CKI(cell_fwd_core(sdt, u, P.Zdec + t * sd.ZT, P.Ydec + t * sd.ZT, P.zr_d + t * R * 2 * Hd,
                          P.hc_d + t * R * Hd, wd, Zn, sd.Cp, st, P.bf16 ? P.x0b_d + (long long)2 * t * sd.PSb : nullptr,
                          P.bf16 ? P.x0c_d + (long long)2 * t * sd.PSb : nullptr,
                          lite ? P.Pb_d + (long long)2 * t * PbS_d : nullptr, t > 0,
                          (lite || P.x3r) && t + 1 < To ? P.x0b_d + (long long)2 * (t + 1) * sd.PSb : nullptr,
                          P.Xp_d ? P.Xp_d + (long long)t * P.nb * R * 4 : nullptr));
        const bool last = t + 1 == To;
        const float* lab = (teacher && teacher[t] && labels) ? labels + (long long)t * N * od : nullptr;
        LAUNCH(k_proj_fwd, dim3(cdiv(R, 4) < 2048 ? cdiv(R, 4) : 2048), dim3(256), 0, st, (const float*)Zn,
               (long long)sd.Cp, p->proj_w, p->proj_b, Hd, od, B, N, output + (long long)t * N * od,
               (long long)To * N * od, (long long)od, last ? (float*)nullptr : Zn,
               last ? (float*)nullptr : P.Ydec + (t + 1) * sd.ZT, (long long)sd.Cp, Hd, lab);
        // go = proj(h') was not known when the decoder's input channels were hoisted: propagate this one channel block now
        if (P.bf16 && !last && !lab) CKI(hoist_inputs(sd, u, Zn, P.Ydec + (t + 1) * sd.ZT, 1, Hd, od, P.xin_b, P.xin_t, st,
                                                     P.Xp_d ? P.Xp_d + (long long)(t + 1) * P.nb * R * 4 : nullptr));
    }
    return 0;
}

static int model_backward(const mcrn_dims_t* d, const mcrn_params_t* p, const int* teacher,
                          const float* d_output, const float* d_hatt, const float* d_query, const float* d_pos,
                          const float* d_neg, char* ws, const mcrn_grads_t* g, hipStream_t st) {
    ModelPlan P;
    plan_model(d, ws, P);
    X3rScope x3scope(P.x3r);
    const int B = d->B, N = d->N, H = d->H, D = d->mem_dim, M = d->mem_num;
    const int Hd = H + D, od = d->output_dim, yd = d->ycov_dim;
    const int Ti = d->T_in, To = d->T_out;
    const Shp &se = P.se, &sd = P.sd;
    const long long R = se.R;
    Sup u = model_sup(P, N);
    CK(hipMemsetAsync(P.dS, 0, (size_t)(P.ds4 ? 4 : 2) * P.nslabS * N * P.ldS * sizeof(float), st));
    CK(hipMemsetAsync(P.dWq_s, 0, (size_t)NSLAB_T * H * D * sizeof(float), st));
    CK(hipMemsetAsync(P.dMem_s, 0, (size_t)NSLAB_T * M * D * sizeof(float), st));
    CK(hipMemsetAsync(P.dWp_s, 0, (size_t)NSLAB_T * od * Hd * sizeof(float), st));
    const bool bh = P.bf16 && P.bwd_hoist;       // hoisted backward: packed state-channel operands + stack-wide input operands
    if (P.bf16 && P.Kp > N) {   // pad rows of the bf16 gradient planes: d-grad writes the N data rows in place
        for (int e_ = 0; e_ < 2; ++e_) {
            const Shp& s_ = e_ ? sd : se;
            const long long ps_ = bh ? s_.PSbh : s_.PSb, ldx = bh ? s_.ldh : s_.ldp;
            const long long np_ = (long long)2 * (e_ ? To : Ti) * P.nb, per = (long long)(s_.Kp - N) * (ldx / 8);
            LAUNCH(k_zero_pad_rows, dim3(cdiv(per * np_, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>(e_ ? P.dPb_d : P.dPb_e),
                   ps_ / 8, N, s_.Kp, (int)(ldx / 8), np_);
            if (s_.lo_dPb > 0)     // (x3r: the lo images too)
                LAUNCH(k_zero_pad_rows, dim3(cdiv(per * np_, 256)), dim3(256), 0, st, reinterpret_cast<uint4*>((e_ ? P.dPb_d : P.dPb_e) + s_.lo_dPb),
                       ps_ / 8, N, s_.Kp, (int)(ldx / 8), np_);
        }
    }
    if (bh) {   // K padding of the input operands (columns beyond the calls' data) must be zero (x3r: hi and lo image, contiguous)
        const size_t lm = P.x3r ? 2 : 1;
        CK(hipMemsetAsync(P.dPin_e, 0, lm * (size_t)(se.lo_dPin > 0 ? se.lo_dPin : (long long)(P.nb * N + 64) * P.kin_e) * sizeof(uint16_t), st));
        CK(hipMemsetAsync(P.dPin_d, 0, lm * (size_t)(sd.lo_dPin > 0 ? sd.lo_dPin : (long long)(P.nb * N + 64) * P.kin_d) * sizeof(uint16_t), st));
    }
    // ---- decoder BPTT
    const Sup& ud = u;
    CellW wd{P.Wf[2], P.Wd[2], p->dec_gate_b, P.Wf[3], P.Wd[3], p->dec_update_b, P.imgf[2], P.imgd[2], P.imgf[3], P.imgd[3]};
    {
        int xu = 0, xg = 0;
        const float *dPprev = nullptr, *dQprev = nullptr;
        bool prev_hoisted = false;                 // the cell processed just before (t + 1) ran its chain on the state columns only
        bool any_teacher = false;                  // (as in model_forward: hoisting only when a step beyond the first is teacher-forced)
        for (int t = 0; t + 1 < To; ++t) any_teacher |= teacher && teacher[t];
        for (int t = To - 1; t >= 0; --t) {
            const bool last = t == To - 1;
            const int use_next = (!last && !(teacher && teacher[t])) ? 1 : 0;
            const int pair = P.flat ? t : t % ModelPlan::NPAIR;
            // hoisted backward (small graphs): nothing reads the propagated gradient of this cell's input channels when its go symbol
            // was known before the stack started - zero (t = 0) or the label of a teacher-forced step (no use_next at t - 1)
            Shp sdt = sd;
            sdt.state_only = sd.hoist_bwd && any_teacher && (t == 0 || teacher[t - 1]);
            float* dPt = P.dPp[pair];
            float* dQt = P.dQp[pair];
            if (last) {
                LAUNCH(k_proj_bwd, dim3(cdiv(R * Hd, 256)), dim3(256), 0, st, d_output + (long long)t * N * od,
                       (long long)To * N * od, (long long)od, (const float*)P.dxin_d, (long long)(od + yd), use_next,
                       p->proj_w, Hd, od, B, N, (const float*)nullptr, P.dhn_d, P.dgo + (long long)t * R * od);
            } else {   // C(t+1) + projection backward(t) + A(t) in one launch
                // the go symbol of step t+1 was this step's projection: its share of the propagated input gradient of cell t+1
                if (bh && use_next) CKI(go_grad_bf16(sd, u, P.dPin_d, P.kin_d, 2 * (t + 1), P.go_tmp, const_cast<float*>(dQprev), const_cast<float*>(dPprev), od, st));
                CKI(cell_bwd_ca(sd, (bh || prev_hoisted) ? Hd : sd.Cp, dPprev, dQprev, P.dTu, P.dTg, xu, xg, d_output + (long long)t * N * od,
                                (long long)To * N * od, (long long)od, use_next, p->proj_w, od, P.dgo + (long long)t * R * od,
                                P.Zdec + t * sd.ZT, P.zr_d + t * R * 2 * Hd, P.hc_d + t * R * Hd,
                                P.dU_d + t * R * Hd, P.dG_d + t * R * 2 * Hd, P.dacc_d, st));
            }
            CKI(cell_bwd_core(sdt, ud, P.Zdec + t * sd.ZT, P.Ydec + t * sd.ZT, P.zr_d + t * R * 2 * Hd, P.hc_d + t * R * Hd,
                              wd, P.dhn_d, P.dU_d + t * R * Hd, P.dG_d + t * R * 2 * Hd, dPt, dQt, P.dacc_d, P.dxin_d, st,
                              P.dTu, P.dTg, /*do_a=*/last, /*do_c=*/t == 0, &xu, &xg,
                              P.bf16 ? P.dPb_d + (long long)2 * t * P.nb * (bh ? sd.PSbh : sd.PSb) : nullptr, pair,
                              bh ? P.dPin_d : nullptr, P.kin_d, 2 * t));
            dPprev = dPt; dQprev = dQt; prev_hoisted = sdt.state_only;
        }
    }
    // Decoder weight/bias/projection gradients depend only on the finished decoder BPTT: run them on the
    // helper stream so they overlap the memory-head and encoder backward below (joined before the adjacency
    // backward).  Tuning / profiling runs keep them in-line.
    hipStream_t ws_ = st;
    float* part_ = P.part;
    if (g_use_side && !g_tuning && g_prof.role < 0) {
        CKI(side_init());
        // Second helper queue on the small graphs.  On the first queue the encoder's adjacency-gradient launches waited behind these
        // ~0.5 ms of HBM streaming, the plane-set guards of the main queue behind them (profiles/r4/experiments.md section 11).
        const bool two = !P.bf16;
        hipStream_t hs = two ? g_side.st2 : g_side.st;
        CK(hipEventRecord(two ? g_side.fork2 : g_side.fork, st));         // (its own event: ready[0] belongs to the per-call launches of the loops)
        CK(hipStreamWaitEvent(hs, two ? g_side.fork2 : g_side.fork, 0));
        ws_ = hs; part_ = two ? P.part3 : P.part2;
        if (two) g_side.any2 = true; else g_side.any = true;
    }
    {   // proj grads: dWp[j][c] = sum_{t,r} dgo[t][r][j] * h'_t[r][c]  (h'_t lives in Zdec[t+1])
        GemmP q = gp();
        q.M = od; q.N = Hd; q.K = (int)(To * R);
        q.A[0] = P.dgo; q.am = plain(1); q.ak = plain(od);
        q.B[0] = P.Zdec + sd.ZT; q.bk = two((int)R, sd.ZT, sd.Cp); q.bk_hi[0] = sd.ZT; q.bn = plain(1);
        q.C[0] = P.dWp_s; q.Cin[0] = P.dWp_s; q.cm = plain(Hd); q.cn = plain(1);
        q.beta = 1.f; q.slab = (long long)od * Hd;
        CKI(gemm(q, false, false, NSLAB_T, ROLE_MISC, ws_));
        LAUNCH(k_reduce_slabs, dim3(cdiv(od * Hd, 64)), dim3(1024), 0, ws_, g->proj_w, (const float*)P.dWp_s, NSLAB_T,
               (long long)od * Hd, (long long)od * Hd, 0);
        CKI(colsum(P.dgo, od, To * R, od, part_, g->proj_b, 0, ws_));
    }
    int ns1 = 0;
    bool on1 = false, on2 = false, on3 = false, on4 = false;
    const bool lite = P.bf16 && P.Pb_e != nullptr;
    const long long PbS_e = (long long)P.nb * N * se.ldh, PbS_d = (long long)P.nb * N * sd.ldh;
    // Workgroups of the decoder's weight-gradient launches when they run beside the encoder BPTT on the helper stream
    // (`dec_budget` below).  At full width (252 workgroups at T = 12) the two launches take every CU for ~0.35 ms and the first
    // encoder cells of the main queue run at half speed; at 120 they take twice as long on half the chip and the main queue
    // keeps the other half: METR-LA 10 950 vs 10 700 / 10 630 and 10 780 / 10 805 vs 10 633 / 10 625 samples/s, PEMS-BAY 6 425 vs
    // 6 314 / 6 299 (two calls; 72 workgroups: slower again - profiles/r4/experiments.md).  The bf16 mode measured no difference
    // (its helper stream runs beside MFMA-bound products) and keeps the full width.
    // Round 4, second step: on their OWN queue (Side::st2) they no longer hold up the encoder's adjacency-gradient launches, and
    // there the narrower the better until they stop fitting the encoder BPTT: 24 workgroups (budget 32) -> METR-LA 11 026 / 10 995
    // vs 10 781 (one helper queue, 120 workgroups) and 10 633 before; PEMS-BAY 6 314 / 6 276 vs 6 184; 12 workgroups: 9 410.
    const bool own_queue = g_side.ok && ws_ == g_side.st2;
    // (bf16 mode re-measured in round 5 - 256 / 192 / 128 / 96 / 72 workgroups: 4 705 / 4 711 / 4 751 / 4 714 / 4 699 samples/s at EXPY-TKY,
    //  means of three runs inside their spread; 48: -3 %, 24: -17 %: the tail waits for them.  Full width kept.)
    const int dec_budget = ws_ != st ? (P.bf16 ? NSLAB_W : (own_queue ? 32 : NSLAB_W / 2)) : NSLAB_W;
    CKI(agcn_wgrad(sd, P.Zdec, sd.ZT, To, P.dG_d, 2 * Hd, P.dWs[2], ws_, &ns1, &on1, lite ? P.Pb_d : nullptr, 2 * PbS_d, (long long)N * sd.ldh, P.Xp_d, dec_budget));
    int ns2 = 0;
    CKI(agcn_wgrad(sd, P.Ydec, sd.ZT, To, P.dU_d, Hd, P.dWs[3], ws_, &ns2, &on2, lite ? P.Pb_d + PbS_d : nullptr, 2 * PbS_d, (long long)N * sd.ldh, P.Xp_d, dec_budget));
    // hoisted backward, adjacency gradient of one cell stack (K-concatenated over every AGCN call): both stacks' products run on the
    // caller's stream in the tail, the encoder's accumulating into the same dA
    auto stack_ds = [&](int e_, hipStream_t st) -> int {
        const Shp& s_ = e_ ? se : sd;
        const int T_ = e_ ? Ti : To;
        const float *Zall = e_ ? P.Zenc : P.Zdec, *Yall = e_ ? P.Yenc : P.Ydec;
        uint16_t* x0ch = e_ ? P.x0c_e : P.x0c_d;
        const long long kin = e_ ? P.kin_e : P.kin_d;
        const int nsamp = N < 64 ? N : 64;
        for (int a = 0; a < 2; ++a) {      // node-centred packed state channels of every call: gate calls (Z), update calls (Y)
            const float* X = a ? Yall : Zall;
            float* mu = u.mu + (long long)a * T_ * s_.ld;
            LAUNCH(k_colsum_sample, dim3(cdiv(s_.ld, 64), T_), dim3(256), 0, st, X, s_.ld, N, (int)s_.ld, nsamp, mu, s_.ZT, s_.ld);
            PackX xh; xh.ny = T_; xh.src_y = s_.ZT; xh.dst_y = 2 * s_.PSbh / 8; xh.mu = mu; xh.mu_y = s_.ld; xh.inv_rows = 1.f / (float)N;
            xh.lo = s_.lo_x0c;
            CKI(pack_cols_bf16(X, 0, s_, 0, s_.H, 1, (int)s_.ldh, x0ch + (long long)a * s_.PSbh, st, xh));
        }
        CKI(ds_bf16_h(s_, u, e_ ? P.dPb_e : P.dPb_d, x0ch, 2 * T_, P.dA, P.ldS, e_ != 0, st));
        // input channels of every call: the centred inputs (the same for the gate and the update call of a step)
        CK(hipMemsetAsync(P.xin_c, 0, (size_t)(N + 64) * kin * sizeof(uint16_t), st));
        if (P.lo_xin_c > 0) CK(hipMemsetAsync(P.xin_c + P.lo_xin_c, 0, (size_t)(N + 64) * kin * sizeof(uint16_t), st));
        for (int a = 0; a < 2; ++a) {
            PackX xi; xi.mu = u.mu; xi.mu_t = s_.ld; xi.inv_rows = 1.f / (float)N; xi.rows = N; xi.tmul = 2; xi.toff = a;
            xi.lo = P.lo_xin_c;
            xi.ncw = (T_ * s_.B * s_.d + 7) & ~7;
            CKI(pack_cols_bf16(Zall, s_.ZT, s_, s_.H, s_.d, T_, (int)kin, P.xin_c, st, xi));
        }
        CKI(ds_bf16_in(s_, u, e_ ? P.dPin_e : P.dPin_d, P.xin_c, kin, 2 * T_ * s_.B * s_.d, P.dA, P.ldS, st, P.lo_xin_c));
        return 0;
    };
    CKI(wunprep(g->dec_gate_w, P.dWs[2], sd, 2 * Hd, ws_, ns1, on1 ? g->dec_gate_b : nullptr));
    CKI(wunprep(g->dec_update_w, P.dWs[3], sd, Hd, ws_, ns2, on2 ? g->dec_update_b : nullptr));
    if (!on1) CKI(colsum(P.dG_d, 2 * Hd, To * R, 2 * Hd, part_, g->dec_gate_b, 0, ws_));
    if (!on2) CKI(colsum(P.dU_d, Hd, To * R, Hd, part_, g->dec_update_b, 0, ws_));
    // ---- memory head backward: dacc_d = d[h_t | value]
    CKI(memory_bwd_rows_launch(P.dacc_d, Hd, H, d_hatt, d_query, P.att_rows, p->Memory, B, N, M, D, P.dval, P.dsc, P.dq, st));
    LAUNCH(k_copy2d, dim3(cdiv(R * H, 256)), dim3(256), 0, st, P.dacc_e, (long long)H, (const float*)P.dacc_d, (long long)Hd, R, H);
    CKI(memory_bwd_gemms(P.Zenc + Ti * se.ZT, se.Cp, p->Wq, R, H, M, D, P.att_rows, P.q_rows, P.dval, P.dsc, P.dq,
                         P.dacc_e, H, true, P.dWq_s, P.dMem_s, st));
    LAUNCH(k_reduce_slabs, dim3(cdiv(H * D, 64)), dim3(1024), 0, st, g->Wq, (const float*)P.dWq_s, NSLAB_T,
           (long long)H * D, (long long)H * D, 0);
    // ---- encoder BPTT
    CellW we{P.Wf[0], P.Wd[0], p->enc_gate_b, P.Wf[1], P.Wd[1], p->enc_update_b, P.imgf[0], P.imgd[0], P.imgf[1], P.imgd[1]};
    {
        int xu = 0, xg = 0;
        for (int t = Ti - 1; t >= 0; --t) {
            const bool first = t == Ti - 1;
            const int pair = P.flat ? To + t : t % ModelPlan::NPAIR;
            const int prev = P.flat ? To + t + 1 : (t + 1) % ModelPlan::NPAIR;           // (the cell processed just before: t + 1)
            float* dPt = P.dPp[pair];
            float* dQt = P.dQp[pair];
            if (!first)   // C(t+1) + A(t) in one launch (dh' of step t IS the accumulated state gradient)
                CKI(cell_bwd_ca(se, bh ? H : se.Cp, P.dPp[prev], P.dQp[prev], P.dTu, P.dTg, xu, xg, nullptr, 0, 0, 0, nullptr, 0, nullptr,
                                P.Zenc + t * se.ZT, P.zr_e + t * R * 2 * H, P.hc_e + t * R * H,
                                P.dU_e + t * R * H, P.dG_e + t * R * 2 * H, P.dacc_e, st));
            CKI(cell_bwd_core(se, u, P.Zenc + t * se.ZT, P.Yenc + t * se.ZT, P.zr_e + t * R * 2 * H, P.hc_e + t * R * H, we,
                              P.dacc_e, P.dU_e + t * R * H, P.dG_e + t * R * 2 * H,
                              dPt, dQt, P.dacc_e, P.dxin_e, st, P.dTu, P.dTg, /*do_a=*/first, /*do_c=*/t == 0, &xu, &xg,
                              P.bf16 ? P.dPb_e + (long long)2 * t * P.nb * (bh ? se.PSbh : se.PSb) : nullptr, pair,
                              bh ? P.dPin_e : nullptr, P.kin_e, 2 * t));
        }
    }
    // Encoder weight / bias gradients: nothing below needs them, and they are HBM-streaming kernels while the adjacency
    // backward below is MFMA work (bf16 mode: ~1 ms of full-chip products) or a chain of tiny launches (small graphs).  In
    // the bf16 mode both go to the helper stream; on the small graphs the gate call's stays on the caller's stream (the
    // helper stream is still finishing the last cell's adjacency gradient) and the update call's goes to the helper stream.
    // The caller's stream then waits only for what the helper stream had BEFORE them (`mid`): the adjacency-gradient slabs.
    const bool tail_side = g_use_side && !g_tuning && g_prof.role < 0;
    hipStream_t wg_st = st, wu_st = st;
    float *part_g = P.part, *part_u = P.part;
    if (tail_side) {
        CKI(side_init());
        CK(hipEventRecord(g_side.mid, g_side.st));
        CK(hipEventRecord(g_side.fork, st));
        CK(hipStreamWaitEvent(g_side.st, g_side.fork, 0));
        wu_st = g_side.st; part_u = P.part2;
        if (P.bf16) { wg_st = g_side.st; part_g = P.part2; }
        g_side.any = true;
    }
    int ns3 = 0;
    CKI(agcn_wgrad(se, P.Zenc, se.ZT, Ti, P.dG_e, 2 * H, P.dWs[0], wg_st, &ns3, &on3, lite ? P.Pb_e : nullptr, 2 * PbS_e, (long long)N * se.ldh, P.Xp_e));
    CKI(wunprep(g->enc_gate_w, P.dWs[0], se, 2 * H, wg_st, ns3, on3 ? g->enc_gate_b : nullptr));
    if (!on3) CKI(colsum(P.dG_e, 2 * H, Ti * R, 2 * H, part_g, g->enc_gate_b, 0, wg_st));
    int ns4 = 0;
    CKI(agcn_wgrad(se, P.Yenc, se.ZT, Ti, P.dU_e, H, P.dWs[1], wu_st, &ns4, &on4, lite ? P.Pb_e + PbS_e : nullptr, 2 * PbS_e, (long long)N * se.ldh, P.Xp_e));
    CKI(wunprep(g->enc_update_w, P.dWs[1], se, H, wu_st, ns4, on4 ? g->enc_update_b : nullptr));
    if (!on4) CKI(colsum(P.dU_e, H, Ti * R, H, part_u, g->enc_update_b, 0, wu_st));
    // ---- adjacency backward (all dS contributions are in the slabs once the helper stream's earlier work is joined)
    if (tail_side) CK(hipStreamWaitEvent(st, g_side.mid, 0));
    else CKI(side_join(st));
    if (P.bf16) {
        // one K-concatenated product per cell stack over every AGCN call's (dP planes, centred input plane), then the
        // chain rule of T2 = 2 S S - I; the S blocks of dA then hold dS1 / dS2
        if (bh) {
            CKI(stack_ds(0, st));
            CKI(stack_ds(1, st));
        } else {
        CKI(centre_planes(sd, u, P.Zdec, P.Ydec, To, P.x0c_d, st));
        CKI(ds_bf16(sd, u, P.dPb_d, P.x0c_d, 2 * To, P.dA, P.ldS, false, st));
        CKI(centre_planes(se, u, P.Zenc, P.Yenc, Ti, P.x0c_e, st));
        CKI(ds_bf16(se, u, P.dPb_e, P.x0c_e, 2 * Ti, P.dA, P.ldS, true, st));
        }
        CKI(t2_backward(P, u, N, d->cheb_k, st));
        CKI(sup_bwd_core(N, M, D, p->We1, p->We2, p->Memory, P.sup, P.sup.g1, P.sup.g2, P.ldS, P.dA,
                         P.dA + (long long)(d->cheb_k - 1) * N * P.ldS, P.ldS, 1, 0, g->We1, g->We2, P.dMem_s, st, NSLAB_T));
    } else if (P.ds4) {
        // fold the slabs of the four blocks (fixed order), then the chain rule of T2 = 2 S S - I onto S in exact fp32 (two batched N^3
        // products of a few MFLOP):   dS_b = dA[d1_b] + dA[e2_b] S_b^T + S_b^T dA[e2_b]        (e2 = 2 d2 carries the factor 2)
        const long long nn = (long long)N * P.ldS;
        LAUNCH(k_reduce_slabs, dim3(cdiv(4 * nn, 64)), dim3(1024), 0, st, P.dAm, (const float*)P.dS, P.nslabS, 4 * nn, 4 * nn, 0);
        {
            ExactFp32 exact_chain;
            for (int which = 0; which < 2; ++which) {
                GemmP q = gp();
                q.M = N; q.N = N; q.K = N; q.nbatch = 2; q.beta = 1.f;
                q.am = plain(P.ldS); q.ak = plain(1);
                if (which == 0) { q.bk = plain(1); q.bn = plain(P.ldS); } else { q.bk = plain(P.ldS); q.bn = plain(1); }
                q.cm = plain(P.ldS); q.cn = plain(1);
                for (int b = 0; b < 2; ++b) {
                    const float* dE = P.dAm + (long long)(2 * b + 1) * nn;
                    q.A[b] = which == 0 ? dE : (b ? P.sup.St2 : P.sup.St1);            // dE S^T   |   S^T dE
                    q.B[b] = which == 0 ? (b ? P.sup.g2 : P.sup.g1) : dE;
                    q.C[b] = P.dAm + (long long)(2 * b) * nn; q.Cin[b] = q.C[b];
                }
                CKI(gemm(q, true, which == 0, 0, ROLE_MISC, st));
            }
        }
        CKI(sup_bwd_core(N, M, D, p->We1, p->We2, p->Memory, P.sup, P.sup.g1, P.sup.g2, P.ldS, P.dAm, P.dAm + 2 * nn, P.ldS, 1, 0,
                         g->We1, g->We2, P.dMem_s, st, NSLAB_T));
    } else
    CKI(sup_bwd_core(N, M, D, p->We1, p->We2, p->Memory, P.sup, P.sup.g1, P.sup.g2, P.ldS, P.dS, P.dS + u.sup_stride,
                     P.ldS, P.nslabS, u.slab, g->We1, g->We2, P.dMem_s, st, NSLAB_T));
    LAUNCH(k_reduce_slabs, dim3(cdiv(M * D, 64)), dim3(1024), 0, st, g->Memory, (const float*)P.dMem_s, NSLAB_T,
           (long long)M * D, (long long)M * D, 0);
    if (d_pos) LAUNCH(k_memory_scatter, dim3(cdiv(R * D, 256)), dim3(256), 0, st, g->Memory, (const int*)P.ind_rows, 0, d_pos, B, N, D);
    if (d_neg) LAUNCH(k_memory_scatter, dim3(cdiv(R * D, 256)), dim3(256), 0, st, g->Memory, (const int*)P.ind_rows, 1, d_neg, B, N, D);
    CKI(side_join(st));
    return 0;
}

// =================================================================================================
// stand-alone ops (batch-major boundary; pack -> node-major core -> unpack)
// =================================================================================================
struct CellPlan {
    Shp s;
    float *Z, *Y, *zr, *hc, *hn_rows;
    float *Wf[2], *Wd[2], *dWs[2];
    float *St1, *St2, *dS;
    uint4* frag[4];
    int nslabS;
    float *dP, *dQ, *dTu, *dTg, *dU, *dG, *dacc, *dxin, *part;
    size_t total;
};
static void plan_cell(int B, int N, int din, int H, int K, char* base, CellPlan& P) {
    Bump b{base, 0};
    P.s = mk_shape(B, N, din, H, K);
    const Shp& s = P.s;
    P.Z = b.take<float>((size_t)s.ZT); P.Y = b.take<float>((size_t)s.ZT);
    P.zr = b.take<float>((size_t)s.R * 2 * H); P.hc = b.take<float>((size_t)s.R * H);
    P.hn_rows = b.take<float>((size_t)s.R * H);
    const int Os[2] = {2 * H, H};
    for (int i = 0; i < 2; ++i) {
        size_t n = (size_t)s.G * s.Cp * Os[i];
        P.Wf[i] = b.take<float>(n); P.Wd[i] = b.take<float>(n); P.dWs[i] = b.take<float>(n * NSLAB_W);
    }
    P.St1 = b.take<float>((size_t)N * N); P.St2 = b.take<float>((size_t)N * N);
    for (int i = 0; i < 4; ++i) P.frag[i] = N <= PROP2_MAX_N ? b.take<uint4>(sfrag_uint4(N)) : nullptr;
    P.nslabS = nslab_S(N);
    P.dS = b.take<float>((size_t)2 * P.nslabS * N * N);
    P.dP = b.take<float>((size_t)s.ZT); P.dQ = b.take<float>((size_t)s.ZT);
    P.dTu = b.take<float>((size_t)s.PS); P.dTg = b.take<float>((size_t)s.PS);
    P.dU = b.take<float>((size_t)s.R * H); P.dG = b.take<float>((size_t)s.R * 2 * H);
    P.dacc = b.take<float>((size_t)s.R * H); P.dxin = b.take<float>((size_t)s.R * (din > 0 ? din : 1));
    P.part = b.take<float>(colsum_part_floats(s.R, 2 * H) + 1024);
    P.total = (b.off + 255) & ~(size_t)255;
}
static int bnc_to_rows(float* dst, long long ld, int col0, int w, const float* src, int B, int N, hipStream_t st) {
    if (w <= 0) return 0;
    LAUNCH(k_bnc_to_rows, dim3(cdiv((long long)B * N * w, 256)), dim3(256), 0, st, dst, ld, col0, w, src, B, N);
    return 0;
}
static int rows_to_bnc(float* dst, const float* src, long long ld, int col0, int w, int B, int N, hipStream_t st) {
    if (w <= 0) return 0;
    LAUNCH(k_rows_to_bnc, dim3(cdiv((long long)B * N * w, 256)), dim3(256), 0, st, dst, src, ld, col0, w, B, N, 0);
    return 0;
}

struct AgcnPlan {
    Shp s;
    float *Z, *y_rows, *Wf, *Wd, *dWs, *St1, *St2, *dS, *dP, *dY, *part;
    uint4* frag[4];
    int nslabS;
    size_t total;
};
static void plan_agcn(int B, int N, int C, int O, int K, char* base, AgcnPlan& P) {
    Bump b{base, 0};
    P.s = mk_shape(B, N, 0, C, K);
    const Shp& s = P.s;
    P.Z = b.take<float>((size_t)s.ZT);
    P.y_rows = b.take<float>((size_t)s.R * O);
    size_t n = (size_t)s.G * s.Cp * O;
    P.Wf = b.take<float>(n); P.Wd = b.take<float>(n); P.dWs = b.take<float>(n * NSLAB_W);
    P.St1 = b.take<float>((size_t)N * N); P.St2 = b.take<float>((size_t)N * N);
    for (int i = 0; i < 4; ++i) P.frag[i] = N <= PROP2_MAX_N ? b.take<uint4>(sfrag_uint4(N)) : nullptr;
    P.nslabS = nslab_S(N);
    P.dS = b.take<float>((size_t)2 * P.nslabS * N * N);
    P.dP = b.take<float>((size_t)s.ZT);
    P.dY = b.take<float>((size_t)s.R * O);
    P.part = b.take<float>(colsum_part_floats(s.R, O) + 1024);
    P.total = (b.off + 255) & ~(size_t)255;
}

struct MemPlan {
    float *h_rows, *q_rows, *att, *dval, *dsc, *dq, *dh_rows, *dWq_s, *dMem_s;
    int* ind;
    size_t total;
};
static void plan_mem(int B, int N, int H, int M, int D, char* base, MemPlan& P) {
    Bump b{base, 0};
    size_t R = (size_t)B * N;
    P.h_rows = b.take<float>(R * H); P.q_rows = b.take<float>(R * D); P.att = b.take<float>(R * M);
    P.ind = b.take<int>(R * 2);
    P.dval = b.take<float>(R * D); P.dsc = b.take<float>(R * M); P.dq = b.take<float>(R * D);
    P.dh_rows = b.take<float>(R * H);
    P.dWq_s = b.take<float>((size_t)NSLAB_T * H * D); P.dMem_s = b.take<float>((size_t)NSLAB_T * M * D);
    P.total = (b.off + 255) & ~(size_t)255;
}

}  // namespace mcrn

using namespace mcrn;

// =================================================================================================
// C ABI
// =================================================================================================
#pragma GCC visibility push(default)   // (-fvisibility=hidden: only the C ABI is exported)
extern "C" {

const char* mcrn_last_error(void) { return g_err; }
int mcrn_version(void) { return 106; }
#ifndef MCRN_BUILD_ID
#define MCRN_BUILD_ID "unknown"
#endif
const char* mcrn_build_id(void) { return MCRN_BUILD_ID; }
int mcrn_last_launch_count(void) { return g_launches; }
// launch histogram (see fam()): "family=count\n" lines, NUL-terminated; returns the bytes needed (incl. the NUL).  reset = 1 clears it.
long long mcrn_launch_histogram(char* buf, long long cap, int reset) {
    static char out[NFAM_MAX * 96];
    size_t n = 0;
    for (int i = 0; i < g_nfam && n + 96 < sizeof out; ++i) n += (size_t)snprintf(out + n, 96, "%s=%lld\n", g_fam_name[i], g_fam_n[i]);
    out[n++] = 0;
    if (buf && cap >= (long long)n) memcpy(buf, out, n);
    if (reset) g_nfam = 0;
    return (long long)n;
}

int mcrn_set_gemm_cfg(int cfg) {
    if (cfg < -1 || cfg >= NCFG) FAIL("gemm cfg %d outside -1..%d", cfg, NCFG - 1);
    g_force_cfg = cfg;
    return 0;
}
int mcrn_set_debug(int bits) { g_debug = bits; return 0; }
#ifdef MCRN_TIMELINE
// measurement-only builds: copy out the in-kernel phase stamps (prop_small.h)
int mcrn_debug_timeline(void* host_out, size_t bytes) {
    CK(timeline_copy(host_out, bytes));
    return 0;
}
#endif

int mcrn_model_autotune(const mcrn_dims_t* d, void* ws, size_t ws_bytes, void* stream) {
    ENTER();
    CKI(check_dims(d));
    if (!ws || ws_bytes < mcrn_model_workspace_bytes(d)) FAIL("autotune: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // scratch parameters / inputs / outputs carved from the END of nothing: allocate temporaries here
    // (this entry point is the one place that may allocate: it runs once per shape, outside any step)
    const int B = d->B, N = d->N, H = d->H, D = d->mem_dim, M = d->mem_num, K = d->cheb_k;
    const int Hd = H + D, od = d->output_dim, yd = d->ycov_dim, din = d->input_dim;
    const size_t Ce = din + H, Cd = od + yd + Hd;
    const size_t np[14] = {(size_t)M * D, (size_t)H * D, (size_t)N * M, (size_t)N * M,
                           2 * K * Ce * 2 * H, (size_t)2 * H, 2 * K * Ce * H, (size_t)H,
                           2 * K * Cd * 2 * Hd, (size_t)2 * Hd, 2 * K * Cd * Hd, (size_t)Hd,
                           (size_t)od * Hd, (size_t)od};
    size_t tot = 0;
    for (int i = 0; i < 14; ++i) tot += (np[i] + 63) / 64 * 64;
    const size_t nx = (size_t)B * d->T_in * N * din, nyc = (size_t)B * d->T_out * N * (yd > 0 ? yd : 1),
                 nout = (size_t)B * d->T_out * N * od, nbd = (size_t)B * N * D;
    float* buf = nullptr;
    const size_t nfl = 2 * tot + nx + nyc + 2 * nout + 5 * nbd;
    CK(hipMalloc(&buf, nfl * sizeof(float)));
    CK(hipMemsetAsync(buf, 0, nfl * sizeof(float), st));
    float* q = buf;
    const float* pp[14];
    float* gg[14];
    for (int i = 0; i < 14; ++i) { pp[i] = q; q += (np[i] + 63) / 64 * 64; }
    for (int i = 0; i < 14; ++i) { gg[i] = q; q += (np[i] + 63) / 64 * 64; }
    float* x = q; q += nx;
    float* yc = q; q += nyc;
    float* out = q; q += nout;
    float* dout = q; q += nout;
    float* o4[4];
    for (int i = 0; i < 4; ++i) { o4[i] = q; q += nbd; }
    float* dq = q; q += nbd;
    mcrn_params_t P = {pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], pp[7], pp[8], pp[9], pp[10], pp[11], pp[12], pp[13]};
    mcrn_grads_t G = {gg[0], gg[1], gg[2], gg[3], gg[4], gg[5], gg[6], gg[7], gg[8], gg[9], gg[10], gg[11], gg[12], gg[13]};
    PrecisionScope prec(d->precision);
    g_tuning = true;
    if (d->precision == MCRN_BF16 || (d->precision == MCRN_BF16X3 && d->N > PROP2_MAX_N)) {   // (the sessions that run gemm_bf16.h products)
        g_flush_bytes = (size_t)192 << 20;
        if (hipMalloc(&g_flush, g_flush_bytes) != hipSuccess) { g_flush = nullptr; g_flush_bytes = 0; (void)hipGetLastError(); }
    }
    int rc = model_forward(d, &P, x, yc, nullptr, nullptr, (char*)ws, out, o4[0], o4[1], o4[2], o4[3], st);
    if (!rc) rc = model_backward(d, &P, nullptr, dout, nullptr, dq, nullptr, nullptr, (char*)ws, &G, st);
    g_tuning = false;
    if (g_flush) { (void)hipStreamSynchronize(st); (void)hipFree(g_flush); g_flush = nullptr; g_flush_bytes = 0; }
    g_force_cfg = -1;                     // a failed trial launch may have left the tuning loop's value behind
    if (rc) { (void)hipStreamSynchronize(st); side_reset(); }
    (void)hipStreamSynchronize(st);
    (void)hipFree(buf);
    return rc;
}
int mcrn_autotune_entries(void) { return (int)(g_tuned.size() + g_tuned_bf16.size() + g_tuned_split.size()); }
int mcrn_autotune_clear(void) { g_tuned.clear(); g_tuned_bf16.clear(); g_tuned_split.clear(); return 0; }
// Tile table as a flat int32 record list: {kind, nkey, key..., cfg}.  Data-parallel ranks tune independently (timing noise
// can pick different tiles, hence different fp32 summation orders); rank 0 exports, the others import (dp / bench.py).
long long mcrn_autotune_export(int* buf, long long cap) {
    std::vector<int> out;
    for (const auto& kv : g_tuned) {
        const int* k = reinterpret_cast<const int*>(&kv.first);
        const int nk = (int)(sizeof(TuneKey) / sizeof(int));
        out.push_back(0); out.push_back(nk);
        for (int i = 0; i < nk; ++i) out.push_back(k[i]);
        out.push_back(kv.second);
    }
    for (const auto& kv : g_tuned_bf16) {
        const int* k = reinterpret_cast<const int*>(&kv.first);
        const int nk = (int)(sizeof(Bf16Key) / sizeof(int));
        out.push_back(1); out.push_back(nk);
        for (int i = 0; i < nk; ++i) out.push_back(k[i]);
        out.push_back(kv.second);
    }
    for (const auto& kv : g_tuned_split) {
        const int* k = reinterpret_cast<const int*>(&kv.first);
        const int nk = (int)(sizeof(SplitKey) / sizeof(int));
        out.push_back(2); out.push_back(nk);
        for (int i = 0; i < nk; ++i) out.push_back(k[i]);
        out.push_back(kv.second);
    }
    if (buf && cap >= (long long)out.size()) memcpy(buf, out.data(), out.size() * sizeof(int));
    return (long long)out.size();
}
int mcrn_autotune_import(const int* buf, long long n) {
    if (!buf || n < 0) FAIL("autotune_import: bad arguments");
    std::map<TuneKey, int> a;
    std::map<Bf16Key, int> b;
    std::map<SplitKey, int> c;
    long long i = 0;
    while (i < n) {
        if (i + 2 > n) FAIL("autotune_import: truncated record");
        const int kind = buf[i], nk = buf[i + 1];
        const int want = kind == 0 ? (int)(sizeof(TuneKey) / sizeof(int)) : kind == 1 ? (int)(sizeof(Bf16Key) / sizeof(int)) :
                         kind == 2 ? (int)(sizeof(SplitKey) / sizeof(int)) : -1;
        if (nk != want || i + 2 + nk + 1 > n) FAIL("autotune_import: malformed record at word %lld", i);
        const int cfg = buf[i + 2 + nk];
        if (kind == 0) {
            if (cfg < 0 || cfg >= NCFG) FAIL("autotune_import: tile configuration %d out of range", cfg);
            TuneKey k; memcpy(&k, buf + i + 2, sizeof k); a[k] = cfg;
        } else if (kind == 1) {
            if (cfg < 0 || cfg >= NCFG_BF16) FAIL("autotune_import: bf16 tile configuration %d out of range", cfg);
            if (bf16_cfg_is_sk(cfg)) FAIL("autotune_import: tile configuration %d is a retired slot", cfg);
            Bf16Key k; memcpy(&k, buf + i + 2, sizeof k);
            b[k] = cfg;
        } else {
            if (cfg < 1 || cfg > 1 + PROPT_MAX_X) FAIL("autotune_import: split count %d out of range", cfg);
            SplitKey k; memcpy(&k, buf + i + 2, sizeof k); c[k] = cfg;
        }
        i += 2 + nk + 1;
    }
    // validated as a whole, then merged: entries of other shapes (an earlier leg of the same process) stay
    for (const auto& kv : a) g_tuned[kv.first] = kv.second;
    for (const auto& kv : b) g_tuned_bf16[kv.first] = kv.second;
    for (const auto& kv : c) g_tuned_split[kv.first] = kv.second;
    return 0;
}
int mcrn_set_precision(int precision) {
    if (precision != MCRN_F32 && precision != MCRN_BF16X3 && precision != MCRN_BF16) FAIL("unsupported precision %d", precision);
    g_precision = precision == MCRN_BF16 ? MCRN_BF16X3 : precision;   // the stand-alone ops have no bf16-resident form
    return 0;
}
int mcrn_get_precision(void) { return g_precision; }
int mcrn_set_side_stream(int enable) { g_use_side = enable != 0; return 0; }

int mcrn_prof_begin(int role) {
    if (role < 0 || role >= PROF_ROLE_COUNT) FAIL("prof: bad role %d", role);
    if (!g_prof.created) {
        for (int i = 0; i < 2 * Prof::MAXEV; ++i) CK(hipEventCreate(&g_prof.ev[i]));
        g_prof.created = true;
    }
    if (!g_prof.clk) CK(hipMalloc(&g_prof.clk, (size_t)Prof::MAXEV * 4 * sizeof(unsigned long long)));
    CK(hipMemset(g_prof.clk, 0, (size_t)Prof::MAXEV * 4 * sizeof(unsigned long long)));
    g_prof.role = role; g_prof.n = 0; g_prof.nclk = 0; g_prof.alg_flops = 0; g_prof.exec_flops = 0;
    return 0;
}
int mcrn_prof_end(double* total_ms, long long* launches, double* alg_flops, double* exec_flops) {
    double tot = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        float ms = 0;
        CK(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        CK(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = g_prof.n;
    if (alg_flops) *alg_flops = g_prof.alg_flops;
    if (exec_flops) *exec_flops = g_prof.exec_flops;
    g_prof.role = -1;
    // shader clock held under the profiled bf16-resident products: sum of the K loops' cycle counts / sum of their wall times
    g_prof.last_mhz = 0; g_prof.last_nclk = 0;
    if (g_prof.clk && g_prof.nclk > 0) {
        static unsigned long long h[Prof::MAXEV * 4];
        CK(hipMemcpy(h, g_prof.clk, (size_t)g_prof.nclk * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double dc = 0, dw = 0;
        for (int i = 0; i < g_prof.nclk; ++i) {
            if (!h[4 * i + 1] || h[4 * i + 3] <= h[4 * i + 1] || h[4 * i + 2] <= h[4 * i]) continue;       // (a launch that ran no K loop in workgroup 0)
            dc += (double)(h[4 * i + 2] - h[4 * i]); dw += (double)(h[4 * i + 3] - h[4 * i + 1]); ++g_prof.last_nclk;
        }
        if (dw > 0) g_prof.last_mhz = dc / dw * 100.0;
    }
    return 0;
}
int mcrn_prof_clock_mhz(double* shader_mhz, long long* launches) {
    if (shader_mhz) *shader_mhz = g_prof.last_mhz;
    if (launches) *launches = g_prof.last_nclk;
    return 0;
}

size_t mcrn_model_workspace_bytes(const mcrn_dims_t* d) {
    if (check_dims(d)) return 0;
    ModelPlan P;
    plan_model(d, nullptr, P);
    return P.total;
}

int mcrn_model_forward(const mcrn_dims_t* d, const mcrn_params_t* p, const float* x, const float* ycov,
                       const float* labels, const int* teacher, void* ws, size_t ws_bytes, float* output,
                       float* h_att, float* query, float* pos, float* neg, void* stream) {
    ENTER();
    CKI(check_dims(d));
    if (!p || !x || !ws || !output || !h_att || !query || !pos || !neg) FAIL("mcrn_model_forward: NULL argument");
    if (d->ycov_dim > 0 && !ycov) FAIL("mcrn_model_forward: ycov is NULL");
    if (ws_bytes < mcrn_model_workspace_bytes(d)) FAIL("workspace too small: %zu < %zu", ws_bytes, mcrn_model_workspace_bytes(d));
    if (teacher && !labels) for (int t = 0; t < d->T_out; ++t) if (teacher[t]) FAIL("teacher forcing requested without labels");
    g_launches = 0;
    PrecisionScope prec(d->precision);   // the session precision of the stand-alone ops is restored on return
    const int rc = model_forward(d, p, x, ycov, labels, teacher, (char*)ws, output, h_att, query, pos, neg, (hipStream_t)stream);
    if (rc) side_reset();
    return rc;
}

int mcrn_model_backward(const mcrn_dims_t* d, const mcrn_params_t* p, const int* teacher, const float* d_output,
                        const float* d_hatt, const float* d_query, const float* d_pos, const float* d_neg, void* ws,
                        size_t ws_bytes, const mcrn_grads_t* grads, void* stream) {
    ENTER();
    CKI(check_dims(d));
    if (!p || !d_output || !ws || !grads) FAIL("mcrn_model_backward: NULL argument");
    if (ws_bytes < mcrn_model_workspace_bytes(d)) FAIL("workspace too small");
    PrecisionScope prec(d->precision);
    const int rc = model_backward(d, p, teacher, d_output, d_hatt, d_query, d_pos, d_neg, (char*)ws, grads, (hipStream_t)stream);
    if (rc) side_reset();
    return rc;
}

// ---- supports -----------------------------------------------------------------------------------
size_t mcrn_supports_workspace_bytes(int N, int M, int D) {
    Bump b{nullptr, 0};
    SupBufs o;
    plan_sup(b, N, M, D, N, o);
    b.take<float>((size_t)M * D);
    return (b.off + 255) & ~(size_t)255;
}
int mcrn_supports_forward(int N, int M, int D, const float* We1, const float* We2, const float* Mem, void* ws,
                          size_t ws_bytes, float* g1, float* g2, void* stream) {
    ENTER();
    if (N < 1 || M < 1 || D < 1) FAIL("supports: bad sizes");
    if (ws_bytes < mcrn_supports_workspace_bytes(N, M, D)) FAIL("workspace too small");
    Bump b{(char*)ws, 0};
    SupBufs o;
    plan_sup(b, N, M, D, N, o);
    return sup_fwd_core(N, M, D, We1, We2, Mem, o, g1, N, g2, false, (hipStream_t)stream);
}
int mcrn_supports_backward(int N, int M, int D, const float* We1, const float* We2, const float* Mem,
                           const float* dg1, const float* dg2, void* ws, size_t ws_bytes, float* dWe1, float* dWe2,
                           float* dMem, void* stream) {
    ENTER();
    if (ws_bytes < mcrn_supports_workspace_bytes(N, M, D)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Bump b{(char*)ws, 0};
    SupBufs o;
    plan_sup(b, N, M, D, N, o);
    // recompute g1,g2 (forward left E1,E2,L1,L2 in ws; g were caller outputs) into St1/St2 scratch
    CKI(relu_softmax_rows(o.L1, o.ldS, o.St1, (long long)N, N, st));
    CKI(relu_softmax_rows(o.L2, o.ldS, o.St2, (long long)N, N, st));
    CK(hipMemsetAsync(dMem, 0, (size_t)M * D * sizeof(float), st));
    return sup_bwd_core(N, M, D, We1, We2, Mem, o, o.St1, o.St2, N, dg1, dg2, N, 1, 0, dWe1, dWe2, dMem, st);
}

// ---- AGCN ---------------------------------------------------------------------------------------
size_t mcrn_agcn_workspace_bytes(int B, int N, int C, int O, int cheb_k) {
    if (B < 1 || N < 1 || C < 1 || O < 1 || (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K)) return 0;
    AgcnPlan P;
    plan_agcn(B, N, C, O, cheb_k, nullptr, P);
    return P.total;
}
int mcrn_agcn_forward(int B, int N, int C, int O, int cheb_k, const float* x, const float* s1, const float* s2,
                      const float* W, const float* bias, void* ws, size_t ws_bytes, float* y, void* stream) {
    ENTER();
    if (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K) FAIL("cheb_k must be in 2 .. %d (got %d)", MCRN_MAX_CHEB_K, cheb_k);
    if (B < 1 || N < 1 || C < 1 || O < 1) FAIL("agcn: bad sizes");
    if (ws_bytes < mcrn_agcn_workspace_bytes(B, N, C, O, cheb_k)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    AgcnPlan P;
    plan_agcn(B, N, C, O, cheb_k, (char*)ws, P);
    const Shp& s = P.s;
    Sup u; u.S[0] = s1; u.S[1] = s2; u.St[0] = P.St1; u.St[1] = P.St2; u.Sf[0] = P.frag[0]; u.Sf[1] = P.frag[1]; u.Stf[0] = P.frag[2]; u.Stf[1] = P.frag[3]; u.Simg[0] = u.Simg[1] = u.Stimg[0] = u.Stimg[1] = nullptr; u.simg_n = 0; u.ldS = N; u.dS = P.dS; u.nslab = P.nslabS; u.slab = (long long)N * N; u.sup_stride = (long long)P.nslabS * N * N;
    CKI(wprep(W, P.Wf, P.Wd, s, O, st));
    CKI(build_frags(s1, s2, N, N, P.frag, st));
    CKI(bnc_to_rows(P.Z, s.Cp, 0, C, x, B, N, st));
    CKI(zero_cols(P.Z, 0, s.Cp, s.C, s.Cp, s.R, 1, st));
    CKI(prop_fwd(s, u, P.Z, st));
    GemmP e = gp();
    e.epi = EPI_BIAS; e.bias = bias; e.C[0] = P.y_rows; e.cm = plain(O); e.cn = plain(1);
    CKI(wp_fwd(s, P.Z, P.Wf, O, e, st));
    return rows_to_bnc(y, P.y_rows, O, 0, O, B, N, st);
}
int mcrn_agcn_backward(int B, int N, int C, int O, int cheb_k, const float* dy, const float* s1, const float* s2,
                       const float* W, void* ws, size_t ws_bytes, float* dx, float* ds1, float* ds2, float* dW,
                       float* db, void* stream) {
    ENTER();
    if (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K) FAIL("cheb_k must be in 2 .. %d (got %d)", MCRN_MAX_CHEB_K, cheb_k);
    if (ws_bytes < mcrn_agcn_workspace_bytes(B, N, C, O, cheb_k)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    AgcnPlan P;
    plan_agcn(B, N, C, O, cheb_k, (char*)ws, P);
    const Shp& s = P.s;
    Sup u; u.S[0] = s1; u.S[1] = s2; u.St[0] = P.St1; u.St[1] = P.St2; u.Sf[0] = P.frag[0]; u.Sf[1] = P.frag[1]; u.Stf[0] = P.frag[2]; u.Stf[1] = P.frag[3]; u.Simg[0] = u.Simg[1] = u.Stimg[0] = u.Stimg[1] = nullptr; u.simg_n = 0; u.ldS = N; u.dS = P.dS; u.nslab = P.nslabS; u.slab = (long long)N * N; u.sup_stride = (long long)P.nslabS * N * N;
    CKI(transpose(P.St1, N, s1, N, nullptr, 0, N, st));
    CKI(transpose(P.St2, N, s2, N, nullptr, 0, N, st));
    CK(hipMemsetAsync(P.dS, 0, (size_t)2 * P.nslabS * N * N * sizeof(float), st));
    CKI(bnc_to_rows(P.dY, O, 0, O, dy, B, N, st));
    CKI(agcn_bwd_core(s, u, P.dY, O, P.Wd, P.Z, P.dP, st));
    CKI(side_join(st));
    int ns5 = 0;
    CKI(agcn_wgrad(s, P.Z, s.ZT, 1, P.dY, O, P.dWs, st, &ns5));
    CKI(wunprep(dW, P.dWs, s, O, st, ns5));
    CKI(colsum(P.dY, O, s.R, O, P.part, db, 0, st));
    CKI(rows_to_bnc(dx, P.dP, s.Cp, 0, C, B, N, st));
    LAUNCH(k_reduce_slabs, dim3(cdiv((long long)N * N, 64)), dim3(1024), 0, st, ds1, (const float*)P.dS, P.nslabS, (long long)N * N, (long long)N * N, 0);
    LAUNCH(k_reduce_slabs, dim3(cdiv((long long)N * N, 64)), dim3(1024), 0, st, ds2, (const float*)(P.dS + (long long)P.nslabS * N * N), P.nslabS, (long long)N * N, (long long)N * N, 0);
    return 0;
}

// ---- cell ---------------------------------------------------------------------------------------
size_t mcrn_cell_workspace_bytes(int B, int N, int din, int H, int cheb_k) {
    if (B < 1 || N < 1 || din < 0 || H < 1 || (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K)) return 0;
    CellPlan P;
    plan_cell(B, N, din, H, cheb_k, nullptr, P);
    return P.total;
}
int mcrn_cell_forward(int B, int N, int din, int H, int cheb_k, const float* x, const float* h, const float* s1,
                      const float* s2, const float* gate_w, const float* gate_b, const float* update_w,
                      const float* update_b, void* ws, size_t ws_bytes, float* hn, void* stream) {
    ENTER();
    if (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K) FAIL("cheb_k must be in 2 .. %d (got %d)", MCRN_MAX_CHEB_K, cheb_k);
    if (B < 1 || N < 1 || din < 0 || H < 1) FAIL("cell: bad sizes");
    if (ws_bytes < mcrn_cell_workspace_bytes(B, N, din, H, cheb_k)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    CellPlan P;
    plan_cell(B, N, din, H, cheb_k, (char*)ws, P);
    const Shp& s = P.s;
    Sup u; u.S[0] = s1; u.S[1] = s2; u.St[0] = P.St1; u.St[1] = P.St2; u.Sf[0] = P.frag[0]; u.Sf[1] = P.frag[1]; u.Stf[0] = P.frag[2]; u.Stf[1] = P.frag[3]; u.Simg[0] = u.Simg[1] = u.Stimg[0] = u.Stimg[1] = nullptr; u.simg_n = 0; u.ldS = N; u.dS = P.dS; u.nslab = P.nslabS; u.slab = (long long)N * N; u.sup_stride = (long long)P.nslabS * N * N;
    CKI(wprep(gate_w, P.Wf[0], P.Wd[0], s, 2 * H, st));
    CKI(wprep(update_w, P.Wf[1], P.Wd[1], s, H, st));
    CKI(build_frags(s1, s2, N, N, P.frag, st));
    CKI(bnc_to_rows(P.Z, s.Cp, 0, H, h, B, N, st));
    CKI(bnc_to_rows(P.Z, s.Cp, H, din, x, B, N, st));
    CKI(bnc_to_rows(P.Y, s.Cp, H, din, x, B, N, st));
    CKI(zero_cols(P.Z, 0, s.Cp, s.C, s.Cp, s.R, 1, st));
    CKI(zero_cols(P.Y, 0, s.Cp, s.C, s.Cp, s.R, 1, st));
    CellW w{P.Wf[0], P.Wd[0], gate_b, P.Wf[1], P.Wd[1], update_b};
    CKI(cell_fwd_core(s, u, P.Z, P.Y, P.zr, P.hc, w, P.hn_rows, H, st));
    return rows_to_bnc(hn, P.hn_rows, H, 0, H, B, N, st);
}
int mcrn_cell_backward(int B, int N, int din, int H, int cheb_k, const float* dhn, const float* s1, const float* s2,
                       const float* gate_w, const float* update_w, void* ws, size_t ws_bytes, float* dx, float* dh,
                       float* ds1, float* ds2, float* dgate_w, float* dgate_b, float* dupdate_w, float* dupdate_b,
                       void* stream) {
    ENTER();
    (void)gate_w; (void)update_w;
    if (cheb_k < 2 || cheb_k > MCRN_MAX_CHEB_K) FAIL("cheb_k must be in 2 .. %d (got %d)", MCRN_MAX_CHEB_K, cheb_k);
    if (ws_bytes < mcrn_cell_workspace_bytes(B, N, din, H, cheb_k)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    CellPlan P;
    plan_cell(B, N, din, H, cheb_k, (char*)ws, P);
    const Shp& s = P.s;
    Sup u; u.S[0] = s1; u.S[1] = s2; u.St[0] = P.St1; u.St[1] = P.St2; u.Sf[0] = P.frag[0]; u.Sf[1] = P.frag[1]; u.Stf[0] = P.frag[2]; u.Stf[1] = P.frag[3]; u.Simg[0] = u.Simg[1] = u.Stimg[0] = u.Stimg[1] = nullptr; u.simg_n = 0; u.ldS = N; u.dS = P.dS; u.nslab = P.nslabS; u.slab = (long long)N * N; u.sup_stride = (long long)P.nslabS * N * N;
    CKI(transpose(P.St1, N, s1, N, nullptr, 0, N, st));
    CKI(transpose(P.St2, N, s2, N, nullptr, 0, N, st));
    CK(hipMemsetAsync(P.dS, 0, (size_t)2 * P.nslabS * N * N * sizeof(float), st));
    CKI(bnc_to_rows(P.dacc, H, 0, H, dhn, B, N, st));
    CellW w{P.Wf[0], P.Wd[0], nullptr, P.Wf[1], P.Wd[1], nullptr};
    CKI(cell_bwd_core(s, u, P.Z, P.Y, P.zr, P.hc, w, P.dacc, P.dU, P.dG, P.dP, P.dQ, P.dacc, P.dxin, st, P.dTu, P.dTg));
    CKI(side_join(st));
    int ns6 = 0;
    CKI(agcn_wgrad(s, P.Z, s.ZT, 1, P.dG, 2 * H, P.dWs[0], st, &ns6));
    int ns7 = 0;
    CKI(agcn_wgrad(s, P.Y, s.ZT, 1, P.dU, H, P.dWs[1], st, &ns7));
    CKI(wunprep(dgate_w, P.dWs[0], s, 2 * H, st, ns6));
    CKI(wunprep(dupdate_w, P.dWs[1], s, H, st, ns7));
    CKI(colsum(P.dG, 2 * H, s.R, 2 * H, P.part, dgate_b, 0, st));
    CKI(colsum(P.dU, H, s.R, H, P.part, dupdate_b, 0, st));
    CKI(rows_to_bnc(dh, P.dacc, H, 0, H, B, N, st));
    if (din > 0) CKI(rows_to_bnc(dx, P.dxin, din, 0, din, B, N, st));
    LAUNCH(k_reduce_slabs, dim3(cdiv((long long)N * N, 64)), dim3(1024), 0, st, ds1, (const float*)P.dS, P.nslabS, (long long)N * N, (long long)N * N, 0);
    LAUNCH(k_reduce_slabs, dim3(cdiv((long long)N * N, 64)), dim3(1024), 0, st, ds2, (const float*)(P.dS + (long long)P.nslabS * N * N), P.nslabS, (long long)N * N, (long long)N * N, 0);
    return 0;
}

// ---- memory head --------------------------------------------------------------------------------
size_t mcrn_memory_workspace_bytes(int B, int N, int H, int M, int D) {
    if (B < 1 || N < 1 || H < 1 || M < 1 || D < 1) return 0;
    MemPlan P;
    plan_mem(B, N, H, M, D, nullptr, P);
    return P.total;
}
int mcrn_memory_forward(int B, int N, int H, int M, int D, const float* h, const float* Mem, const float* Wq,
                        void* ws, size_t ws_bytes, float* value, float* query, float* pos, float* neg, int* ind,
                        void* stream) {
    ENTER();
    if (B < 1 || N < 1 || H < 1 || M < 1 || D < 1) FAIL("memory: bad sizes");
    if (ws_bytes < mcrn_memory_workspace_bytes(B, N, H, M, D)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    MemPlan P;
    plan_mem(B, N, H, M, D, (char*)ws, P);
    CKI(bnc_to_rows(P.h_rows, H, 0, H, h, B, N, st));
    return memory_fwd_launch(P.h_rows, H, Wq, Mem, B, N, H, M, D, P.q_rows, P.att, P.dsc, P.ind, nullptr, 0, value, query,
                             pos, neg, ind, st);
}
int mcrn_memory_backward(int B, int N, int H, int M, int D, const float* h, const float* Mem, const float* Wq,
                         const float* dvalue, const float* dquery, const float* dpos, const float* dneg, void* ws,
                         size_t ws_bytes, float* dh, float* dMem, float* dWq, void* stream) {
    ENTER();
    (void)h;
    if (ws_bytes < mcrn_memory_workspace_bytes(B, N, H, M, D)) FAIL("workspace too small");
    hipStream_t st = (hipStream_t)stream;
    MemPlan P;
    plan_mem(B, N, H, M, D, (char*)ws, P);
    const long long R = (long long)B * N;
    CK(hipMemsetAsync(P.dWq_s, 0, (size_t)NSLAB_T * H * D * sizeof(float), st));
    CK(hipMemsetAsync(P.dMem_s, 0, (size_t)NSLAB_T * M * D * sizeof(float), st));
    CKI(memory_bwd_rows_launch(nullptr, 0, 0, dvalue, dquery, P.att, Mem, B, N, M, D, P.dval, P.dsc, P.dq, st));
    CKI(memory_bwd_gemms(P.h_rows, H, Wq, R, H, M, D, P.att, P.q_rows, P.dval, P.dsc, P.dq, P.dh_rows, H, false,
                         P.dWq_s, P.dMem_s, st));
    LAUNCH(k_reduce_slabs, dim3(cdiv(H * D, 64)), dim3(1024), 0, st, dWq, (const float*)P.dWq_s, NSLAB_T, (long long)H * D, (long long)H * D, 0);
    LAUNCH(k_reduce_slabs, dim3(cdiv(M * D, 64)), dim3(1024), 0, st, dMem, (const float*)P.dMem_s, NSLAB_T, (long long)M * D, (long long)M * D, 0);
    if (dpos) LAUNCH(k_memory_scatter, dim3(cdiv(R * D, 256)), dim3(256), 0, st, dMem, (const int*)P.ind, 0, dpos, B, N, D);
    if (dneg) LAUNCH(k_memory_scatter, dim3(cdiv(R * D, 256)), dim3(256), 0, st, dMem, (const int*)P.ind, 1, dneg, B, N, D);
    return rows_to_bnc(dh, P.dh_rows, H, 0, H, B, N, st);
}

// ---- optimizer tail -----------------------------------------------------------------------------
int mcrn_flat_clip_adam(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                        float eps, int step, float max_norm, float grad_scale, float* scratch, float* total_norm_out,
                        void* stream) {
    ENTER();
    if (!p || !g || !m || !v || !scratch || n < 1 || step < 1) FAIL("flat_clip_adam: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int nblk = cdiv(n, 256 * 8);
    if (nblk > 1000) nblk = 1000;
    LAUNCH(k_sumsq_stage1, dim3(nblk), dim3(256), 0, st, (const float*)g, n, grad_scale, scratch);
    LAUNCH(k_sumsq_stage2, dim3(1), dim3(64), 0, st, (const float*)scratch, nblk, scratch + 1000);
    // bias corrections in double like torch (Python floats): 1 - 0.999f^step in fp32 loses ~6e-5 to cancellation
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    LAUNCH(k_clip_adam, dim3(cdiv(n, 256)), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, max_norm,
           grad_scale, (const float*)(scratch + 1000), total_norm_out);
    return 0;
}

// ---- fused trainer loss ---------------------------------------------------------------------------
int mcrn_loss_fwd_bwd(int B, int T, int N, int od, int D, const float* output, const float* labels,
                      const float* query, const float* pos, const float* neg, float mean, float stdv, float lamb,
                      float lamb1, float margin, float* scratch, float* losses, float* d_output, float* d_query,
                      void* stream) {
    ENTER();
    if (B < 1 || T < 1 || N < 1 || od < 1 || D < 1 || !output || !labels || !query || !pos || !neg || !scratch ||
        !losses || !d_output || !d_query)
        FAIL("loss: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long long nout = (long long)B * T * N * od, rows = (long long)B * N;
    int nblk = (int)std::max<long long>(cdiv(nout, 256 * 4), cdiv(rows, 4 * 4));   // ~4 rows per wave in the triplet part
    if (nblk > 1024) nblk = 1024;
    LAUNCH(k_loss_stage1, dim3(nblk), dim3(256), 0, st, output, labels, nout, query, pos, neg, rows, D, mean, stdv,
           lamb, lamb1, margin, scratch, d_query);
    LAUNCH(k_loss_stage2, dim3(1), dim3(64), 0, st, (const float*)scratch, nblk, nout, rows, D, lamb, lamb1, losses,
           scratch + 4096);
    LAUNCH(k_loss_dout, dim3(cdiv(nout, 256)), dim3(256), 0, st, output, labels, nout, mean, stdv,
           (const float*)(scratch + 4096), d_output);
    return 0;
}

// ---- evaluation metrics, one launch per batch (model/traintest_MegaCRN.py:63-93) -------------------
int mcrn_eval_metrics(int B, int T, int N, int od, int D, const float* output, const float* labels,
                      const float* query, const float* pos, const float* neg, float mean, float stdv, float lamb,
                      float lamb1, float margin, const int* horizons, int nh, float* scratch, float* acc,
                      void* stream) {
    ENTER();
    if (B < 1 || T < 1 || N < 1 || od < 1 || D < 1 || !output || !labels || !query || !pos || !neg || !scratch || !acc ||
        nh < 0 || nh > 3 || (nh > 0 && !horizons))
        FAIL("eval_metrics: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int h[3] = {-1, -1, -1};
    for (int i = 0; i < nh; ++i) {
        if (horizons[i] < 1 || horizons[i] > T) FAIL("eval_metrics: horizon %d outside 1..%d", horizons[i], T);
        h[i] = horizons[i] - 1;
    }
    const long long nout = (long long)B * T * N * od, rows = (long long)B * N;
    int nblk = (int)std::max<long long>(cdiv(nout, 256 * 4), cdiv(rows, 4 * 4));
    if (nblk > 1024) nblk = 1024;
    // scratch: [0] ticket (zero before the first call; the kernel re-arms it), [64 ..) per-block partials
    LAUNCH(k_eval_metrics, dim3(nblk), dim3(256), 0, st, output, labels, nout, T, (long long)N * od, h[0], h[1], h[2], query,
           pos, neg, rows, D, mean, stdv, lamb, lamb1, margin, scratch + 64, reinterpret_cast<unsigned*>(scratch), acc);
    return 0;
}

// ---- GEMM test hook -----------------------------------------------------------------------------
int mcrn_gemm_f32(int M, int N, int K, int transA, int transB, const float* A, const float* B, float* C, float alpha,
                  float beta, int nsplit, float* slabs, void* stream) {
    ENTER();
    hipStream_t st = (hipStream_t)stream;
    GemmP p = gp();
    p.M = M; p.N = N; p.K = K;
    p.A[0] = A; p.B[0] = B;
    if (!transA) { p.am = plain(K); p.ak = plain(1); } else { p.am = plain(1); p.ak = plain(M); }
    if (!transB) { p.bk = plain(N); p.bn = plain(1); } else { p.bk = plain(1); p.bn = plain(K); }
    p.cm = plain(N); p.cn = plain(1);
    p.alpha = alpha;
    if (nsplit > 1) {
        if (!slabs) FAIL("gemm: nsplit > 1 needs slabs");
        CK(hipMemsetAsync(slabs, 0, (size_t)nsplit * M * N * sizeof(float), st));
        p.C[0] = slabs; p.Cin[0] = slabs; p.beta = 1.f; p.slab = (long long)M * N;
        CKI(gemm(p, !transA, transB != 0, nsplit, ROLE_MISC, st));
        // C = alpha*sum(slabs) (alpha already applied) + beta*C
        if (beta == 0.f) {
            LAUNCH(k_reduce_slabs, dim3(cdiv((long long)M * N, 64)), dim3(1024), 0, st, C, (const float*)slabs, nsplit, (long long)M * N, (long long)M * N, 0);
        } else {
            if (beta != 1.f) FAIL("gemm: split-K supports beta in {0,1}");
            LAUNCH(k_reduce_slabs, dim3(cdiv((long long)M * N, 64)), dim3(1024), 0, st, C, (const float*)slabs, nsplit, (long long)M * N, (long long)M * N, 1);
        }
        return 0;
    }
    p.C[0] = C;
    if (beta != 0.f) { p.Cin[0] = C; p.beta = beta; }
    return gemm(p, !transA, transB != 0, 0, ROLE_MISC, st);
}

}  // extern "C"
#pragma GCC visibility pop
