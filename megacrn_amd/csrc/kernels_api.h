// kernels_api.h - parameter blocks, enums and host entry points of the kernel translation units.
// engine.hip (orchestration + C ABI) includes ONLY this file, gemm_bf16_api.h, wgrad_stream_api.h and ops.h, so that it
// rebuilds in seconds; the kernels live in their own units (tiled_unit.hip: gemm_f32.h + gemm_bf16x3.h;
// prop_unit.hip: prop_small.h; dgrad_unit.hip: dgrad_stream.h), each of which also includes this file for the structs.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

namespace mcrn {

// ---- tiled GEMM (gemm_f32.h / gemm_bf16x3.h) ------------------------------------------------------------------
struct Dim2 {          // off(i) = (i / inner) * hi + (i % inner) * lo ; inner <= 0 means plain i*lo
    int inner;
    long long hi;
    long long lo;
};
static inline Dim2 plain(long long lo) { return Dim2{0, 0, lo}; }
static inline Dim2 two(int inner, long long hi, long long lo) { return Dim2{inner, hi, lo}; }

enum Epi { EPI_STORE = 0, EPI_GATE = 1, EPI_UPDATE = 2, EPI_BIAS = 3 };

// Role tag: only changes the kernel's symbol name so that rocprofv3 --stats reports the hot
// contractions separately (all roles share one body).
enum Role {
    ROLE_MISC = 0,    // adjacency / memory head / proj odds and ends
    ROLE_PROP = 1,    // K-hop propagation  S x Z[g]            (model/MegaCRN.py:25)   <- north_star kernel
    ROLE_WP = 2,      // weight pool + fused GRU epilogue        (model/MegaCRN.py:27,43-47)
    ROLE_DGRAD = 3,   // dY x W^T -> plane gradients
    ROLE_PROPT = 4,   // S^T x dZ[g]  (backward propagation)
    ROLE_DS = 5,      // adjacency gradient  dZ x Z^T  (split-K)
    ROLE_WGRAD = 6,   // deferred weight gradient Z^T x dY (split-K)
    ROLE_COUNT = 7,
    // timing-only id (mcrn_prof_begin): the hoisted once-per-stack products of the input channels in the bf16 mode.  They
    // run the ROLE_PROP kernels (same epilogue) but are narrow (T*B*d columns, split-K): timed apart, so that role 1 is the
    // per-step propagation alone
    PROF_ROLE_PROP_IN = 7,
    PROF_ROLE_COUNT = 8
};

struct GemmP {
    const float* A[2];
    const float* B[2];
    float* C[2];
    const float* Cin[2];        // nullable (beta ignored then)
    long long ak_hi[2], bk_hi[2];   // per-batch override of ak.hi / bk.hi
    Dim2 am, ak, bk, bn, cm, cn;
    int M, N, K;
    int nbatch, nsplit, kchunk;     // grid.z = nbatch*nsplit ; kchunk multiple of 16
    long long slab;                 // C/Cin offset per split
    float alpha, beta;
    int epi;
    // fused GRU epilogues (model/MegaCRN.py:43-47)
    const float* bias;
    const float* hsrc; long long hsrc_ld;     // previous state h[r*hsrc_ld + c]
    float* out2;       long long out2_ld;     // GATE: z*h  ; UPDATE: new state
    const float* zr;                           // UPDATE: sigmoid gates (R x 2H)
    int H;
    double alg_flops;   // host-side bookkeeping only (algorithmic flops of this launch)
    int vec;            // bf16x3 path: bit0 = A rows float4-loadable, bit1 = B
    // bf16x3 path: optional pre-split image of the B operand (static weights, gemm_bf16x3.h: k_bimg_build):
    //   Bimg[((kt*4 + kg)*2 + hl) * bimg_n + n] = 8 bf16 (hi | lo) of B[k = 32 kt + 8 kg + 0..7][n]
    const uint4* Bimg;
    int bimg_n;
    // same for a static A operand (the adjacency of the tiled propagation at N > 256), per batch, KC layout
    const uint4* Aimg[2];
    int aimg_n;
};

struct GemmStats { long long launches; double flops; };
extern GemmStats g_gemm_stats;
extern int g_force_cfg;   // >= 0: force this tile configuration (tuning sweeps)
extern int g_debug;       // ablation bits for the bf16x3 kernel (0 in production)

// tile configurations: {BM, BN, waves in M, waves in N}
static const int NCFG = 7;
static const int kCfg[NCFG][4] = {{128, 128, 2, 2}, {64, 128, 2, 2}, {128, 64, 2, 2}, {64, 64, 2, 2},
                                  {32, 128, 1, 4},  {256, 64, 4, 1}, {64, 256, 1, 4}};
static inline size_t bimg_uint4(int K, int N) { return (size_t)((K + 31) / 32) * 4 * 2 * ((N + 3) & ~3); }

// ---- adjacency-stationary kernels for small graphs (prop_small.h) ----------------------------------------------
struct PropP {
    const uint4* Sf[2][2];      // [batch][segment] fragment-ordered split adjacency
    const float* X[2][2];       // [batch][segment] right operand plane (N x ncols, row stride ld)
    float* C[2];
    const float* Cin[2];        // nullable
    int nseg;                   // 1 or 2 (K-concatenated [S_a | S_b] x [X_a ; X_b])
    int N, ncols;
    long long ld;
    float alpha, beta;
};

struct Prop2P {
    const uint4* Sf[2];         // forward: S1,S2 fragments ; backward: S1^T,S2^T fragments
    float* base;                // plane set (Z for forward, dP for backward)
    float* extra;               // backward: support 1 stores S_2^T d1t_2 here (consumers add it to dP[0])
    long long PS, ld;
    int N, ncols;
    void *ev0, *ev1;            // host side only: events attached to the dispatch itself (roofline leg, see Bf16GemmP); nullptr otherwise
    // forward, state columns only (nunits > 0): unit u covers the 64 columns  (u / cps) * cstride + (u % cps) * 64 ..  - the
    // H = 64 * cps state channels of sample u / cps inside its row of cstride = Cp floats; the input / pad channels of the planes
    // 1 .. 4 come from a once-per-stack product (engine.hip: hoist_inputs_small)
    int nunits, cps, cstride;
    // backward: 1 = d1t_s = d1_s + S_s^T e2_s is NOT written back to plane 1 + 2s.  The adjacency gradient of the fused model path reads
    // the raw d-grad planes (round 5: dS = d1 x0^T and e2 x0^T as four blocks, the chain rule of 2 S S once per step), so d1t lives only
    // in this kernel's LDS image between its two hops: two plane writes less per launch.
    int no_d1;
    int nblk;                   // forward, set by the launcher: unit ranges per support (the grid is 1-D, see prop2_fwd_kernel)
};

struct DsP {
    // [output block][segment].  Feature-recursion form (stand-alone ops, cheb_k = 2): block = support, 2 segments per AGCN call at
    // cheb_k = 3 (d1t x0^T, e2 x1^T).  Fused model path at cheb_k = 3 (round 5): FOUR blocks [d1_a, e2_a, d1_b, e2_b] x x0^T, one segment
    // per call - plane 0 is the only right operand (5 planes read instead of 7).  A cell's two calls share one launch.
    const float* A[4][4];
    const float* B[4][4];
    float* C[4];                // slab 0 of the block; slab z at + z*slab
    long long slab;
    int nseg, N, ncols, kchunk; // kchunk multiple of 16
    long long ld, ldc;
};

static inline bool prop_small_ok(int N, long long ld, int ncols) {
    return N <= 256 && (ncols % 4) == 0 && (ld % 4) == 0;
}
// the fused two-hop kernels (cheb_k = 3) also take 256 < N <= 352 (lo fragments of S streamed, see PropBlock::WIDE)
static const int PROP2_MAX_N = 352;
static inline bool prop2_ok(int N, long long ld, int ncols) {
    return N <= PROP2_MAX_N && (ncols % 4) == 0 && (ld % 4) == 0;
}
static inline size_t sfrag_uint4(int N) {
    const int NF = (N + 31) / 32;
    return (size_t)NF * 2 * NF * 2 * 64;
}

// ---- streaming d-grad (dgrad_stream.h) -----------------------------------------------------------------------
static inline size_t wfrag_uint4(int rows, int K) { return (size_t)((rows + 31) / 32) * ((K + 15) / 16) * 2 * 64; }
static inline bool dgrad_stream_ok(int O) { return O >= 16 && ((O % 16 == 0 && O <= 128) || (O % 32 == 0 && O <= 256)); }

struct DgradP {
    const float* dY;        // [R][O]
    const uint4* Wfrag;     // k_wfrag_build image of Wd [(g, c')][o]
    float* dP;              // [G][R][Cp]  (plane stride PS)
    unsigned short* dPb;    // MCRN_BF16: planes 1.. are written as bf16 here ([G-1][.][Cp] rows, plane stride PSb) and
    long long PSb;          //   NOT to dP: they are only ever read as bf16 operands (S^T product, adjacency gradient)
    long long R, PS;
    int O, ncols, Cp;       // ncols = G*Cp
    int ncf, parts, cf_per_part;
    int dbg;                // -DMCRN_ABLATE builds only (MCRN_DEBUG bits): 1 = no stores, 2 = no MFMA, 4 = no B staging
    // MCRN_BF16, hoisted backward (H > 0): the state channels c < H of the planes 1.. go to dPb PACKED as [plane][R][H]
    // (plane stride PSb): the [k][n] operand of the S^T product and the K-contiguous operand of the adjacency gradient cover
    // the state channels only.  The d input channels of those planes go, as bf16, straight into the stack-wide operand of
    // the adjacency gradient's input part:  dPin[((g - 1) * N + n) * kin + in_col0 + b * d + j]   (r = n * B + b);
    // pad channels are not written at all.
    int H, d, B;
    unsigned short* dPin;
    long long kin, in_plane;  // row length of dPin ; (g - 1) * in_plane = N * kin floats between the planes' row blocks
    int in_col0;
    long long dPb_lo, dPin_lo;   // > 0 (hoisted backward, hi/lo operand pairs: gemm_bf16.h nterm = 3): the bf16 residuals are written that many
                                 // elements behind dPb / dPin
};


// ---- host entry points defined by the kernel units ------------------------------------------------------------
namespace ext {
hipError_t launch_gemm_f32(GemmP p, bool akc, bool bkc, int max_split, hipStream_t st);
hipError_t launch_gemm_x3(GemmP p, bool akc, bool bkc, int max_split, int role, hipStream_t st);
// pre-split tile image of a static B operand (see k_bimg_build in gemm_bf16x3.h)
hipError_t launch_bimg_build(const float* src, long long sk, long long sn, int K, int N, int npad, int kc_layout, uint4* img, hipStream_t st);
hipError_t launch_prop_small(const PropP& p, int nbatch, hipStream_t st);
hipError_t launch_prop2_fwd(const Prop2P& p, hipStream_t st);
hipError_t launch_prop2_bwd(const Prop2P& p, hipStream_t st);
hipError_t launch_ds_small(DsP p, int nslab, hipStream_t st, int nblk = 2);
// fragment images of n <= 8 matrices (S or S^T each) in one launch
hipError_t launch_sfrag_multi(const float* const* S, const int* transpose, uint4* const* out, int n, long long ldS, int N, hipStream_t st);
hipError_t launch_sfrag(const float* S, long long ldS, int N, int transpose, uint4* out, hipStream_t st);
hipError_t launch_dgrad_stream(DgradP p, hipStream_t st);
// B-fragment image of Wd for the streaming d-grad (k_wfrag_build in dgrad_stream.h)
hipError_t launch_wfrag_build(const float* W, long long ld, int rows, int K, int KS, uint4* out, long long tot, hipStream_t st);
#ifdef MCRN_TIMELINE
hipError_t timeline_copy(void* host_out, size_t bytes);      // in-kernel phase stamps of the prop_small.h kernels
#endif
}  // namespace ext

}  // namespace mcrn
