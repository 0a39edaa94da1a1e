// gemm_f32.h - exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), wave64.
//
// One kernel serves every contraction of the MegaCRN hot path:
//   C[r,c] = epi( alpha * sum_k A(r,k) * B(k,c) + beta * Cin[r,c] )
// A, B, C are addressed through two-level strided index maps (Dim2), which is what lets the
// node-major plane layout Z[g][n][b][c] be consumed directly as
//   * the (N x B*Cp) right operand of the K-hop propagation  S x Z[g]
//   * the (N*B x G*Cp) left operand of the weight-pool contraction (concat folded into K)
//   * both operands of the N x N adjacency-gradient contraction
// without any cat/transpose pass (reference: model/MegaCRN.py:24-27 materialises both).
//
// Tiling: 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) built from 32x32
// MFMA fragments; BK = 16.  Operands are staged through LDS with a register prefetch of the next
// K-tile.  LDS images are conflict-free for ds_read_b32 / ds_write_b32 (stride 17 for
// K-contiguous sources, unit stride for row-contiguous ones).
//
// fp32 MFMA runs at 64 FLOP/clk/SIMD (157 TF chip peak), so one ds_read_b32 per operand per
// 64-cycle MFMA leaves the LDS and the global path far from critical; the kernel is MFMA-issue
// bound once tiles are full.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_api.h"

namespace mcrn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Measurement-only build (-DMCRN_TIMELINE=1|2, `make timeline [FENCE=2]`): thread 0 of every workgroup stamps the
// 100 MHz wall clock at phase boundaries; =2 also drains the memory counters first so that a phase's
// time includes the latency of what it issued.  kind 0 = prop2_fwd, 1 = prop2_bwd, 2 = ds_small,
// 3 + ROLE = gemm_bf16x3_kernel (accumulated per-phase times of thread 0, see tools/timeline.py).
#ifdef MCRN_TIMELINE
__device__ unsigned long long g_tl[10][512][12];   // kinds 0-2: prop_small.h kernels; 3 + ROLE: tiled bf16x3 GEMM
#define MCRN_TL(kind, i)                                                                            \
    do {                                                                                            \
        if (MCRN_TIMELINE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         \
        if (threadIdx.x == 0) g_tl[kind][(blockIdx.y * gridDim.x + blockIdx.x) & 511][i] = wall_clock64(); \
    } while (0)
#else
#define MCRN_TL(kind, i)
#endif

__device__ __forceinline__ long long d2off(int inner, long long hi, long long lo, int i) {
    if (inner <= 0) return (long long)i * lo;
    int q = i / inner;
    return (long long)q * hi + (long long)(i - q * inner) * lo;
}

// ---- shared epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5)
// Loads never sit under a per-element branch: a fragment's 16 (or 32) operand loads are issued back to back from
// clamped, always-valid addresses and only the stores are predicated.  (With `if (r < M) { load; ...; store }`
// per element every load got its own basic block and a full memory round trip: 8.7 of 31.7 us per workgroup
// in the weight-pool GEMM, measured with the in-kernel timeline.)  `rows_in` is wave-uniform.
#define MCRN_EROW(v) (((v) & 3) + 8 * ((v) >> 2))
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[FM][FN], int batch, int split,
                                              int r_base, int c_base) {
    float* __restrict__ Cb = p.C[batch] + (long long)split * p.slab;
    const float* __restrict__ Cin = p.Cin[batch] ? p.Cin[batch] + (long long)split * p.slab : nullptr;
    const int rw = __builtin_amdgcn_readfirstlane(r_base - 4 * ((int)(threadIdx.x & 63) >> 5));   // first row of the wave's block
    const int mlast = p.M - 1;
    // One fragment (i, j) at a time.  `ld*` are re-read through an opaque move per fragment and a scheduling
    // barrier closes it, so the address arithmetic of all FM*FN fragments is not hoisted in front of the first
    // one (that spilled the accumulators at 128x128).
#define MCRN_EPI_FRAGS(BODY)                                                                        \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                               \
        const int c = c_base + j * 32;                                                              \
        const bool cok = c < p.N;                                                                   \
        const int cc = cok ? c : p.N - 1;                                                           \
        (void)cc;                                                                                   \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                           \
            const int r0 = r_base + i * 32;                                                         \
            const bool rows_in = rw + i * 32 + 32 <= p.M;                                           \
            long long ld0 = LD0, ld1 = LD1, ld2 = LD2;                                              \
            asm volatile("" : "+s"(ld0), "+s"(ld1), "+s"(ld2));                                     \
            (void)ld1; (void)ld2;                                                                   \
            BODY                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
    }
    // row used for ADDRESSES (clamped into range) and the store predicate of accumulator register v
#define MCRN_RA(v) (rows_in ? r0 + MCRN_EROW(v) : min(r0 + MCRN_EROW(v), mlast))
#define MCRN_LIVE(v) (cok && (rows_in || r0 + MCRN_EROW(v) <= mlast))
    if (p.epi == EPI_STORE) {
#define LD0 p.cm.lo
#define LD1 0
#define LD2 0
        if (Cin) {
            MCRN_EPI_FRAGS({
                const long long co = d2off(p.cn.inner, p.cn.hi, p.cn.lo, cc);   /* C rows are always plain-strided */
                float cin[16];
                _Pragma("unroll") for (int v = 0; v < 16; ++v) cin[v] = Cin[MCRN_RA(v) * ld0 + co];
                _Pragma("unroll") for (int v = 0; v < 16; ++v)
                    if (MCRN_LIVE(v)) Cb[(r0 + MCRN_EROW(v)) * ld0 + co] = p.alpha * acc[i][j][v] + p.beta * cin[v];
            })
        } else {
            MCRN_EPI_FRAGS({
                const long long co = d2off(p.cn.inner, p.cn.hi, p.cn.lo, cc);
                _Pragma("unroll") for (int v = 0; v < 16; ++v)
                    if (MCRN_LIVE(v)) Cb[(r0 + MCRN_EROW(v)) * ld0 + co] = p.alpha * acc[i][j][v];
            })
        }
    } else if (p.epi == EPI_BIAS) {
        MCRN_EPI_FRAGS({
            const long long co = d2off(p.cn.inner, p.cn.hi, p.cn.lo, cc);
            const float bj = p.bias[cc];
            _Pragma("unroll") for (int v = 0; v < 16; ++v)
                if (MCRN_LIVE(v)) Cb[(r0 + MCRN_EROW(v)) * ld0 + co] = acc[i][j][v] + bj;
        })
#undef LD0
#undef LD1
#undef LD2
    } else if (p.epi == EPI_GATE) {
        // z_r = sigmoid(AGCN_gate) ; candidate state input = z*h   (MegaCRN.py:43-45)
#define LD0 (long long)(2 * p.H)
#define LD1 p.hsrc_ld
#define LD2 p.out2_ld
        MCRN_EPI_FRAGS({
            const float bj = p.bias[cc];
            const bool isz = c < p.H;
            const int ch = isz ? c : 0;
            float h[16];
            _Pragma("unroll") for (int v = 0; v < 16; ++v) h[v] = p.hsrc[MCRN_RA(v) * ld1 + ch];
            _Pragma("unroll") for (int v = 0; v < 16; ++v) {
                const float g = 1.f / (1.f + expf(-(acc[i][j][v] + bj)));
                if (MCRN_LIVE(v)) {
                    const int r = r0 + MCRN_EROW(v);
                    Cb[r * ld0 + c] = g;
                    if (isz) p.out2[r * ld2 + c] = g * h[v];
                }
            }
        })
#undef LD0
    } else {
        // hc = tanh(AGCN_update) ; h' = r*h + (1-r)*hc               (MegaCRN.py:46-47)
#define LD0 (long long)p.H
        MCRN_EPI_FRAGS({
            const float bj = p.bias[cc];
            float rg[16];
            float h[16];
            _Pragma("unroll") for (int v = 0; v < 16; ++v) {
                const int ra = MCRN_RA(v);
                rg[v] = p.zr[ra * (2 * ld0) + ld0 + cc];
                h[v] = p.hsrc[ra * ld1 + cc];
            }
            _Pragma("unroll") for (int v = 0; v < 16; ++v) {
                const float hc = tanhf(acc[i][j][v] + bj);
                if (MCRN_LIVE(v)) {
                    const int r = r0 + MCRN_EROW(v);
                    Cb[r * ld0 + c] = hc;
                    p.out2[r * ld2 + c] = rg[v] * h[v] + (1.f - rg[v]) * hc;
                }
            }
        })
#undef LD0
#undef LD1
#undef LD2
    }
#undef MCRN_RA
#undef MCRN_LIVE
#undef MCRN_EPI_FRAGS
}

// ---- operand tile: E rows (m or n) x 16 k ------------------------------------------------
// KC: memory is contiguous along k.  thread -> k = tid&15, rows (tid>>4) + 16q.
// RC: memory is contiguous along the row index.  thread -> row = (tid&31)+32q, k = (tid>>5)+8p.
template <int E, bool KC>
struct Tile {
    static constexpr int NE = E / 16;                      // elements per thread
    static constexpr int SZ = KC ? E * 17 : 16 * E;        // LDS floats
    long long roff[KC ? E / 16 : E / 32];
    float reg[NE];

    __device__ __forceinline__ void init_rows(int tid, int row0, int nrows, const Dim2& d) {
        if (KC) {
#pragma unroll
            for (int q = 0; q < E / 16; ++q) {
                int r = row0 + (tid >> 4) + 16 * q;
                roff[q] = r < nrows ? d2off(d.inner, d.hi, d.lo, r) : -1;
            }
        } else {
#pragma unroll
            for (int q = 0; q < E / 32; ++q) {
                int r = row0 + (tid & 31) + 32 * q;
                roff[q] = r < nrows ? d2off(d.inner, d.hi, d.lo, r) : -1;
            }
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ base, int tid, int k0, int kend,
                                         int kinner, long long khi, long long klo) {
        if (KC) {
            int k = k0 + (tid & 15);
            bool kv = k < kend;
            long long ko = kv ? d2off(kinner, khi, klo, k) : 0;
#pragma unroll
            for (int q = 0; q < E / 16; ++q)
                reg[q] = (kv && roff[q] >= 0) ? base[roff[q] + ko] : 0.f;
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                int k = k0 + (tid >> 5) + 8 * p;
                bool kv = k < kend;
                long long ko = kv ? d2off(kinner, khi, klo, k) : 0;
#pragma unroll
                for (int q = 0; q < E / 32; ++q)
                    reg[p * (E / 32) + q] = (kv && roff[q] >= 0) ? base[roff[q] + ko] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* s, int tid) const {
        if (KC) {
#pragma unroll
            for (int q = 0; q < E / 16; ++q) s[((tid >> 4) + 16 * q) * 17 + (tid & 15)] = reg[q];
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < E / 32; ++q)
                    s[((tid >> 5) + 8 * p) * E + (tid & 31) + 32 * q] = reg[p * (E / 32) + q];
        }
    }
    static __device__ __forceinline__ float at(const float* s, int row, int k) {
        return KC ? s[row * 17 + k] : s[k * E + row];
    }
};

template <int BM, int BN, int WGM, int WGN, bool AKC, bool BKC, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmP p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    using TA = Tile<BM, AKC>;
    using TB = Tile<BN, BKC>;
    __shared__ float smem[TA::SZ + TB::SZ];
    float* sA = smem;
    float* sB = smem + TA::SZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile_m = blockIdx.x % tiles_m, tile_n = blockIdx.x / tiles_m;
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int z = blockIdx.z;
    const int batch = z / p.nsplit, split = z - batch * p.nsplit;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    if (kbeg >= kend) return;

    const float* __restrict__ Ab = p.A[batch];
    const float* __restrict__ Bb = p.B[batch];
    const long long akhi = p.ak_hi[batch], bkhi = p.bk_hi[batch];

    TA ta;
    TB tb;
    ta.init_rows(tid, m_blk, p.M, p.am);
    tb.init_rows(tid, n_blk, p.N, p.bn);

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    ta.load(Ab, tid, kbeg, kend, p.ak.inner, akhi, p.ak.lo);
    tb.load(Bb, tid, kbeg, kend, p.bk.inner, bkhi, p.bk.lo);
    ta.store(sA, tid);
    tb.store(sB, tid);
    __syncthreads();

    const int l31 = lane & 31, kh = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const bool more = k0 + 16 < kend;
        if (more) {
            ta.load(Ab, tid, k0 + 16, kend, p.ak.inner, akhi, p.ak.lo);
            tb.load(Bb, tid, k0 + 16, kend, p.bk.inner, bkhi, p.bk.lo);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = TA::at(sA, wm * WM + i * 32 + l31, 2 * s + kh);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = TB::at(sB, wn * WN + j * 32 + l31, 2 * s + kh);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            ta.store(sA, tid);
            tb.store(sB, tid);
            __syncthreads();
        }
    }

    gemm_epilogue<FM, FN>(p, acc, batch, split, m_blk + wm * WM + 4 * kh, n_blk + wn * WN + l31);
}

// ---- host launcher ------------------------------------------------------------------------
#ifndef MCRN_PROBE
template <int BM, int BN, int WGM, int WGN, bool AKC, bool BKC, int ROLE>
static inline hipError_t launch_one(const GemmP& p, hipStream_t st) {
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, p.nbatch * p.nsplit);
    (void)hipGetLastError();
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WGM, WGN, AKC, BKC, ROLE>), grid, dim3(256), 0, st, p);
    return hipGetLastError();
}
template <bool AKC, bool BKC, int ROLE>
static inline hipError_t launch_cfg(const GemmP& p, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_one<128, 128, 2, 2, AKC, BKC, ROLE>(p, st);
        case 1: return launch_one<64, 128, 2, 2, AKC, BKC, ROLE>(p, st);
        case 2: return launch_one<128, 64, 2, 2, AKC, BKC, ROLE>(p, st);
        case 3: return launch_one<64, 64, 2, 2, AKC, BKC, ROLE>(p, st);
        case 4: return launch_one<32, 128, 1, 4, AKC, BKC, ROLE>(p, st);
        case 5: return launch_one<256, 64, 4, 1, AKC, BKC, ROLE>(p, st);
        default: return launch_one<64, 256, 1, 4, AKC, BKC, ROLE>(p, st);
    }
}
static inline hipError_t launch_f32(const GemmP& p, bool akc, bool bkc, int cfg, hipStream_t st) {
    if (akc && !bkc) return launch_cfg<true, false, ROLE_MISC>(p, cfg, st);
    if (akc && bkc) return launch_cfg<true, true, ROLE_MISC>(p, cfg, st);
    if (!akc && !bkc) return launch_cfg<false, false, ROLE_MISC>(p, cfg, st);
    return launch_cfg<false, true, ROLE_MISC>(p, cfg, st);
}

// Pick tile shape and split-K.  max_split: 0 = never split (C is a real output), otherwise the
// number of slabs the caller provisioned behind C (stride p.slab).  kgran = K-tile depth.
static inline int choose_cfg(GemmP& p, int max_split, int kgran) {
    if (p.nbatch <= 0) p.nbatch = 1;
    // Cost model (seconds): the larger of
    //   matrix-core time  : ceil(workgroups / 256 CUs) * tile MACs / per-CU rate
    //   operand traffic   : A is re-read once per N-tile, B once per M-tile (operands > one XCD L2
    //                       come from MALL/HBM at ~4.5 TB/s, small ones from L2 at ~20 TB/s) + C
    // plus a fixed per-K-tile latency term for the un-overlapped part of the pipeline.
    const double cu_rate = (kgran == 32 ? 1365.0 : 256.0) * 2.2e9;     // flop/s per CU (bf16x3 / f32)
    int best = 3, best_split = 1;
    double best_t = 1e300;
    const double Abytes = 4.0 * p.M * (double)p.K * p.nbatch, Bbytes = 4.0 * p.N * (double)p.K * p.nbatch;
    for (int i = 0; i < NCFG; ++i) {
        if (g_force_cfg >= 0 && i != g_force_cfg) continue;
        const int bm = kCfg[i][0], bn = kCfg[i][1];
        const long long tm = (p.M + bm - 1) / bm, tn = (p.N + bn - 1) / bn;
        const long long tiles = tm * tn * p.nbatch;
        int ns = 1;
        if (max_split > 1) {
            ns = (int)((512 + tiles - 1) / tiles);
            if (ns > max_split) ns = max_split;
            int maxk = p.K / 64;
            if (maxk < 1) maxk = 1;
            if (ns > maxk) ns = maxk;
            if (ns < 1) ns = 1;
        }
        const double kc = (double)p.K / ns;
        const double rounds = ceil((double)tiles * ns / 256.0);
        const double t_mma = rounds * (2.0 * bm * bn * kc) / cu_rate;
        const double bwA = Abytes < 2.0e6 ? 20e12 : 4.5e12, bwB = Bbytes < 2.0e6 ? 20e12 : 4.5e12;
        const double t_mem = Abytes * tn / bwA + Bbytes * tm / bwB + 4.0 * p.M * (double)p.N * p.nbatch * ns * (p.Cin[0] ? 2 : 1) / 4.5e12;
        const double t_lat = rounds * (kc / kgran) * 0.35e-6 + 2e-6;
        const double t = (t_mma > t_mem ? t_mma : t_mem) + t_lat;
        if (t < best_t) { best_t = t; best = i; best_split = ns; }
    }
    int kchunk = ((p.K + best_split - 1) / best_split + kgran - 1) / kgran * kgran;
    p.nsplit = (p.K + kchunk - 1) / kchunk;
    p.kchunk = kchunk;
    g_gemm_stats.launches++;
    g_gemm_stats.flops += 2.0 * p.M * p.N * (double)p.K * p.nbatch;
    return best;
}

static inline hipError_t launch_gemm_f32(GemmP p, bool akc, bool bkc, int max_split, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipSuccess;
    const int cfg = choose_cfg(p, max_split, 16);
    return launch_f32(p, akc, bkc, cfg, st);
}

#endif  // MCRN_PROBE

}  // namespace mcrn
