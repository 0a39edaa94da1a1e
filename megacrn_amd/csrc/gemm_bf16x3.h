// gemm_bf16x3.h - fp32-equivalent GEMM on the bf16 matrix cores of gfx950 (wave64).
//
// Every fp32 operand element x is split on the fly into two bf16 values  hi = bf16(x),
// lo = bf16(x - hi)  (16 mantissa bits together) and the product is evaluated as
//     A*B ~= Ahi*Bhi + Ahi*Blo + Alo*Bhi          (fp32 accumulate in the MFMA)
// with v_mfma_f32_32x32x16_bf16: three MFMAs at 16x the fp32-MFMA rate = 5.3x the throughput of
// the exact v_mfma_f32_32x32x2_f32 path (gemm_f32.h) at an error of ~1e-5 relative (the dropped
// lo*lo term is 2^-16 of the product), an order of magnitude inside the 1e-4 parity budget.
// Storage stays fp32; the split costs ~3 VALU ops per loaded element and every element is reused
// BM (or BN) times.
//
// Same GemmP interface / index maps / fused epilogues as gemm_f32.h.  Differences:
//   BK = 32.  A thread owns (row, 8 consecutive k) of an operand tile: K-contiguous sources are
//   fetched as two float4 per thread, row-contiguous sources as 8 coalesced dwords (the transposition
//   to "k-contiguous per lane" that the bf16 MFMA wants happens in registers), converted, and stored
//   as one 16-byte hi and one 16-byte lo vector into LDS images [row][32 bf16 + 8 pad] (80-byte
//   rows: conflict-free for ds_write_b128 and for the ds_read_b128 fragment reads).
#pragma once
#include "gemm_f32.h"

namespace mcrn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8 fp32 -> 8 bf16 hi + 8 bf16 lo (round-to-nearest-even twice)
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned r = cvt_pk_bf16(v[2 * j], v[2 * j + 1]);
        const float h0 = __uint_as_float(r << 16), h1 = __uint_as_float(r & 0xFFFF0000u);
        h[j] = r;
        l[j] = cvt_pk_bf16(v[2 * j] - h0, v[2 * j + 1] - h1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Pre-split image of a static B operand (weights): built once per step, consumed by every tile load of every
// launch that uses it.  element(k, n) = src[k*sk + n*sn]; k >= K or n >= N -> 0.
static __global__ void k_bimg_build(const float* __restrict__ src, long long sk, long long sn, int K, int N, int npad,
                             int kc_layout, uint4* __restrict__ img) {
    const int nkt = (K + 31) / 32;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nkt * 4 * npad) return;
    const int n = (int)(i % npad);
    const int kg = (int)((i / npad) & 3), kt = (int)(i / npad / 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * kt + 8 * kg + j;
        v[j] = (k < K && n < N) ? src[(long long)k * sk + (long long)n * sn] : 0.f;
    }
    uint4 h, l;
    split8(v, h, l);
    if (kc_layout) {            // consumer threads are (n, kg) with kg fastest: [kt][n][kg][hi|lo], 32 B per lane
        img[(((long long)kt * npad + n) * 4 + kg) * 2 + 0] = h;
        img[(((long long)kt * npad + n) * 4 + kg) * 2 + 1] = l;
    } else {                    // consumer lanes run along n: [kt][kg][hi|lo][n]
        img[((long long)(kt * 4 + kg) * 2 + 0) * npad + n] = h;
        img[((long long)(kt * 4 + kg) * 2 + 1) * npad + n] = l;
    }
}

// ---- operand tile: E rows x 32 k ; thread owns NP (row, k-group-of-8) pairs -------------------
// Fast path (host-verified, see fast_ok): no divisions, no bounds branches inside the K loop.
//   rows are clamped to the last valid row (their results are never stored), per-thread pointers
//   advance by 32 k per tile with an incremental wrap for two-level k maps.
//   KC: two float4 units per pair (each 16-B aligned, never straddling `inner`);
//   RC: 8 dwords k..k+7 of one row (coalesced across lanes), never straddling `inner`.
// Slow path: fully general per-element address map + bounds (K-tail tile, odd strides).
template <int E, bool KC>
struct TileX {
    static constexpr int NP = (E * 4 + 255) / 256;
    static constexpr int NU = KC ? 2 : 1;       // independently tracked units per pair
    static constexpr int SZ = E * 5;            // uint4 per image (80-byte rows)
    const float* ptr[NP][NU];
    int rem[NP][NU];
    long long roff[NP];
    int row[NP], kg[NP];
    struct Regs { float v[NP][8]; };     // one pipeline stage of raw fp32 operand data

    __device__ __forceinline__ void init(const float* __restrict__ base, int tid, int row0, int nrows,
                                         const Dim2& d, int kbeg, int kinner, long long khi, long long klo) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int pi = tid + 256 * q;
            int r, g;
            if (KC) { r = pi >> 2; g = pi & 3; } else { r = pi % E; g = pi / E; }
            if (r >= E) { r = E - 1; }
            if (g > 3) g = 3;
            row[q] = r; kg[q] = g;
            int gr = row0 + r;
            const bool valid = gr < nrows;
            if (!valid) gr = nrows - 1;                       // clamp: finite data, result discarded
            const long long ro = d2off(d.inner, d.hi, d.lo, gr);
            roff[q] = valid ? ro : -1;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int k = kbeg + 8 * g + 4 * u;
                int qd = 0, rm = k;
                if (kinner > 0) { qd = k / kinner; rm = k - qd * kinner; }
                ptr[q][u] = base + ro + (long long)qd * khi + (long long)rm * klo;
                rem[q][u] = rm;
            }
        }
    }
    __device__ __forceinline__ void advance(int kinner, long long khi, long long klo) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                ptr[q][u] += 32 * klo;
                if (kinner > 0) {
                    rem[q][u] += 32;
                    while (rem[q][u] >= kinner) { rem[q][u] -= kinner; ptr[q][u] += khi - (long long)kinner * klo; }
                }
            }
    }
    // full tile, fast path: the pointers already address this tile
    __device__ __forceinline__ void load_fast(Regs& R, long long klo) {
        float (&v)[NP][8] = R.v;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (KC) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 t = *reinterpret_cast<const float4*>(ptr[q][u]);
                    v[q][4 * u] = t.x; v[q][4 * u + 1] = t.y; v[q][4 * u + 2] = t.z; v[q][4 * u + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[q][j] = ptr[q][0][(long long)j * klo];
            }
        }
    }
    // any tile, general path (one division per group + bounds); independent of the incremental pointers
    __device__ __forceinline__ void load_slow(Regs& R, const float* __restrict__ base, int k0, int kend, int kinner,
                                              long long khi, long long klo) {
        float (&v)[NP][8] = R.v;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int k = k0 + 8 * kg[q];
            const bool rv = roff[q] >= 0;
            const float* __restrict__ rp = base + (rv ? roff[q] : 0);
            int kc = k < kend ? k : kend - 1;                 // clamp, then mask: no OOB address is formed
            int qd = 0, rm = kc;
            if (kinner > 0) { qd = kc / kinner; rm = kc - qd * kinner; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = rp[(long long)qd * khi + (long long)rm * klo];
                v[q][j] = (rv && k + j < kend) ? x : 0.f;
                if (k + j + 1 < kend) {                       // advance to the next valid k only
                    ++rm;
                    if (kinner > 0 && rm == kinner) { rm = 0; ++qd; }
                }
            }
        }
    }
    // pre-split image source: the registers carry the hi (v[0..3]) and lo (v[4..7]) vectors as raw bits
    __device__ __forceinline__ void load_img(Regs& R, const uint4* __restrict__ img, int kt, int row0, int npad,
                                             int tid) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            int n = row0 + row[q];
            if (n >= npad) n = npad - 1;                       // padded columns hold zeros / discarded outputs
            uint4 h, l;
            if (KC) {
                const uint4* b = img + (((long long)kt * npad + n) * 4 + kg[q]) * 2;
                h = b[0]; l = b[1];
            } else {
                const uint4* b = img + ((long long)(kt * 4 + kg[q]) * 2) * npad + n;
                h = b[0]; l = b[npad];
            }
            R.v[q][0] = __uint_as_float(h.x); R.v[q][1] = __uint_as_float(h.y);
            R.v[q][2] = __uint_as_float(h.z); R.v[q][3] = __uint_as_float(h.w);
            R.v[q][4] = __uint_as_float(l.x); R.v[q][5] = __uint_as_float(l.y);
            R.v[q][6] = __uint_as_float(l.z); R.v[q][7] = __uint_as_float(l.w);
        }
    }
    __device__ __forceinline__ void store_img(const Regs& R, uint4* sHi, uint4* sLo, int tid) const {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (tid + 256 * q < E * 4) {
                sHi[row[q] * 5 + kg[q]] = make_uint4(__float_as_uint(R.v[q][0]), __float_as_uint(R.v[q][1]),
                                                     __float_as_uint(R.v[q][2]), __float_as_uint(R.v[q][3]));
                sLo[row[q] * 5 + kg[q]] = make_uint4(__float_as_uint(R.v[q][4]), __float_as_uint(R.v[q][5]),
                                                     __float_as_uint(R.v[q][6]), __float_as_uint(R.v[q][7]));
            }
        }
    }
    __device__ __forceinline__ void store(const Regs& R, uint4* sHi, uint4* sLo, int tid) const {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (tid + 256 * q < E * 4) {
                uint4 h, l;
                split8(R.v[q], h, l);
                sHi[row[q] * 5 + kg[q]] = h;
                sLo[row[q] * 5 + kg[q]] = l;
            }
        }
    }
};

template <int BM, int BN, int WGM, int WGN, bool AKC, bool BKC, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const GemmP p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 32, FN = WN / 32;
    using TA = TileX<BM, AKC>;
    using TB = TileX<BN, BKC>;
    constexpr int STAGE = 2 * TA::SZ + 2 * TB::SZ;         // uint4 per LDS stage (A hi, A lo, B hi, B lo)
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // 2 stages

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int tile_m, tile_n;
    if (p.vec & (1 << 12)) {                     // debug bit 16: plain column-major tile order
        tile_m = blockIdx.x % tiles_m; tile_n = blockIdx.x / tiles_m;
    } else {
        // XCD-aware order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so give XCD x
        // the contiguous logical range [x*per, (x+1)*per) and walk that range in GM-row groups: the blocks
        // resident on one XCD then cover a compact patch of C and share A row-panels / B column-panels in L2.
        const int nblk = gridDim.x, per = nblk >> 3, full = per << 3;
        const int bid = blockIdx.x;
        const int L = bid < full ? (bid & 7) * per + (bid >> 3) : bid;
        constexpr int GM = 4;
        const int tiles_n = (p.N + BN - 1) / BN;
        const int width = GM * tiles_n;
        const int grp = L / width, first_m = grp * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int r = L - grp * width;
        tile_m = first_m + r % gsz; tile_n = r / gsz;
    }
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int z = blockIdx.z;
    const int batch = z / p.nsplit, split = z - batch * p.nsplit;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    if (kbeg >= kend) return;

    const float* __restrict__ Ab = p.A[batch];
    const float* __restrict__ Bb = p.B[batch];
    const long long akhi = p.ak_hi[batch], bkhi = p.bk_hi[batch];
    const bool afast = (p.vec & 1) != 0, bfast = (p.vec & 2) != 0;
    const uint4* __restrict__ bimg = p.Bimg;          // non-null: B comes pre-split (static weights)
    const uint4* __restrict__ aimg = p.Aimg[batch];   // non-null: A comes pre-split (static adjacency)
#ifdef MCRN_ABLATE
    const int dbg = p.vec >> 8;      // ablation knob (tools/ablate.py, -DMCRN_ABLATE builds only): 1 = no MFMA, 2 = no global loads, 4 = no convert/LDS store, 8 = no barrier
#else
    constexpr int dbg = 0;
#endif

#ifdef MCRN_TIMELINE
    unsigned long long tg_[6] = {0, 0, 0, 0, 0, 0}, tgt_ = wall_clock64();
    const unsigned long long tg0_ = tgt_;
#define MCRN_TG(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); tg_[i] += n_ - tgt_; tgt_ = n_; } } while (0)
#else
#define MCRN_TG(i)
#endif
    TA ta;
    TB tb;
    ta.init(Ab, tid, m_blk, p.M, p.am, kbeg, p.ak.inner, akhi, p.ak.lo);
    tb.init(Bb, tid, n_blk, p.N, p.bn, kbeg, p.bk.inner, bkhi, p.bk.lo);

    // small tiles (<= 2 fragments per wave) keep the two low-order products in their own accumulators;
    // larger ones share `acc` (the product-outer loop order already spaces dependent MFMAs FM*FN apart)
    constexpr bool SPLIT_ACC = FM * FN <= 2;
    f32x16 acc[FM][FN], accA_[SPLIT_ACC ? FM : 1][SPLIT_ACC ? FN : 1], accB_[SPLIT_ACC ? FM : 1][SPLIT_ACC ? FN : 1];
    f32x16 (&accA)[FM][FN] = *reinterpret_cast<f32x16 (*)[FM][FN]>(SPLIT_ACC ? &accA_ : (void*)&acc);
    f32x16 (&accB)[FM][FN] = *reinterpret_cast<f32x16 (*)[FM][FN]>(SPLIT_ACC ? &accB_ : (void*)&acc);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                acc[i][j][v] = 0.f;
                if (SPLIT_ACC) { accA[i][j][v] = 0.f; accB[i][j][v] = 0.f; }
            }

    // Software pipeline: tile t is computed from LDS stage t&1 while tile t+1 sits in registers (converted
    // and stored to the other LDS stage at the end of the iteration) and tile t+2 is in flight from
    // L2/HBM - every global load has two iterations to land.  One barrier per K-tile.
    const int kfull = kbeg + ((kend - kbeg) & ~31);        // tiles below kfull have 32 valid k
    const int nt = (kend - kbeg + 31) >> 5;
    int kload = kbeg;                                       // first k of the next tile to request
#define MCRN_LOAD_TILE(RA, RB)                                                                      \
    do {                                                                                            \
        const bool full_ = kload < kfull;                                                           \
        if (aimg) ta.load_img(RA, aimg, kload >> 5, m_blk, p.aimg_n, tid);                          \
        else if (afast && full_) ta.load_fast(RA, p.ak.lo);                                         \
        else ta.load_slow(RA, Ab, kload, kend, p.ak.inner, akhi, p.ak.lo);                          \
        if (bimg) tb.load_img(RB, bimg, kload >> 5, n_blk, p.bimg_n, tid);                          \
        else if (bfast && full_) tb.load_fast(RB, p.bk.lo);                                         \
        else tb.load_slow(RB, Bb, kload, kend, p.bk.inner, bkhi, p.bk.lo);                          \
        kload += 32;                                                                                \
        if (afast && !aimg) ta.advance(p.ak.inner, akhi, p.ak.lo);                                  \
        if (bfast && !bimg) tb.advance(p.bk.inner, bkhi, p.bk.lo);                                  \
    } while (0)
#define MCRN_STORE_TILE(RA, RB, STG)                                                                \
    do {                                                                                            \
        uint4* b_ = smem + (STG) * STAGE;                                                           \
        if (aimg) ta.store_img(RA, b_, b_ + TA::SZ, tid); else ta.store(RA, b_, b_ + TA::SZ, tid);  \
        if (bimg) tb.store_img(RB, b_ + 2 * TA::SZ, b_ + 2 * TA::SZ + TB::SZ, tid);                 \
        else tb.store(RB, b_ + 2 * TA::SZ, b_ + 2 * TA::SZ + TB::SZ, tid);                          \
    } while (0)
    const int l31 = lane & 31, kq = lane >> 5;
#define MCRN_COMPUTE(STG)                                                                           \
    do {                                                                                            \
        const uint4* sAh = smem + (STG) * STAGE;                                                    \
        const uint4* sAl = sAh + TA::SZ;                                                            \
        const uint4* sBh = sAh + 2 * TA::SZ;                                                        \
        const uint4* sBl = sBh + TB::SZ;                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                         \
            bf16x8 ah[FM], al[FM], bh[FN], bl[FN];                                                  \
            _Pragma("unroll") for (int i = 0; i < FM; ++i) {                                       \
                const int o = (wm * WM + i * 32 + l31) * 5 + ks * 2 + kq;                           \
                ah[i] = __builtin_bit_cast(bf16x8, sAh[o]);                                         \
                al[i] = __builtin_bit_cast(bf16x8, sAl[o]);                                         \
            }                                                                                       \
            _Pragma("unroll") for (int j = 0; j < FN; ++j) {                                       \
                const int o = (wn * WN + j * 32 + l31) * 5 + ks * 2 + kq;                           \
                bh[j] = __builtin_bit_cast(bf16x8, sBh[o]);                                         \
                bl[j] = __builtin_bit_cast(bf16x8, sBl[o]);                                         \
            }                                                                                       \
            /* three independent accumulator chains (one per split product) so no MFMA waits on the previous one */ \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < FN; ++j)                                     \
                    accA[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], accA[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < FN; ++j)                                     \
                    accB[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], accB[i][j], 0, 0, 0); \
            _Pragma("unroll") for (int i = 0; i < FM; ++i)                                         \
                _Pragma("unroll") for (int j = 0; j < FN; ++j)                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0); \
        }                                                                                           \
    } while (0)

    // register stages: NS tiles are in flight between "requested" and "stored to LDS".  Bytes in flight per CU
    // set the achievable fill bandwidth (Little's law: ~2 us loaded latency), so the small tiles, which have
    // registers to spare, run 4 stages deep; the big ones 2.
    constexpr int NS = (BM + BN <= 128) ? 4 : 2;
    typename TA::Regs ra[NS];
    typename TB::Regs rb[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
        if (i < nt) MCRN_LOAD_TILE(ra[i], rb[i]);            // tiles 0 .. NS-1
    MCRN_STORE_TILE(ra[0], rb[0], 0);
    __syncthreads();
    MCRN_TG(0);
    for (int t0 = 0; t0 < nt; t0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {                       // static register-set indices: set = tile % NS
            const int t = t0 + u;
            if (t >= nt) break;
            if (t + NS < nt && !(dbg & 2)) MCRN_LOAD_TILE(ra[u], rb[u]);   // set u held tile t (already in LDS) -> tile t+NS
            MCRN_TG(1);
            // convert + store tile t+1 into the other LDS stage; independent of the MFMA block below, emitted
            // first so the scheduler can interleave its VALU/DS work with the matrix instructions
            if (t + 1 < nt && !(dbg & 4)) {
                if (u & 1) MCRN_STORE_TILE(ra[(u + 1) % NS], rb[(u + 1) % NS], 0);
                else MCRN_STORE_TILE(ra[(u + 1) % NS], rb[(u + 1) % NS], 1);
            }
            MCRN_TG(2);
            if (!(dbg & 1)) { if (u & 1) MCRN_COMPUTE(1); else MCRN_COMPUTE(0); }   // NS is even: LDS stage = t & 1 = u & 1
            MCRN_TG(3);
            if (!(dbg & 8)) __syncthreads();
            MCRN_TG(4);
        }
    }
#undef MCRN_LOAD_TILE
#undef MCRN_STORE_TILE
#undef MCRN_COMPUTE
    if (SPLIT_ACC) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] += accA[i][j][v] + accB[i][j][v];
    }
    gemm_epilogue<FM, FN>(p, acc, batch, split, m_blk + wm * WM + 4 * kq, n_blk + wn * WN + l31);
#ifdef MCRN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MCRN_TG(5);
    if (tid == 0) {
        unsigned long long* o = g_tl[3 + ROLE][(blockIdx.z * gridDim.x + blockIdx.x) & 511];
        for (int i = 0; i < 6; ++i) o[i] = tg_[i];
        o[6] = (unsigned long long)nt; o[7] = (unsigned long long)(BM * 1000 + BN); o[8] = tg0_; o[9] = tgt_;
        o[10] = (unsigned long long)gridDim.x * gridDim.z;
    }
#endif
#undef MCRN_TG
}

// ---- host launcher ------------------------------------------------------------------------
#ifndef MCRN_PROBE   // compile-only probes (tools/probe_prop.hip) skip the launchers: they instantiate every tile configuration
template <int BM, int BN, int WGM, int WGN, bool AKC, bool BKC, int ROLE>
static inline hipError_t launch_one_x3(const GemmP& p, hipStream_t st) {
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, p.nbatch * p.nsplit);
    constexpr size_t lds = 2 * (2 * TileX<BM, AKC>::SZ + 2 * TileX<BN, BKC>::SZ) * sizeof(uint4);   // 2 stages
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<BM, BN, WGM, WGN, AKC, BKC, ROLE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, WGM, WGN, AKC, BKC, ROLE>), grid, dim3(256), lds, st, p);
    return hipGetLastError();
}
template <bool AKC, bool BKC, int ROLE>
static inline hipError_t launch_cfg_x3(const GemmP& p, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_one_x3<128, 128, 2, 2, AKC, BKC, ROLE>(p, st);
        case 1: return launch_one_x3<64, 128, 2, 2, AKC, BKC, ROLE>(p, st);
        case 2: return launch_one_x3<128, 64, 2, 2, AKC, BKC, ROLE>(p, st);
        case 3: return launch_one_x3<64, 64, 2, 2, AKC, BKC, ROLE>(p, st);
        case 4: return launch_one_x3<32, 128, 1, 4, AKC, BKC, ROLE>(p, st);
        case 5: return launch_one_x3<256, 64, 4, 1, AKC, BKC, ROLE>(p, st);
        default: return launch_one_x3<64, 256, 1, 4, AKC, BKC, ROLE>(p, st);
    }
}
static inline hipError_t launch_role_x3(const GemmP& p, bool akc, bool bkc, int role, int cfg, hipStream_t st) {
    // each hot role uses exactly one operand-contiguity combination; anything else goes to MISC
    if (role == ROLE_PROP && akc && !bkc) return launch_cfg_x3<true, false, ROLE_PROP>(p, cfg, st);
    if (role == ROLE_WP && akc && !bkc) return launch_cfg_x3<true, false, ROLE_WP>(p, cfg, st);
    if (role == ROLE_DGRAD && akc && bkc) return launch_cfg_x3<true, true, ROLE_DGRAD>(p, cfg, st);
    if (role == ROLE_PROPT && akc && !bkc) return launch_cfg_x3<true, false, ROLE_PROPT>(p, cfg, st);
    if (role == ROLE_DS && akc && bkc) return launch_cfg_x3<true, true, ROLE_DS>(p, cfg, st);
    if (role == ROLE_WGRAD && !akc && !bkc) return launch_cfg_x3<false, false, ROLE_WGRAD>(p, cfg, st);
    if (akc && !bkc) return launch_cfg_x3<true, false, ROLE_MISC>(p, cfg, st);
    if (akc && bkc) return launch_cfg_x3<true, true, ROLE_MISC>(p, cfg, st);
    if (!akc && !bkc) return launch_cfg_x3<false, false, ROLE_MISC>(p, cfg, st);
    return launch_cfg_x3<false, true, ROLE_MISC>(p, cfg, st);
}

// Fast-path eligibility of an operand (see TileX).  K-contiguous: unit k stride, every aligned
// 4-group of k contiguous and 16-byte aligned for every row of every batch.  Row-contiguous: an
// aligned 8-group of k never straddles the inner extent of a two-level k map.
static inline bool fast_ok(bool kc, const float* const* bases, int nbatch, const Dim2& rows, const Dim2& k,
                           const long long* khi) {
    if (!kc) return k.inner <= 0 || (k.inner % 8) == 0;
    if (k.lo != 1) return false;
    for (int b = 0; b < nbatch; ++b) {
        if (((uintptr_t)bases[b]) & 15) return false;
        if (k.inner > 0 && (khi[b] & 3)) return false;
    }
    if (k.inner > 0 && (k.inner & 3)) return false;
    if (rows.lo & 3) return false;
    if (rows.inner > 0 && (rows.hi & 3)) return false;
    return true;
}

static inline hipError_t launch_gemm_x3(GemmP p, bool akc, bool bkc, int max_split, int role, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipSuccess;
    const int cfg = choose_cfg(p, max_split, 32);
    p.vec = g_debug << 8;
    if (fast_ok(akc, p.A, p.nbatch, p.am, p.ak, p.ak_hi)) p.vec |= 1;
    if (fast_ok(bkc, p.B, p.nbatch, p.bn, p.bk, p.bk_hi)) p.vec |= 2;
    return launch_role_x3(p, akc, bkc, role, cfg, st);
}

#endif  // MCRN_PROBE

}  // namespace mcrn
