// gemm_bf16.h - plain-bf16 GEMM for gfx950 (MI355X): bf16-RESIDENT operands, one v_mfma_f32_32x32x16_bf16 per
// product, fp32 accumulate.  This is the MCRN_BF16 arithmetic of the large-graph path (N >= 512): the K-hop
// propagation  [S1; T2(S1); S2; T2(S2)] x X  (model/MegaCRN.py:20-25), its transpose in the backward pass and the
// adjacency gradient  dP x X^T.  (bf16x3 - gemm_bf16x3.h - stays the 1e-4 parity arithmetic)
//
//   C[m][n] = alpha * sum_k A[m][k] * B(k, n)  (+ beta * Cin[m][n])
//
// A is always K-contiguous (row m = 64-element runs of k).  B comes in two storage forms:
//   BTR = false ("NT"): B stored [n][k], K-contiguous like A                     (adjacency gradient: both operands
//                       are node-major planes, contracted over their columns)
//   BTR = true  ("NN"): B stored [k][n], n contiguous                            (propagation: B = a plane, k = node)
//                       -> the MFMA B fragment (8 consecutive k per lane) is produced by ds_read_b64_tr_b16, the
//                       LDS transpose read of gfx950, from an image of [4 k][16 n] blocks.
// Operand tiles travel HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, no VGPR round trip):
// the LDS destination of a wave instruction is linear (base + 16*lane), so every layout below is expressed by
// WHICH 16-byte chunk a lane fetches (per-lane source address), never by a scattered destination.
//   * K-contiguous tiles (BM x 64 k, 128-byte rows): chunk (row, c) sits in slot  (row>>1)*16 + ((8*(row&1)+c) ^
//     ((row>>1)&15)) - a 16-slot XOR swizzle over row pairs, conflict-free for the ds_read_b128 lane groups.
//   * [k][n] tiles (64 k x BN n): 128-byte blocks of [4 k][16 n], blocks ordered [k/4][n/16]; the two 16-lane
//     groups of a half wave read two adjacent blocks = one full 256-byte bank row.
//   * rows / columns / k beyond the operand, and tail chunks, are fetched from a 16-byte ZERO PAGE instead of being
//     predicated: every lane always issues its load, no divergent control flow in the K loop.
// K is a list of equal segments (k-tile kt -> segment kt / tps): one segment = one Chebyshev block of the stacked
// transposed adjacency (backward propagation) or one time step (deferred adjacency gradient).
// Pipeline: 2 LDS stages; the loads of tile t+1 are issued before the MFMA block of tile t, one barrier per tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16_api.h"

namespace mcrn {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] = *gsrc.  Issued from inline asm ON PURPOSE: hipcc treats the
// builtin form as a pending LDS write that may alias every later ds_read and drains it (s_waitcnt vmcnt(0)) in
// front of the MFMA block it was meant to overlap.  The asm form is invisible to that bookkeeping; the K loop
// waits for it itself (one s_waitcnt vmcnt(0) in front of the barrier that publishes the tile).  M0 carries the
// wave-uniform LDS address and is compiler-reserved: saved and restored inside the same statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// One workgroup = WGM x WGN waves, wave tile (BM/WGM) x (BN/WGN) built from 32x32 fragments, BK = 64.
template <int BM, int BN, int WGM, int WGN, bool BTR>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(const Bf16GemmP p) {
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 32, FN = WN / 32;
    constexpr int ASLOTS = BM * 8, BSLOTS = BN * 8;            // 16-byte chunks per tile
    constexpr int AJ = ASLOTS / NT, BJ = BSLOTS / NT;          // chunks per thread
    static_assert(ASLOTS % NT == 0 && BSLOTS % NT == 0, "tile / workgroup mismatch");
    constexpr int STAGE = (ASLOTS + BSLOTS) * 16;              // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf16[];   // 2 stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_bf16);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, kq = lane >> 5;

    // ---- tile of this workgroup
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    {
        const int nblk = tiles_m * tiles_n;
        int L = blockIdx.x;
        if (p.xcd) {   // workgroups are dealt round-robin to the 8 XCDs: give XCD x a contiguous range of tiles (bijective)
            const int q = nblk >> 3, r = nblk & 7, x = L & 7, i = L >> 3;
            L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        }
        constexpr int GM = 4;                                   // walk GM row tiles per column tile: compact C patches
        const int width = GM * tiles_n;
        const int grp = L / width, first_m = grp * GM;
        const int gsz = min(tiles_m - first_m, GM);
        const int rr = L - grp * width;
        tile_m = first_m + rr % gsz; tile_n = rr / gsz;
    }
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int split = blockIdx.z;
    const int nkt = p.nseg * p.tps;
    const int kt_beg = split * p.tiles_per_split;
    const int kt_end = min(nkt, kt_beg + p.tiles_per_split);
    if (kt_beg >= kt_end) return;

    // ---- per-thread source description of its chunks (constant over the K loop)
    long long offA[AJ], offB[BJ];
    int cA[AJ], kB[BJ];                                         // A / NT-B: chunk index c (k = 8c) ; NN-B: k row inside the tile
    bool okA[AJ], okB[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int s = j * NT + tid;
        const int R = s >> 4, sw = (s & 15) ^ (R & 15);
        const int row = 2 * R + (sw >> 3), c = sw & 7;
        const int gr = m_blk + row;
        okA[j] = gr < p.M;
        cA[j] = c;
        offA[j] = rm_off(p.am, okA[j] ? gr : 0) + 8 * c;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int s = j * NT + tid;
        if (BTR) {
            const int blk = s >> 3, kr = (s >> 1) & 3, half = s & 1;
            const int kb = blk / (BN / 16), nb = blk - kb * (BN / 16);
            const int k = 4 * kb + kr, n = n_blk + 16 * nb + 8 * half;
            okB[j] = n < p.N;
            kB[j] = k;
            offB[j] = (long long)k * p.ldb + (okB[j] ? n : 0);
        } else {
            const int R = s >> 4, sw = (s & 15) ^ (R & 15);
            const int row = 2 * R + (sw >> 3), c = sw & 7;
            const int gn = n_blk + row;
            okB[j] = gn < p.N;
            kB[j] = c;
            offB[j] = rm_off(p.bm, okB[j] ? gn : 0) + 8 * c;
        }
    }

    auto stage = [&](int kt, int stg) {
        const int seg = kt / p.tps, lt = kt - seg * p.tps;
        const int kl0 = lt * 64;
        const int krem = p.seg_len - kl0;                        // valid k in this tile (>= 1; < 64 only in a tail tile)
        const uint16_t* __restrict__ Ab = p.A + (long long)seg * p.a_seg + kl0;
        const uint16_t* __restrict__ Bb = BTR ? p.B + (long long)seg * p.b_seg + (long long)kl0 * p.ldb
                                              : p.B + (long long)seg * p.b_seg + kl0;
        const unsigned sA = lds_base + stg * STAGE + wave * 1024;   // this wave's 64 slots of pass j = 0
        const unsigned sB = sA + ASLOTS * 16;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const bool ok = okA[j] && 8 * cA[j] < krem;
            const uint16_t* src = ok ? Ab + offA[j] : p.zero;
            glds16(src, sA + j * NT * 16);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const bool ok = okB[j] && (BTR ? kB[j] : 8 * kB[j]) < krem;
            const uint16_t* src = ok ? Bb + offB[j] : p.zero;
            glds16(src, sB + j * NT * 16);
        }
    };

    // ---- fragment read addresses (bytes inside a stage's A / B image)
    int a0, b0;
    {
        const int row = wm * WM + l31;                           // fragment i adds 32 rows = 16 row pairs: same XOR key
        const int R = row >> 1, q = R & 15;
        a0 = R * 256 + (((8 * (row & 1) + kq) ^ q) << 4);
        if (BTR) {
            const int g = lane >> 4, i = lane & 15;
            const int kq2 = g >> 1, nhalf = g & 1;
            b0 = ((2 * kq2) * (BN / 16) + (wn * WN) / 16 + nhalf) * 128 + (i >> 2) * 32 + (i & 3) * 8;
        } else {
            const int rowb = wn * WN + l31;
            const int Rb = rowb >> 1, qb = Rb & 15;
            b0 = Rb * 256 + (((8 * (rowb & 1) + kq) ^ qb) << 4);
        }
    }

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    stage(kt_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's part of the tile has landed ...
    __syncthreads();                                             // ... and so has everybody else's
    int cur = 0;
    for (int kt = kt_beg; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) stage(kt + 1, cur ^ 1);             // in flight during the MFMA block below
        const unsigned char* sA = smem_bf16 + cur * STAGE;
        const unsigned char* sB = sA + ASLOTS * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a[i] = *reinterpret_cast<const bf16x8_t*>(sA + ((a0 ^ (ks << 5)) + i * 4096));
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (BTR) {
                    const unsigned char* q = sB + b0 + (4 * ks) * (BN / 16) * 128 + 2 * j * 128;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(q));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(q + (BN / 16) * 128));
                    typedef short s16x8_t __attribute__((ext_vector_type(8)));
                    const s16x8_t w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[j] = __builtin_bit_cast(bf16x8_t, w);
                } else {
                    b[j] = *reinterpret_cast<const bf16x8_t*>(sB + ((b0 ^ (ks << 5)) + j * 4096));
                }
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // tile kt+1: this wave's LDS-DMA has landed
        __syncthreads();                                         // tile kt consumed by every wave, tile kt+1 visible to all
        cur ^= 1;
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
    float* __restrict__ C = p.C ? p.C + (long long)split * p.slab : nullptr;
    const float* __restrict__ Cin = p.Cin ? p.Cin + (long long)split * p.slab : nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int r0 = m_blk + wm * WM + i * 32 + 4 * kq;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = r0 + (v & 3) + 8 * (v >> 2);
            if (r >= p.M) continue;
            const long long ro = rm_off(p.cm, r);
            const long long rob = p.Cb ? rm_off(p.cbm, r) : 0;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int c = n_blk + wn * WN + j * 32 + l31;
                if (c >= p.N) continue;
                float o = p.alpha * acc[i][j][v];
                if (Cin) o += p.beta * Cin[ro + c];
                if (C) C[ro + c] = o;
                if (p.Cb) {
                    unsigned u = __float_as_uint(o);
                    u += 0x7FFFu + ((u >> 16) & 1u);             // round to nearest even (finite values)
                    p.Cb[rob + c] = (uint16_t)(u >> 16);
                }
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------

template <int BM, int BN, int WGM, int WGN, bool BTR>
static inline hipError_t launch_one_bf16(const Bf16GemmP& p, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr size_t lds = 2 * (BM * 8 + BN * 8) * 16;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, WGM, WGN, BTR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, BTR>), dim3(tiles, 1, p.nsplit), dim3(64 * WGM * WGN), lds, st, p);
    return hipGetLastError();
}
template <bool BTR>
static inline hipError_t launch_cfg_bf16(const Bf16GemmP& p, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_one_bf16<128, 128, 2, 2, BTR>(p, st);
        case 1: return launch_one_bf16<256, 128, 4, 2, BTR>(p, st);
        case 2: return launch_one_bf16<128, 256, 2, 4, BTR>(p, st);
        default: return launch_one_bf16<256, 256, 2, 4, BTR>(p, st);
    }
}
// fills the derived fields (tps, split ranges) and launches
hipError_t launch_gemm_bf16(Bf16GemmP p, bool btr, int cfg, int nsplit, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.nseg <= 0 || p.seg_len <= 0) return hipSuccess;
    // NT: whole 16-byte chunks of k.  NN: whole chunks of n; A rows must be readable (and zero) up to the next multiple
    // of 8 beyond seg_len (the stacked adjacency is stored with rows padded to a multiple of 64).
    if ((!btr && (p.seg_len & 7)) || (btr && (p.N & 7)) || !p.zero) return hipErrorInvalidValue;
    p.tps = (p.seg_len + 63) / 64;
    const int nkt = p.nseg * p.tps;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nkt) nsplit = nkt;
    p.tiles_per_split = (nkt + nsplit - 1) / nsplit;
    p.nsplit = (nkt + p.tiles_per_split - 1) / p.tiles_per_split;
    return btr ? launch_cfg_bf16<true>(p, cfg, st) : launch_cfg_bf16<false>(p, cfg, st);
}


}  // namespace mcrn
