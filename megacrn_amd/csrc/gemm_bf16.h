// gemm_bf16.h - plain-bf16 GEMM for gfx950 (MI355X): bf16-RESIDENT operands, one v_mfma_f32_32x32x16_bf16 per
// product, fp32 accumulate.  This is the MCRN_BF16 arithmetic of the large-graph path: the K-hop propagation
// [S1; T2(S1); S2; T2(S2)] x X  (model/MegaCRN.py:20-25), its transpose in the backward pass and the adjacency
// gradient  dP x X^T.  (bf16x3 - gemm_bf16x3.h - stays the 1e-4 parity arithmetic)
//
//   C[m][n] = alpha * sum_k A[m][k] * B(k, n)  (+ beta * Cin[m][n])
//
// A is always K-contiguous (row m = runs of k).  B comes in two storage forms:
//   BTR = false ("NT"): B stored [n][k], K-contiguous like A     (adjacency gradient: both operands are node-major
//                       planes, contracted over their columns)
//   BTR = true  ("NN"): B stored [k][n], n contiguous            (propagation: B = a plane, k = node)
//                       -> the MFMA B fragment (8 consecutive k per lane) is produced by ds_read_b64_tr_b16, the LDS
//                       transpose read of gfx950, from an image of [4 k][16 n] blocks.
// OPERAND CONTRACT (what lets the K loop run without a single predicate or per-lane address update):
//   * K-contiguous operands are readable and ZERO from seg_len up to the next multiple of 64 in every segment
//     (the stacked adjacency is built that way; bf16 planes carry zero pad columns);
//   * [k][n] operands are readable and FINITE for k up to the next multiple of 64 of seg_len (zero pad rows);
//   * rows m >= M / columns n >= N are CLAMPED to the last valid one when fetched (their products are never stored).
// Operand tiles travel L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, no VGPR round trip), in the
// SGPR-base + 32-bit lane-offset form: the lane offsets are constants of the thread, a tile costs a handful of SALU
// instructions.  The LDS destination of a wave instruction is linear (base + 16*lane), so every layout is expressed by
// WHICH 16-byte chunk a lane fetches:
//   * K-contiguous tiles (rows of CH = BK/8 chunks, RP = 16/CH rows per 256-byte bank row): chunk (row, c) sits in
//     slot  Q*16 + ((CH*(row % RP) + c) ^ (Q & 15)),  Q = row / RP   - conflict-free for the ds_read_b128 lane groups
//     (SQ_LDS_BANK_CONFLICT = 0, profiles/r2);
//   * [k][n] tiles: 128-byte blocks of [4 k][16 n], blocks ordered [k/4][n/16]; the two 16-lane groups of a half
//     wave read two adjacent blocks = one full 256-byte bank row.
// K is a list of equal segments (k-tile kt -> segment kt / tps): one segment = one Chebyshev block of the stacked
// transposed adjacency (backward propagation) or one AGCN call (deferred adjacency gradient).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "gemm_bf16_api.h"

namespace mcrn {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] = *(sbase + voff).  Issued from inline asm ON PURPOSE: hipcc
// treats the builtin form as a pending LDS write that may alias every later ds_read and drains it (s_waitcnt vmcnt(0))
// in front of the MFMA block it was meant to overlap.  The asm form is invisible to that bookkeeping; the K loops
// below count it themselves (s_waitcnt vmcnt(N)).  M0 carries the wave-uniform LDS address and is compiler-reserved:
// saved and restored inside the same statement.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
#define MCRN_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// -DMCRN_BF16_ABL=bits (tools/kbench/bf16_gemm_test only; results are wrong by construction): 1 = no operand DMA inside the
// K loop, 2 = fragments read from LDS once, 4 = no MFMA.  What a K loop costs without one of its three streams.
#ifndef MCRN_BF16_ABL
#define MCRN_BF16_ABL 0
#endif
// where the DMA pieces of a K tile go (A/B builds: make EXTRA=-DMCRN_BF16_ILV=n): bit 1 = between the MFMAs in the ring kernel
// (wave tiles of 8+ fragments), bit 2 = between the MFMAs of the computing group in the ping-pong kernel; 0 = behind the MFMAs
#ifndef MCRN_BF16_ILV
#define MCRN_BF16_ILV 3
#endif
// Accumulator preload (round 5: the split-0 workgroups of the transposed propagation start their K loop FROM the addend instead of adding it in
// the epilogue).  OFF since round 6 (A/B build: -DMCRN_BF16_PRELOAD=1 compiles it into the ROLE 4 kernels): its address arithmetic took the ring
// kernel from 84 to 104 SGPRs and cost every instantiation 6 - 7 % per launch, taken or not - more than it saved where it was taken
// (transposed product at N = 1843: 55.1 - 56.1 us with it in the ROLE 4 kernels only, 52.9 - 53.2 us without; profiles/r6/experiments.md section 8).
#ifndef MCRN_BF16_PRELOAD
#define MCRN_BF16_PRELOAD 0
#endif
#ifndef MCRN_BF16_G0C
#define MCRN_BF16_G0C 0
#endif
#ifndef MCRN_BF16_G1L
#define MCRN_BF16_G1L 1
#endif
// production probe (Bf16GemmP::clk, null outside the roofline leg): one thread of workgroup 0, two scalar clock reads and a 16-byte store
#define MCRN_CLK_STAMP(P, i) do { if ((P).clk && blockIdx.x == 0 && threadIdx.x == 0) { (P).clk[2 * (i)] = clock64(); (P).clk[2 * (i) + 1] = wall_clock64(); } } while (0)
#if MCRN_BF16_ABL & 8
// bit 8: workgroup 0 records the shader-clock counter and the 100 MHz wall clock at both ends of the kernel (the clock the
// chip really runs at under this load = d(clock64) / d(wall_clock64) x 100 MHz)
__device__ unsigned long long g_bf16_clk[4];
#define MCRN_CLK_PROBE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_bf16_clk[2 * (i)] = clock64(); g_bf16_clk[2 * (i) + 1] = wall_clock64(); } } while (0)
#else
#define MCRN_CLK_PROBE(i)
#endif

// ---- pieces shared by the two kernels ------------------------------------------------------------------------
// X3 (round 6): hi/lo operand PAIRS (Bf16GemmP::nterm == 3: the lo copies sit a_lo / b_lo elements behind the hi ones).  A K tile is FOUR
// images - A_hi, A_lo, B_hi, B_lo - fetched ONCE each, and the three MFMA blocks
//     A_hi x B_hi + A_hi x B_lo + A_lo x B_hi
// are issued from LDS: the bf16x3 arithmetic of the library (fp32 operands as bf16 hi + lo, fp32 accumulate, ~4e-6) on bf16-RESIDENT
// operands with 4 tile fetches and 0.5 fragment reads per MFMA.  (Round 5 walked every K segment three times with a full A + B fetch per
// term: 6 fetches where 4 distinct tiles exist, 2.78 x the algorithmic bytes through the fabric ports at N = 1843.)
template <int BM, int BN, int BK, int NT, bool BTR, bool X3 = false>
struct Bf16Tile {
    static constexpr int CH = BK / 8, RP = 16 / CH, KS = BK / 16;
    static constexpr int ASLOTS = BM * CH, BSLOTS = BN * CH;            // 16-byte chunks per tile image
    static constexpr int AJ = (ASLOTS + NT - 1) / NT, BJ = (BSLOTS + NT - 1) / NT;
    static constexpr int NOP = X3 ? 2 : 1;                              // images per operand: hi (+ lo)
    static constexpr int NTM = X3 ? 3 : 1;                              // MFMA terms per fragment pair
    static constexpr int NLD_MIN = NOP * (ASLOTS / NT + BSLOTS / NT);   // DMA instructions every wave issues per tile
    static constexpr int A_IMG = ASLOTS * 16, B_IMG = BSLOTS * 16;      // bytes of one image
    static constexpr int B_OFF = NOP * A_IMG;                           // stage layout: [A_hi | A_lo | B_hi | B_lo]
    static constexpr int STAGE = NOP * (A_IMG + B_IMG);                 // bytes
    static_assert(ASLOTS % 64 == 0 && BSLOTS % 64 == 0, "whole waves per pass");
    static_assert(BK == 16 || BK == 32 || BK == 64, "BK");     // (16: one MFMA step per K tile - the hi/lo form of the 256 x 256 tile, four stages in 128 KB)

    unsigned offA[AJ], offB[BJ];          // byte offsets of this thread's chunks from the tile's scalar base
    const uint16_t *baseA, *baseB;        // scalar: first element of the workgroup's rows / columns
    long long stepA_seg, stepB_seg, stepB_k;   // scalar strides (elements)
    int tps;
    int iss_seg, iss_lt;                  // K tile the next DMA will fetch
    long long a_lo, b_lo;                 // X3: element offsets of the lo images

    // (Every workgroup walks its K tiles IN ORDER.  Round 5 tried a per-tile rotated start - workgroups that share an operand panel on one
    //  XCD would then not ask for the same lines at the same time - and measured 20 - 25 % SLOWER on every product, harness and model: the
    //  lock-step is what makes one fetch from the memory side serve every workgroup of the XCD.  profiles/r5/experiments.md section 3.)
    __device__ __forceinline__ void init(const Bf16GemmP& p, int tid, int m_blk, int n_blk, int kt_beg) {
        const long long a0 = rm_off(p.am, m_blk);
        baseA = p.A + a0;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int s = min(j * NT + tid, ASLOTS - 1);        // (a partial last pass is skipped by whole waves)
            const int Q = s >> 4, sw = (s & 15) ^ (Q & 15);
            const int row = Q * RP + sw / CH, c = sw % CH;
            const int gr = min(m_blk + row, p.M - 1);
            offA[j] = (unsigned)((rm_off(p.am, gr) - a0 + 8 * c) * 2);
        }
        if (BTR) {
            baseB = p.B + n_blk;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int s = min(j * NT + tid, BSLOTS - 1);
                const int blk = s >> 3, kr = (s >> 1) & 3, half = s & 1;
                const int kb = blk / (BN / 16), nb = blk - kb * (BN / 16);
                const int k = 4 * kb + kr;
                const int n = min(n_blk + 16 * nb + 8 * half, p.N - 8);
                offB[j] = (unsigned)(((long long)k * p.ldb + (n - n_blk)) * 2);
            }
            stepB_k = p.ldb;
        } else {
            const long long b0 = rm_off(p.bm, n_blk);
            baseB = p.B + b0;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int s = min(j * NT + tid, BSLOTS - 1);
                const int Q = s >> 4, sw = (s & 15) ^ (Q & 15);
                const int row = Q * RP + sw / CH, c = sw % CH;
                const int gn = min(n_blk + row, p.N - 1);
                offB[j] = (unsigned)((rm_off(p.bm, gn) - b0 + 8 * c) * 2);
            }
            stepB_k = 1;
        }
        stepA_seg = p.a_seg; stepB_seg = p.b_seg; tps = p.tps;
        a_lo = p.a_lo; b_lo = p.b_lo;
        iss_seg = kt_beg / p.tps;
        iss_lt = kt_beg - iss_seg * p.tps;
    }
    __device__ __forceinline__ void advance() {
        if (++iss_lt == tps) { iss_lt = 0; ++iss_seg; }
    }
    // DMA of the next K tile into LDS stage `stg`
    __device__ __forceinline__ void issue(unsigned lds_base, int stg, int wave) {
        issue_begin(lds_base, stg, wave);
#pragma unroll
        for (int J = 0; J < NPIECE; ++J) issue_piece(J, wave);
        issue_end();
    }
    // The same DMA in PIECES, for K loops that slip them between the MFMAs of the tile being multiplied: a wave that issues
    // its 6 - 8 pieces back to back waits 60 - 180 cycles per piece for the address path with the matrix pipe idle
    // (profiles/r4/experiments.md section 13: the loop without its DMA ran 12 - 14 % faster); one piece every few MFMAs
    // issues in the shadow of the 32-cycle MFMA in flight.   issue_begin ; issue_piece(0 .. NPIECE-1) ; issue_end
    // piece order: A_hi pieces, (X3: A_lo pieces,) B_hi pieces, (X3: B_lo pieces)
    static constexpr int NPIECE = NOP * (AJ + BJ);
    const uint16_t *pcA, *pcB;
    unsigned pcsA, pcsB;
    __device__ __forceinline__ void issue_begin(unsigned lds_base, int stg, int wave) {
        pcA = baseA + (long long)iss_seg * stepA_seg + iss_lt * BK;
        pcB = baseB + (long long)iss_seg * stepB_seg + (long long)(iss_lt * BK) * stepB_k;
        pcsA = lds_base + stg * STAGE + wave * 1024;
        pcsB = pcsA + B_OFF;
    }
    __device__ __forceinline__ void issue_piece(int J, int wave) {          // J is a constant after unrolling
        if (J < NOP * AJ) {
            const int img = J / AJ, j = J - img * AJ;
            if (ASLOTS % NT == 0 || j * NT + wave * 64 < ASLOTS) glds16(img ? pcA + a_lo : pcA, offA[j], pcsA + img * A_IMG + j * NT * 16);
        } else if (J < NPIECE) {
            const int q = J - NOP * AJ, img = q / BJ, j = q - img * BJ;
            if (BSLOTS % NT == 0 || j * NT + wave * 64 < BSLOTS) glds16(img ? pcB + b_lo : pcB, offB[j], pcsB + img * B_IMG + j * NT * 16);
        }
    }
    __device__ __forceinline__ void issue_end() { advance(); }
};
// which DMA piece (if any) follows MFMA number m of a tile's NM: the NP pieces are spread evenly, the first one early
__device__ __forceinline__ constexpr int bf16_piece_after(int m, int NM, int NP) {
    // piece q follows MFMA number floor(q * NM / NP) (0-based): q = ceil(m * NP / NM) if that maps back to m
    const int q = (m * NP + NM - 1) / NM;
    return (q < NP && q * NM / NP == m) ? q : -1;
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); XCD x gets a contiguous
// range of tiles (bijective for any tile count), walked GM row tiles per column tile: the tiles resident on one XCD
// cover a compact patch of C and share A row panels / B column panels in its L2.
// walk order inside a range of tiles: GM row tiles per column tile
__device__ __forceinline__ void bf16_tile_coords(int L, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    constexpr int GM = 4;
    const int width = GM * tiles_n;
    const int grp = L / width, first_m = grp * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int rr = L - grp * width;
    tile_m = first_m + rr % gsz; tile_n = rr / gsz;
}
// Split-K launches are one-dimensional, nsplit * tiles workgroups: the (split, tile) space, split slowest, is what gets dealt
// to the XCDs in contiguous ranges - an XCD then works on ONE K range of a patch of tiles and streams 1/nsplit of the A row
// panels and of B through its L2.  (With the split in blockIdx.z every XCD walked the whole K of its tiles: 2.5 x the L2 miss
// traffic at N=1843 with 4 splits, and the transposed propagation ran at 57 us in the model against 37 us in the tuner's
// bursts, where the infinity cache hides it.)
__device__ __forceinline__ void bf16_tile_of(const Bf16GemmP& p, int BM, int BN, int& tile_m, int& tile_n, int& split) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nblk = tiles_m * tiles_n, total = nblk * p.nsplit;
    int L = blockIdx.x;
    if (p.xcd) {
        const int q = total >> 3, r = total & 7, x = L & 7, i = L >> 3;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    split = __builtin_amdgcn_readfirstlane(L / nblk);
    bf16_tile_coords(L - split * nblk, tiles_m, tiles_n, tile_m, tile_n);
}

template <int FM, int FN, int BN, int CH, int RP, bool BTR>
__device__ __forceinline__ void bf16_frag_offsets(int row_a0, int row_b0, int lane, int (&aoff)[FM], int (&boff)[FN]) {
    const int l31 = lane & 31, kq = lane >> 5;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int row = row_a0 + i * 32 + l31;
        const int Q = row / RP;
        aoff[i] = Q * 256 + (((CH * (row % RP) + kq) ^ (Q & 15)) << 4);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        if (BTR) {
            const int g = lane >> 4, i = lane & 15;
            const int kq2 = g >> 1, nhalf = g & 1;
            boff[j] = ((2 * kq2) * (BN / 16) + row_b0 / 16 + 2 * j + nhalf) * 128 + (i >> 2) * 32 + (i & 3) * 8;
        } else {
            const int row = row_b0 + j * 32 + l31;
            const int Q = row / RP;
            boff[j] = Q * 256 + (((CH * (row % RP) + kq) ^ (Q & 15)) << 4);
        }
    }
}
template <int BN, bool BTR>
__device__ __forceinline__ bf16x8_t bf16_read_b(const unsigned char* sB, int boff, int ks) {
    if (BTR) {
        const unsigned char* q = sB + boff + (4 * ks) * (BN / 16) * 128;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(q + (BN / 16) * 128));
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        const s16x8_t w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, w);
    }
    return *reinterpret_cast<const bf16x8_t*>(sB + (boff ^ (ks << 5)));
}

// epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5).
// Output rows are plain-strided in every use of this library (stacked planes / stacked N x N blocks are contiguous:
// the host folds such two-level maps into a plain stride), so a fragment is addressed as a SCALAR 64-bit base (its
// first row) + a 32-bit lane offset, and fragments that lie completely inside the matrix - wave-uniform test - store
// unpredicated.  (The first version evaluated a two-level row map with an integer division per element and a bounds
// branch per store: ~6000 instructions, 15-25 % of a K = 1843 launch.)
// bf16-only output (C == null, Cb != null: the bf16-resident planes of the propagation) through LDS: the C/D layout gives a
// lane ONE column of 16 rows, i.e. 2-byte stores; staged in the wave's private LDS region (rows of 128 bytes, the 64-byte
// halves of rows 4..7 (mod 8) swapped so that the two lane halves of a ds_write_b16 hit different banks) every lane
// then stores 16 contiguous bytes: 8 x fewer store instructions, full 128-byte lines.  Only whole-tile fragments, FN == 2.
template <int FM, int FN>
__device__ __forceinline__ bool bf16_epilogue_wide(const Bf16GemmP& p, f32x16_t (&acc)[FM][FN], int r_base, int c_base, int lane,
                                                   unsigned char* wave_lds, uint16_t* __restrict__ out, int ldb) {
    if (FN != 2) return false;
    const int rw = __builtin_amdgcn_readfirstlane(r_base), cw = __builtin_amdgcn_readfirstlane(c_base);
    if (rw + 32 * FM > p.M || cw + 64 > p.N) return false;                 // wave-uniform
    const int l31 = lane & 31, kq = lane >> 5;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 32 * i + 4 * kq + (v & 3) + 8 * (v >> 2);
                unsigned u = __float_as_uint(p.alpha * acc[i][j][v]);
                u += 0x7FFFu + ((u >> 16) & 1u);
                *reinterpret_cast<uint16_t*>(wave_lds + row * 128 + ((64 * j + 2 * l31) ^ (kq << 6))) = (uint16_t)(u >> 16);   // ((row >> 2) & 1) == kq
            }
    uint16_t* __restrict__ dst = out + (long long)rw * ldb + cw;
#pragma unroll
    for (int n = 0; n < 4 * FM; ++n) {
        const int q = lane + 64 * n, row = q >> 3, c = q & 7;
        const uint4 val = *reinterpret_cast<const uint4*>(wave_lds + row * 128 + ((16 * c) ^ (((row >> 2) & 1) << 6)));
        *reinterpret_cast<uint4*>(dst + (long long)row * ldb + 8 * c) = val;
    }
    return true;
}

// Accumulators initialised with the addend (Bf16GemmP::cin_pre, round 5: the split-0 workgroups of the transposed propagation):
//   acc = Cin   (alpha = beta = 1; elements outside the matrix: 0)
// issued at kernel start, in front of the operand DMA of the prologue, so that the loads fly while the first K tiles arrive and the
// epilogue only stores - the read-modify-write epilogue of these workgroups was the tail of the whole launch (profiles/r5/experiments.md
// section 2).  Same addressing as bf16_epilogue; the loads are older than every DMA piece, so the kernels' counted vmcnt waits cover them.
template <int FM, int FN>
__device__ __forceinline__ void bf16_acc_preload(const Bf16GemmP& p, f32x16_t (&acc)[FM][FN], int r_base, int c_base, int lane) {
    const int l31 = lane & 31, kq = lane >> 5;
    const float* __restrict__ Cin = p.Cin;
    const int rw = __builtin_amdgcn_readfirstlane(r_base), cw = __builtin_amdgcn_readfirstlane(c_base);
    const int ld = (int)p.cm.lo;
    const bool cols_in = cw + 32 * FN <= p.N;
    int cof[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int cj = cw + 32 * j;
        cof[j] = p.cn_inner > 0 ? (cj / p.cn_inner) * p.cn_hi + (cj - (cj / p.cn_inner) * p.cn_inner) : cj;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int rb = rw + 32 * i;
        const float* __restrict__ Cif = Cin + (long long)min(rb, p.M - 1) * ld;
        const unsigned lo0 = (unsigned)(4 * kq * ld + l31);
        if (cols_in && rb + 32 <= p.M) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = Cif[lo0 + (unsigned)(((v & 3) + 8 * (v >> 2)) * ld) + cof[j]];
        } else {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dr = (v & 3) + 8 * (v >> 2);
                    const bool ok = rb + 4 * kq + dr < p.M && cw + 32 * j + l31 < p.N;
                    acc[i][j][v] = ok ? Cin[(long long)(rb + 4 * kq + dr) * ld + l31 + cof[j]] : 0.f;
                }
        }
    }
}

template <int FM, int FN>
__device__ __forceinline__ void bf16_epilogue(const Bf16GemmP& p, f32x16_t (&acc)[FM][FN], int split, int r_base, int c_base,
                                              int lane) {
    const int l31 = lane & 31, kq = lane >> 5;
    const long long soff = p.slab2 > 0 ? (split > 0 ? p.slab + (long long)(split - 1) * p.slab2 : 0) : (long long)split * p.slab;
    float* __restrict__ C = p.C ? p.C + soff : nullptr;
    const float* __restrict__ Cin = (p.Cin && !(p.cin_first_only && split > 0) && !p.cin_pre) ? p.Cin + soff : nullptr;   // (cin_pre: already in acc)
    const int rw = __builtin_amdgcn_readfirstlane(r_base), cw = __builtin_amdgcn_readfirstlane(c_base);
    const int ld = (int)p.cm.lo, ldb = (int)p.cbm.lo;
    const bool cols_in = cw + 32 * FN <= p.N;
    // column offset of fragment j (scalar): contiguous columns, or the two-level map of cn_inner / cn_hi - a 32-column
    // fragment never straddles an inner block (cn_inner % 32 == 0, fragments start at multiples of 32)
    int cof[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int cj = cw + 32 * j;
        cof[j] = p.cn_inner > 0 ? (cj / p.cn_inner) * p.cn_hi + (cj - (cj / p.cn_inner) * p.cn_inner) : cj;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int rb = rw + 32 * i;                                  // first row of the fragment (scalar)
        if (rb >= p.M) break;
        const long long base = (long long)rb * ld;
        const unsigned lo0 = (unsigned)(4 * kq * ld + l31);
        float* __restrict__ Cf = C ? C + base : nullptr;
        const float* __restrict__ Cif = Cin ? Cin + base : nullptr;
        uint16_t* __restrict__ Cbf = p.Cb ? p.Cb + (long long)rb * ldb + cw : nullptr;
        const unsigned lb0 = (unsigned)(4 * kq * ldb + l31);
        if (cols_in && rb + 32 <= p.M) {                             // whole fragment inside: straight-line code
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                float x[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) x[v] = p.alpha * acc[i][j][v];
                if (Cif) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) x[v] += p.beta * Cif[lo0 + (unsigned)(((v & 3) + 8 * (v >> 2)) * ld) + cof[j]];
                }
                if (Cf) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) Cf[lo0 + (unsigned)(((v & 3) + 8 * (v >> 2)) * ld) + cof[j]] = x[v];
                }
                if (Cbf) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        unsigned u = __float_as_uint(x[v]);
                        u += 0x7FFFu + ((u >> 16) & 1u);             // round to nearest even (finite values)
                        Cbf[lb0 + (unsigned)(((v & 3) + 8 * (v >> 2)) * ldb) + 32 * j] = (uint16_t)(u >> 16);
                    }
                }
            }
        } else {                                                     // edge fragment: per-element predicate
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dr = (v & 3) + 8 * (v >> 2);
                    if (rb + 4 * kq + dr < p.M && cw + 32 * j + l31 < p.N) {
                        const unsigned o = lo0 + (unsigned)(dr * ld) + cof[j];
                        float x = p.alpha * acc[i][j][v];
                        if (Cif) x += p.beta * Cif[o];
                        if (Cf) Cf[o] = x;
                        if (Cbf) {
                            unsigned u = __float_as_uint(x);
                            u += 0x7FFFu + ((u >> 16) & 1u);
                            Cbf[lb0 + (unsigned)(dr * ldb) + 32 * j] = (uint16_t)(u >> 16);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 1: every wave does the same thing.  NSTAGE LDS stages form a ring; per K tile:
//     wait(tile t landed for this wave) ; barrier (landed for everybody, tile t-1 consumed by everybody) ;
//     multiply tile t, fragment reads and MFMAs interleaved by the compiler ; issue the DMA of tile t+NSTAGE-1 into the
//     stage tile t-1 just left (after the MFMAs in program order: they run while the addresses are formed).
// ---------------------------------------------------------------------------------------------------------
// (ROLE only changes the symbol name, so that rocprofv3 reports the forward propagation, its transpose and the
//  adjacency gradient separately: 1 / 4 / 5 as in gemm_f32.h, 0 = everything else)
template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, bool BTR, int ROLE, bool X3 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(const Bf16GemmP p) {
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 32, FN = WN / 32;
    using T = Bf16Tile<BM, BN, BK, NT, BTR, X3>;
    constexpr int NOP = T::NOP, NTM = T::NTM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf16[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_bf16);
    const int wm = wave / WGN, wn = wave % WGN;
    int tile_m, tile_n;
    int split;
    bf16_tile_of(p, BM, BN, tile_m, tile_n, split);
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int nkt = p.nseg * p.tps;
    const int kt_beg = split * p.tiles_per_split;
    const int kt_end = min(nkt, kt_beg + p.tiles_per_split);
    if (kt_beg >= kt_end) return;
    const int nt = kt_end - kt_beg;

    T tl;
    tl.init(p, tid, m_blk, n_blk, kt_beg);
    int aoff[FM], boff[FN];
    bf16_frag_offsets<FM, FN, BN, T::CH, T::RP, BTR>(wm * WM, wn * WN, lane, aoff, boff);

    f32x16_t acc[FM][FN];
    // (see MCRN_BF16_PRELOAD: round 5 had this branch in EVERY instantiation - the "encoder tile regression" of the round-5 review, bisected in round 6)
    bool preloaded = false;
    if constexpr (ROLE == 4 && MCRN_BF16_PRELOAD != 0) {
        if (p.cin_pre && split == 0) { bf16_acc_preload<FM, FN>(p, acc, m_blk + wm * WM, n_blk + wn * WN, lane); preloaded = true; }   // workgroup-uniform
    }
    if (!preloaded) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    }

    MCRN_CLK_STAMP(p, 0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nt) tl.issue(lds_base, s, wave);
    int rd = 0;
#if MCRN_BF16_ABL & 2
    bf16x8_t a[2][NOP][FM], b[2][NOP][FN];
#endif
    // MFMA term tm of a fragment pair: 0 = hi x hi ; X3: 1 = hi x lo, 2 = lo x hi
    for (int t = 0; t < nt; ++t) {
        // tile t has landed once at most the DMA of the NSTAGE-2 younger tiles remains in flight (steady state);
        // in the tail fewer tiles are in flight: drain
        if (NSTAGE > 2 && t + NSTAGE - 2 < nt) MCRN_VMCNT((NSTAGE - 2) * T::NLD_MIN);
        else MCRN_VMCNT(0);
        __syncthreads();                                         // tile t visible to all; tile t-1 consumed by all
        const unsigned char* sA = smem_bf16 + rd * T::STAGE;
        const unsigned char* sB = sA + T::B_OFF;
        // Wave tiles of 8+ fragments (128 x 64): the DMA pieces go BETWEEN the MFMAs (see Bf16Tile::issue_piece; -5 .. -8 % per
        // launch).  The 64 x 64 wave tile of the 256 x 128 eight-wave form has only 16 MFMAs per tile for 6 pieces and two
        // waves per SIMD already overlap each other's issue: it keeps the DMA behind its MFMAs (interleaved: +1 %).
        constexpr bool ILV = (MCRN_BF16_ILV & 1) && FM * FN >= 8;
        if constexpr (!ILV) {
#pragma unroll
            for (int ks = 0; ks < T::KS; ++ks) {
                bf16x8_t a1[NOP][FM], b1[NOP][FN];
#pragma unroll
                for (int o = 0; o < NOP; ++o) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) a1[o][i] = *reinterpret_cast<const bf16x8_t*>(sA + o * T::A_IMG + (aoff[i] ^ (ks << 5)));
#pragma unroll
                    for (int j = 0; j < FN; ++j) b1[o][j] = bf16_read_b<BN, BTR>(sB + o * T::B_IMG, boff[j], ks);
                }
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            const bf16x8_t& av = a1[tm == 2 ? 1 : 0][i];
                            const bf16x8_t& bv = b1[tm == 1 ? 1 : 0][j];
                            if (MCRN_BF16_ABL & 4) { asm volatile("" ::"v"(av), "v"(bv)); }
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
                        }
            }
            if (!(MCRN_BF16_ABL & 1) && t + NSTAGE - 1 < nt) {
                int wr = rd + NSTAGE - 1;
                if (wr >= NSTAGE) wr -= NSTAGE;
                tl.issue(lds_base, wr, wave);                    // refill the stage tile t-1 left (all waves passed the barrier)
            }
        } else {
            // the DMA of tile t+NSTAGE-1 refills the stage tile t-1 left (all waves passed the barrier): issued piece by piece
            // between the MFMAs of tile t
            const bool refill = !(MCRN_BF16_ABL & 1) && t + NSTAGE - 1 < nt;
            if (refill) {
                int wr = rd + NSTAGE - 1;
                if (wr >= NSTAGE) wr -= NSTAGE;
                tl.issue_begin(lds_base, wr, wave);
            }
            constexpr int NM = T::KS * NTM * FM * FN;
            static_assert(T::NPIECE <= NM, "at most one DMA piece per MFMA");
            // fragments of sub-step ks+1 are requested before the MFMAs of sub-step ks (two register sets)
#if MCRN_BF16_ABL & 2
            if (t == 0) {
#else
            bf16x8_t a[2][NOP][FM], b[2][NOP][FN];
            {
#endif
#pragma unroll
                for (int o = 0; o < NOP; ++o) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) a[0][o][i] = *reinterpret_cast<const bf16x8_t*>(sA + o * T::A_IMG + aoff[i]);
#pragma unroll
                    for (int j = 0; j < FN; ++j) b[0][o][j] = bf16_read_b<BN, BTR>(sB + o * T::B_IMG, boff[j], 0);
                }
            }
#pragma unroll
            for (int ks = 0; ks < T::KS; ++ks) {
                const int cur = (MCRN_BF16_ABL & 2) ? 0 : (ks & 1), nxt = cur ^ 1;
                if (!(MCRN_BF16_ABL & 2) && ks + 1 < T::KS) {
#pragma unroll
                    for (int o = 0; o < NOP; ++o) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) a[nxt][o][i] = *reinterpret_cast<const bf16x8_t*>(sA + o * T::A_IMG + (aoff[i] ^ ((ks + 1) << 5)));
#pragma unroll
                        for (int j = 0; j < FN; ++j) b[nxt][o][j] = bf16_read_b<BN, BTR>(sB + o * T::B_IMG, boff[j], ks + 1);
                    }
                }
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            const bf16x8_t& av = a[cur][tm == 2 ? 1 : 0][i];
                            const bf16x8_t& bv = b[cur][tm == 1 ? 1 : 0][j];
                            if (MCRN_BF16_ABL & 4) { asm volatile("" ::"v"(av), "v"(bv)); }
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
                            const int q = bf16_piece_after(((ks * NTM + tm) * FM + i) * FN + j, NM, T::NPIECE);
                            if (q >= 0) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (refill) tl.issue_piece(q, wave);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
            }
            if (refill) tl.issue_end();
        }
        if (++rd == NSTAGE) rd = 0;
    }
    MCRN_CLK_STAMP(p, 1);
    if (p.wide_cb && FN == 2 && (size_t)NW * WM * 128 <= (size_t)NSTAGE * T::STAGE) {   // (uniform: same barrier count for every wave)
        __syncthreads();                                         // every wave is done reading the operand stages
        if (bf16_epilogue_wide<FM, FN>(p, acc, m_blk + wm * WM, n_blk + wn * WN, lane, smem_bf16 + wave * (WM * 128), p.Cb, (int)p.cbm.lo)) return;
    }
    bf16_epilogue<FM, FN>(p, acc, split, m_blk + wm * WM, n_blk + wn * WN, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Kernel 2, ping-pong (8 waves = two groups of four, one wave of each group per SIMD).  The K loop is a sequence of
// barrier-separated PHASES; a group alternates a LOAD phase (all fragments of a K tile: LDS -> registers) with a
// COMPUTE phase (back-to-back MFMAs from registers), and group 1 runs one phase behind group 0: whenever one wave of
// a SIMD feeds the matrix pipe, its partner uses the LDS and issues the DMA of a later tile.
//     phase 2t   : G0 load(t)    | G1 compute(t-1)      every wave issues the DMA of tile t+NSTAGE-1 (its stage was
//     phase 2t+1 : G0 compute(t) | G1 load(t)           released by the barrier that ended phase 2t-1)
// and waits (counted vmcnt) for its share of tile t+1 before the barrier that ends phase 2t+1.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int NSTAGE, bool BTR, bool X3 = false>
struct PpLoop {
    static constexpr int WGM = 2, WGN = 4, NT = 512;
    static constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 32, FN = WN / 32;
    static_assert(BM % 64 == 0 && BN % 128 == 0, "tile shape");
    using T = Bf16Tile<BM, BN, BK, NT, BTR, X3>;
    static constexpr int KS = T::KS, NOP = T::NOP, NTM = T::NTM;

    // acc = sum over the K tiles [kt_beg, kt_beg + nt) of the C tile at (m_blk, n_blk).  Ends with every fragment in
    // registers; the caller puts a barrier between the use of acc and the next run() (LDS stages are reused).
    static __device__ __forceinline__ void run(const Bf16GemmP& p, unsigned char* smem, unsigned lds_base, int tid, int wave,
                                               int m_blk, int n_blk, int kt_beg, int nt, const int (&aoff)[FM],
                                               const int (&boff)[FN], f32x16_t (&acc)[FM][FN]) {
        const int grp = wave / WGN;                              // waves 0-3: group 0 (upper half of the tile), 4-7: group 1
        T tl;
        tl.init(p, tid, m_blk, n_blk, kt_beg);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

        bf16x8_t fa[NOP][FM][KS], fb[NOP][FN][KS];
        bool first_load = true;
        auto load_frags = [&](int stg) {
            if ((MCRN_BF16_ABL & 2) && !first_load) return;
            first_load = false;
            const unsigned char* sA = smem + stg * T::STAGE;
            const unsigned char* sB = sA + T::B_OFF;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int o = 0; o < NOP; ++o) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) fa[o][i][ks] = *reinterpret_cast<const bf16x8_t*>(sA + o * T::A_IMG + (aoff[i] ^ (ks << 5)));
#pragma unroll
                    for (int j = 0; j < FN; ++j) fb[o][j][ks] = bf16_read_b<BN, BTR>(sB + o * T::B_IMG, boff[j], ks);
                }
            }
        };
        constexpr int NM = KS * NTM * FM * FN;
        static_assert(T::NPIECE <= NM, "at most one DMA piece per MFMA");
        // dma_tag: std::true_type = the pieces of the tile opened with issue_begin go between the MFMAs (when `refill`)
        auto compute = [&](auto dma_tag, bool refill) {
            if (MCRN_BF16_ABL & 4) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int o = 0; o < NOP; ++o) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(fa[o][i][ks]));
#pragma unroll
                        for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(fb[o][j][ks]));
                    }
                return;
            }
            __builtin_amdgcn_s_setprio(1);
            // term tm of a fragment pair: 0 = hi x hi ; X3: 1 = hi x lo, 2 = lo x hi (FM * FN independent MFMAs between two on one accumulator)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm == 2 ? 1 : 0][i][ks], fb[tm == 1 ? 1 : 0][j][ks], acc[i][j], 0, 0, 0);
                            if constexpr (decltype(dma_tag)::value) {
                                const int q = bf16_piece_after(((ks * NTM + tm) * FM + i) * FN + j, NM, T::NPIECE);
                                if (q >= 0) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (refill) tl.issue_piece(q, wave);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
            __builtin_amdgcn_s_setprio(0);
        };
        // wait until this wave's share of tile u has landed: issued so far are tiles <= u + NSTAGE - 2
        auto wait_landed = [&](int u) {
            if (NSTAGE > 2 && u + NSTAGE - 2 < nt) MCRN_VMCNT((NSTAGE - 2) * T::NLD_MIN);
            else MCRN_VMCNT(0);
        };

#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nt) tl.issue(lds_base, s, wave);
        wait_landed(0);
        __syncthreads();
        int rd = 0, wr = NSTAGE - 1;                             // stage read next / stage the next DMA fills
        // G0C (A/B build only, -DMCRN_BF16_G0C=1; NSTAGE >= 3): group 0 also sends its share of a refill out BETWEEN the MFMAs of its compute
        // phase instead of behind the fragment reads of its load phase.  Measured 4 - 10 % SLOWER on every product (profiles/r6/experiments.md
        // section 6): a DMA piece between MFMAs stalls the wave's in-order issue - the opposite move, G1L below, is the one that pays.
        constexpr bool G0C = (MCRN_BF16_G0C != 0) && NSTAGE >= 3 && (MCRN_BF16_ILV & 2) != 0;
        // (measured per tile, profiles/r6/experiments.md section 6: 5 - 17 % fewer microseconds on every hi/lo tile and on the plain 256 x 256 /
        //  320 x 256 / 192 x 256 tiles; the plain 256 x 128 x 32 tile - 8 MFMAs per phase, the shortest - is 4 - 7 % slower with it and keeps round 4's form)
        constexpr bool G1L = (MCRN_BF16_G1L != 0) && NSTAGE >= 3 && !G0C && KS * NTM * FM * FN >= 16;
        if (grp == 0) {
            for (int t = 0; t < nt; ++t) {
                const bool refill = !(MCRN_BF16_ABL & 1) && t + NSTAGE - 1 < nt;
                load_frags(rd);                                  // phase 2t
                // (issuing the refill BEFORE the fragment reads of this phase - it goes to another stage - measured 4 - 10 % slower:
                //  profiles/r6/experiments.md section 4)
                if (!G0C && refill) { tl.issue(lds_base, wr, wave); if (++wr == NSTAGE) wr = 0; }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (G0C) {                             // phase 2t+1
                    if (refill) tl.issue_begin(lds_base, wr, wave);
                    compute(std::true_type{}, refill);
                    if (refill) { tl.issue_end(); if (++wr == NSTAGE) wr = 0; }
                } else compute(std::false_type{}, false);
                if (t + 1 < nt) wait_landed(t + 1);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (++rd == NSTAGE) rd = 0;
            }
        } else if constexpr (G1L) {
            // (round 6, NSTAGE >= 3) group 1 runs group 0's schedule one phase later: its share of tile t + NSTAGE - 1 goes out in its LOAD
            // phase, into the stage tile t - 1 left (every read of that stage retired before the last barrier: no race, no extra wait).
            // Between the MFMAs of the compute phase - round 4's form, kept where only two stages fit - every DMA piece holds the wave's
            // in-order issue for 60 - 180 cycles at the address path: ~90 cycles of idle matrix pipe per piece, a fifth of the loop
            // (profiles/r6/experiments.md section 6).  Here both compute phases are MFMA-only.
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                load_frags(rd);                                  // phase 2t+1
                if (!(MCRN_BF16_ABL & 1) && t + NSTAGE - 1 < nt) { tl.issue(lds_base, wr, wave); if (++wr == NSTAGE) wr = 0; }
                if (t + 1 < nt) wait_landed(t + 1);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                compute(std::false_type{}, false);               // phase 2t+2
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nt) __syncthreads();
                if (++rd == NSTAGE) rd = 0;
            }
        } else {
            if (!(MCRN_BF16_ABL & 1) && NSTAGE - 1 < nt) { tl.issue(lds_base, wr, wave); if (++wr == NSTAGE) wr = 0; }   // phase 0: nothing to compute yet
            __syncthreads();
            for (int t = 0; t < nt; ++t) {
                load_frags(rd);                                  // phase 2t+1
                if (t + 1 < nt) wait_landed(t + 1);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                // phase 2t+2: this group's share of tile t+NSTAGE goes out between its MFMAs (after them, the matrix pipe of
                // the SIMD sat idle while the wave queued 8 pieces at the address path)
                const bool refill = !(MCRN_BF16_ABL & 1) && t + NSTAGE < nt;
                if constexpr ((MCRN_BF16_ILV & 2) != 0) {
                    if (refill) tl.issue_begin(lds_base, wr, wave);
                    compute(std::true_type{}, refill);
                    if (refill) { tl.issue_end(); if (++wr == NSTAGE) wr = 0; }
                } else {
                    compute(std::false_type{}, false);
                    if (refill) { tl.issue(lds_base, wr, wave); if (++wr == NSTAGE) wr = 0; }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nt) __syncthreads();
                if (++rd == NSTAGE) rd = 0;
            }
        }
    }
};

template <int BM, int BN, int BK, int NSTAGE, bool BTR, int ROLE, bool X3 = false>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const Bf16GemmP p) {
    using L = PpLoop<BM, BN, BK, NSTAGE, BTR, X3>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf16[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_bf16);
    const int wm = wave / L::WGN, wn = wave % L::WGN;
    int tile_m, tile_n;
    int split;
    bf16_tile_of(p, BM, BN, tile_m, tile_n, split);
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int nkt = p.nseg * p.tps;
    const int kt_beg = split * p.tiles_per_split;
    const int kt_end = min(nkt, kt_beg + p.tiles_per_split);
    if (kt_beg >= kt_end) return;
    int aoff[L::FM], boff[L::FN];
    bf16_frag_offsets<L::FM, L::FN, BN, L::T::CH, L::T::RP, BTR>(wm * L::WM, wn * L::WN, lane, aoff, boff);
    f32x16_t acc[L::FM][L::FN];
    MCRN_CLK_PROBE(0);
    MCRN_CLK_STAMP(p, 0);
    // (no accumulator preload here: this kernel sits at the 256-VGPR limit of two waves per SIMD, and the preload's address arithmetic
    //  pushed it into 556 bytes of scratch per lane - found in round 5 after the fact: launch_one_bf16_pp clears cin_pre)
    L::run(p, smem_bf16, lds_base, tid, wave, m_blk, n_blk, kt_beg, kt_end - kt_beg, aoff, boff, acc);
    MCRN_CLK_PROBE(1);
    MCRN_CLK_STAMP(p, 1);
    if (p.wide_cb && L::FN == 2 && (size_t)8 * L::WM * 128 <= (size_t)NSTAGE * L::T::STAGE) {
        __syncthreads();                                         // both groups: every fragment read of the K loop has retired
        if (bf16_epilogue_wide<L::FM, L::FN>(p, acc, m_blk + wm * L::WM, n_blk + wn * L::WN, lane, smem_bf16 + wave * (L::WM * 128), p.Cb,
                                             (int)p.cbm.lo)) return;
    }
    bf16_epilogue<L::FM, L::FN>(p, acc, split, m_blk + wm * L::WM, n_blk + wn * L::WN, lane);
}

// ---- host side ----------------------------------------------------------------------------------
static inline void bf16_split_plan(Bf16GemmP& p, int BK, int nsplit) {
    p.tps = (p.seg_len + BK - 1) / BK;
    const int nkt = p.nseg * p.tps;                   // (hi/lo pairs: the three terms of a K tile are ONE tile of the walk)
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nkt) nsplit = nkt;
    p.tiles_per_split = (nkt + nsplit - 1) / nsplit;
    p.nsplit = (nkt + p.tiles_per_split - 1) / p.tiles_per_split;
}
template <int BM, int BN, int WGM, int WGN, int BK, int NSTAGE, bool BTR, int ROLE, bool X3 = false>
static inline hipError_t launch_one_bf16(Bf16GemmP p, int nsplit, hipStream_t st) {
    if (ROLE != 4 || MCRN_BF16_PRELOAD == 0) p.cin_pre = 0;   // (only the ROLE 4 instantiations carry the preload: everybody else adds Cin in the epilogue)
    bf16_split_plan(p, BK, nsplit);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * (BK / 8) * 16 * (X3 ? 2 : 1);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<BM, BN, WGM, WGN, BK, NSTAGE, BTR, ROLE, X3>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    (void)hipGetLastError();
    if (p.ev0 && p.ev1)
        hipExtLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, BK, NSTAGE, BTR, ROLE, X3>), dim3(tiles * p.nsplit), dim3(64 * WGM * WGN), lds, st,
                              (hipEvent_t)p.ev0, (hipEvent_t)p.ev1, 0, p);
    else
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WGM, WGN, BK, NSTAGE, BTR, ROLE, X3>), dim3(tiles * p.nsplit), dim3(64 * WGM * WGN), lds, st, p);
    return hipGetLastError();
}
template <int BM, int BN, int BK, int NSTAGE, bool BTR, int ROLE, bool X3 = false>
static inline hipError_t launch_one_bf16_pp(Bf16GemmP p, int nsplit, hipStream_t st) {
    p.cin_pre = 0;                                   // the ping-pong kernel keeps the read-modify-write epilogue (register budget, see the kernel)
    bf16_split_plan(p, BK, nsplit);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr size_t lds = (size_t)NSTAGE * (BM + BN) * (BK / 8) * 16 * (X3 ? 2 : 1);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<BM, BN, BK, NSTAGE, BTR, ROLE, X3>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    (void)hipGetLastError();
    if (p.ev0 && p.ev1)
        hipExtLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, BK, NSTAGE, BTR, ROLE, X3>), dim3(tiles * p.nsplit), dim3(512), lds, st,
                              (hipEvent_t)p.ev0, (hipEvent_t)p.ev1, 0, p);
    else
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<BM, BN, BK, NSTAGE, BTR, ROLE, X3>), dim3(tiles * p.nsplit), dim3(512), lds, st, p);
    return hipGetLastError();
}
// Translation units: the plain and the hi/lo kernels compile separately (gemm_bf16_unit.hip: MCRN_BF16_PART 1, gemm_bf16_x3_unit.hip: 2;
// ~2.5 min each); the stand-alone harness includes both (0).
#ifndef MCRN_BF16_PART
#define MCRN_BF16_PART 0
#endif
// harness builds only (tools/kbench, -DMCRN_BF16_CFGMASK=bits): instantiate just these tile slots (a one-slot A/B build compiles in seconds)
#ifndef MCRN_BF16_CFGMASK
#define MCRN_BF16_CFGMASK 0xFFFF
#endif
#define MCRN_CFG_ON(n) if constexpr (!(((MCRN_BF16_CFGMASK) >> (n)) & 1)) return hipErrorInvalidValue; else
hipError_t launch_gemm_bf16_hilo(const Bf16GemmP& p, bool btr, int cfg, int nsplit, int role, hipStream_t st);
#if MCRN_BF16_PART != 2
template <bool BTR, int ROLE>
static inline hipError_t launch_cfg_bf16(const Bf16GemmP& p, int cfg, int nsplit, hipStream_t st) {
    switch (cfg) {                                   //  BM   BN  waves  BK stages        LDS   workgroups / CU
        case 0: MCRN_CFG_ON(0) return launch_one_bf16<128, 128, 2, 2, 64, 2, BTR, ROLE>(p, nsplit, st);   //  64 KB   2
        case 1: MCRN_CFG_ON(1) return launch_one_bf16<256, 128, 4, 2, 64, 3, BTR, ROLE>(p, nsplit, st);   // 144 KB   1
        case 2: MCRN_CFG_ON(2) return launch_one_bf16<256, 256, 2, 4, 64, 2, BTR, ROLE>(p, nsplit, st);   // 128 KB   1
        case 3: MCRN_CFG_ON(3) return launch_one_bf16_pp<256, 256, 32, 4, BTR, ROLE>(p, nsplit, st);      // 128 KB   1   ping-pong
        case 4: MCRN_CFG_ON(4) return launch_one_bf16_pp<256, 256, 64, 2, BTR, ROLE>(p, nsplit, st);      // 128 KB   1   ping-pong, 64-deep phases
        case 5: MCRN_CFG_ON(5) return launch_one_bf16_pp<320, 256, 32, 4, BTR, ROLE>(p, nsplit, st);      // 144 KB   1   ping-pong
        case 6: MCRN_CFG_ON(6) return launch_one_bf16_pp<192, 256, 32, 4, BTR, ROLE>(p, nsplit, st);      // 112 KB   1   ping-pong
        case 7: MCRN_CFG_ON(7) return launch_one_bf16_pp<256, 128, 32, 4, BTR, ROLE>(p, nsplit, st);      //  96 KB   1   ping-pong
        case 8: MCRN_CFG_ON(8) return launch_one_bf16_pp<192, 256, 64, 2, BTR, ROLE>(p, nsplit, st);      // 112 KB   1   ping-pong, 64-deep phases
        case 9: MCRN_CFG_ON(9) return launch_one_bf16_pp<256, 128, 64, 2, BTR, ROLE>(p, nsplit, st);      //  96 KB   1   ping-pong, 64-deep phases
        // (round 6) three-stage ping-pong tiles of 32K outputs: every DMA instruction in a load phase (G1L needs a third stage), 64-deep phases.
        // 128 x 256: 64 x 64 wave tiles, 1.0 fragment reads per MFMA where 256 x 128 (128 x 32 wave tiles) has 1.25
        case 10: MCRN_CFG_ON(10) return launch_one_bf16_pp<128, 256, 64, 3, BTR, ROLE>(p, nsplit, st);    // 144 KB   1   ping-pong
        case 11: MCRN_CFG_ON(11) return launch_one_bf16_pp<256, 128, 64, 3, BTR, ROLE>(p, nsplit, st);    // 144 KB   1   ping-pong
        case 12: return hipErrorInvalidValue;                                             // retired slot (round 2's stream-K tiles, removed in round 5)
        // FOUR waves (one per SIMD), 128 x 64 wave tiles: a 32K-output tile - the size that fills 256 CUs in one round on
        // the hoisted N = 1843 encoder product (7372 x 1024: 232 tiles) - read with 0.75 LDS fragment reads per MFMA, like
        // the 256 x 256 eight-wave tile (the eight-wave 256 x 128 forms have 64 x 64 or 128 x 32 wave tiles: 1.0 / 1.25)
        case 13: MCRN_CFG_ON(13) return launch_one_bf16<256, 128, 2, 2, 64, 3, BTR, ROLE>(p, nsplit, st);  // 144 KB   1
        case 14: MCRN_CFG_ON(14) return launch_one_bf16<128, 256, 1, 4, 64, 3, BTR, ROLE>(p, nsplit, st);  // 144 KB   1
        default: MCRN_CFG_ON(15) return launch_one_bf16<256, 192, 2, 2, 64, 2, BTR, ROLE>(p, nsplit, st);  // 112 KB   1   128 x 96 wave tiles
    }
}
#endif
#if MCRN_BF16_PART != 1
// hi/lo operand pairs (nterm == 3): the same tile shapes, every K tile FOUR images wide (A_hi, A_lo, B_hi, B_lo) and 16 or 32 deep, so a stage
// holds as many bytes as a 32- / 64-deep plain tile.
template <bool BTR, int ROLE>
static inline hipError_t launch_cfg_bf16_x3(const Bf16GemmP& p, int cfg, int nsplit, hipStream_t st) {
    switch (cfg) {                                   //  BM   BN  waves  BK stages        LDS   workgroups / CU
        case 0: MCRN_CFG_ON(0) return launch_one_bf16<128, 128, 2, 2, 32, 2, BTR, ROLE, true>(p, nsplit, st);   //  64 KB   2
        case 1: MCRN_CFG_ON(1) return launch_one_bf16<256, 128, 4, 2, 32, 3, BTR, ROLE, true>(p, nsplit, st);   // 144 KB   1
        case 2: MCRN_CFG_ON(2) return launch_one_bf16<256, 256, 2, 4, 32, 2, BTR, ROLE, true>(p, nsplit, st);   // 128 KB   1
        // ping-pong, 16-deep K tiles (one MFMA step of three terms per tile): FOUR stages where the 32-deep form has two - a refill is issued
        // 5 phases before it is read and group 0 sends its share between its MFMAs too (PpLoop G0C)
        case 3: MCRN_CFG_ON(3) return launch_one_bf16_pp<256, 256, 16, 4, BTR, ROLE, true>(p, nsplit, st);      // 128 KB   1   ping-pong
        case 4: MCRN_CFG_ON(4) return launch_one_bf16_pp<256, 256, 32, 2, BTR, ROLE, true>(p, nsplit, st);      // 128 KB   1   ping-pong, 32-deep phases
        case 5: MCRN_CFG_ON(5) return launch_one_bf16_pp<320, 256, 16, 4, BTR, ROLE, true>(p, nsplit, st);      // 144 KB   1   ping-pong
        case 6: MCRN_CFG_ON(6) return launch_one_bf16_pp<192, 256, 16, 4, BTR, ROLE, true>(p, nsplit, st);      // 112 KB   1   ping-pong
        case 7: MCRN_CFG_ON(7) return launch_one_bf16_pp<256, 128, 32, 3, BTR, ROLE, true>(p, nsplit, st);      // 144 KB   1   ping-pong
        case 8: MCRN_CFG_ON(8) return launch_one_bf16_pp<192, 256, 32, 2, BTR, ROLE, true>(p, nsplit, st);      // 112 KB   1   ping-pong, 32-deep phases
        case 9: MCRN_CFG_ON(9) return launch_one_bf16_pp<256, 128, 16, 4, BTR, ROLE, true>(p, nsplit, st);      //  96 KB   1   ping-pong
        case 10: MCRN_CFG_ON(10) return launch_one_bf16_pp<128, 256, 32, 3, BTR, ROLE, true>(p, nsplit, st);     // 144 KB   1   ping-pong
        case 11: MCRN_CFG_ON(11) return launch_one_bf16_pp<128, 256, 16, 4, BTR, ROLE, true>(p, nsplit, st);     //  96 KB   1   ping-pong
        case 13: MCRN_CFG_ON(13) return launch_one_bf16<256, 128, 2, 2, 32, 3, BTR, ROLE, true>(p, nsplit, st);  // 144 KB   1
        case 14: MCRN_CFG_ON(14) return launch_one_bf16<128, 256, 1, 4, 32, 3, BTR, ROLE, true>(p, nsplit, st);  // 144 KB   1
        case 15: MCRN_CFG_ON(15) return launch_one_bf16<256, 192, 2, 2, 32, 2, BTR, ROLE, true>(p, nsplit, st);  // 112 KB   1
        default: return hipErrorInvalidValue;                                              // retired slots
    }
}
hipError_t launch_gemm_bf16_hilo(const Bf16GemmP& p, bool btr, int cfg, int nsplit, int role, hipStream_t st) {
    // each hot role uses one storage form of B; everything else is "misc"
    if (role == 1 && btr) return launch_cfg_bf16_x3<true, 1>(p, cfg, nsplit, st);
    if (role == 4 && btr) return launch_cfg_bf16_x3<true, 4>(p, cfg, nsplit, st);
    if (role == 5 && !btr) return launch_cfg_bf16_x3<false, 5>(p, cfg, nsplit, st);
    return btr ? launch_cfg_bf16_x3<true, 0>(p, cfg, nsplit, st) : launch_cfg_bf16_x3<false, 0>(p, cfg, nsplit, st);
}
#endif
#if MCRN_BF16_PART != 2
hipError_t launch_gemm_bf16(Bf16GemmP p, bool btr, int cfg, int nsplit, int role, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.nseg <= 0 || p.seg_len <= 0) return hipSuccess;
    if (p.nterm != 0 && p.nterm != 1 && p.nterm != 3) return hipErrorInvalidValue;
    const bool x3 = p.nterm == 3;
    if (!bf16_cfg_ok(cfg, x3)) return hipErrorInvalidValue;
    // stacked outputs whose blocks are contiguous ((r / inner) * hi + (r % inner) * lo with hi == inner * lo) are plain rows
    if (p.cm.inner > 0 && p.cm.hi == (long long)p.cm.inner * p.cm.lo) p.cm = rm_plain(p.cm.lo);
    if (p.cbm.inner > 0 && p.cbm.hi == (long long)p.cbm.inner * p.cbm.lo) p.cbm = rm_plain(p.cbm.lo);
    if (p.cm.inner > 0 || (p.Cb && p.cbm.inner > 0)) return hipErrorInvalidValue;   // the epilogue addresses plain-strided rows
    if (btr && ((p.N & 7) || p.N < 8)) return hipErrorInvalidValue;     // [k][n] operands are fetched in 8-column chunks
    if (p.cn_inner > 0 && ((p.cn_inner & 31) || p.Cb)) return hipErrorInvalidValue;   // a fragment must not straddle an inner block
    {
        p.wide_cb = (p.Cb && !p.C && !p.Cin && p.nsplit <= 1 && nsplit <= 1 && (p.cbm.lo & 7) == 0 &&
                     ((uintptr_t)p.Cb & 15) == 0) ? 1 : 0;
    }
    if (p.cin_pre && !(p.Cin && p.C && p.alpha == 1.f && p.beta == 1.f && (p.nsplit <= 1 || p.cin_first_only) && (nsplit <= 1 || p.cin_first_only)))
        return hipErrorInvalidValue;                                    // (the preload IS the addend: no scaling, split 0 only)
    // each hot role uses one storage form of B; everything else is "misc"
    if (x3) return launch_gemm_bf16_hilo(p, btr, cfg, nsplit, role, st);
    if (role == 1 && btr) return launch_cfg_bf16<true, 1>(p, cfg, nsplit, st);
    if (role == 4 && btr) return launch_cfg_bf16<true, 4>(p, cfg, nsplit, st);
    if (role == 5 && !btr) return launch_cfg_bf16<false, 5>(p, cfg, nsplit, st);
    return btr ? launch_cfg_bf16<true, 0>(p, cfg, nsplit, st) : launch_cfg_bf16<false, 0>(p, cfg, nsplit, st);
}
#endif

}  // namespace mcrn
