// prop_small.h - adjacency-stationary K-hop propagation for small graphs (N <= 256), gfx950.
//
//   OUT[b] = alpha * sum_seg  S_seg[b] (N x N)  x  X_seg[b] (N x ncols)  + beta * Cin[b]
//
// This is model/MegaCRN.py:25 (einsum 'nm,bmc->bnc') and its transpose in the backward pass, on the
// node-major planes (X row n = all (batch, channel) columns of node n).  At METR-LA / PEMS-BAY sizes
// the generic tiled GEMM is instruction- and latency-bound: K = N is only 7-11 tiles deep, so
// per-tile address math, fp32->bf16 splitting and prologue/epilogue dominate (measured: 47 VALU
// instructions per MFMA).  Here instead:
//   * the adjacency is pre-split ONCE per step into bf16 hi/lo in MFMA A-fragment order (k_sfrag_build);
//     a wave loads its 32 rows of S for the WHOLE K extent straight into registers with perfectly
//     coalesced 16-byte loads (NF*2 k-steps * 2 * 4 VGPRs = 112 VGPRs at N <= 224) - S never touches LDS;
//   * a workgroup = NF waves (one per 32-row fragment) owns 64 columns of X: every X element is
//     fetched once (float4, coalesced), split once, written to LDS in B-fragment order (lane-linear,
//     conflict-free), then all NF waves run 2 tiles * KS k-steps * 3 bf16 MFMAs back to back;
//   * the column -> lane slot permutation col = 4*(s&7) + (s>>3) makes the staging writes conflict-free.
// Arithmetic: bf16x3 split (see gemm_bf16x3.h), fp32 accumulate.
#pragma once
#include <stdlib.h>
#include "gemm_bf16x3.h"

namespace mcrn {

// Sfrag[((i*KS + ks)*2 + hl)*64 + lane] = 8 bf16 of  A[row = 32 i + (lane&31)][k = 16 ks + 8 (lane>>5) + 0..7]
// A = S (transpose == 0) or S^T (transpose == 1); out-of-range rows / k are zero.
static __global__ void k_sfrag_build(const float* __restrict__ S, long long ldS, int N, int NF, int transpose,
                              uint4* __restrict__ out) {
    const int KS = 2 * NF;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NF * KS * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % KS, i = (idx >> 6) / KS;
    const int row = 32 * i + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = (row < N && k < N) ? (transpose ? S[(long long)k * ldS + row] : S[(long long)row * ldS + k]) : 0.f;
    }
    uint4 h, l;
    split8(v, h, l);
    out[((long long)(i * KS + ks) * 2 + 0) * 64 + lane] = h;
    out[((long long)(i * KS + ks) * 2 + 1) * 64 + lane] = l;
}

// fragment images (k_sfrag_build layout) of up to 8 matrices in ONE launch: blockIdx.y = which
struct SfragMultiP {
    const float* S[8];
    uint4* out[8];
    int transpose[8];
    long long ldS;
    int N, NF, n;
};
static __global__ void k_sfrag_build_multi(const SfragMultiP p) {
    const int KS = 2 * p.NF;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.NF * KS * 64) return;
    const int which = blockIdx.y;
    const float* __restrict__ S = p.S[which];
    const int transpose = p.transpose[which];
    const int lane = idx & 63, ks = (idx >> 6) % KS, i = (idx >> 6) / KS;
    const int row = 32 * i + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = (row < p.N && k < p.N) ? (transpose ? S[(long long)k * p.ldS + row] : S[(long long)row * p.ldS + k]) : 0.f;
    }
    uint4 h, l;
    split8(v, h, l);
    uint4* __restrict__ out = p.out[which];
    out[((long long)(i * KS + ks) * 2 + 0) * 64 + lane] = h;
    out[((long long)(i * KS + ks) * 2 + 1) * 64 + lane] = l;
}
static inline hipError_t launch_sfrag_multi(const SfragMultiP& p, hipStream_t st) {
    (void)hipGetLastError();
    const int tot = p.NF * 2 * p.NF * 64;
    hipLaunchKernelGGL(k_sfrag_build_multi, dim3((tot + 255) / 256, p.n), dim3(256), 0, st, p);
    return hipGetLastError();
}

template <int NF>
__global__ __launch_bounds__(64 * NF) void prop_small_kernel(const PropP p) {
    constexpr int KS = 2 * NF;                       // k-steps of 16
    __shared__ uint4 img[2 * KS * 2 * 64];           // [coltile][ks][hi/lo][slot]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int batch = blockIdx.y;
    const int colbase = blockIdx.x * 64;
    const int l31 = lane & 31, kq = lane >> 5;

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    for (int seg = 0; seg < p.nseg; ++seg) {
        // ---- this wave's 32 rows of the adjacency for the whole K extent: registers, straight from L2
        const uint4* __restrict__ sf = p.Sf[batch][seg] + (long long)w * KS * 2 * 64 + lane;
        uint4 ah[KS], al[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ah[ks] = sf[(ks * 2 + 0) * 64];
            al[ks] = sf[(ks * 2 + 1) * 64];
        }
        // ---- stage 64 columns of X: thread = (4-column group, 8-k group); exactly 64*NF work items
        if (seg > 0) __syncthreads();                // previous segment's image fully consumed
        {
            const float* __restrict__ X = p.X[batch][seg];
            const int cg = tid & 15, kg = tid >> 4;
            const int col = colbase + 4 * cg;
            const bool cv = col < p.ncols;           // ncols % 4 == 0: a float4 is all-in or all-out
            float v[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 8 * kg + i;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cv && k < p.N) t = *reinterpret_cast<const float4*>(X + (long long)k * p.ld + col);
                v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
            }
            const int ct = cg >> 3, ks = kg >> 1, kqq = kg & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float col8[8] = {v[0][c], v[1][c], v[2][c], v[3][c], v[4][c], v[5][c], v[6][c], v[7][c]};
                uint4 h, l;
                split8(col8, h, l);
                const int slot = c * 8 + (cg & 7) + 32 * kqq;      // column 4*(cg&7)+c of tile ct
                img[((ct * KS + ks) * 2 + 0) * 64 + slot] = h;
                img[((ct * KS + ks) * 2 + 1) * 64 + slot] = l;
            }
        }
        __syncthreads();
        // ---- 2 column tiles x KS k-steps x 3 MFMAs, operands: A in registers, B lane-linear in LDS
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, img[((t * KS + ks) * 2 + 0) * 64 + lane]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, img[((t * KS + ks) * 2 + 1) * 64 + lane]);
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[ks]);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, al[ks]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: C/D layout row = (v&3) + 8*(v>>2) + 4*(lane>>5), column slot = lane&31
    float* __restrict__ C = p.C[batch];
    const float* __restrict__ Cin = p.Cin[batch];
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = colbase + 32 * t + cperm;
        if (col >= p.ncols) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * w + (v & 3) + 8 * (v >> 2) + 4 * kq;
            if (r < p.N) {
                const long long off = (long long)r * p.ld + col;
                float o = p.alpha * acc[t][v];
                if (Cin) o += p.beta * Cin[off];
                C[off] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused two-hop variants (cheb_k = 3).  A workgroup owns ALL N rows of its 64 columns, so the second
// hop needs nothing from other workgroups: the hop-1 result is converted from the accumulators
// straight into the next B image (each lane owns 4 consecutive rows = one 8-byte half of a B-fragment
// vector) and S stays in registers for both hops.
// ---------------------------------------------------------------------------------------------
template <int NF, int CT, bool STREAM = (NF > 8)>      // CT = 32-column tiles per workgroup unit (1 .. 4)
struct PropBlock {
    static constexpr int KS = 2 * NF;
    static constexpr int IMG = CT * KS * 2 * 64;     // uint4
    // 256 < N <= 352 (NF = 9..11): a wave's S fragments for the whole K extent would be 16 NF = 176 VGPRs and the workgroup
    // has 11 waves (3 per SIMD: 168 VGPRs each) - S is not register-stationary there: every k-step's hi and lo fragments
    // are streamed from L2 (coalesced 2 KB per wave and k-step; the 0.5 MB image is shared by every workgroup) through a
    // small register ring, a few k-steps ahead of the MFMAs that use them.
    // (STREAM may also be asked for at NF <= 8 - prop_mform.h: ~110 instead of ~220 VGPRs, two workgroups per CU)
    static constexpr bool WIDE = STREAM;
    static constexpr int NAL = WIDE ? 1 : KS;        // fragments held in registers (hi and lo alike)
    static constexpr int RD = 3;                     // ring depth of the streamed fragments
    static_assert(!(WIDE && CT == 1), "the wide variant is built for 2 / 3 column tiles");
    // stage 32*CT columns of a plane into the B image: thread = (4-column group, 8-k group).
    // The 8 loads are UNCONDITIONAL from clamped in-range addresses and zeroed afterwards: a load under a
    // divergent branch gets its own basic block and (measured with the in-kernel timeline) one full memory
    // round trip each.  Offsets are 32-bit: base lane offset + scalar multiples of the row stride.
    // cstep: floats between consecutive column QUADS of the unit (4 = contiguous columns; Cp = the gathered form: quad q is the 4
    // input / pad channels of sample q0 + q, colbase = q0 * Cp + H)
    static __device__ __forceinline__ void stage(uint4* img, const float* __restrict__ X, int ld, int nlast /* N-1 */,
                                                 int ncols, int colbase, int tid, int cstep = 4) {
        // items = (8*CT column quads) x (4*NF k-groups); CT == 1: half of the threads have no item, CT == 3: a second round
#pragma unroll
        for (int it0 = 0; it0 < 8 * CT * 4 * NF; it0 += 64 * NF) {
            const int it = it0 + tid;
            if (it >= 8 * CT * 4 * NF) break;
            const int cg = it % (8 * CT), kg = it / (8 * CT);
            const int col = colbase + cstep * cg;
            const bool cv = col < ncols;              // ncols % 4 == 0: a float4 is all-in or all-out
            const int colc = cv ? col : 0;
            float v[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 8 * kg + i;
                const float4 t = *reinterpret_cast<const float4*>(X + (unsigned)(min(k, nlast) * ld + colc));
                const bool ok = k <= nlast;
                v[i][0] = ok ? t.x : 0.f; v[i][1] = ok ? t.y : 0.f; v[i][2] = ok ? t.z : 0.f; v[i][3] = ok ? t.w : 0.f;
            }
            const int ct = cg >> 3, ks = kg >> 1, kqq = kg & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float col8[8] = {v[0][c], v[1][c], v[2][c], v[3][c], v[4][c], v[5][c], v[6][c], v[7][c]};
                uint4 h, l;
                split8(col8, h, l);
                if (!cv) { h = make_uint4(0u, 0u, 0u, 0u); l = h; }
                const int slot = c * 8 + (cg & 7) + 32 * kqq;
                img[((ct * KS + ks) * 2 + 0) * 64 + slot] = h;
                img[((ct * KS + ks) * 2 + 1) * 64 + slot] = l;
            }
        }
    }
    static __device__ __forceinline__ void load_a(const uint4* __restrict__ sfw, uint4 (&ah)[NAL], uint4 (&al)[NAL]) {
        if constexpr (!WIDE) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ah[ks] = sfw[(ks * 2 + 0) * 64];
                al[ks] = sfw[(ks * 2 + 1) * 64];
            }
        }
    }
    // acc[t] = A x img[t].  Independent accumulator chains hide the MFMA dependent-issue latency:
    // CT >= 2: the column tiles; CT == 1: the three split products, summed at the end.
    // INIT: acc already holds the addend (loaded straight into the accumulator registers: costs no extra VGPRs)
    template <bool INIT = false>
    static __device__ __forceinline__ void mma(const uint4* img, const uint4 (&ah)[NAL], const uint4 (&al)[NAL],
                                               f32x16 (&acc)[CT], int lane, const uint4* __restrict__ sfw, int ks0 = 0) {
        if constexpr (CT >= 2) {
            if constexpr (!INIT) {
#pragma unroll
                for (int t = 0; t < CT; ++t)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
            }
            // WIDE: hi and lo fragments of every k-step come from L2 through a ring of RD register pairs.  The loads are
            // inline asm (SGPR base + the lane's 32-bit offset) with explicit vmcnt waits: written as ordinary loads the
            // compiler - at the register limit here - sinks each one to just before its use (one exposed L2 round trip per
            // k-step: 60 us per launch instead of ~35).  The k-steps are walked from a per-workgroup starting point ks0
            // (wrapping): all workgroups stream the SAME 0.5 MB image in lockstep, and with a common order every CU asks
            // the same L2 channel at the same time.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 rh[WIDE ? RD : 1], rl[WIDE ? RD : 1];
            const unsigned char* sbase = nullptr;
            unsigned voff = 0;
            auto kof = [&](int i) { int k = i + ks0; return k >= KS ? k - KS : k; };
            auto issue = [&](int slot, int i) {
                const unsigned char* b0 = sbase + (long long)kof(i) * 2048;      // k-step image: [hi | lo] x 64 lanes x 16 B
                // ("s" operands need readfirstlane even when uniform by construction; folded away where the compiler knows it)
                const unsigned b_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)b0);
                const unsigned b_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b0 >> 32));
                const unsigned char* b = reinterpret_cast<const unsigned char*>(((unsigned long long)b_hi << 32) | (unsigned long long)b_lo);
                // s_nop 4: the compiler may produce `b` with VALU instructions (v_readfirstlane / v_readlane of a spilled offset, a
                // 64-bit VALU add when it runs out of SGPRs) right in front of this block.  A VMEM instruction that reads an SGPR
                // written by a VALU instruction needs 5 wait states, and the hazard recogniser cannot see into inline asm: without
                // them the load uses the STALE register pair - one k-step of wrong fragments (found with tools/kbench/prop1_test:
                // 3e-2 .. 6e-2 errors in exactly the variants whose ISA had such a pair; tools/isa_hazards.py scans for it).
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                             : "=&v"(rh[slot]), "=&v"(rl[slot]) : "v"(voff), "s"(b));
            };
            if constexpr (WIDE) {
                // sfw = wave base + lane: split into the wave-uniform base and the lane's byte offset
                const unsigned lane_b = (unsigned)(threadIdx.x & 63) * 16u;
                sbase = reinterpret_cast<const unsigned char*>(sfw) - lane_b;
                // (readfirstlane returns int: without the unsigned casts a low half with bit 31 set sign-extends into the high half)
                const unsigned sb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)sbase);
                const unsigned sb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)sbase >> 32));
                sbase = reinterpret_cast<const unsigned char*>(((unsigned long long)sb_hi << 32) | (unsigned long long)sb_lo);
                voff = lane_b;
#pragma unroll
                for (int i = 0; i < RD; ++i) issue(i, i);
            }
#pragma unroll
            for (int ksi = 0; ksi < KS; ++ksi) {
                const int ks = WIDE ? kof(ksi) : ksi;
                uint4 hi4, lo4;
                if constexpr (WIDE) {
                    // pair ksi has landed once at most the younger pairs (issued up to ksi + RD - 1) remain in flight
                    constexpr int dummy = 0; (void)dummy;
                    const int slot = ksi % RD;
                    if (ksi + RD <= KS) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rh[slot]), "+v"(rl[slot]) : "n"(2 * (RD - 1)));
                    else if (ksi + 2 == KS) asm volatile("s_waitcnt vmcnt(2)" : "+v"(rh[slot]), "+v"(rl[slot]));
                    else if (ksi + 1 == KS) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rh[slot]), "+v"(rl[slot]));
                    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rh[slot]), "+v"(rl[slot]) : "n"(2 * (RD - 1)));
                    hi4 = make_uint4(rh[slot][0], rh[slot][1], rh[slot][2], rh[slot][3]);
                    lo4 = make_uint4(rl[slot][0], rl[slot][1], rl[slot][2], rl[slot][3]);
                } else {
                    hi4 = ah[ks]; lo4 = al[ks];
                }
                const bf16x8 xh = __builtin_bit_cast(bf16x8, hi4);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, lo4);
                bf16x8 bh[CT], bl[CT];
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    bh[t] = __builtin_bit_cast(bf16x8, img[((t * KS + ks) * 2 + 0) * 64 + lane]);
                    bl[t] = __builtin_bit_cast(bf16x8, img[((t * KS + ks) * 2 + 1) * 64 + lane]);
                }
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl[t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh[t], acc[t], 0, 0, 0);
                if constexpr (WIDE) {
                    if (ksi + RD < KS) issue(ksi % RD, ksi + RD);    // refill the slot the MFMAs above have just read
                }
            }
        } else {
            f32x16 a0, a1, a2;
#pragma unroll
            for (int v = 0; v < 16; ++v) { a0[v] = 0.f; a1[v] = 0.f; a2[v] = INIT ? acc[0][v] : 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[WIDE ? 0 : ks]);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, al[WIDE ? 0 : ks]);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, img[(ks * 2 + 0) * 64 + lane]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, img[(ks * 2 + 1) * 64 + lane]);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, a2, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[0][v] = (a0[v] + a1[v]) + a2[v];
        }
    }
    // values val[t][v] in C/D layout (wave w = rows 32w..32w+31) -> B image of the next hop.
    // lane (slot, kq) holds rows 8g + 4kq + 0..3 for g = v>>2: the kq-th 8-byte half of the vector
    // (k-step 2w + (g>>1), k-half g&1, slot).
    static __device__ __forceinline__ void to_img(uint4* img, const f32x16 (&val)[CT], int w, int lane) {
        const int l31 = lane & 31, kq = lane >> 5;
        uint2* img2 = reinterpret_cast<uint2*>(img);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float a0 = val[t][4 * g], a1 = val[t][4 * g + 1], a2 = val[t][4 * g + 2], a3 = val[t][4 * g + 3];
                const unsigned h01 = cvt_pk_bf16(a0, a1), h23 = cvt_pk_bf16(a2, a3);
                const unsigned l01 = cvt_pk_bf16(a0 - __uint_as_float(h01 << 16), a1 - __uint_as_float(h01 & 0xFFFF0000u));
                const unsigned l23 = cvt_pk_bf16(a2 - __uint_as_float(h23 << 16), a3 - __uint_as_float(h23 & 0xFFFF0000u));
                const int ks = 2 * w + (g >> 1);
                const int slot = l31 + 32 * (g & 1);
                img2[(((t * KS + ks) * 2 + 0) * 64 + slot) * 2 + kq] = make_uint2(h01, h23);
                img2[(((t * KS + ks) * 2 + 1) * 64 + slot) * 2 + kq] = make_uint2(l01, l23);
            }
    }
};

// Element addressing shared by the fused kernels.  Lane (l31, kq) of wave w owns, per column tile t and
// accumulator register v, the element (row 32w + 4kq + (v&3) + 8(v>>2), column colbase + 32t + cperm).
// Offsets are 32-bit (< 2^31: one plane of an N <= 256 graph) = lane base + a SCALAR multiple of the row
// stride; the stride is re-read through an opaque move every unit so the 32 row offsets are not hoisted
// out of the unit loop into 64 live VGPRs (that spilled, and every reload drained vmcnt: 12 us per unit).
#define MCRN_ROW_OF(v) (((v) & 3) + 8 * ((v) >> 2))
// re-read before every load / store phase: offsets are recomputed (16 adds) instead of kept live across the MFMA phases
#define MCRN_FRESH(x) asm volatile("" : "+s"(x))

// forward, one workgroup per (32*CT columns, support s = blockIdx.y):
//   plane[1+2s] = S_s plane[0] ;  plane[2+2s] = 2 S_s plane[1+2s] - plane[0]      (model/MegaCRN.py:20-25)
// (body of one launch for the workgroup (bxi, s), S fragments already in ah / al: shared by prop2_fwd_kernel and by the persistent
//  multi-cell prototype of tools/kbench/prop_chain_test.hip, which calls it once per cell between grid barriers)
template <int NF, int CT>
__device__ __forceinline__ void prop2_fwd_units(const Prop2P& p, float* __restrict__ base, uint4* const img, const int bxi, const int s,
                                                const uint4 (&ah)[PropBlock<NF, CT>::NAL], const uint4 (&al)[PropBlock<NF, CT>::NAL],
                                                const uint4* __restrict__ sfw, const int ks0) {
    using PB = PropBlock<NF, CT>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
    const float* __restrict__ X0 = base;
    float* __restrict__ X1 = base + (long long)(1 + 2 * s) * p.PS;
    float* __restrict__ X2 = base + (long long)(2 + 2 * s) * p.PS;
    const int row0 = 32 * w + 4 * kq;
    const bool rows_in = 32 * w + 32 <= p.N;           // wave-uniform: every row of this wave exists
    const int nunits = p.nunits > 0 ? p.nunits : (p.ncols + 32 * CT - 1) / (32 * CT);
    const int u0 = (int)(((long long)bxi * nunits) / p.nblk), u1 = (int)(((long long)(bxi + 1) * nunits) / p.nblk);
    for (int unit = u0; unit < u1; ++unit) {
        // (state columns only: the unit's 64 columns sit inside one sample's row of cstride floats, see Prop2P)
        const int colbase = p.nunits > 0 ? (unit / p.cps) * p.cstride + (unit % p.cps) * 32 * CT : unit * 32 * CT;
        int ld = (int)p.ld;
        asm volatile("" : "+s"(ld));                   // see MCRN_ROW_OF
        if (unit > u0) __syncthreads();                // previous unit's hop-2 image fully consumed
        MCRN_TL(0, 2);
        int nlast = p.N - 1;
        int tidv = tid;
        MCRN_FRESH(ld); MCRN_FRESH(nlast);              // nothing the loads need is hoisted out of the unit loop (and spilled)
        asm volatile("" : "+v"(tidv));                  // ... nor the per-thread LDS store addresses of the staging
        PB::stage(img, X0, ld, nlast, p.ncols, colbase, tidv);
        MCRN_TL(0, 3);
        __syncthreads();
        MCRN_TL(0, 4);
        f32x16 acc[CT];
        PB::mma(img, ah, al, acc, lane, sfw, ks0);
        // X1 out (fp32) + next image
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 32 * t + cperm;
            const unsigned o = (unsigned)(row0 * ld + col);
            if (col < p.ncols) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) X1[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = acc[t][v];
            }
        }
        MCRN_TL(0, 5);
        __syncthreads();                               // every wave finished reading the hop-1 image
        MCRN_TL(0, 6);
        PB::to_img(img, acc, w, lane);
        // x2 = 2 S x1 - x0 = 2 (S x1 - x0/2): x0 is loaded STRAIGHT INTO the accumulators (dead after to_img) as the
        // initial value of hop 2 - no extra registers, and the loads (cache hits: the lines were staged) fly during the
        // barrier.  Clamped addresses, no predicate: out-of-range elements are never stored.
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = min(colbase + 32 * t + cperm, p.ncols - 1);
            if (rows_in) {
                const unsigned o = (unsigned)(row0 * ld + col);
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = X0[o + (unsigned)(MCRN_ROW_OF(v) * ld)];
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = X0[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ld + col)];
            }
        }
        __syncthreads();
        MCRN_TL(0, 7);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] *= -0.5f;
        PB::template mma<true>(img, ah, al, acc, lane, sfw, ks0);
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 32 * t + cperm;
            const unsigned o = (unsigned)(row0 * ld + col);
            if (col < p.ncols) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) X2[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = 2.f * acc[t][v];
            }
        }
        MCRN_TL(0, 8);
    }   // unit loop
}
template <int NF, int CT>
__global__ __launch_bounds__(64 * NF) void prop2_fwd_kernel(const Prop2P p) {
    using PB = PropBlock<NF, CT>;
    constexpr int KS = 2 * NF;
    extern __shared__ __attribute__((aligned(16))) uint4 prop2_img[];   // PB::IMG uint4 (up to 132 KB at N = 352: dynamic)
    uint4* const img = prop2_img;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // 1-D grid of 16 * ceil(nblk / 8) workgroups: block b -> XCD b % 8, slot b / 8 -> (support = slot & 1, unit range (slot / 2) * 8 + XCD).
    // The two supports of a unit range stage the SAME columns of plane 0: they share one XCD's L2 (round 5; with grid.y = support they
    // landed on different XCDs whenever the range count is not a multiple of 8 - the METR-LA encoder: 68)
    const int s = (int)((blockIdx.x >> 3) & 1);
    const int bxi = (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7));
    if (bxi >= p.nblk) return;                         // (padding of the last round; uniform over the workgroup)

    // S stays in registers while the workgroup walks its balanced range of column units: the grid is capped
    // so that all workgroups are resident at once (no second round when #units is just above #CUs)
    uint4 ah[PB::NAL], al[PB::NAL];
    const uint4* __restrict__ sfw0 = p.Sf[s] + (long long)w * KS * 2 * 64 + lane;
    const uint4* __restrict__ sfw = PB::WIDE ? sfw0 : nullptr;   // (kept live only where the fragments are streamed)
    const int ks0 = PB::WIDE ? (int)((bxi * 7 + s * 3) % KS) : 0;
    MCRN_TL(0, 0);
    PB::load_a(sfw0, ah, al);
    MCRN_TL(0, 1);
    prop2_fwd_units<NF, CT>(p, p.base, img, bxi, s, ah, al, sfw, ks0);
#ifdef MCRN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) g_tl[0][blockIdx.x & 511][9] = wall_clock64();
#endif
}

// backward, one workgroup per (32*CT columns, support s = blockIdx.y):
//   d1t_s = dP[1+2s] + S_s^T dP[2+2s]   (written back to dP[1+2s])
//   s = 0:  dP[0] += S_1^T d1t_1        (read-modify-write)
//   s = 1:  extra  = S_2^T d1t_2        (plain store; the consumers of dP[0] add it)
// Each element has exactly one writer, so the result is deterministic while both supports run on different CUs.
// The addends dP[1+2s] and dP[0] are loaded (clamped, unpredicated) straight into the accumulators before the MFMA
// chain that adds to them, so no memory round trip sits between an MFMA phase and its stores.
template <int NF, int CT>
__device__ __forceinline__ void prop2_bwd_units(const Prop2P& p, float* __restrict__ base, float* __restrict__ EX, uint4* const img,
                                                const int bx, const int nbx, const int s,
                                                const uint4 (&ah)[PropBlock<NF, CT>::NAL], const uint4 (&al)[PropBlock<NF, CT>::NAL],
                                                const uint4* __restrict__ sfw, const int ks0) {
    using PB = PropBlock<NF, CT>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
    float* __restrict__ D0 = base;
    float* __restrict__ D1 = base + (long long)(1 + 2 * s) * p.PS;
    const float* __restrict__ E2 = base + (long long)(2 + 2 * s) * p.PS;
    const int row0 = 32 * w + 4 * kq;
    const bool rows_in = 32 * w + 32 <= p.N;
    const int nunits = p.nunits > 0 ? p.nunits : (p.ncols + 32 * CT - 1) / (32 * CT);
    const int u0 = (int)(((long long)bx * nunits) / nbx), u1 = (int)(((long long)(bx + 1) * nunits) / nbx);
    for (int unit = u0; unit < u1; ++unit) {
        // (state columns only, see Prop2P: the first hop on the input channels - d1 += S^T e2, which the adjacency gradient reads -
        //  is a gathered single-hop launch on the helper stream: prop_mform.h, Prop1P::cstep)
        const int colbase = p.nunits > 0 ? (unit / p.cps) * p.cstride + (unit % p.cps) * 32 * CT : unit * 32 * CT;
        int ld = (int)p.ld;
        asm volatile("" : "+s"(ld));                   // see MCRN_ROW_OF
        if (unit > u0) __syncthreads();
        MCRN_TL(1, 2);
        int nlast = p.N - 1;
        int tidv = tid;
        MCRN_FRESH(ld); MCRN_FRESH(nlast);
        asm volatile("" : "+v"(tidv));
        PB::stage(img, E2, ld, nlast, p.ncols, colbase, tidv);
        MCRN_TL(1, 3);
        // The addends (dP[1+2s] for hop 1, dP[0] for hop 2 of support 0) are loaded STRAIGHT INTO the accumulators as
        // the MFMA chain's initial value: no extra registers, and the loads fly during the barrier that follows.
        auto load_acc = [&](const float* __restrict__ src, f32x16 (&dst)[CT], int ldv) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int col = min(colbase + 32 * t + cperm, p.ncols - 1);
                if (rows_in) {
                    const unsigned o = (unsigned)(row0 * ldv + col);
#pragma unroll
                    for (int v = 0; v < 16; ++v) dst[t][v] = src[o + (unsigned)(MCRN_ROW_OF(v) * ldv)];
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) dst[t][v] = src[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ldv + col)];
                }
            }
        };
        f32x16 acc[CT];
        MCRN_FRESH(ld);
        load_acc(D1, acc, ld);
        __syncthreads();
        MCRN_TL(1, 4);
        PB::template mma<true>(img, ah, al, acc, lane, sfw, ks0);
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 32 * t + cperm;
            const unsigned o = (unsigned)(row0 * ld + col);
            const bool cok = col < p.ncols;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const bool ok = cok && (rows_in || row0 + MCRN_ROW_OF(v) < p.N);
                const float d = ok ? acc[t][v] : 0.f;                 // zero rows / columns that do not exist
                acc[t][v] = d;
                if (ok && !p.no_d1) D1[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = d;   // (no_d1: nobody reads d1t - see Prop2P)
            }
        }
        MCRN_TL(1, 5);
        __syncthreads();
        MCRN_TL(1, 6);
        PB::to_img(img, acc, w, lane);
        MCRN_FRESH(ld);
        if (s == 0) {
            load_acc(D0, acc, ld);
        } else {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        }
        __syncthreads();
        MCRN_TL(1, 7);
        PB::template mma<true>(img, ah, al, acc, lane, sfw, ks0);
        float* __restrict__ OUT = s == 0 ? D0 : EX;
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 32 * t + cperm;
            const unsigned o = (unsigned)(row0 * ld + col);
            if (col < p.ncols) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) OUT[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = acc[t][v];
            }
        }
        MCRN_TL(1, 8);
    }   // unit loop
}
template <int NF, int CT>
__global__ __launch_bounds__(64 * NF) void prop2_bwd_kernel(const Prop2P p) {
    using PB = PropBlock<NF, CT>;
    constexpr int KS = 2 * NF;
    extern __shared__ __attribute__((aligned(16))) uint4 prop2_img[];   // PB::IMG uint4 (up to 132 KB at N = 352: dynamic)
    uint4* const img = prop2_img;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = blockIdx.y;
    uint4 ah[PB::NAL], al[PB::NAL];
    const uint4* __restrict__ sfw0 = p.Sf[s] + (long long)w * KS * 2 * 64 + lane;
    const uint4* __restrict__ sfw = PB::WIDE ? sfw0 : nullptr;
    const int ks0 = PB::WIDE ? (int)((blockIdx.x * 7 + blockIdx.y * 3) % KS) : 0;
    MCRN_TL(1, 0);
    PB::load_a(sfw0, ah, al);
    MCRN_TL(1, 1);
    prop2_bwd_units<NF, CT>(p, p.base, p.extra, img, (int)blockIdx.x, (int)gridDim.x, s, ah, al, sfw, ks0);
#ifdef MCRN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) g_tl[1][(blockIdx.y * gridDim.x + blockIdx.x) & 511][9] = wall_clock64();
#endif
}

#define MCRN_LAUNCH_PROP2(KERN, NF_, CT_, GRID, P)                                                        \
    do {                                                                                                  \
        constexpr size_t lds_ = (size_t)PropBlock<NF_, CT_>::IMG * sizeof(uint4);                         \
        static bool set_ = false;                                                                         \
        if (lds_ > 64 * 1024 && !set_) {                                                                  \
            hipError_t e_ = hipFuncSetAttribute((const void*)KERN<NF_, CT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
            if (e_ != hipSuccess) return e_;                                                              \
            set_ = true;                                                                                  \
        }                                                                                                 \
        if ((P).ev1)   /* (ev1 alone: a completion event for another stream's wait, attached to the dispatch instead of a marker packet) */ \
            hipExtLaunchKernelGGL((KERN<NF_, CT_>), GRID, dim3(64 * NF_), lds_, st, (hipEvent_t)(P).ev0, (hipEvent_t)(P).ev1, 0, P); \
        else                                                                                              \
        hipLaunchKernelGGL((KERN<NF_, CT_>), GRID, dim3(64 * NF_), lds_, st, P);                          \
    } while (0)
#define MCRN_NF_SWITCH(KERN, CT_, GRID, P)                                                   \
    switch (NF) {                                                                            \
        case 1: MCRN_LAUNCH_PROP2(KERN, 1, CT_, GRID, P); break;                             \
        case 2: MCRN_LAUNCH_PROP2(KERN, 2, CT_, GRID, P); break;                             \
        case 3: MCRN_LAUNCH_PROP2(KERN, 3, CT_, GRID, P); break;                             \
        case 4: MCRN_LAUNCH_PROP2(KERN, 4, CT_, GRID, P); break;                             \
        case 5: MCRN_LAUNCH_PROP2(KERN, 5, CT_, GRID, P); break;                             \
        case 6: MCRN_LAUNCH_PROP2(KERN, 6, CT_, GRID, P); break;                             \
        case 7: MCRN_LAUNCH_PROP2(KERN, 7, CT_, GRID, P); break;                             \
        case 8: MCRN_LAUNCH_PROP2(KERN, 8, CT_, GRID, P); break;                             \
        case 9: MCRN_LAUNCH_PROP2(KERN, 9, CT_, GRID, P); break;                             \
        case 10: MCRN_LAUNCH_PROP2(KERN, 10, CT_, GRID, P); break;                           \
        default: MCRN_LAUNCH_PROP2(KERN, 11, CT_, GRID, P); break;                           \
    }
#define MCRN_NF_SWITCH8(KERN, CT_, GRID, P)                                                  \
    switch (NF) {                                                                            \
        case 1: MCRN_LAUNCH_PROP2(KERN, 1, CT_, GRID, P); break;                             \
        case 2: MCRN_LAUNCH_PROP2(KERN, 2, CT_, GRID, P); break;                             \
        case 3: MCRN_LAUNCH_PROP2(KERN, 3, CT_, GRID, P); break;                             \
        case 4: MCRN_LAUNCH_PROP2(KERN, 4, CT_, GRID, P); break;                             \
        case 5: MCRN_LAUNCH_PROP2(KERN, 5, CT_, GRID, P); break;                             \
        case 6: MCRN_LAUNCH_PROP2(KERN, 6, CT_, GRID, P); break;                             \
        case 7: MCRN_LAUNCH_PROP2(KERN, 7, CT_, GRID, P); break;                             \
        default: MCRN_LAUNCH_PROP2(KERN, 8, CT_, GRID, P); break;                            \
    }
// column tiles per workgroup: minimise (rounds over 256 CUs) x (work per workgroup)
static inline int pick_ct(int ncols, int ny) {
    const long long b2 = (long long)((ncols + 63) / 64) * ny, b1 = (long long)((ncols + 31) / 32) * ny;
    const long long t2 = ((b2 + 255) / 256) * 2, t1 = ((b1 + 255) / 256) * 1;
    return t1 < t2 ? 1 : 2;
}
// Unit width and workgroups per support.  Both supports together must be resident at once on the 256 CUs (<= 128
// workgroups each) and a launch costs `passes` unit-times, so: 64-column units when they fit in one pass; else
// 96-column units if THOSE fit in one pass (METR-LA decoder: 136 -> 91 units, ~22 us instead of 2 x 16 us); else
// 64-column units over just enough workgroups for the pass count (the rest of the CUs stay free for the side stream).
static inline void prop2_shape(int ncols, int& ct, int& blocks, int NF) {
    const int u2 = (ncols + 63) / 64, u3 = (ncols + 95) / 96;
    if (u2 > 128 && u3 <= 128 && NF <= 8) { ct = 3; blocks = u3; return; }   // (the wide variant has no registers for 3 tiles)
    const int passes = (u2 + 127) / 128;
    ct = 2; blocks = (u2 + passes - 1) / passes;
}
static inline hipError_t launch_prop2_fwd(const Prop2P& p, hipStream_t st) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    int ct, blocks;
    prop2_shape(p.ncols, ct, blocks, NF);
    if (p.nunits > 0) {             // state columns only: 64-column units by construction
        const int passes = (p.nunits + 127) / 128;
        ct = 2; blocks = (p.nunits + passes - 1) / passes;
    }
    Prop2P q = p;
    q.nblk = blocks;
    dim3 grid(16 * ((blocks + 7) / 8));
    if (ct == 3) { MCRN_NF_SWITCH8(prop2_fwd_kernel, 3, grid, q) }
    else { MCRN_NF_SWITCH(prop2_fwd_kernel, 2, grid, q) }
    return hipGetLastError();
}
static inline hipError_t launch_prop2_bwd(const Prop2P& p, hipStream_t st) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    int ct, blocks;
    prop2_shape(p.ncols, ct, blocks, NF);
    if (p.nunits > 0) {             // state columns only: 64-column units by construction
        const int passes = (p.nunits + 127) / 128;
        ct = 2; blocks = (p.nunits + passes - 1) / passes;
    }
    dim3 grid(blocks, 2);
    if (ct == 3) { MCRN_NF_SWITCH8(prop2_bwd_kernel, 3, grid, p) }
    else { MCRN_NF_SWITCH(prop2_bwd_kernel, 2, grid, p) }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adjacency gradient for small graphs (N <= 256), output-stationary:
//   dS[n][m] += sum_seg sum_k  A_seg[n][k] * B_seg[m][k]      (k = the B*Cp columns of a plane)
// (dS = d1t x0^T + e2 x1^T).  The N x N result lives entirely in accumulators:
// wave w owns output rows 32w.. and all NF column fragments (NF x 16 VGPRs), a workgroup owns one K-chunk
// and adds its partial into its own slab (split-K, reduced later in a fixed order).  Both operands are
// K-contiguous plane rows, so A- and B-fragments are the same load (32 bytes per lane) + bf16 split;
// the B fragments of the NF waves are exchanged through a double-buffered LDS image.  Every converted
// fragment feeds NF x 3 MFMAs (21 at N = 207) instead of 3 in the tiled GEMM: MFMA-bound by design.
// ---------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(64 * NF) void ds_small_kernel(const DsP p) {
    // LDS image of one 32-column panel of both operands, in MFMA fragment order, double-buffered:
    //   img[stage][op][frag j = row/32][ks = (k/16)&1][hi/lo][slot = row%32 + 32*((k/8)&1)] : 8 bf16 (k%8)
    // filled with 8-byte granules (4 consecutive k) by threads that load coalesced float4s: 8 lanes cover
    // one row's 128-byte panel line.  Panel n+1 is published into the other stage while panel n is multiplied:
    // one barrier per panel.
    __shared__ uint4 img[2][2][NF][2][2][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    const int sup = blockIdx.y, z = blockIdx.x;
    const int kbeg = z * p.kchunk;
    const int kend = min(p.ncols, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int npan = (kend - kbeg + 31) >> 5;               // 32-column panels per segment
    const int total = npan * p.nseg;

    // loader role of this thread: rows r0 + 8*NF*i (i < 4), column quad cq (4 floats) of the panel
    const int cq = tid & 7, r0 = tid >> 3;                  // 8*NF rows per pass, 4 passes cover 32*NF rows
    float4 va[4], vb[4];
    auto fetch = [&](int pn) {
        const int seg = pn / npan, pi = pn - seg * npan;
        const int k = kbeg + 32 * pi + 4 * cq;
        const bool kv = k < kend;                           // ncols % 4 == 0: a float4 is all-in or all-out
        const int kc = kv ? k : kbeg;                       // unpredicated loads from an in-range address, zeroed after
        const float* __restrict__ A = p.A[sup][seg];
        const float* __restrict__ B = p.B[sup][seg];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(r0 + 8 * NF * i, p.N - 1);    // clamped rows only feed outputs that are not stored
            va[i] = *reinterpret_cast<const float4*>(A + (long long)r * p.ld + kc);
            vb[i] = *reinterpret_cast<const float4*>(B + (long long)r * p.ld + kc);
        }
        if (!kv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { va[i] = make_float4(0.f, 0.f, 0.f, 0.f); vb[i] = va[i]; }
        }
    };
    auto publish = [&](int stg) {
        uint2* g = reinterpret_cast<uint2*>(&img[stg][0][0][0][0][0]);
        const int ks = cq >> 2, kq8 = (cq >> 1) & 1, half = cq & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 8 * NF * i;
            const int fj = r >> 5, slot = (r & 31) + 32 * kq8;
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                const float4 x = op ? vb[i] : va[i];
                const unsigned h01 = cvt_pk_bf16(x.x, x.y), h23 = cvt_pk_bf16(x.z, x.w);
                const unsigned l01 = cvt_pk_bf16(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xFFFF0000u));
                const unsigned l23 = cvt_pk_bf16(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xFFFF0000u));
                const int base = (((op * NF + fj) * 2 + ks) * 2) * 64;          // uint4 index of [op][fj][ks][hi][0]
                // slot permutation inside the 64-slot array: the 8-byte stores of the four (ks, k-half) lanes of one
                // row would otherwise fall on the same banks (sub-array strides are multiples of 128 bytes): 4-way
                // conflicts, 1.07 M conflict cycles per launch (profiles/r1).  The readers apply the same XOR.
                const int ps = slot ^ (kq8 << 1) ^ (ks << 2);
                g[(base + ps) * 2 + half] = make_uint2(h01, h23);
                g[(base + 64 + ps) * 2 + half] = make_uint2(l01, l23);
            }
        }
    };
#ifdef MCRN_TIMELINE
    unsigned long long tl_acc[5] = {0, 0, 0, 0, 0}, tl_t = 0;
#define MCRN_TLA(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); tl_acc[i] += n_ - tl_t; tl_t = n_; } } while (0)
    MCRN_TL(2, 0);
    if (threadIdx.x == 0) tl_t = wall_clock64();
#else
#define MCRN_TLA(i)
#endif
    fetch(0);
    // The slab already holds the partial sums of earlier launches: they are loaded STRAIGHT INTO the accumulators
    // as the MFMA chains' initial value and the result is written back with plain stores.  This element of this
    // slab has exactly one writer (this workgroup) per launch and launches are stream-ordered, so no atomics are
    // needed (the former 112 global_atomic_add_f32 per lane cost 10 of 27 us per workgroup) and the result stays
    // deterministic.  Clamped addresses, no predicate: out-of-range elements are never stored.
    float* __restrict__ C = p.C[sup] + (long long)z * p.slab;
    f32x16 acc[NF];
    {
        const int rows_in = 32 * w + 32 <= p.N;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int c = min(32 * j + l31, p.N - 1);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * w + (v & 3) + 8 * (v >> 2) + 4 * kq;
                acc[j][v] = C[(long long)(rows_in ? r : min(r, p.N - 1)) * p.ldc + c];
            }
        }
    }
    publish(0);
    if (total > 1) fetch(1);
    __syncthreads();
    for (int pn = 0; pn < total; ++pn) {
        const int stg = pn & 1;
        MCRN_TLA(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pl = lane ^ (kq << 1) ^ (ks << 2);              // see publish()
            const bf16x8 xh = __builtin_bit_cast(bf16x8, img[stg][0][w][ks][0][pl]);
            const bf16x8 xl = __builtin_bit_cast(bf16x8, img[stg][0][w][ks][1][pl]);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, img[stg][1][j][ks][0][pl]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, img[stg][1][j][ks][1][pl]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc[j], 0, 0, 0);
            }
        }
        MCRN_TLA(3);
        if (pn + 1 < total) {
            publish(stg ^ 1);                                // panel pn+1 (in registers) -> the other stage
            MCRN_TLA(1);
            if (pn + 2 < total) fetch(pn + 2);               // panel pn+2's loads fly during the next MFMA block
        }
        MCRN_TLA(2);
        __syncthreads();                                     // stage stg free again, stage stg^1 visible
        MCRN_TLA(4);
    }
    MCRN_TL(2, 1);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int c = 32 * j + l31;
        if (c >= p.N) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * w + (v & 3) + 8 * (v >> 2) + 4 * kq;
            if (r < p.N) C[(long long)r * p.ldc + c] = acc[j][v];
        }
    }
#ifdef MCRN_TIMELINE
    MCRN_TL(2, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        unsigned long long* o = g_tl[2][(blockIdx.y * gridDim.x + blockIdx.x) & 511];
        o[3] = wall_clock64();
        for (int i = 0; i < 5; ++i) o[4 + i] = tl_acc[i];
        o[9] = (unsigned long long)total;
    }
#endif
#undef MCRN_TLA
}
// The same output-stationary adjacency gradient for 256 < N <= 352 (NF = 9 .. 11: PEMS-BAY).  The N x N block no longer fits
// one workgroup's accumulators (NF x NF fragments = 121 x 16 VGPRs over 11 waves, and the two-operand image of a panel would
// be 176 KB of LDS), so a workgroup owns NJ = 4 of the NF column fragments (blockIdx.z = column group): wave w accumulates
// the fragments (w, j0 .. j0 + NJ - 1) in 64 VGPRs, the LDS image holds all NF fragments of the A panel and the NJ fragments
// of the B panel it needs ((NF + NJ) x 8 KB, double-buffered: 120 KB at NF = 11).  Every thread still fetches its rows of
// BOTH operands (the rows of B outside the group are not published: their loads hit the lines the group's neighbours use);
// slab layout, split-K over column chunks, addend-in-accumulator and the one-writer rule are those of ds_small_kernel.
template <int NF, int NJ>
__global__ __launch_bounds__(64 * NF) void ds_wide_kernel(const DsP p) {
    __shared__ uint4 img[2][NF + NJ][2][2][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    const int sup = blockIdx.y, z = blockIdx.x;
    const int j0 = blockIdx.z * NJ;                         // first column fragment of this workgroup
    const int kbeg = z * p.kchunk;
    const int kend = min(p.ncols, kbeg + p.kchunk);
    if (kbeg >= kend) return;
    const int npan = (kend - kbeg + 31) >> 5;
    const int total = npan * p.nseg;
    const int cq = tid & 7, r0 = tid >> 3;
    float4 va[4], vb[4];
    auto fetch = [&](int pn) {
        const int seg = pn / npan, pi = pn - seg * npan;
        const int k = kbeg + 32 * pi + 4 * cq;
        const bool kv = k < kend;
        const int kc = kv ? k : kbeg;
        const float* __restrict__ A = sup == 0 ? p.A[0][seg] : sup == 1 ? p.A[1][seg] : sup == 2 ? p.A[2][seg] : p.A[3][seg];
        const float* __restrict__ B = sup == 0 ? p.B[0][seg] : sup == 1 ? p.B[1][seg] : sup == 2 ? p.B[2][seg] : p.B[3][seg];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(r0 + 8 * NF * i, p.N - 1);
            va[i] = *reinterpret_cast<const float4*>(A + (long long)r * p.ld + kc);
            vb[i] = *reinterpret_cast<const float4*>(B + (long long)r * p.ld + kc);
        }
        if (!kv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { va[i] = make_float4(0.f, 0.f, 0.f, 0.f); vb[i] = va[i]; }
        }
    };
    auto publish = [&](int stg) {
        uint2* g = reinterpret_cast<uint2*>(&img[stg][0][0][0][0]);
        const int ks = cq >> 2, kq8 = (cq >> 1) & 1, half = cq & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 8 * NF * i;
            const int fj = r >> 5, slot = (r & 31) + 32 * kq8;
            const int ps = slot ^ (kq8 << 1) ^ (ks << 2);    // bank-conflict-free slot order (see ds_small_kernel)
#pragma unroll
            for (int op = 0; op < 2; ++op) {
                const int fi = op ? NF + fj - j0 : fj;       // fragment slot inside the image
                if (op && (fj < j0 || fj >= j0 + NJ)) continue;
                const float4 x = op ? vb[i] : va[i];
                const unsigned h01 = cvt_pk_bf16(x.x, x.y), h23 = cvt_pk_bf16(x.z, x.w);
                const unsigned l01 = cvt_pk_bf16(x.x - __uint_as_float(h01 << 16), x.y - __uint_as_float(h01 & 0xFFFF0000u));
                const unsigned l23 = cvt_pk_bf16(x.z - __uint_as_float(h23 << 16), x.w - __uint_as_float(h23 & 0xFFFF0000u));
                const int base = ((fi * 2 + ks) * 2) * 64;                        // uint4 index of [fi][ks][hi][0]
                g[(base + ps) * 2 + half] = make_uint2(h01, h23);
                g[(base + 64 + ps) * 2 + half] = make_uint2(l01, l23);
            }
        }
    };
    fetch(0);
    float* __restrict__ C = (sup == 0 ? p.C[0] : sup == 1 ? p.C[1] : sup == 2 ? p.C[2] : p.C[3]) + (long long)z * p.slab;
    const int nj = min(NJ, NF - j0);                        // the last column group is ragged (NF = 9, 10, 11 over groups of 4): its
                                                            // missing fragments are never published and must not be multiplied
    f32x16 acc[NJ];
    {
        const int rows_in = 32 * w + 32 <= p.N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = min(32 * (j0 + j) + l31, p.N - 1);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 * w + (v & 3) + 8 * (v >> 2) + 4 * kq;
                acc[j][v] = C[(long long)(rows_in ? r : min(r, p.N - 1)) * p.ldc + c];
            }
        }
    }
    publish(0);
    if (total > 1) fetch(1);
    __syncthreads();
    for (int pn = 0; pn < total; ++pn) {
        const int stg = pn & 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int pl = lane ^ (kq << 1) ^ (ks << 2);
            const bf16x8 xh = __builtin_bit_cast(bf16x8, img[stg][w][ks][0][pl]);
            const bf16x8 xl = __builtin_bit_cast(bf16x8, img[stg][w][ks][1][pl]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j >= nj) break;                          // block-uniform
                const bf16x8 bh = __builtin_bit_cast(bf16x8, img[stg][NF + j][ks][0][pl]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, img[stg][NF + j][ks][1][pl]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, acc[j], 0, 0, 0);
            }
        }
        if (pn + 1 < total) {
            publish(stg ^ 1);
            if (pn + 2 < total) fetch(pn + 2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 32 * (j0 + j) + l31;
        if (c >= p.N) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = 32 * w + (v & 3) + 8 * (v >> 2) + 4 * kq;
            if (r < p.N) C[(long long)r * p.ldc + c] = acc[j][v];
        }
    }
}
static const int DS_WIDE_NJ = 4;
static inline hipError_t launch_ds_small(DsP p, int nslab, hipStream_t st, int nblk = 2) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    int kc = ((p.ncols + nslab - 1) / nslab + 31) / 32 * 32;
    if (kc < 64) kc = 64;
    p.kchunk = kc;
    dim3 grid((p.ncols + kc - 1) / kc, nblk);
    switch (NF) {
        case 1: hipLaunchKernelGGL(ds_small_kernel<1>, grid, dim3(64), 0, st, p); break;
        case 2: hipLaunchKernelGGL(ds_small_kernel<2>, grid, dim3(128), 0, st, p); break;
        case 3: hipLaunchKernelGGL(ds_small_kernel<3>, grid, dim3(192), 0, st, p); break;
        case 4: hipLaunchKernelGGL(ds_small_kernel<4>, grid, dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL(ds_small_kernel<5>, grid, dim3(320), 0, st, p); break;
        case 6: hipLaunchKernelGGL(ds_small_kernel<6>, grid, dim3(384), 0, st, p); break;
        case 7: hipLaunchKernelGGL(ds_small_kernel<7>, grid, dim3(448), 0, st, p); break;
        case 8: hipLaunchKernelGGL(ds_small_kernel<8>, grid, dim3(512), 0, st, p); break;
        // 256 < N <= 352: column groups of DS_WIDE_NJ fragments (grid.z)
        case 9: grid.z = (9 + DS_WIDE_NJ - 1) / DS_WIDE_NJ; hipLaunchKernelGGL((ds_wide_kernel<9, DS_WIDE_NJ>), grid, dim3(576), 0, st, p); break;
        case 10: grid.z = (10 + DS_WIDE_NJ - 1) / DS_WIDE_NJ; hipLaunchKernelGGL((ds_wide_kernel<10, DS_WIDE_NJ>), grid, dim3(640), 0, st, p); break;
        case 11: grid.z = (11 + DS_WIDE_NJ - 1) / DS_WIDE_NJ; hipLaunchKernelGGL((ds_wide_kernel<11, DS_WIDE_NJ>), grid, dim3(704), 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

static inline hipError_t launch_sfrag(const float* S, long long ldS, int N, int transpose, uint4* out, hipStream_t st) {
    const int NF = (N + 31) / 32;
    const int tot = NF * 2 * NF * 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_sfrag_build, dim3((tot + 255) / 256), dim3(256), 0, st, S, ldS, N, NF, transpose, out);
    return hipGetLastError();
}
static inline hipError_t launch_prop_small(const PropP& p, int nbatch, hipStream_t st) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    dim3 grid((p.ncols + 63) / 64, nbatch);
    switch (NF) {
        case 1: hipLaunchKernelGGL(prop_small_kernel<1>, grid, dim3(64), 0, st, p); break;
        case 2: hipLaunchKernelGGL(prop_small_kernel<2>, grid, dim3(128), 0, st, p); break;
        case 3: hipLaunchKernelGGL(prop_small_kernel<3>, grid, dim3(192), 0, st, p); break;
        case 4: hipLaunchKernelGGL(prop_small_kernel<4>, grid, dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL(prop_small_kernel<5>, grid, dim3(320), 0, st, p); break;
        case 6: hipLaunchKernelGGL(prop_small_kernel<6>, grid, dim3(384), 0, st, p); break;
        case 7: hipLaunchKernelGGL(prop_small_kernel<7>, grid, dim3(448), 0, st, p); break;
        default: hipLaunchKernelGGL(prop_small_kernel<8>, grid, dim3(512), 0, st, p); break;
    }
    return hipGetLastError();
}

}  // namespace mcrn
