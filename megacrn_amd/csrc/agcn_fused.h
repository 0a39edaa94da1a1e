// agcn_fused.h - one AGCN call of a small graph (N <= 256) as ONE launch: both Chebyshev hops of both supports AND the
// weight pool with its GRU epilogue (model/MegaCRN.py:16-28 and :42-47), gfx950, bf16x3 arithmetic (the 1e-4 parity mode).
//
// Why: at N = 207 a train step was ~360 dependent launches of 15-35 us.  The forward of one AGCN call was two of them -
// the fused two-hop propagation (prop_small.h) WRITES four fp32 planes (14 / 28 MB), the weight pool READS all five back
// (18 / 35 MB) - each a serial chain load -> convert -> barrier -> MFMA -> store on a quarter of the CUs, with the planes'
// write-back sitting between them.  Here the workgroup that owns the 64 state channels of one sample multiplies every plane
// into the weight pool while the plane is still in its accumulators:
//   * workgroup = (sample b, block of 32*NBF output columns); NF waves, wave w owns node rows 32w .. 32w+31 throughout;
//   * per 64-channel chunk of the state and per support: stage X0 -> LDS image, hop 1 (S in registers), hop 2 - exactly the
//     pipeline of prop2_fwd_kernel (PropBlock) - and after each hop the accumulator tile (C/D layout: lane = channel,
//     registers = 16 node rows) goes through a wave-private LDS tile of [4 channel][16 node] blocks from which
//     ds_read_b64_tr_b16 returns MFMA A fragments (row = node, 8 consecutive channels): the weight-pool MFMAs need no
//     barrier, no global round trip;
//   * the weight slabs (64 channels x 32*NBF outputs, bf16 hi/lo in B-fragment order, the image of wp_stream.h) travel by
//     LDS-DMA into a ring of three LDS buffers, one phase ahead of their use;
//   * the planes are still written (the backward pass reads them: weight gradient, adjacency gradient) - by the workgroups
//     of output block 0 only - but nobody waits for them;
//   * the input channels (d = 1 .. 3 per plane) are not part of the per-step propagation at all: they are propagated once per
//     stack for every step (hoisted) and enter as one 16-deep k-step gathered from the planes.
// The redundant propagation of the 2 .. 4 workgroups that share a sample is free: 64 samples alone would leave 3/4 of the
// CUs idle.
#pragma once
#include "prop_small.h"

namespace mcrn {

namespace agf {
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
// LDS-DMA of 16 bytes per lane (see glds16 in gemm_bf16.h): LDS[lds_dst + 16*lane] = *(sbase + voff)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
}  // namespace agf

// wave-private transpose tile: 32 channels x 32 nodes (bf16), rows of 4 channels = 2 blocks [4 ch][16 node] of 128 B, padded
static constexpr int AGF_ROW = 256 + 16;                 // bytes per 4-channel row (pad: the 8 rows start in different banks)
static constexpr int AGF_TILE = 8 * AGF_ROW;             // one 32 x 32 tile (hi or lo)

template <int NF, int NBF, int EPI>
__global__ __launch_bounds__(64 * NF) void agcn_fwd_kernel(const AgcnFP p) {
    using PB = PropBlock<NF, 2>;
    static_assert(!PB::WIDE, "S register-stationary (N <= 256)");
    constexpr int KS = 2 * NF;
    constexpr int SLAB = 4 * NBF * 2 * 1024;             // weight slab of one (plane, 64-channel chunk): 4 k-steps
    extern __shared__ __attribute__((aligned(16))) uint4 agf_lds[];
    uint4* const img = agf_lds;                                                        // PB::IMG uint4
    unsigned char* const trb = reinterpret_cast<unsigned char*>(agf_lds + PB::IMG);    // NF waves x (hi | lo) tiles
    unsigned char* const slabs = trb + NF * 2 * AGF_TILE;                              // 3 weight slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kq = lane >> 5;
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
    const int b = blockIdx.x, cb = blockIdx.y;
    const bool writer = cb == 0;                                                      // this workgroup also stores the planes
    const int KSH = p.H / 16, KSt = 5 * KSH + 1, nchunk = p.H / 64;
    const unsigned lds_slab0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)slabs);
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wimg) + (long long)cb * KSt * NBF * 2 * 1024;
    unsigned char* const mytr = trb + w * 2 * AGF_TILE;
    const int row0 = 32 * w + 4 * kq;
    const bool rows_in = 32 * w + 32 <= p.N;
    const int nlast_c = p.N - 1;

    f32x16 wacc[NBF];
#pragma unroll
    for (int j = 0; j < NBF; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) wacc[j][v] = 0.f;

    // weight slab of k-steps [ks0, ks0 + nks) -> LDS slab buffer `buf` (pieces of 1 KB dealt to the waves)
    auto slab_issue = [&](int ks0, int nks, int buf) {
        const int npiece = nks * NBF * 2;
        const unsigned char* src = wsrc + (long long)ks0 * NBF * 2 * 1024;
        for (int piece = w; piece < npiece; piece += NF) {
            const unsigned long long a = (unsigned long long)(uintptr_t)(src + (long long)piece * 1024);
            const unsigned alo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
            const unsigned ahi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_slab0 + buf * SLAB + piece * 1024));
            agf::glds16(reinterpret_cast<const void*>(((unsigned long long)ahi << 32) | alo), (unsigned)lane * 16u, dst);
        }
    };
    // accumulate  wacc += tile values (C/D layout, scaled) x W slab : the two 32-channel tiles of a chunk, 2 k-steps each
    auto wp_tiles = [&](const f32x16 (&val)[2], float scale, int buf) {
        const uint4* sl = reinterpret_cast<const uint4*>(slabs + buf * SLAB);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // accumulators -> wave-private tile: lane = channel cperm, register group gq = 4 consecutive nodes
            const int cq = cperm >> 2, c4 = cperm & 3;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int m0 = 4 * kq + 8 * gq;                      // first of 4 consecutive node rows (within the wave's 32)
                const float a0 = scale * val[t][4 * gq], a1 = scale * val[t][4 * gq + 1], a2 = scale * val[t][4 * gq + 2], a3 = scale * val[t][4 * gq + 3];
                const unsigned h01 = cvt_pk_bf16(a0, a1), h23 = cvt_pk_bf16(a2, a3);
                const unsigned l01 = cvt_pk_bf16(a0 - __uint_as_float(h01 << 16), a1 - __uint_as_float(h01 & 0xFFFF0000u));
                const unsigned l23 = cvt_pk_bf16(a2 - __uint_as_float(h23 << 16), a3 - __uint_as_float(h23 & 0xFFFF0000u));
                const int off = cq * AGF_ROW + (m0 >> 4) * 128 + c4 * 32 + (m0 & 15) * 2;
                *reinterpret_cast<uint2*>(mytr + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(mytr + AGF_TILE + off) = make_uint2(l01, l23);
            }
            // (same wave wrote and reads: LDS operations of a wave complete in order)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int i = lane & 15, nhalf = (lane >> 4) & 1;
                const int o0 = (4 * ks + 2 * kq) * AGF_ROW + nhalf * 128 + (i >> 2) * 32 + (i & 3) * 8;
                const agf::s16x4_t h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agf::s16x4_t*)(mytr + o0));
                const agf::s16x4_t h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agf::s16x4_t*)(mytr + o0 + AGF_ROW));
                const agf::s16x4_t g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agf::s16x4_t*)(mytr + AGF_TILE + o0));
                const agf::s16x4_t g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) agf::s16x4_t*)(mytr + AGF_TILE + o0 + AGF_ROW));
                const agf::s16x8_t ahv = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                const agf::s16x8_t alv = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const bf16x8 ah = __builtin_bit_cast(bf16x8, ahv), al = __builtin_bit_cast(bf16x8, alv);
                const int kk = 2 * t + ks;                            // k-step inside the slab
#pragma unroll
                for (int j = 0; j < NBF; ++j) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, sl[((kk * NBF + j) * 2 + 0) * 64 + lane]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, sl[((kk * NBF + j) * 2 + 1) * 64 + lane]);
                    wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, wacc[j], 0, 0, 0);
                    wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, wacc[j], 0, 0, 0);
                    wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, wacc[j], 0, 0, 0);
                }
            }
        }
    };

    // ---- input channels of all 5 planes: one k-step (rows = this wave's nodes of sample b), weights straight from L2
    {
        const int nin = 5 * p.d;
        const long long node = min(32 * w + l31, nlast_c);
        float ain[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = 8 * kq + i;
            const int qc = q < nin ? q : 0;
            const int g = qc / p.d, j = qc - g * p.d;
            const float x = p.Z[(long long)g * p.PS + node * p.ld + (long long)b * p.Cp + p.H + j];
            ain[i] = q < nin ? x : 0.f;
        }
        uint4 h4, l4;
        split8(ain, h4, l4);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h4), al = __builtin_bit_cast(bf16x8, l4);
        const uint4* wi = reinterpret_cast<const uint4*>(wsrc + (long long)(5 * KSH) * NBF * 2 * 1024);
#pragma unroll
        for (int j = 0; j < NBF; ++j) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, wi[(j * 2 + 0) * 64 + lane]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, wi[(j * 2 + 1) * 64 + lane]);
            wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, wacc[j], 0, 0, 0);
            wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, wacc[j], 0, 0, 0);
            wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, wacc[j], 0, 0, 0);
        }
    }

    const float* __restrict__ X0 = p.Z;
    int sbuf = 0;                                                     // slab ring: buffer the NEXT slab goes to
    for (int s = 0; s < 2; ++s) {
        // the adjacency fragments of a support are loaded ONCE per workgroup (every workgroup streams the same 0.5 MB image:
        // reloading it per chunk multiplied the L2 traffic of the launch)
        uint4 ah[PB::NAL], al[PB::NAL];
        PB::load_a(p.Sf[s] + (long long)w * KS * 2 * 64 + lane, ah, al);
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            const int colbase = b * p.Cp + 64 * chunk;                // first column of the chunk inside a plane row
            float* __restrict__ X1 = p.Z + (long long)(1 + 2 * s) * p.PS;
            float* __restrict__ X2 = p.Z + (long long)(2 + 2 * s) * p.PS;
            const int b1 = sbuf;                                      // slab of plane 1 + 2s
            slab_issue((1 + 2 * s) * KSH + 4 * chunk, 4, b1);
            sbuf = sbuf == 2 ? 0 : sbuf + 1;
            int ld = (int)p.ld;
            asm volatile("" : "+s"(ld));
            int nlast = p.N - 1, tidv = tid;
            MCRN_FRESH(ld); MCRN_FRESH(nlast);
            asm volatile("" : "+v"(tidv));
            __syncthreads();                                          // previous phase done with img
            PB::stage(img, X0, ld, nlast, (int)p.ld, colbase, tidv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                          // X0 image complete; slab b1 landed
            f32x16 acc[2];
            PB::mma(img, ah, al, acc, lane, nullptr, 0);
            if (writer) {
                MCRN_FRESH(ld);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned o = (unsigned)(row0 * ld + colbase + 32 * t + cperm);
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) X1[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = acc[t][v];
                }
            }
            __syncthreads();                                          // every wave finished reading the hop-1 image
            PB::to_img(img, acc, w, lane);
            wp_tiles(acc, 1.f, b1);
            // next slab: plane 0 (first support only: its accumulators hold X0 below), else plane 2 + 2s
            const int b2 = sbuf;
            slab_issue((s == 0 ? 0 : 2 + 2 * s) * KSH + 4 * chunk, 4, b2);
            sbuf = sbuf == 2 ? 0 : sbuf + 1;
            MCRN_FRESH(ld);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = colbase + 32 * t + cperm;
                if (rows_in) {
                    const unsigned o = (unsigned)(row0 * ld + col);
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][v] = X0[o + (unsigned)(MCRN_ROW_OF(v) * ld)];
                } else {
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][v] = X0[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ld + col)];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                          // hop-2 image complete; slab b2 landed
            int b3 = b2;
            if (s == 0) {
                if (!rows_in) {                                       // rows beyond N were loaded from a clamped row: they are not data
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int v = 0; v < 16; ++v) if (row0 + MCRN_ROW_OF(v) >= p.N) acc[t][v] = 0.f;
                }
                wp_tiles(acc, 1.f, b2);                               // plane 0: the state itself
                b3 = sbuf;
                slab_issue(2 * KSH + 4 * chunk, 4, b3);               // plane 2 (T2 of the first support)
                sbuf = sbuf == 2 ? 0 : sbuf + 1;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] *= -0.5f;
            PB::template mma<true>(img, ah, al, acc, lane, nullptr, 0);
            if (writer) {
                MCRN_FRESH(ld);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned o = (unsigned)(row0 * ld + colbase + 32 * t + cperm);
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) X2[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = 2.f * acc[t][v];
                }
            }
            if (s == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                      // slab b3 landed
            }
            wp_tiles(acc, 2.f, b3);
        }
    }

    // ---- fused GRU epilogue (as in wp_stream.h).  C/D layout: column = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
    const int H = p.H;
#pragma unroll
    for (int j = 0; j < NBF; ++j) {
        const int c = (cb * NBF + j) * 32 + l31;
        const float bj = p.bias[c];
        if (EPI == AGF_GATE) {
            const bool isz = c < H;
            const int ch = isz ? c : 0;
            float hv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long n = min(row0 + MCRN_ROW_OF(v), p.N - 1);
                hv[v] = p.Z[n * p.ld + (long long)b * p.Cp + ch];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = row0 + MCRN_ROW_OF(v);
                const long long r = (long long)n * p.B + b;
                const float g = 1.f / (1.f + expf(-(wacc[j][v] + bj)));
                if (rows_in || n < p.N) {
                    p.out[r * (2 * H) + c] = g;
                    if (isz) p.out2[r * p.out2_ld + c] = g * hv[v];
                }
            }
        } else {
            float hv[16], rg[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long n = min(row0 + MCRN_ROW_OF(v), p.N - 1);
                const long long r = n * p.B + b;
                hv[v] = p.hsrc[r * p.hsrc_ld + c];
                rg[v] = p.zr[r * (2 * H) + H + c];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int n = row0 + MCRN_ROW_OF(v);
                const long long r = (long long)n * p.B + b;
                const float hc = tanhf(wacc[j][v] + bj);
                if (rows_in || n < p.N) {
                    p.out[r * H + c] = hc;
                    p.out2[r * p.out2_ld + c] = rg[v] * hv[v] + (1.f - rg[v]) * hc;
                }
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static inline bool agcn_fused_ok_(int N, int H, int d, int O, long long ld, int Cp) {
    return N <= 256 && (H == 64 || H == 128) && d >= 1 && 5 * d <= 16 && O % 64 == 0 && (ld % 4) == 0 && (Cp % 4) == 0;
}
// output columns per workgroup: 64 (NBF = 2) keeps the kernel inside 256 VGPRs next to the register-resident adjacency
static inline int agcn_fused_nbf_(int O) { return O % 64 == 0 ? 2 : 1; }
template <int NF, int EPI>
static inline hipError_t launch_agcn_one(const AgcnFP& p, hipStream_t st) {
    constexpr int NBF = 2;
    using PB = PropBlock<NF, 2>;
    constexpr size_t lds = (size_t)PB::IMG * 16 + (size_t)NF * 2 * AGF_TILE + (size_t)3 * 4 * NBF * 2 * 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)agcn_fwd_kernel<NF, NBF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((agcn_fwd_kernel<NF, NBF, EPI>), dim3(p.B, p.O / (32 * NBF)), dim3(64 * NF), lds, st, p);
    return hipGetLastError();
}
template <int EPI>
static inline hipError_t launch_agcn_nf(const AgcnFP& p, hipStream_t st) {
    switch ((p.N + 31) / 32) {
        case 1: return launch_agcn_one<1, EPI>(p, st);
        case 2: return launch_agcn_one<2, EPI>(p, st);
        case 3: return launch_agcn_one<3, EPI>(p, st);
        case 4: return launch_agcn_one<4, EPI>(p, st);
        case 5: return launch_agcn_one<5, EPI>(p, st);
        case 6: return launch_agcn_one<6, EPI>(p, st);
        case 7: return launch_agcn_one<7, EPI>(p, st);
        default: return launch_agcn_one<8, EPI>(p, st);
    }
}
static inline hipError_t launch_agcn_fused(const AgcnFP& p, hipStream_t st) {
    if (!agcn_fused_ok_(p.N, p.H, p.d, p.O, p.ld, p.Cp)) return hipErrorInvalidValue;
    if ((((uintptr_t)p.Z) | ((uintptr_t)p.Wimg)) & 15) return hipErrorInvalidValue;
    if (p.epi == AGF_GATE ? p.O != 2 * p.H : p.O != p.H) return hipErrorInvalidValue;
    return p.epi == AGF_GATE ? launch_agcn_nf<AGF_GATE>(p, st) : launch_agcn_nf<AGF_UPDATE>(p, st);
}

}  // namespace mcrn
