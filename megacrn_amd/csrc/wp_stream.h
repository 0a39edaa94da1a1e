// wp_stream.h - weight pool of the large-graph (MCRN_BF16, hoisted) path as a STREAMING row-block kernel (gfx950).
//
//   out[r][o] = epi( sum_g sum_c P_g[r][c] * W_g[c][o] + b[o] )          (model/MegaCRN.py:26-27, GRU algebra :43-47)
//
// r = (node, sample) runs over N*B = 59 000 ... 262 000 rows, the contraction is only G*(H + d) = 165 ... 660 deep and the
// output 32 ... 256 wide: every operand byte is needed once, so the floor is the HBM read of the propagated planes.
// The tiled GEMM this replaces in the bf16 mode (gemm_bf16x3.h, role 2) staged fp32 planes through LDS tile by tile and
// ran at 60-90 TF / 1.2 TB/s (profiles/r2).  Here:
//   * the planes 1 .. nb are bf16-RESIDENT, written by the propagation GEMM itself ([nb][N*B][H], exactly H channels
//     wide: the per-step propagation covers the state channels only).  A lane's MFMA A fragment (8
//     consecutive channels of one row) is ONE 16-byte load; a wave issues every load of its 32 rows up front
//     (<= ~100 VGPRs) - no LDS staging, no conversion, maximal memory-level parallelism;
//   * plane 0 (the state itself) stays fp32 and is split on the fly into bf16 hi + lo: the identity term keeps the
//     library's fp32-equivalent arithmetic (3 MFMAs per product);
//   * the d input channels of all G planes (raw inputs + their hoisted propagation, fp32) form one extra 16-deep k-step;
//   * the weights are pre-split ONCE per step into bf16 hi/lo images in MFMA B-fragment order (k_wp_img_build) and travel
//     plane by plane through a double-buffered LDS slab by LDS-DMA (a column block of NB <= 128 outputs): bf16 planes x
//     (W_hi + W_lo) = 2 MFMAs;
//   * the same kernel serves the small graphs in the bf16x3 (1e-4 parity) mode with fp32 planes 1 .. nb (PBF16 = false):
//     there the tiled GEMM spent its time staging 128-row tiles of a 340 .. 660-deep, 64 .. 256-wide product;
//   * the GRU epilogues are fused and ALSO emit the packed bf16 [N][B*H] operand of the next propagation GEMM
//     (gate: z*h -> the candidate call; update: h' -> the next step's gate call): no separate pack pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "wp_stream_api.h"

namespace mcrn {
namespace wps {
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 8 fp32 -> bf16 hi (8 x 2 bytes) + bf16 lo of the remainders
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
        l[i] = pk_bf16(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xFFFF0000u));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] = *(sbase + voff)  (see glds16 in gemm_bf16.h: inline asm, M0 saved)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
}  // namespace wps

// ---- weight image ------------------------------------------------------------------------------------------------
// k index list of the contraction (KS_tot = G*KSH + 1 k-steps of 16):
//   [0, G*H)          plane g = k / H, state channel c = k % H                    -> Wf row g*Cp + c
//   [G*H, G*H + 16)   input group q = k - G*H: plane g = q / d, input channel j = q % d (q < G*d) -> Wf row g*Cp + H + j
// img[((cb*KS_tot + ks)*NBF + jf)*2 + hl][lane] = 8 bf16 (hi | lo) of W[k = 16 ks + 8 (lane >> 5) + 0..7][o = cb*32*NBF + 32 jf + (lane & 31)]
__global__ void k_wp_img_build(const float* __restrict__ Wf, int Cp, int H, int d, int G, int O, int NBF, uint4* __restrict__ img,
                               long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long q = idx >> 6;
    const int jf = (int)(q % NBF); q /= NBF;
    const int KSt = G * (H / 16) + 1;
    const int ks = (int)(q % KSt);
    const int cb = (int)(q / KSt);
    const int o = (cb * NBF + jf) * 32 + (lane & 31);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 16 * ks + 8 * (lane >> 5) + i;
        int row = -1;
        if (k < G * H) row = (k / H) * Cp + (k % H);
        else { const int qq = k - G * H; if (qq < G * d) row = (qq / d) * Cp + H + (qq % d); }
        v[i] = (row >= 0 && o < O) ? Wf[(long long)row * O + o] : 0.f;
    }
    uint4 h, l;
    wps::split8(v, h, l);
    const long long base = (((long long)cb * KSt + ks) * NBF + jf) * 2;
    img[(base + 0) * 64 + lane] = h;
    img[(base + 1) * 64 + lane] = l;
}

// ---- the kernel --------------------------------------------------------------------------------------------------
// 256 threads = 4 waves; wave w owns rows [128 blockIdx.x + 32 w, +32) and the NBF column fragments of column block
// blockIdx.y.  KSH = H / 16, NBP = number of propagated planes (2 (K - 1)), EPI = WP_GATE / WP_UPDATE,
// PBF16: planes 1 .. NBP are bf16-resident (p.Pb; MCRN_BF16 mode) - otherwise fp32 planes of the plane set (bf16x3 mode).
// The contraction is walked STAGE by stage (stage g < G = plane g, KSH k-steps; stage G = the input group, one k-step):
//     wait (everything issued has landed) ; barrier ; issue stage g+1 (weight slab by LDS-DMA into the other LDS buffer,
//     operand rows into the other register set) ; multiply stage g
// so the loads of a stage fly during the MFMAs of the previous one, with ONE barrier per stage: the barrier that publishes
// slab g is also the one after which nobody reads the buffer slab g+1 goes to.
template <int KSH, int NBP, int NBF, int EPI, bool PBF16>
__global__ __launch_bounds__(256) void wp_stream_kernel(const WpP p) {
    using namespace wps;
    constexpr int G = NBP + 1, H = 16 * KSH;
    constexpr int SLAB = KSH * NBF * 2 * 1024;                              // bytes of one plane's weight slab
    extern __shared__ __attribute__((aligned(16))) uint4 wp_lds[];          // 2 buffers of SLAB bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kq = lane >> 5;
    // The column blocks of a 128-row block run on ONE XCD (block b runs on XCD b % 8): they read the same operand rows - once from the
    // memory side, then from that XCD's L2 (round 5, the d-grad's lesson: profiles/r5/experiments.md section 4).  1-D grid of
    // ceil(row blocks / 8) * 8 * column blocks.
    const int xcd = (int)(blockIdx.x & 7);
    const long long slot = blockIdx.x >> 3;
    // (a division by a run-time value is VALU work: its uniform result is moved back to SGPRs HERE, far from the first LDS-DMA issue -
    //  left in VGPRs, every slab address below would be a v_readfirstlane 3 - 4 wait states before the asm load that reads it, and a
    //  VMEM instruction needs 5 after a VALU write of its SGPR base: tools/isa_hazards.py found exactly that in round 5)
    const int cb = __builtin_amdgcn_readfirstlane((int)(slot % p.ncb));
    const long long rbk = (long long)__builtin_amdgcn_readfirstlane((int)(slot / p.ncb)) * 8 + xcd;
    if (rbk * 128 >= p.R) return;                                           // (padding of the last round; uniform over the workgroup)
    const long long r0 = rbk * 128 + 32 * w;
    const long long rl = p.R - 1;
    const long long ra = min(r0 + l31, rl);                                 // row of this lane's A fragments (clamped)
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.Wimg) + (long long)cb * (G * KSH + 1) * NBF * 2 * 1024;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) uint4*)wp_lds);

    // operand registers of one stage: fp32 rows (8 floats per k-step) or bf16 rows (one uint4 per k-step)
    float4 af[2][KSH][2];
    uint4 ab[2][PBF16 ? KSH : 1];
    float ain[8];
    auto issue_w = [&](int g, int buf) {                                    // weight slab of stage g -> LDS buffer buf
        const int npiece = (g < G ? KSH : 1) * NBF * 2;
        const unsigned char* src = wsrc + (long long)g * SLAB;
#pragma unroll
        for (int i = 0; i < (KSH * NBF * 2 + 3) / 4; ++i) {
            const int piece = 4 * i + w;                                    // wave-uniform
            if (piece < npiece) {
                // (wave-uniform by construction; readfirstlane tells the compiler so: the asm operands are SGPRs)
                const unsigned long long a = (unsigned long long)(uintptr_t)(src + (long long)piece * 1024);
                const unsigned alo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
                const unsigned ahi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + buf * SLAB + piece * 1024));
                glds16(reinterpret_cast<const void*>(((unsigned long long)ahi << 32) | alo), (unsigned)lane * 16u, dst);
            }
        }
    };
    auto issue_a = [&](int g, int set) {                                    // operand rows of stage g -> register set
        if (g == G) {
            const int nin = G * p.d;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = 8 * kq + i;
                const int qc = q < nin ? q : 0;
                const int gg = qc / p.d, j = qc - gg * p.d;
                const float* __restrict__ src = (p.Xc && gg >= 1) ? p.Xc + (long long)(gg - 1) * p.xc_plane + ra * 4 + j
                                                                  : p.Z + (long long)gg * p.PS + ra * p.Cp + H + j;
                const float x = *src;
                ain[i] = q < nin ? x : 0.f;                                 // (a select on the loaded value, not a branch)
            }
        } else if (PBF16 && g > 0) {
            const uint16_t* __restrict__ pb = p.Pb + (long long)(g - 1) * p.PSh + ra * H + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) ab[set][PBF16 ? ks : 0] = *reinterpret_cast<const uint4*>(pb + 16 * ks);
        } else {
            const float* __restrict__ z = p.Z + (long long)g * p.PS + ra * p.Cp + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                af[set][ks][0] = *reinterpret_cast<const float4*>(z + 16 * ks);
                af[set][ks][1] = *reinterpret_cast<const float4*>(z + 16 * ks + 4);
            }
        }
    };

    f32x16_t acc[NBF];
#pragma unroll
    for (int j = 0; j < NBF; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

    issue_w(0, 0);
    issue_a(0, 0);
#pragma unroll
    for (int g = 0; g <= G; ++g) {
        const int buf = g & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's share of slab g (and the rows of stage g)
        __syncthreads();                                                    // slab g complete; buffer buf ^ 1 no longer read
        if (g < G) { issue_w(g + 1, buf ^ 1); issue_a(g + 1, buf ^ 1); }
        const uint4* sl = wp_lds + buf * (SLAB / 16);
        auto bfrag = [&](int ks, int j, int hl) { return __builtin_bit_cast(bf16x8_t, sl[((ks * NBF + j) * 2 + hl) * 64 + lane]); };
        if (g == G) {                                                       // input channels of all planes: one k-step, bf16x3
            uint4 h4, l4;
            split8(ain, h4, l4);
            const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, h4), al = __builtin_bit_cast(bf16x8_t, l4);
#pragma unroll
            for (int j = 0; j < NBF; ++j) {
                const bf16x8_t bh = bfrag(0, j, 0), bl = bfrag(0, j, 1);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        } else if (PBF16 && g > 0) {                                        // bf16-resident plane x (W_hi + W_lo)
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                const bf16x8_t a = __builtin_bit_cast(bf16x8_t, ab[buf][PBF16 ? ks : 0]);
#pragma unroll
                for (int j = 0; j < NBF; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag(ks, j, 1), acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag(ks, j, 0), acc[j], 0, 0, 0);
                }
            }
        } else {                                                            // fp32 plane, bf16x3
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                const float4 x0 = af[buf][ks][0], x1 = af[buf][ks][1];
                const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                uint4 h4, l4;
                split8(v, h4, l4);
                const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, h4), al = __builtin_bit_cast(bf16x8_t, l4);
#pragma unroll
                for (int j = 0; j < NBF; ++j) {
                    const bf16x8_t bh = bfrag(ks, j, 0), bl = bfrag(ks, j, 1);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- fused GRU epilogue.  C/D layout: column = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
    const bool rows_in = r0 + 32 <= p.R;                                    // wave-uniform
#pragma unroll
    for (int j = 0; j < NBF; ++j) {
        const int c = (cb * NBF + j) * 32 + l31;                            // output column (< O: O % 32 == 0)
        const float bj = p.bias[c];
        if (EPI == WP_GATE) {
            // z_r = sigmoid(AGCN_gate): zr[r][2H] ; z columns (c < H) also emit the candidate state z*h (MegaCRN.py:43-45)
            const bool isz = c < H;                                         // fragment-uniform (H % 32 == 0)
            const int ch = isz ? c : 0;
            float hv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = min(r0 + 4 * kq + (v & 3) + 8 * (v >> 2), rl);
                hv[v] = p.Z[r * p.Cp + ch];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = r0 + 4 * kq + (v & 3) + 8 * (v >> 2);
                const float g = 1.f / (1.f + expf(-(acc[j][v] + bj)));
                if (rows_in || r <= rl) {
                    p.out[r * (2 * H) + c] = g;
                    if (isz) {
                        const float zh = g * hv[v];
                        p.out2[r * p.out2_ld + c] = zh;
                        if (p.out2b) {
                            const unsigned short hb = bf16_rne(zh);
                            p.out2b[r * H + c] = hb;
                            if (p.out2b_lo > 0) p.out2b[p.out2b_lo + r * H + c] = bf16_rne(zh - __uint_as_float((unsigned)hb << 16));
                        }
                    }
                }
            }
        } else {
            // hc = tanh(AGCN_update) ; h' = r*h + (1-r)*hc (MegaCRN.py:46-47)
            float hv[16], rg[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = min(r0 + 4 * kq + (v & 3) + 8 * (v >> 2), rl);
                hv[v] = p.hsrc[r * p.hsrc_ld + c];
                rg[v] = p.zr[r * (2 * H) + H + c];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long long r = r0 + 4 * kq + (v & 3) + 8 * (v >> 2);
                const float hc = tanhf(acc[j][v] + bj);
                if (rows_in || r <= rl) {
                    p.out[r * H + c] = hc;
                    const float hn = rg[v] * hv[v] + (1.f - rg[v]) * hc;
                    p.out2[r * p.out2_ld + c] = hn;
                    if (p.out2b) {
                        const unsigned short hb = bf16_rne(hn);
                        p.out2b[r * H + c] = hb;
                        if (p.out2b_lo > 0) p.out2b[p.out2b_lo + r * H + c] = bf16_rne(hn - __uint_as_float((unsigned)hb << 16));
                    }
                }
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
// column fragments per workgroup: as many as the output has (the operand rows are then read once), fewer when the grid
// would otherwise leave most of the chip idle (small graphs: the rows re-read come from L2)
static inline int wp_pick_nbf(int H, int O, long long R) {
    const long long rb = (R + 127) / 128;
    int best = 1;
    for (int nbf = 4; nbf >= 1; nbf >>= 1) {
        if (O % (32 * nbf)) continue;
        best = nbf;
        // (measured, METR-LA: 104 row blocks -> 208 workgroups per launch 6.26 ms/step, 416 6.12, 832 6.31; PEMS-BAY and
        //  EXPY-TKY unchanged between 160 and 300: with two workgroups per CU one's stage waits hide behind the other's MFMAs)
        const int minwg = 300;
        if (rb * (O / (32 * nbf)) >= minwg) break;
    }
    return best;
}
bool wp_stream_ok(int H, int d, int nbp, int O) {
    return (H == 32 || H == 64 || H == 128) && (nbp == 2 || nbp == 4) && d >= 1 && (nbp + 1) * d <= 16 && O % 32 == 0 && O >= 32;
}
size_t wp_img_uint4(int H, int nbp, int O) {
    const int KSt = (nbp + 1) * (H / 16) + 1;
    return (size_t)KSt * (O / 32) * 2 * 64;
}
hipError_t launch_wp_img_build(const float* Wf, int Cp, int H, int d, int nbp, int O, long long R, uint4* img, hipStream_t st, int nbf_force) {
    if (!wp_stream_ok(H, d, nbp, O)) return hipErrorInvalidValue;
    if (nbf_force > 0 && O % (32 * nbf_force)) return hipErrorInvalidValue;
    const int nbf = nbf_force > 0 ? nbf_force : wp_pick_nbf(H, O, R);
    const long long total = (long long)wp_img_uint4(H, nbp, O) / 2;          // one thread writes the hi AND the lo fragment word
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_wp_img_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Wf, Cp, H, d, nbp + 1, O, nbf, img, total);
    return hipGetLastError();
}
template <int KSH, int NBP, int NBF, int EPI, bool PBF16>
static inline hipError_t launch_wp_one(const WpP& p, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * KSH * NBF * 2 * 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)wp_stream_kernel<KSH, NBP, NBF, EPI, PBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    WpP q = p;
    q.ncb = p.O / (32 * NBF);
    const long long nrb = (p.R + 127) / 128;
    dim3 grid((unsigned)(((nrb + 7) / 8) * 8 * q.ncb));
    (void)hipGetLastError();
    hipLaunchKernelGGL((wp_stream_kernel<KSH, NBP, NBF, EPI, PBF16>), grid, dim3(256), lds, st, q);
    return hipGetLastError();
}
template <int KSH, int NBP, int EPI, bool PBF16>
static inline hipError_t launch_wp_nbf(const WpP& p, int nbf, hipStream_t st) {
    switch (nbf) {
        case 4: return launch_wp_one<KSH, NBP, 4, EPI, PBF16>(p, st);
        case 2: return launch_wp_one<KSH, NBP, 2, EPI, PBF16>(p, st);
        default: return launch_wp_one<KSH, NBP, 1, EPI, PBF16>(p, st);
    }
}
template <int EPI, bool PBF16>
static inline hipError_t launch_wp_epi(const WpP& p, int nbf, hipStream_t st) {
    const int ksh = p.H / 16;
    if (p.nbp == 4) {
        if (ksh == 2) return launch_wp_nbf<2, 4, EPI, PBF16>(p, nbf, st);
        if (ksh == 4) return launch_wp_nbf<4, 4, EPI, PBF16>(p, nbf, st);
        return launch_wp_nbf<8, 4, EPI, PBF16>(p, nbf, st);
    }
    if (ksh == 2) return launch_wp_nbf<2, 2, EPI, PBF16>(p, nbf, st);
    if (ksh == 4) return launch_wp_nbf<4, 2, EPI, PBF16>(p, nbf, st);
    return launch_wp_nbf<8, 2, EPI, PBF16>(p, nbf, st);
}
hipError_t launch_wp_stream(const WpP& p, hipStream_t st) {
    if (!wp_stream_ok(p.H, p.d, p.nbp, p.O) || p.R <= 0) return hipErrorInvalidValue;
    if ((((uintptr_t)p.Z) | ((uintptr_t)p.Pb) | ((uintptr_t)p.Wimg)) & 15) return hipErrorInvalidValue;
    if ((p.Cp & 3) || (p.PS & 3) || (p.PSh & 7)) return hipErrorInvalidValue;
    if (p.epi == WP_GATE ? p.O != 2 * p.H : p.O != p.H) return hipErrorInvalidValue;
    const int nbf = wp_pick_nbf(p.H, p.O, p.R);                             // (the image was built for the same choice)
    if (p.Pb) return p.epi == WP_GATE ? launch_wp_epi<WP_GATE, true>(p, nbf, st) : launch_wp_epi<WP_UPDATE, true>(p, nbf, st);
    return p.epi == WP_GATE ? launch_wp_epi<WP_GATE, false>(p, nbf, st) : launch_wp_epi<WP_UPDATE, false>(p, nbf, st);
}

}  // namespace mcrn
