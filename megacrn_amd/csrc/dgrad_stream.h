// dgrad_stream.h - plane-gradient GEMM of the AGCN backward, gfx950, bf16x3 arithmetic.
//
//   dP[g][r][c'] = sum_o dY[r][o] * Wd[(g, c')][o]          r < R = N*B rows, (g, c') < G*Cp columns, o < O <= 256
//
// Small K (O = 2H or H), wide N (G*Cp = 340 / 680 at METR-LA), huge M: the product is bound by its 18 / 36 MB of
// output.  In the tiled GEMM a 128x128 workgroup spent 3.1 us in its prologue and 4.4 us in its store epilogue around
// a 5.8 us K loop (in-kernel timeline, profiles/r1), all workgroups in lock-step: the memory system idled during
// the compute phases and saturated during the store phases.  Here every WAVE is its own pipeline:
//   * it owns one 32-row fragment of dY for the whole K extent: loaded once (two float4 per k-step and lane),
//     split into bf16 hi/lo once, kept in registers (8*KS VGPRs);
//   * the static operand Wd is pre-split once per step in MFMA B-fragment order (k_wfrag_build); a 32-column
//     fragment (16 KB at O = 128) is copied L2 -> LDS by the 4 waves of a workgroup together, no conversion, and read
//     back lane-linearly (conflict-free).  (A first version let every wave fetch its own fragments from L2:
//     146 MB per decoder launch, no faster than the tiled GEMM.)
//   * the workgroup walks a range of column fragments, double-buffered: the loads of fragment j+1 fly while fragment
//     j is multiplied (3*KS MFMAs) and stored, one barrier per fragment, so stores leave in a steady stream instead
//     of bursts;
//   * ~3 workgroups per CU are resident at once, no second round, no tail.
#pragma once
#include "gemm_bf16x3.h"

namespace mcrn {

// Wfrag[((j*KS + ks)*2 + hl)*64 + lane] = 8 bf16 (hi | lo) of W[n = 32 j + (lane&31)][k = 16 ks + 8 (lane>>5) + 0..7]
// (W row-major [rows][ld], zero beyond rows / K)
static __global__ void k_wfrag_build(const float* __restrict__ W, long long ld, int rows, int K, int KS,
                              uint4* __restrict__ out, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long long q = idx >> 6;
    const int ks = (int)(q % KS);
    const int j = (int)(q / KS);
    const int n = 32 * j + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (n < rows && k0 + i < K) ? W[(long long)n * ld + k0 + i] : 0.f;
    uint4 h, l;
    split8(v, h, l);
    out[((long long)(j * KS + ks) * 2 + 0) * 64 + lane] = h;
    out[((long long)(j * KS + ks) * 2 + 1) * 64 + lane] = l;
}
// KH = 2 (128 < O <= 256, the decoder gate of the H = 64 configurations): the K extent is walked in two halves of KS
// k-steps; the wave keeps BOTH halves of its dY rows resident (16 KS VGPRs), a column fragment is staged and multiplied
// half by half into the same accumulators and stored after the second.
// OUT (round 6): which output forms the instantiation carries.  0 = fp32 gradient planes only (the small graphs, the tiled large-graph path),
// 1 = + bf16 gradient planes, 2 = + the hoisted backward's packed state channels / stack-wide input operand (hi and hi/lo).  Until round 6 every
// instantiation carried all three: the fp32-only kernels of METR-LA paid for the other two with 65 VGPRs and 30 SGPRs (<8, 1>: 195 -> 130 VGPRs, a third
// wave per SIMD; <8, 2>: 256 + 12 B of scratch -> 194) - at the register limit a path that is never taken is not free (profiles/r6/experiments.md 8, 9).
template <int KS, int KH = 1, int OUT = 0>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 3))) void dgrad_stream_kernel(const DgradP p) {   // <= 3 waves per SIMD: keeps the compiler from spilling the staging registers to reach a higher occupancy
    constexpr int FRAG = KS * 2 * 64;                                 // uint4 per staged piece: one column fragment, one K half (hi and lo)
    constexpr int NLD = (FRAG + 255) / 256;                           // uint4 per thread to stage one piece
    constexpr int OH = 16 * KS, O = OH * KH;                          // columns of dY per half / in all
    // the A bounce buffer takes the wave's 32 rows in column CHUNKS of CW <= 64 (round 6): at O = 128 the whole-row buffer was 68 KB per workgroup
    // and held the kernel at two workgroups per CU whatever its registers; in 64-column chunks it is 35 KB (the two B stages: 32 KB)
    constexpr int NCH = (OH > 64 && KS % 2 == 0) ? 2 : 1, CW = OH / NCH;
    constexpr int AROW = CW + 4;                                      // padded row (floats) of the A bounce buffer
    constexpr int ABYTES = 4 * 32 * AROW * 4, BBYTES = 2 * FRAG * 16;
    // one LDS region: first the four waves' A bounce buffers, then (after a barrier) the two B stages
    __shared__ __attribute__((aligned(16))) unsigned char smem_[ABYTES > BBYTES ? ABYTES : BBYTES];
    uint4 (*sB)[FRAG] = reinterpret_cast<uint4 (*)[FRAG]>(smem_);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    // workgroup = (group of 4 row fragments, column part); wave w owns row fragment 4*rg + w; the workgroup walks the
    // column fragments of its part together so that every B fragment is fetched from L2 once per 128 rows
    // The `parts` workgroups of a row group run on ONE XCD (block b runs on XCD b % 8): they all read the same 128 rows of dY - once
    // from the memory side, then from that XCD's L2 (round 5: 25 -> 20 us per launch at METR-LA, +3 % on the step; with the parts
    // dealt round-robin over the XCDs every one of them fetched the rows through the fabric: 79 MB per launch for 25 MB of work).
    const int xcd = (int)(blockIdx.x & 7);
    const long long slot = blockIdx.x >> 3;
    // (the quotient and remainder of a run-time divisor come out of VALU code: back to SGPRs once, here)
    const int part = __builtin_amdgcn_readfirstlane((int)(slot % p.parts));
    const long long rg = (long long)__builtin_amdgcn_readfirstlane((int)(slot / p.parts)) * 8 + xcd;
    if (rg * 128 >= p.R) return;                                      // (grid padded to 8 row groups per round; uniform over the workgroup)
    const long long rf = rg * 4 + wave;
    const bool live = rf * 32 < p.R;                                  // whole wave beyond the last row: helps staging only
#ifdef MCRN_ABLATE
    const int dbg = p.dbg;
#else
    constexpr int dbg = 0;
#endif

    // ---- A: this wave's 32 rows for the whole K extent, split once, resident.  The 32 rows are ONE contiguous
    // 128*O-byte block of dY: it is read with fully coalesced 16-byte loads and bounced through LDS into fragment
    // order.  (Reading it directly in fragment order - 32 bytes per lane from 32 different rows - used a quarter of
    // every 128-byte line it touched and alone cost 20 of the kernel's 25 us.)
    uint4 ah[KH * KS], al[KH * KS];
    {
        float* __restrict__ sa = reinterpret_cast<float*>(smem_) + wave * 32 * AROW;
        const long long lim = p.R * O;                                // floats in dY
#pragma unroll
        for (int h = 0; h < KH; ++h)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int i = 0; i < CW / 8; ++i) {                        // 32*CW/4 float4 over 64 lanes
                const int e = lane + 64 * i;                          // float4 index inside the (32 x CW) block
                const int row = (4 * e) / CW, k = (4 * e) % CW;
                long long src = (rf * 32 + row) * O + h * OH + ch * CW + k;   // (a row's chunk: CW contiguous floats)
                if (src + 4 > lim) src = lim - 4;                     // rows beyond R: any valid address, never stored
                if (src < 0) src = 0;
                const float4 x = *reinterpret_cast<const float4*>(p.dY + src);
                *reinterpret_cast<float4*>(sa + row * AROW + k) = x;
            }
            // same wave wrote and reads (and overwrites for the next chunk / half): LDS operations of a wave complete in order
#pragma unroll
            for (int kc = 0; kc < CW / 16; ++kc) {
                const int ks = ch * (CW / 16) + kc;
                const float4 x = *reinterpret_cast<const float4*>(sa + l31 * AROW + 16 * kc + 8 * kq);
                const float4 y = *reinterpret_cast<const float4*>(sa + l31 * AROW + 16 * kc + 8 * kq + 4);
                const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
                split8(v, ah[h * KS + ks], al[h * KS + ks]);
            }
        }
    }
    __syncthreads();                                                  // the B stages reuse the bounce buffers
    const int j0 = part * p.cf_per_part;
    const int j1 = min(p.ncf, j0 + p.cf_per_part);
    if (j0 >= j1) return;                                             // uniform over the workgroup
    const long long r0 = rf * 32 + 4 * kq;
    const bool rows_in = rf * 32 + 32 <= p.R;                         // wave-uniform

    // staging registers as four scalars (NLD <= 4): as an array - lambda-captured or not - they were kept in scratch
    uint4 s0 = make_uint4(0u, 0u, 0u, 0u), s1 = s0, s2 = s0, s3 = s0;
    static_assert(NLD <= 4, "O <= 128 per half");
#define MCRN_DG_E(i) ((FRAG % 256 == 0 || tid + 256 * (i) < FRAG) ? tid + 256 * (i) : FRAG - 1)
#define MCRN_DG_FETCH(J)                                                                       \
    do {                                                                                       \
        const uint4* __restrict__ w_ = p.Wfrag + (long long)(J) * FRAG;                        \
        s0 = w_[MCRN_DG_E(0)];                                                                 \
        if constexpr (NLD > 1) s1 = w_[MCRN_DG_E(1)];                                          \
        if constexpr (NLD > 2) s2 = w_[MCRN_DG_E(2)];                                          \
        if constexpr (NLD > 3) s3 = w_[MCRN_DG_E(3)];                                          \
    } while (0)
#define MCRN_DG_OK(i) (FRAG % 256 == 0 || tid + 256 * (i) < FRAG)
#define MCRN_DG_PUBLISH(S)                                                                     \
    do {                                                                                       \
        if (MCRN_DG_OK(0)) sB[S][tid] = s0;                                                    \
        if constexpr (NLD > 1) { if (MCRN_DG_OK(1)) sB[S][tid + 256] = s1; }                   \
        if constexpr (NLD > 2) { if (MCRN_DG_OK(2)) sB[S][tid + 512] = s2; }                   \
        if constexpr (NLD > 3) { if (MCRN_DG_OK(3)) sB[S][tid + 768] = s3; }                   \
    } while (0)
    MCRN_DG_FETCH(j0 * KH);                                           // (pieces are numbered j * KH + half, contiguous in Wfrag)
    MCRN_DG_PUBLISH(0);
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
        f32x16 acc, acx;                                               // main product / the two cross products
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[v] = 0.f; acx[v] = 0.f; }
#pragma unroll
        for (int h = 0; h < KH; ++h) {
        const int s = KH == 1 ? ((j - j0) & 1) : h;                    // (an even number of pieces per fragment: the parity is the half)
        const bool more = h + 1 < KH || j + 1 < j1;
        if (more && !(dbg & 4)) MCRN_DG_FETCH(j * KH + h + 1);       // in flight during the MFMA chain and the stores
        if (!(dbg & 2))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[h * KS + ks]), xl = __builtin_bit_cast(bf16x8, al[h * KS + ks]);
            const bf16x8 yh = __builtin_bit_cast(bf16x8, sB[s][(ks * 2 + 0) * 64 + lane]);
            const bf16x8 yl = __builtin_bit_cast(bf16x8, sB[s][(ks * 2 + 1) * 64 + lane]);
            acx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acx, 0, 0, 0);
        }
        const int n = 32 * j + l31;                                    // C/D layout: column = lane & 31
        if (h == KH - 1 && live && n < p.ncols && !(dbg & 1)) {
            const int g = n / p.Cp;
            int cp = p.Cp;
            asm volatile("" : "+s"(cp));                               // row offsets stay scalar multiples, not 16 live VGPR pairs
            if (OUT == 2 && p.dPb && g > 0 && p.H > 0) {               // hoisted backward: packed state channels / stack-wide input operand
                const int cc = n - g * p.Cp;
                if (cc < p.H) {
                    unsigned short* __restrict__ c = p.dPb + (long long)(g - 1) * p.PSb + cc + r0 * p.H;
                    int hh = p.H;
                    asm volatile("" : "+s"(hh));
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int dr = (v & 3) + 8 * (v >> 2);
                        const float val = acc[v] + acx[v];
                        unsigned u = __float_as_uint(val);
                        u += 0x7FFFu + ((u >> 16) & 1u);
                        if (rows_in || r0 + dr < p.R) {
                            c[dr * hh] = (unsigned short)(u >> 16);
                            if (p.dPb_lo > 0) {                      // hi/lo operand pairs (gemm_bf16.h, nterm = 3): the residual as a second bf16
                                unsigned ul = __float_as_uint(val - __uint_as_float(u & 0xFFFF0000u));
                                ul += 0x7FFFu + ((ul >> 16) & 1u);
                                c[p.dPb_lo + dr * hh] = (unsigned short)(ul >> 16);
                            }
                        }
                    }
                } else if (cc < p.H + p.d && p.dPin) {
                    // row r = n * B + b -> element (n, in_col0 + b * d + j): one division, then the 28 row steps by carry
                    unsigned nn = (unsigned)r0 / (unsigned)p.B, bb = (unsigned)r0 - nn * (unsigned)p.B;
                    unsigned short* __restrict__ c = p.dPin + (long long)(g - 1) * p.in_plane + p.in_col0 + (cc - p.H);
#pragma unroll
                    for (int dr = 0; dr < 28; ++dr) {
                        if ((dr & 4) == 0) {
                            const int v = (dr & 3) + 4 * (dr >> 3);
                            if (rows_in || r0 + dr < p.R) {
                                const float val = acc[v] + acx[v];
                                unsigned u = __float_as_uint(val);
                                u += 0x7FFFu + ((u >> 16) & 1u);
                                c[(long long)nn * p.kin + bb * p.d] = (unsigned short)(u >> 16);
                                if (p.dPin_lo > 0) {
                                    unsigned ul = __float_as_uint(val - __uint_as_float(u & 0xFFFF0000u));
                                    ul += 0x7FFFu + ((ul >> 16) & 1u);
                                    c[p.dPin_lo + (long long)nn * p.kin + bb * p.d] = (unsigned short)(ul >> 16);
                                }
                            }
                        }
                        if (++bb == (unsigned)p.B) { bb = 0; ++nn; }
                    }
                }
            } else if (OUT == 1 && p.dPb && g > 0) {                   // bf16 gradient plane (same row / column indexing)
                unsigned short* __restrict__ c = p.dPb + (long long)(g - 1) * p.PSb + (n - g * p.Cp) + r0 * p.Cp;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dr = (v & 3) + 8 * (v >> 2);
                    unsigned u = __float_as_uint(acc[v] + acx[v]);
                    u += 0x7FFFu + ((u >> 16) & 1u);                   // round to nearest even
                    if (rows_in || r0 + dr < p.R) c[dr * cp] = (unsigned short)(u >> 16);
                }
            } else {
                float* __restrict__ c = p.dP + (long long)g * p.PS + (n - g * p.Cp) + r0 * p.Cp;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dr = (v & 3) + 8 * (v >> 2);
                    if (rows_in || r0 + dr < p.R) c[dr * cp] = acc[v] + acx[v];
                }
            }
        }
        if (more) MCRN_DG_PUBLISH(s ^ 1);                              // every wave left stage s^1 before the previous barrier
        __syncthreads();
        }   // halves
    }
#undef MCRN_DG_FETCH
#undef MCRN_DG_PUBLISH
#undef MCRN_DG_E
#undef MCRN_DG_OK
}

static inline hipError_t launch_dgrad_stream(DgradP p, hipStream_t st) {
    (void)hipGetLastError();
    const long long nrg = (p.R + 127) / 128;                           // groups of 4 row fragments
    p.ncf = (p.ncols + 31) / 32;
    // ~3 workgroups per CU resident at once: split the column fragments of a row group over `parts` workgroups
    long long parts = (768 + nrg - 1) / (nrg > 0 ? nrg : 1);
    if (parts < 1) parts = 1;
    if (parts > p.ncf) parts = p.ncf;
    p.cf_per_part = (int)((p.ncf + parts - 1) / parts);
    p.parts = (p.ncf + p.cf_per_part - 1) / p.cf_per_part;
    dim3 grid((unsigned)(((nrg + 7) / 8) * 8 * p.parts));             // 8 row groups (one per XCD) per round of `parts` workgroups each
    // output form of this launch (see OUT): decided by the fields the caller filled in
    const int out = p.dPb ? (p.H > 0 ? 2 : 1) : 0;
#define MCRN_DG_LAUNCH(KS_, KH_)                                                                                            \
    do {                                                                                                                    \
        if (out == 2) hipLaunchKernelGGL((dgrad_stream_kernel<KS_, KH_, 2>), grid, dim3(256), 0, st, p);                    \
        else if (out == 1) hipLaunchKernelGGL((dgrad_stream_kernel<KS_, KH_, 1>), grid, dim3(256), 0, st, p);               \
        else hipLaunchKernelGGL((dgrad_stream_kernel<KS_, KH_, 0>), grid, dim3(256), 0, st, p);                             \
    } while (0)
    switch (p.O / 16) {
        case 1: MCRN_DG_LAUNCH(1, 1); break;
        case 2: MCRN_DG_LAUNCH(2, 1); break;
        case 3: MCRN_DG_LAUNCH(3, 1); break;
        case 4: MCRN_DG_LAUNCH(4, 1); break;
        case 5: MCRN_DG_LAUNCH(5, 1); break;
        case 6: MCRN_DG_LAUNCH(6, 1); break;
        case 7: MCRN_DG_LAUNCH(7, 1); break;
        case 8: MCRN_DG_LAUNCH(8, 1); break;
        case 10: MCRN_DG_LAUNCH(5, 2); break;
        case 12: MCRN_DG_LAUNCH(6, 2); break;
        case 14: MCRN_DG_LAUNCH(7, 2); break;
        case 16: MCRN_DG_LAUNCH(8, 2); break;
        default: return hipErrorInvalidValue;
    }
#undef MCRN_DG_LAUNCH
    return hipGetLastError();
}

}  // namespace mcrn
