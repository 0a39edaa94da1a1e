// dgrad_stream.h - plane-gradient GEMM of the AGCN backward, gfx950, bf16x3 arithmetic.
//
//   dP[g][r][c'] = sum_o dY[r][o] * Wd[(g, c')][o]          r < R = N*B rows, (g, c') < G*Cp columns, o < O <= 128
//
// Small K (O = 2H or H), wide N (G*Cp = 340 / 680 at METR-LA), huge M: the product is bound by its 18 / 36 MB of
// output.  In the tiled GEMM a 128x128 workgroup spent 3.1 us in its prologue and 4.4 us in its store epilogue around
// a 5.8 us K loop (in-kernel timeline, profiles/r1), all workgroups in lock-step: the memory system idled during
// the compute phases and saturated during the store phases.  Here every WAVE is its own pipeline:
//   * it owns one 32-row fragment of dY for the whole K extent: loaded once (two float4 per k-step and lane),
//     split into bf16 hi/lo once, kept in registers (8*KS VGPRs);
//   * the static operand Wd is pre-split once per step in MFMA B-fragment order (k_wfrag_build), so a column
//     fragment is 2*KS lane-linear 16-byte loads straight from L2 - no LDS, no barriers, no conversion;
//   * it walks a range of 32-column fragments, double-buffered in registers: the loads of fragment j+1 fly while
//     fragment j is multiplied (3*KS MFMAs) and stored, so stores leave in a steady stream instead of bursts;
//   * all waves of the launch are resident at once (2 per SIMD), no second round, no tail.
#pragma once
#include "gemm_bf16x3.h"

namespace mcrn {

// Wfrag[((j*KS + ks)*2 + hl)*64 + lane] = 8 bf16 (hi | lo) of W[n = 32 j + (lane&31)][k = 16 ks + 8 (lane>>5) + 0..7]
// (W row-major [rows][ld], zero beyond rows / K)
__global__ void k_wfrag_build(const float* __restrict__ W, long long ld, int rows, int K, int KS,
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
static inline size_t wfrag_uint4(int rows, int K) { return (size_t)((rows + 31) / 32) * ((K + 15) / 16) * 2 * 64; }
static inline bool dgrad_stream_ok(int O) { return O % 16 == 0 && O >= 16 && O <= 128; }

struct DgradP {
    const float* dY;        // [R][O]
    const uint4* Wfrag;     // k_wfrag_build image of Wd [(g, c')][o]
    float* dP;              // [G][R][Cp]  (plane stride PS)
    long long R, PS;
    int O, ncols, Cp;       // ncols = G*Cp
    int ncf, parts, cf_per_part;
};

template <int KS>
__global__ __launch_bounds__(256, 2) void dgrad_stream_kernel(const DgradP p) {   // 2 workgroups per CU = 2 waves per SIMD: <= 256 VGPRs+AGPRs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kq = lane >> 5;
    const long long task = (long long)blockIdx.x * 4 + wave;          // (row fragment, column part); the 4 waves of a
    const int part = (int)(task % p.parts);                           // workgroup share the row fragment when parts % 4 == 0
    const long long rf = task / p.parts;
    if (rf * 32 >= p.R) return;

    // ---- A: this wave's 32 rows for the whole K extent, split once, resident
    uint4 ah[KS], al[KS];
    {
        const long long row = min(rf * 32 + l31, p.R - 1);            // clamped rows are never stored
        const float* __restrict__ a = p.dY + row * p.O + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 x = *reinterpret_cast<const float4*>(a + 16 * ks);
            const float4 y = *reinterpret_cast<const float4*>(a + 16 * ks + 4);
            const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
            split8(v, ah[ks], al[ks]);
        }
    }
    const int j0 = part * p.cf_per_part;
    const int j1 = min(p.ncf, j0 + p.cf_per_part);
    if (j0 >= j1) return;
    const long long r0 = rf * 32 + 4 * kq;
    const bool rows_in = rf * 32 + 32 <= p.R;                         // wave-uniform

    uint4 b0h[KS], b0l[KS], b1h[KS], b1l[KS];
    auto load_b = [&](int j, uint4 (&bh)[KS], uint4 (&bl)[KS]) {
        const uint4* __restrict__ w = p.Wfrag + (long long)j * KS * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bh[ks] = w[(ks * 2 + 0) * 64];
            bl[ks] = w[(ks * 2 + 1) * 64];
        }
    };
    auto mma_store = [&](int j, const uint4 (&bh)[KS], const uint4 (&bl)[KS]) {
        f32x16 acc, acx;                                               // main product / the two cross products
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[v] = 0.f; acx[v] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[ks]), xl = __builtin_bit_cast(bf16x8, al[ks]);
            const bf16x8 yh = __builtin_bit_cast(bf16x8, bh[ks]), yl = __builtin_bit_cast(bf16x8, bl[ks]);
            acx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, yh, acx, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yh, acc, 0, 0, 0);
            acx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, yl, acx, 0, 0, 0);
        }
        const int n = 32 * j + l31;                                    // C/D layout: column = lane & 31
        if (n < p.ncols) {
            const int g = n / p.Cp;
            float* __restrict__ c = p.dP + (long long)g * p.PS + (n - g * p.Cp) + r0 * p.Cp;
            int cp = p.Cp;
            asm volatile("" : "+s"(cp));                               // row offsets stay scalar multiples, not 16 live VGPR pairs
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int dr = (v & 3) + 8 * (v >> 2);
                if (rows_in || r0 + dr < p.R) c[dr * cp] = acc[v] + acx[v];
            }
        }
    };
    load_b(j0, b0h, b0l);
    for (int j = j0; j < j1; j += 2) {
        if (j + 1 < j1) load_b(j + 1, b1h, b1l);
        mma_store(j, b0h, b0l);
        if (j + 2 < j1) load_b(j + 2, b0h, b0l);
        if (j + 1 < j1) mma_store(j + 1, b1h, b1l);
    }
}

static inline hipError_t launch_dgrad_stream(DgradP p, hipStream_t st) {
    (void)hipGetLastError();
    const long long nrf = (p.R + 31) / 32;
    p.ncf = (p.ncols + 31) / 32;
    // ~2048 resident waves (256 CUs x 4 SIMDs x 2): split the column fragments of a row fragment over `parts` waves
    long long parts = 2048 / (nrf > 0 ? nrf : 1);
    if (parts < 1) parts = 1;
    if (parts > p.ncf) parts = p.ncf;
    if (parts >= 4) parts &= ~3LL;                                     // the 4 waves of a workgroup then share their A rows
    p.cf_per_part = (int)((p.ncf + parts - 1) / parts);
    p.parts = (p.ncf + p.cf_per_part - 1) / p.cf_per_part;
    const long long tasks = nrf * p.parts;
    dim3 grid((unsigned)((tasks + 3) / 4));
    switch (p.O / 16) {
        case 1: hipLaunchKernelGGL(dgrad_stream_kernel<1>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(dgrad_stream_kernel<2>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(dgrad_stream_kernel<3>, grid, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(dgrad_stream_kernel<4>, grid, dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL(dgrad_stream_kernel<5>, grid, dim3(256), 0, st, p); break;
        case 6: hipLaunchKernelGGL(dgrad_stream_kernel<6>, grid, dim3(256), 0, st, p); break;
        case 7: hipLaunchKernelGGL(dgrad_stream_kernel<7>, grid, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(dgrad_stream_kernel<8>, grid, dim3(256), 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mcrn
