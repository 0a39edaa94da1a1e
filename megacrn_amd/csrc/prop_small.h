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
#include "gemm_bf16x3.h"

namespace mcrn {

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

// Sfrag[((i*KS + ks)*2 + hl)*64 + lane] = 8 bf16 of  A[row = 32 i + (lane&31)][k = 16 ks + 8 (lane>>5) + 0..7]
// A = S (transpose == 0) or S^T (transpose == 1); out-of-range rows / k are zero.
__global__ void k_sfrag_build(const float* __restrict__ S, long long ldS, int N, int NF, int transpose,
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

static inline bool prop_small_ok(int N, long long ld, int ncols) {
    return N <= 256 && (ncols % 4) == 0 && (ld % 4) == 0;
}
static inline size_t sfrag_uint4(int N) {
    const int NF = (N + 31) / 32;
    return (size_t)NF * 2 * NF * 2 * 64;
}
static inline hipError_t launch_sfrag(const float* S, long long ldS, int N, int transpose, uint4* out, hipStream_t st) {
    const int NF = (N + 31) / 32;
    const int tot = NF * 2 * NF * 64;
    hipLaunchKernelGGL(k_sfrag_build, dim3((tot + 255) / 256), dim3(256), 0, st, S, ldS, N, NF, transpose, out);
    return hipGetLastError();
}
static inline hipError_t launch_prop_small(const PropP& p, int nbatch, hipStream_t st) {
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
