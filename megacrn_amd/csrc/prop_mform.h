// prop_mform.h - matrix-form Chebyshev propagation for small graphs (N <= 352), gfx950.
//
// model/MegaCRN.py:20-25 builds the support set [I, S, 2 S S - I] per support as MATRICES and multiplies every one of them
// by the same input.  prop_small.h's fused kernels run the feature recursion instead (x1 = S x0, x2 = 2 S x1 - x0): two
// SERIAL hops per workgroup.  Here the reference's own formulation is used: M2 = 2 S S is built once per step (exact fp32,
// 18 MFLOP at N = 207) and every AGCN call is ONE single-hop product with independent row blocks
//
//   forward :  plane[1 + k] = A_k x plane[0]  (- plane[0] for the T2 blocks)      A = [S1, M2_1, S2, M2_2]
//   backward:  plane[0] += sum_k A_k^T plane[1 + k]  (- plane[2] - plane[4])       K = nb*N, split over grid.y groups
//
// so a launch is (units x nb) workgroups of half the serial depth (no mid-kernel barrier / re-image / store-reload), and
// the adjacency gradient needs plane 0 as its only right operand:  dA_k = dP_k x X0^T  (dsm: ds_small_kernel with nb
// output blocks).  The "- I" of T2 = 2 S S - I never enters a bf16 split: it is applied in fp32 as an addend.
#pragma once
#include "prop_small.h"

namespace mcrn {

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

// One single-hop product per workgroup group:  out[y] = sum_{seg < nseg} A[y*nseg + seg] x src[y*nseg + seg] + c0 add0[y] + c1 add1[y]
// A workgroup = NF waves (one per 32-row fragment of the block) owns 32*CT columns; grid = (column units (capped), ny).
// STREAM at NF <= 8 asks for TWO co-resident workgroups per CU: 2 NF waves on 4 SIMDs (NF = 7: 4 waves per SIMD, 128 VGPRs).
// (the parameter block is only ever indexed through uniform selects: a dynamic index would move it to scratch)
template <class T>
static __device__ __forceinline__ T pick4(const T (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }
template <int NF, int CT, bool STREAM>
__global__ __launch_bounds__(64 * NF, (STREAM && NF <= 8 && CT * NF <= 20 /* two images fit the 160 KB LDS */) ? (2 * NF + 3) / 4 : 1)
void prop1_kernel(const Prop1P p) {
    using PB = PropBlock<NF, CT, STREAM>;
    constexpr int KS = 2 * NF;
    extern __shared__ __attribute__((aligned(16))) uint4 prop2_img[];   // PB::IMG uint4
    uint4* const img = prop2_img;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int y = blockIdx.y;
    const int l31 = lane & 31, kq = lane >> 5;
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
    const int row0 = 32 * w + 4 * kq;
    const bool rows_in = 32 * w + 32 <= p.N;           // wave-uniform: every row of this wave exists
    const int ks0 = STREAM ? (int)((blockIdx.x * 7 + blockIdx.y * 3) % KS) : 0;
    uint4 ah[PB::NAL], al[PB::NAL];
    const int nseg = p.nseg;
    if (!STREAM && nseg == 1) PB::load_a(pick4(p.Sf, y) + (long long)w * KS * 2 * 64 + lane, ah, al);   // register-stationary over the units
    const int cstep = p.cstep > 4 ? p.cstep : 4;       // floats between the column quads of a unit (Prop1P::cstep)
    const int nunits = p.cstep > 4 ? p.nunits : (p.ncols + 32 * CT - 1) / (32 * CT);
    const int u0 = (int)(((long long)blockIdx.x * nunits) / gridDim.x), u1 = (int)(((long long)(blockIdx.x + 1) * nunits) / gridDim.x);
    // column of output slot s = 32 t + cperm of a unit: quad s >> 2, element s & 3 (contiguous units: colbase + s)
    const int cq0 = (cperm >> 2) * cstep + (cperm & 3);
    float* __restrict__ OUT = pick4(p.out, y);
    const float* __restrict__ ad0 = pick4(p.add0, y);
    const float* __restrict__ ad1 = pick4(p.add1, y);
    const float c0 = pick4(p.coef0, y), c1 = pick4(p.coef1, y);
    for (int unit = u0; unit < u1; ++unit) {
        const int colbase = p.cstep > 4 ? p.col0 + unit * 8 * CT * cstep : unit * 32 * CT;
        int ld = (int)p.ld;
        asm volatile("" : "+s"(ld));                   // see MCRN_ROW_OF
        f32x16 acc[CT], adv0[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[t][v] = 0.f; adv0[t][v] = 0.f; }
        for (int seg = 0; seg < nseg; ++seg) {
            const int k = y * nseg + seg;
            if (unit > u0 || seg > 0) __syncthreads(); // previous image fully consumed
            const uint4* __restrict__ sfw = pick4(p.Sf, k) + (long long)w * KS * 2 * 64 + lane;
            if (!STREAM && nseg > 1) PB::load_a(sfw, ah, al);
            int nlast = p.N - 1;
            int tidv = tid;
            MCRN_FRESH(ld); MCRN_FRESH(nlast);
            asm volatile("" : "+v"(tidv));
            PB::stage(img, pick4(p.src, k), ld, nlast, p.ncols, colbase, tidv, cstep);
            if (seg == 0) {
                // fp32 addends (the accumulating plane, the "- I" term of a T2 block): clamped, unpredicated loads issued BEFORE
                // the barrier and the MFMA phase and only consumed by the epilogue, so they fly during both (in the streamed
                // variant they are older than the ring's loads: its counted waits cover them)
                MCRN_FRESH(ld);
                auto ldadd = [&](const float* __restrict__ ad, f32x16 (&dst)[CT]) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        const int col = min(colbase + 8 * t * cstep + cq0, p.ncols - 1);
                        if (rows_in) {
                            const unsigned o = (unsigned)(row0 * ld + col);
#pragma unroll
                            for (int v = 0; v < 16; ++v) dst[t][v] = ad[o + (unsigned)(MCRN_ROW_OF(v) * ld)];
                        } else {
#pragma unroll
                            for (int v = 0; v < 16; ++v) dst[t][v] = ad[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ld + col)];
                        }
                    }
                };
                if (ad0) ldadd(ad0, adv0);
            }
            __syncthreads();
            PB::template mma<true>(img, ah, al, acc, lane, sfw, ks0);
        }
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][v] += c0 * adv0[t][v];
        if (ad1) {   // a second addend (two-blocks-per-group variants only) is fetched here, exposed
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int col = min(colbase + 8 * t * cstep + cq0, p.ncols - 1);
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] += c1 * ad1[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ld + col)];
            }
        }
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 8 * t * cstep + cq0;
            const unsigned o = (unsigned)(row0 * ld + col);
            if (col < p.ncols) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) OUT[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = acc[t][v];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward propagation with the second Chebyshev term in MATRIX form inside the fused kernel's workgroup shape (round 4):
//   plane[1+2s] = S_s x0  (fragments of S register-stationary, as in prop2_fwd_kernel)
//   plane[2+2s] = (2 S_s S_s) x0 - x0   (fragments of M2_s = 2 S_s S_s streamed from L2 through the register ring)
// Both products read the SAME staged image of x0, so the serial part of prop2_fwd_kernel's second hop - barrier, accumulators ->
// B image (to_img), barrier: 3.6 of 18 us per workgroup in the in-kernel timeline (profiles/r1/timeline_metrla.txt) - is gone,
// while the grid stays that of the fused kernel (units x 2 supports: one round).  The planes are the same quantities as the
// feature recursion's (to rounding), so the backward pass keeps the recursion form (folded d-grad weights, prop2_bwd_kernel).
// MEASURED, NOT SHIPPED (tools/kbench/prop1_test only - the library does not instantiate it; profiles/r4/experiments.md section 7):
// correct to 1.3e-6, but at N = 207 the fragments of M2 streamed through the ring cost more than the barrier + to_img they
// replace (16.7 vs 15.7 us, 23.3 vs 22.3 us per launch); at N = 325, where S is streamed anyway, 28.0 vs 28.8 / 52.0 vs 55.6 us.
// ---------------------------------------------------------------------------------------------------------------------
template <int NF, int CT>
__global__ __launch_bounds__(64 * NF) void prop2m_fwd_kernel(const Prop2P p) {
    using PB = PropBlock<NF, CT>;                      // S: register-stationary (streamed at NF > 8)
    using PS = PropBlock<NF, CT, true>;                // M2: always streamed
    constexpr int KS = 2 * NF;
    extern __shared__ __attribute__((aligned(16))) uint4 prop2_img[];
    uint4* const img = prop2_img;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = blockIdx.y;
    const int l31 = lane & 31, kq = lane >> 5;
    const int cperm = 4 * (l31 & 7) + (l31 >> 3);
    const float* __restrict__ X0 = p.base;
    float* __restrict__ X1 = p.base + (long long)(1 + 2 * s) * p.PS;
    float* __restrict__ X2 = p.base + (long long)(2 + 2 * s) * p.PS;
    const int row0 = 32 * w + 4 * kq;
    const bool rows_in = 32 * w + 32 <= p.N;
    uint4 ah[PB::NAL], al[PB::NAL], dh[PS::NAL], dl[PS::NAL];
    const uint4* __restrict__ sfw0 = (s == 0 ? p.Sf[0] : p.Sf[1]) + (long long)w * KS * 2 * 64 + lane;
    const uint4* __restrict__ mfw = (s == 0 ? p.Mf[0] : p.Mf[1]) + (long long)w * KS * 2 * 64 + lane;
    const uint4* __restrict__ sfw = PB::WIDE ? sfw0 : nullptr;
    const int ks0 = (int)((blockIdx.x * 7 + blockIdx.y * 3) % KS);
    PB::load_a(sfw0, ah, al);
    const int nunits = p.nunits > 0 ? p.nunits : (p.ncols + 32 * CT - 1) / (32 * CT);
    const int u0 = (int)(((long long)blockIdx.x * nunits) / gridDim.x), u1 = (int)(((long long)(blockIdx.x + 1) * nunits) / gridDim.x);
    for (int unit = u0; unit < u1; ++unit) {
        const int colbase = p.nunits > 0 ? (unit / p.cps) * p.cstride + (unit % p.cps) * 32 * CT : unit * 32 * CT;
        int ld = (int)p.ld;
        asm volatile("" : "+s"(ld));
        if (unit > u0) __syncthreads();                // previous unit's image fully consumed
        int nlast = p.N - 1;
        int tidv = tid;
        MCRN_FRESH(ld); MCRN_FRESH(nlast);
        asm volatile("" : "+v"(tidv));
        PB::stage(img, X0, ld, nlast, p.ncols, colbase, tidv);
        __syncthreads();
        f32x16 acc[CT];
        PB::mma(img, ah, al, acc, lane, sfw, PB::WIDE ? ks0 : 0);
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
        // second term: acc = -x0 (the lines were just staged: cache hits), then the M2 block on the same image
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = min(colbase + 32 * t + cperm, p.ncols - 1);
            if (rows_in) {
                const unsigned o = (unsigned)(row0 * ld + col);
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = -X0[o + (unsigned)(MCRN_ROW_OF(v) * ld)];
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[t][v] = -X0[(unsigned)(min(row0 + MCRN_ROW_OF(v), p.N - 1) * ld + col)];
            }
        }
        PS::template mma<true>(img, dh, dl, acc, lane, mfw, ks0);
        MCRN_FRESH(ld);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = colbase + 32 * t + cperm;
            const unsigned o = (unsigned)(row0 * ld + col);
            if (col < p.ncols) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (rows_in || row0 + MCRN_ROW_OF(v) < p.N) X2[o + (unsigned)(MCRN_ROW_OF(v) * ld)] = acc[t][v];
            }
        }
    }
}
static inline hipError_t launch_prop2m_fwd(const Prop2P& p, hipStream_t st) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    int ct, blocks;
    prop2_shape(p.ncols, ct, blocks, NF);
    if (p.nunits > 0) {
        const int passes = (p.nunits + 127) / 128;
        ct = 2; blocks = (p.nunits + passes - 1) / passes;
    }
    dim3 grid(blocks, 2);
    if (ct == 3) { MCRN_NF_SWITCH8(prop2m_fwd_kernel, 3, grid, p) }
    else { MCRN_NF_SWITCH(prop2m_fwd_kernel, 2, grid, p) }
    return hipGetLastError();
}

#define MCRN_LAUNCH_PROP1(NF_, CT_, ST_, GRID, P)                                                          \
    do {                                                                                                  \
        constexpr size_t lds_ = (size_t)PropBlock<NF_, CT_, ST_>::IMG * sizeof(uint4);                    \
        static bool set_ = false;                                                                         \
        if (lds_ > 64 * 1024 && !set_) {                                                                  \
            hipError_t e_ = hipFuncSetAttribute((const void*)prop1_kernel<NF_, CT_, ST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); \
            if (e_ != hipSuccess) return e_;                                                              \
            set_ = true;                                                                                  \
        }                                                                                                 \
        if ((P).ev0 && (P).ev1)                                                                           \
            hipExtLaunchKernelGGL((prop1_kernel<NF_, CT_, ST_>), GRID, dim3(64 * NF_), lds_, st, (hipEvent_t)(P).ev0, (hipEvent_t)(P).ev1, 0, P); \
        else                                                                                              \
        hipLaunchKernelGGL((prop1_kernel<NF_, CT_, ST_>), GRID, dim3(64 * NF_), lds_, st, P);             \
    } while (0)
#define MCRN_PROP1_NF(CT_, ST_, GRID, P)                                                     \
    switch (NF) {                                                                            \
        case 1: MCRN_LAUNCH_PROP1(1, CT_, ST_, GRID, P); break;                              \
        case 2: MCRN_LAUNCH_PROP1(2, CT_, ST_, GRID, P); break;                              \
        case 3: MCRN_LAUNCH_PROP1(3, CT_, ST_, GRID, P); break;                              \
        case 4: MCRN_LAUNCH_PROP1(4, CT_, ST_, GRID, P); break;                              \
        case 5: MCRN_LAUNCH_PROP1(5, CT_, ST_, GRID, P); break;                              \
        case 6: MCRN_LAUNCH_PROP1(6, CT_, ST_, GRID, P); break;                              \
        case 7: MCRN_LAUNCH_PROP1(7, CT_, ST_, GRID, P); break;                              \
        default: MCRN_LAUNCH_PROP1(8, CT_, ST_, GRID, P); break;                             \
    }
// variant: ct = column tiles per unit (2 .. 4), stream = adjacency fragments through the register ring (always at N > 256),
// cap = most workgroups per group (0: one per unit)
static inline hipError_t launch_prop1(const Prop1P& p, int ct, bool stream, int cap, hipStream_t st) {
    (void)hipGetLastError();
    const int NF = (p.N + 31) / 32;
    if (NF > 8) { ct = 2; stream = true; }
    const int nunits = p.cstep > 4 ? p.nunits : (p.ncols + 32 * ct - 1) / (32 * ct);     // (gathered: the caller counted units of 8*ct quads)
    dim3 grid(cap > 0 && cap < nunits ? cap : nunits, p.ny);
    if (NF > 8) {
        switch (NF) {
            case 9: MCRN_LAUNCH_PROP1(9, 2, true, grid, p); break;
            case 10: MCRN_LAUNCH_PROP1(10, 2, true, grid, p); break;
            default: MCRN_LAUNCH_PROP1(11, 2, true, grid, p); break;
        }
    } else if (stream) {
        if (ct == 2) { MCRN_PROP1_NF(2, true, grid, p) }
        else if (ct == 3) { MCRN_PROP1_NF(3, true, grid, p) }
        else { MCRN_PROP1_NF(4, true, grid, p) }
    } else {
        if (ct == 2) { MCRN_PROP1_NF(2, false, grid, p) }
        else if (ct == 3) { MCRN_PROP1_NF(3, false, grid, p) }
        else { MCRN_PROP1_NF(4, false, grid, p) }
    }
    return hipGetLastError();
}

}  // namespace mcrn
