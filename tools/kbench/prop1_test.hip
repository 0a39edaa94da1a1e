// Stand-alone correctness + timing harness of the small-graph propagation kernels (no torch): the fused two-hop kernels of
// megacrn_amd/csrc/prop_small.h, forward and backward (with and without the d1t write-back), OUTSIDE the model's launch sequence.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -o prop1_test prop1_test.hip
//   ./prop1_test N ncols [reps]
// Prints, per kernel: max relative error vs a float64 CPU product of the same inputs and the average launch time over `reps`
// launches that rotate through NSET plane sets (operands come from the memory side, as in a train step).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#define MCRN_PROBE 1
#include "../../megacrn_amd/csrc/prop_small.h"
using namespace mcrn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifdef MCRN_TIMELINE
// -DMCRN_TIMELINE=1|2 build (`hipcc ... -DMCRN_TIMELINE=2 -o prop1_test_tl`): thread 0 of every workgroup stamps the 100 MHz clock at the
// kernel's phase boundaries (prop_small.h MCRN_TL); =2 drains the memory counters first, so a phase includes the latency of what it issued.
// One launch on a plane set the previous launches did not touch last; prints, per phase, the mean / max over the workgroups.
template <class F>
static hipError_t tl_dump(int kind, int nstamp, F launch) {
    static unsigned long long h[10][512][12];
    hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) return e;
    memset(h, 0, sizeof h);
    e = hipMemcpyToSymbol(HIP_SYMBOL(mcrn::g_tl), h, sizeof h); if (e != hipSuccess) return e;
    e = launch(); if (e != hipSuccess) return e;
    e = hipDeviceSynchronize(); if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(h, HIP_SYMBOL(mcrn::g_tl), sizeof h); if (e != hipSuccess) return e;
    unsigned long long t0 = ~0ull, t1 = 0; int nwg = 0;
    for (int b = 0; b < 512; ++b) {
        if (!h[kind][b][0]) continue;
        ++nwg;
        for (int i = 0; i < nstamp; ++i) if (h[kind][b][i]) { t0 = t0 < h[kind][b][i] ? t0 : h[kind][b][i]; t1 = t1 > h[kind][b][i] ? t1 : h[kind][b][i]; }
    }
    printf("    timeline kind %d: %d workgroups stamped, first stamp -> last stamp %.2f us\n", kind, nwg, (t1 - t0) / 100.0);
    for (int i = 0; i + 1 < nstamp; ++i) {
        double sum = 0, mx = 0, st = 0; int n = 0;
        for (int b = 0; b < 512; ++b) {
            if (!h[kind][b][i] || !h[kind][b][i + 1] || h[kind][b][i + 1] < h[kind][b][i]) continue;
            const double d = (h[kind][b][i + 1] - h[kind][b][i]) / 100.0;
            sum += d; mx = d > mx ? d : mx; st += (h[kind][b][i] - t0) / 100.0; ++n;
        }
        if (n) printf("      phase %d -> %d: mean %.2f us  max %.2f us   (mean start +%.2f us, %d wgs)\n", i, i + 1, sum / n, mx, st / n, n);
    }
    return hipSuccess;
}
#endif

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: prop1_test N ncols [reps]\n"); return 1; }
    const int N = atoi(argv[1]), ncols = atoi(argv[2]), reps = argc > 3 ? atoi(argv[3]) : 40;
    const int NSET = 6, nb = 4;
    const long long ld = ncols, PS = (long long)N * ld, ZT = 5 * PS;
    const int NF = (N + 31) / 32;
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    // blocks: row-stochastic S1, S2 and M2_s = 2 S_s S_s (float64 product rounded to fp32)
    std::vector<float> hA((size_t)nb * N * N);
    for (int s = 0; s < 2; ++s) {
        float* S = hA.data() + (size_t)(2 * s) * N * N;
        for (int i = 0; i < N; ++i) {
            double sum = 0;
            for (int j = 0; j < N; ++j) { S[(size_t)i * N + j] = expf(2.f * U(rng)); sum += S[(size_t)i * N + j]; }
            for (int j = 0; j < N; ++j) S[(size_t)i * N + j] = (float)(S[(size_t)i * N + j] / sum);
        }
        float* M2 = hA.data() + (size_t)(2 * s + 1) * N * N;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double a = 0;
                for (int k = 0; k < N; ++k) a += (double)S[(size_t)i * N + k] * S[(size_t)k * N + j];
                M2[(size_t)i * N + j] = (float)(2.0 * a);
            }
    }
    std::vector<float> hZ((size_t)NSET * ZT);
    for (auto& v : hZ) v = U(rng);
    float *dA, *dZ, *dZ0, *dX;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dZ, hZ.size() * 4)); CK(hipMalloc(&dZ0, hZ.size() * 4)); CK(hipMalloc(&dX, (size_t)3 * PS * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dZ0, hZ.data(), hZ.size() * 4, hipMemcpyHostToDevice));
    uint4* frag[8];
    {
        const float* S[8]; int tr[8];
        for (int i = 0; i < 8; ++i) { CK(hipMalloc(&frag[i], sfrag_uint4(N) * 16)); S[i] = dA + (size_t)(i & 3) * N * N; tr[i] = i >> 2; }
        SfragMultiP q;
        for (int i = 0; i < 8; ++i) { q.S[i] = S[i]; q.out[i] = frag[i]; q.transpose[i] = tr[i]; }
        q.ldS = N; q.N = N; q.NF = NF; q.n = 8;
        CK(launch_sfrag_multi(q, 0));
    }
    // old-path fragments: S1, S2, S1^T, S2^T = blocks 0, 2, 4, 6
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() { return hipMemcpy(dZ, dZ0, hZ.size() * 4, hipMemcpyDeviceToDevice); };

    // ---- CPU references on a sample of columns (float64)
    const int ncheck = 96;
    std::vector<int> cols(ncheck);
    for (int i = 0; i < ncheck; ++i) cols[i] = (int)(((long long)i * 9973 + 17) % ncols);
    cols[0] = 0; cols[1] = ncols - 1;
    const float* Z0 = hZ.data();    // set 0
    std::vector<double> refF((size_t)nb * N * ncheck), refB((size_t)N * ncheck);
    for (int k = 0; k < nb; ++k)
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < ncheck; ++c) {
                double a = 0;
                for (int j = 0; j < N; ++j) a += (double)hA[((size_t)k * N + i) * N + j] * Z0[(size_t)j * ld + cols[c]];
                if (k & 1) a -= Z0[(size_t)i * ld + cols[c]];
                refF[((size_t)k * N + i) * ncheck + c] = a;
            }
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < ncheck; ++c) {
            double a = Z0[(size_t)i * ld + cols[c]];
            for (int k = 0; k < nb; ++k) {
                const float* P = Z0 + (size_t)(1 + k) * PS;
                for (int j = 0; j < N; ++j) a += (double)hA[((size_t)k * N + j) * N + i] * P[(size_t)j * ld + cols[c]];
                if (k & 1) a -= P[(size_t)i * ld + cols[c]];
            }
            refB[(size_t)i * ncheck + c] = a;
        }
    std::vector<float> got((size_t)ZT), gx((size_t)3 * PS);
    auto check_fwd = [&]() -> double {
        double e = 0, m = 0;
        for (int k = 0; k < nb; ++k)
            for (int i = 0; i < N; ++i)
                for (int c = 0; c < ncheck; ++c) {
                    const double r = refF[((size_t)k * N + i) * ncheck + c], g = got[(size_t)(1 + k) * PS + (size_t)i * ld + cols[c]];
                    e = fmax(e, fabs(r - g)); m = fmax(m, fabs(r));
                }
        return e / m;
    };
    auto check_bwd = [&](int nx) -> double {
        double e = 0, m = 0;
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < ncheck; ++c) {
                const double r = refB[(size_t)i * ncheck + c];
                double g = got[(size_t)i * ld + cols[c]];
                for (int x = 0; x < nx; ++x) g += gx[(size_t)x * PS + (size_t)i * ld + cols[c]];
                e = fmax(e, fabs(r - g)); m = fmax(m, fabs(r));
            }
        return e / m;
    };
    // recursion-form reference of prop2_bwd on the sampled columns:  d1t_s = D1_s + S_s^T E2_s ;  d0 = D0 + sum_s S_s^T d1t_s
    std::vector<double> ref_d1t((size_t)2 * N * ncheck), ref_d0((size_t)N * ncheck);
    for (int sidx = 0; sidx < 2; ++sidx) {
        const float* S = hA.data() + (size_t)(2 * sidx) * N * N;
        const float *D1 = Z0 + (size_t)(1 + 2 * sidx) * PS, *E2 = Z0 + (size_t)(2 + 2 * sidx) * PS;
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < ncheck; ++c) {
                double a = D1[(size_t)i * ld + cols[c]];
                for (int j = 0; j < N; ++j) a += (double)S[(size_t)j * N + i] * E2[(size_t)j * ld + cols[c]];
                ref_d1t[((size_t)sidx * N + i) * ncheck + c] = a;
            }
    }
    for (int i = 0; i < N; ++i)
        for (int c = 0; c < ncheck; ++c) {
            double a = Z0[(size_t)i * ld + cols[c]];
            for (int sidx = 0; sidx < 2; ++sidx) {
                const float* S = hA.data() + (size_t)(2 * sidx) * N * N;
                for (int j = 0; j < N; ++j) a += (double)S[(size_t)j * N + i] * ref_d1t[((size_t)sidx * N + j) * ncheck + c];
            }
            ref_d0[(size_t)i * ncheck + c] = a;
        }
    auto check_bwd2 = [&](bool d1_written) -> double {      // got = plane set after prop2_bwd, gx = the extra plane
        double e1 = 0, m1 = 0, e0 = 0, m0 = 0;
        for (int sidx = 0; sidx < 2; ++sidx)
            for (int i = 0; i < N; ++i)
                for (int c = 0; c < ncheck; ++c) {
                    const double want = d1_written ? ref_d1t[((size_t)sidx * N + i) * ncheck + c] : (double)Z0[(size_t)(1 + 2 * sidx) * PS + (size_t)i * ld + cols[c]];
                    e1 = fmax(e1, fabs(want - got[(size_t)(1 + 2 * sidx) * PS + (size_t)i * ld + cols[c]])); m1 = fmax(m1, fabs(want));
                }
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < ncheck; ++c) {
                const double g = (double)got[(size_t)i * ld + cols[c]] + gx[(size_t)i * ld + cols[c]];
                e0 = fmax(e0, fabs(ref_d0[(size_t)i * ncheck + c] - g)); m0 = fmax(m0, fabs(ref_d0[(size_t)i * ncheck + c]));
            }
        return fmax(e1 / m1, e0 / m0);
    };
    printf("N=%d ncols=%d NF=%d reps=%d (plane %.1f MB, %d rotating sets)\n", N, ncols, NF, reps, PS * 4 / 1e6, NSET);
    // ---- baseline: fused two-hop kernels (feature recursion; numerically the same planes)
    if (N <= PROP2_MAX_N) {
        CK(reset());
        Prop2P q; memset(&q, 0, sizeof q); q.Sf[0] = frag[0]; q.Sf[1] = frag[2]; q.base = dZ; q.extra = nullptr; q.PS = PS; q.ld = ld; q.N = N; q.ncols = ncols;
        CK(launch_prop2_fwd(q, 0));
        CK(hipMemcpy(got.data(), dZ, ZT * 4, hipMemcpyDeviceToHost));
        const double ef = check_fwd();
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) { q.base = dZ + (size_t)(r % NSET) * ZT; CK(launch_prop2_fwd(q, 0)); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  prop2_fwd (2 serial hops)           err %.2e   %.1f us\n", ef, 1e3 * ms / reps);
#ifdef MCRN_TIMELINE
        CK(tl_dump(0, 10, [&]() { q.base = dZ + (size_t)3 * ZT; return launch_prop2_fwd(q, 0); }));
#endif
        if (N <= 352 && ncols % 132 == 0) {          // state columns only (decoder geometry: H = 128 of Cp = 132)
            Prop2P q2 = q; q2.cps = 2; q2.cstride = 132; q2.nunits = (ncols / 132) * 2;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) { q2.base = dZ + (size_t)(r % NSET) * ZT; CK(launch_prop2_fwd(q2, 0)); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  prop2_fwd,  state columns only (%d units)              %.1f us\n", q2.nunits, 1e3 * ms / reps);
        }
        CK(reset());
        q.Sf[0] = frag[4]; q.Sf[1] = frag[6]; q.extra = dX; q.base = dZ;
        CK(launch_prop2_bwd(q, 0));
        CK(hipMemcpy(got.data(), dZ, ZT * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gx.data(), dX, (size_t)PS * 4, hipMemcpyDeviceToHost));
        const double eb = check_bwd2(true);
        CK(reset());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) { q.base = dZ + (size_t)(r % NSET) * ZT; CK(launch_prop2_bwd(q, 0)); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  prop2_bwd (2 serial hops)           err %.2e   %.1f us\n", eb, 1e3 * ms / reps);
#ifdef MCRN_TIMELINE
        CK(tl_dump(1, 10, [&]() { q.base = dZ + (size_t)3 * ZT; return launch_prop2_bwd(q, 0); }));
#endif
    }
    if (N <= PROP2_MAX_N) {   // the same chain without the d1t write-back (Prop2P::no_d1: the model path's form since round 5)
        CK(reset());
        Prop2P q; memset(&q, 0, sizeof q); q.Sf[0] = frag[4]; q.Sf[1] = frag[6]; q.base = dZ; q.extra = dX; q.PS = PS; q.ld = ld; q.N = N; q.ncols = ncols;
        q.no_d1 = 1;
        CK(launch_prop2_bwd(q, 0));
        CK(hipMemcpy(got.data(), dZ, ZT * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gx.data(), dX, (size_t)PS * 4, hipMemcpyDeviceToHost));
        const double en = check_bwd2(false);             // planes 1 and 3 must be untouched, plane 0 + extra as before
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        CK(hipEventRecord(a0, 0));
        for (int r = 0; r < reps; ++r) { q.base = dZ + (size_t)(r % NSET) * ZT; CK(launch_prop2_bwd(q, 0)); }
        CK(hipEventRecord(a1, 0)); CK(hipEventSynchronize(a1));
        float ms = 0; CK(hipEventElapsedTime(&ms, a0, a1));
        printf("  prop2_bwd, no d1t write-back        err %.2e   %.1f us\n", en, 1e3 * ms / reps);
    }
    return 0;
}
