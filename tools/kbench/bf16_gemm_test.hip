// Stand-alone correctness + timing harness for megacrn_amd/csrc/gemm_bf16.h (no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bf16_gemm_test bf16_gemm_test.hip
//   ./bf16_gemm_test probe                      # print the ds_read_b64_tr_b16 lane mapping
//   ./bf16_gemm_test M N seglen nseg nn|nt cfg nsplit reps [with_bf16_copy]
//   env X3=1: hi/lo operand pairs (Bf16GemmP::nterm = 3: A_hi B_hi + A_hi B_lo + A_lo B_hi) checked against the fp32 values' float64 product;
//   env CB_ONLY=1: time the bf16-only output form; NO_OUT=1: time without any store.  Builds with -DMCRN_BF16_ABL=<bits> take one
//   stream out of the K loop / add the in-kernel clock probe (gemm_bf16.h; profiles/r4/experiments.md section 13).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include "../../megacrn_amd/csrc/gemm_bf16.h"
using namespace mcrn;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void k_tr_probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // each lane passes the address of ITS 8 bytes: lane i of a 16-lane group -> elements 4i .. 4i+3 of the group's 128-byte block
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + (lane >> 4) * 64 + (lane & 15) * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "probe")) {
        short* d; CK(hipMalloc(&d, 256 * 2));
        hipLaunchKernelGGL(k_tr_probe, dim3(1), dim3(64), 0, 0, d);
        CK(hipDeviceSynchronize());
        short h[256]; CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
        int okc = 0;
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) { printf(" %4d", h[l * 4 + j]); okc += h[l * 4 + j] == (l & 15) + 16 * j + 64 * (l >> 4); }
            printf("\n");
        }
        printf("tr16_b64 matches [row j][col lane&15] of the group's 4x16 block: %d / 256\n", okc);
        return okc == 256 ? 0 : 1;
    }
    const int M = argc > 1 ? atoi(argv[1]) : 7372, N = argc > 2 ? atoi(argv[2]) : 2176;
    const int seglen = argc > 3 ? atoi(argv[3]) : 1843, nseg = argc > 4 ? atoi(argv[4]) : 1;
    const bool btr = argc > 5 ? !strcmp(argv[5], "nn") : true;
    const int cfg = argc > 6 ? atoi(argv[6]) : 0, nsplit = argc > 7 ? atoi(argv[7]) : 1, reps = argc > 8 ? atoi(argv[8]) : 20;
    const int with_cb = argc > 9 ? atoi(argv[9]) : 0;
    const int Kp = (seglen + 63) / 64 * 64;            // padded segment (A rows are zero beyond seglen)
    const long long lda = (long long)nseg * Kp;
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    const bool x3 = getenv("X3") != nullptr;
    // X3: the operands are fp32 values held as bf16 hi + lo pairs; the lo images sit behind the hi ones (vectors twice as long)
    std::vector<float> fA, fB;
    auto putf = [&](std::vector<uint16_t>& h, std::vector<float>& f, size_t i, size_t lo_off, float v) {
        const uint16_t hi = f2bf(v);
        h[i] = hi;
        if (x3) { f[i] = v; h[lo_off + i] = f2bf(v - bf2f(hi)); } 
    };
    const size_t nA = (size_t)M * lda;
    std::vector<uint16_t> hA(nA * (x3 ? 2 : 1), 0), hB;
    if (x3) fA.assign(nA, 0.f);
    for (int m = 0; m < M; ++m)
        for (int s = 0; s < nseg; ++s)
            for (int k = 0; k < seglen; ++k) putf(hA, fA, (size_t)m * lda + (size_t)s * Kp + k, nA, U(rng));
    long long ldb, bseg;
    if (btr) {   // B[seg][k][n], n contiguous, k rows padded to Kp with zeros
        ldb = N; bseg = (long long)Kp * N;
        hB.assign((size_t)nseg * Kp * N * (x3 ? 2 : 1), 0);
        if (x3) fB.assign((size_t)nseg * Kp * N, 0.f);
        for (int s = 0; s < nseg; ++s)
            for (int k = 0; k < seglen; ++k)
                for (int n = 0; n < N; ++n) putf(hB, fB, (size_t)s * bseg + (size_t)k * ldb + n, (size_t)nseg * Kp * N, U(rng));
    } else {     // B[n][seg][k], k padded to Kp with zeros (same layout as A)
        ldb = lda; bseg = Kp;
        hB.assign((size_t)N * ldb * (x3 ? 2 : 1), 0);
        if (x3) fB.assign((size_t)N * ldb, 0.f);
        for (int n = 0; n < N; ++n)
            for (int s = 0; s < nseg; ++s)
                for (int k = 0; k < seglen; ++k) putf(hB, fB, (size_t)n * ldb + (size_t)s * Kp + k, (size_t)N * ldb, U(rng));
    }
    uint16_t *dA, *dB, *dZ; float *dC; uint16_t* dCb;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2 + 256)); CK(hipMalloc(&dZ, 256));
    const int ns = nsplit < 1 ? 1 : nsplit;
    CK(hipMalloc(&dC, (size_t)ns * M * N * 4)); CK(hipMalloc(&dCb, (size_t)M * N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dZ, 0, 256)); CK(hipMemset(dC, 0xFF, (size_t)ns * M * N * 4));
    Bf16GemmP p; memset(&p, 0, sizeof p);
    p.A = dA; p.B = dB; p.am = rm_plain(lda); p.bm = rm_plain(ldb); p.ldb = ldb;
    p.nseg = nseg; p.seg_len = seglen; p.a_seg = Kp; p.b_seg = bseg; p.M = M; p.N = N;
    p.C = dC; p.cm = rm_plain(N); p.alpha = 1.f; p.beta = 0.f; p.slab = (long long)M * N;
    p.Cb = (ns == 1 && with_cb) ? dCb : nullptr; p.cbm = rm_plain(N); p.xcd = 1;
    if (x3) { p.nterm = 3; p.a_lo = (long long)nA; p.b_lo = (long long)(hB.size() / 2); }
    hipError_t e = launch_gemm_bf16(p, btr, cfg, nsplit, 0, 0);
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 2; }
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)ns * M * N); std::vector<uint16_t> hCb((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hCb.data(), dCb, hCb.size() * 2, hipMemcpyDeviceToHost));
    // reference on sampled rows (all columns), double accumulation of the bf16-rounded operands
    double maxerr = 0, maxref = 0, maxerr_b = 0; int rows_checked = 0;
    std::vector<int> rows = {0, 1, 31, 32, 63, 64, 127, 128, 255, 256, M / 2, M - 2, M - 1};
    for (int i = 0; i < 24; ++i) rows.push_back((int)(rng() % M));
    for (int m = 5; m < M; m += 61) rows.push_back(m);             // every row tile of every configuration
    for (int m : rows) {
        if (m < 0 || m >= M) continue;
        ++rows_checked;
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int sg = 0; sg < nseg; ++sg)
                for (int k = 0; k < seglen; ++k) {
                    const size_t ia = (size_t)m * lda + (size_t)sg * Kp + k, ib = btr ? (size_t)sg * bseg + (size_t)k * ldb + n : (size_t)n * ldb + (size_t)sg * bseg + k;
                    const double a = x3 ? (double)fA[ia] : (double)bf2f(hA[ia]);
                    const double b = x3 ? (double)fB[ib] : (double)bf2f(hB[ib]);
                    s += a * b;
                }
            double got = 0;
            for (int z = 0; z < ns; ++z) got += hC[(size_t)z * M * N + (size_t)m * N + n];
            maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s));
            if (ns == 1 && with_cb) maxerr_b = fmax(maxerr_b, fabs(bf2f(hCb[(size_t)m * N + n]) - s));
        }
    }
    printf("M=%d N=%d K=%dx%d %s cfg=%d split=%d : max|err|=%.3e (max|ref|=%.3e, rel %.2e) bf16copy err %.3e rows %d\n", M, N, nseg, seglen,
           btr ? "nn" : "nt", cfg, ns, maxerr, maxref, maxerr / maxref, maxerr_b, rows_checked);
    const bool ok = maxerr / maxref < 2e-5;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (getenv("CB_ONLY") && ns == 1) { p.C = nullptr; p.Cb = dCb; }     // time the bf16-only output form (the model's forward propagation)
    if (getenv("NO_OUT")) { p.C = nullptr; p.Cb = nullptr; }             // no stores at all (K loop + launch cost)
    for (int i = 0; i < 3; ++i) launch_gemm_bf16(p, btr, cfg, nsplit, 0, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch_gemm_bf16(p, btr, cfg, nsplit, 0, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * M * N * (double)nseg * seglen;        // algorithmic (X3: three MFMAs per product)
    printf("  %s  %.2f us/launch  %.1f TFLOP/s\n", ok ? "OK " : "BAD", 1e3 * ms / reps, fl / (ms / reps * 1e-3) / 1e12);
#if MCRN_BF16_ABL & 8
    {
        unsigned long long c[4];
        CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(mcrn::g_bf16_clk), sizeof c));
        const double dc = (double)(c[2] - c[0]), dw = (double)(c[3] - c[1]);
        printf("  clock probe (workgroup 0, K loop): %.0f shader cycles in %.2f us -> %.0f MHz\n", dc, dw / 100.0, dw > 0 ? dc / dw * 100.0 : 0.0);
    }
#endif
    return ok ? 0 : 1;
}
