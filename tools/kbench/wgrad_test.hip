// Stand-alone correctness + timing harness for megacrn_amd/csrc/wgrad_stream.h (no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-vectorize -o wgrad_test wgrad_test.hip    (the library's flags)
//   ./wgrad_test T R G Cp O cpt reps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#define MCRN_WGS_DEBUG 1
#include "../../megacrn_amd/csrc/wgrad_stream.h"
#include "../../megacrn_amd/csrc/gemm_bf16.h"
using namespace mcrn;
// POISON: fills the whole LDS of every CU with NaN bit patterns and exits (the next kernel's workgroups inherit it)
__global__ __launch_bounds__(1024) void k_lds_poison(unsigned* sink) {
    extern __shared__ unsigned pz[];
    for (int i = threadIdx.x; i < 40000; i += 1024) pz[i] = 0x7FC07FC0u;
    __syncthreads();
    if (pz[threadIdx.x] == 1u) sink[0] = 1;
}
// MFMA-only neighbour without LDS: co-resident with every variant of the kernel under test
__global__ __launch_bounds__(256) void k_mfma_neighbour(int spins, unsigned* sink) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v c = {0};
    uint4 u = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    for (int i = 0; i < spins; ++i)
#pragma unroll 4
        for (int k = 0; k < 64; ++k) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, u), __builtin_bit_cast(b8, u), c, 0, 0, 0);
    if ((unsigned)c[0] == 0xFFFFFFFFu) sink[0] = 1;
}
// holds 64 KB of LDS per workgroup for a while; touches none of it (HOLD=1) or fills its own region only (HOLD=2)
__global__ __launch_bounds__(256) void k_lds_holder(int mode, int spins, unsigned* sink) {
    __shared__ unsigned buf[16384];
    unsigned acc = 0;
    for (int i = 0; i < spins; ++i) {
        if (mode == 2) buf[(threadIdx.x + 256 * i) & 16383] = i;
        if (mode == 3) {   // hammer: full-rate 16-byte reads and writes inside the workgroup's own 64 KB
            uint4* b4 = reinterpret_cast<uint4*>(buf);
#pragma unroll 8
            for (int k = 0; k < 64; ++k) { uint4 v = b4[(threadIdx.x + 37 * k + i) & 4095]; v.x += k; b4[(threadIdx.x * 3 + 11 * k + i) & 4095] = v; acc += v.y; }
            continue;
        }
        if (mode == 4 || mode == 5) {   // MFMA (+ 16-byte LDS reads feeding it, mode 5)
            typedef __bf16 b8 __attribute__((ext_vector_type(8)));
            typedef float f16v __attribute__((ext_vector_type(16)));
            uint4* b4 = reinterpret_cast<uint4*>(buf);
            f16v c = {0};
            uint4 u = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                if (mode == 5) u = b4[(threadIdx.x + 64 * k + i) & 4095];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, u), __builtin_bit_cast(b8, u), c, 0, 0, 0);
            }
            acc += (unsigned)c[0];
            continue;
        }
        __builtin_amdgcn_s_sleep(8);
        acc += i;
    }
    if (mode == 2) acc += buf[threadIdx.x];
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    if (argc < 7) { printf("usage: wgrad_test T R G Cp O cpt [reps]\n"); return 1; }
    const int T = atoi(argv[1]); const long long R = atoll(argv[2]); const int G = atoi(argv[3]), Cp = atoi(argv[4]), O = atoi(argv[5]);
    const int cpt = atoi(argv[6]), reps = argc > 7 ? atoi(argv[7]) : 10;
    const long long PS = R * Cp, ZT = (long long)G * PS;
    const int M0 = G * Cp, M = M0 + 1;      // + the all-ones column of X (bias gradient row)
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> hX((size_t)T * ZT), hY((size_t)T * R * O);
    for (auto& v : hX) v = U(rng);
    for (auto& v : hY) v = U(rng);
    float *dX, *dYd, *dS;
    const int kch = (int)(((R + cpt - 1) / cpt + 31) / 32 * 32);
    const int nslab = T * cpt;
    CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dYd, hY.size() * 4)); CK(hipMalloc(&dS, (size_t)nslab * M * O * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dYd, hY.data(), hY.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dS, 0xFF, (size_t)nslab * M * O * 4));
    WgradP p; memset(&p, 0, sizeof p); p.X = dX; p.step_stride = ZT; p.PS = PS; p.Cp = Cp; p.G = G; p.T = T; p.R = R; p.dY = dYd; p.O = O; p.slabs = dS;
    p.cpt = cpt; p.kch = kch; p.ones = 1;
    hipError_t e = launch_wgrad_stream(p, 0);
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 2; }
    CK(hipDeviceSynchronize());
    std::vector<float> hS((size_t)nslab * M * O);
    CK(hipMemcpy(hS.data(), dS, hS.size() * 4, hipMemcpyDeviceToHost));
    // reference on sampled output rows
    double maxerr = 0, maxref = 0;
    std::vector<int> rows = {0, 1, Cp - 1, Cp, M0 / 2, M0 - 1, M0};
    for (int i = 0; i < 10; ++i) rows.push_back((int)(rng() % M0));
    for (int m : rows) {
        if (m < 0 || m >= M) continue;
        const int g = m < M0 ? m / Cp : 0, c = m < M0 ? m % Cp : 0;
        for (int o = 0; o < O; ++o) {
            double s = 0;
            for (int t = 0; t < T; ++t)
                for (long long r = 0; r < R; ++r) s += (m < M0 ? (double)hX[(size_t)t * ZT + (size_t)g * PS + r * Cp + c] : 1.0) * hY[((size_t)t * R + r) * O + o];
            double got = 0;
            for (int z = 0; z < nslab; ++z) got += hS[((size_t)z * M + m) * O + o];
            maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s));
        }
    }
    printf("T=%d R=%lld G=%d Cp=%d O=%d cpt=%d kch=%d: max|err|=%.3e max|ref|=%.3e rel %.2e %s\n", T, R, G, Cp, O, cpt, kch, maxerr, maxref,
           maxerr / maxref, maxerr / maxref < 2e-5 ? "OK" : "FAIL");
    if (getenv("CONC")) {
        // interference check: the same launch must give bit-identical slabs while an LDS-heavy GEMM runs on another stream
        const int nrep = atoi(getenv("CONC"));
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        const int GM = 2048; uint16_t *gA, *gB; float* gC;
        CK(hipMalloc(&gA, (size_t)GM * GM * 2)); CK(hipMalloc(&gB, (size_t)GM * GM * 2)); CK(hipMalloc(&gC, (size_t)GM * GM * 4));
        CK(hipMemset(gA, 0x3c, (size_t)GM * GM * 2)); CK(hipMemset(gB, 0x3c, (size_t)GM * GM * 2));
        Bf16GemmP g; memset(&g, 0, sizeof g);
        g.A = gA; g.B = gB; g.am = rm_plain(GM); g.bm = rm_plain(GM); g.ldb = GM; g.nseg = 1; g.seg_len = GM; g.M = GM; g.N = GM;
        g.C = gC; g.cm = rm_plain(GM); g.alpha = 1.f; g.xcd = 1;
        if (getenv("VICTIM")) {   // the bf16 GEMM as the kernel under test: configuration VICTIM, 1024^3, against the MFMA neighbour
            const int vc = atoi(getenv("VICTIM")), VM = 1024;
            Bf16GemmP v = g; v.M = VM; v.N = VM; v.seg_len = VM;
            std::vector<float> c0((size_t)VM * GM), c1((size_t)VM * GM);
            CK(hipMemsetAsync(gC, 0, (size_t)GM * GM * 4, s1));
            std::vector<uint16_t> ra((size_t)GM * GM); for (auto& x : ra) x = (uint16_t)(0x3c00 + (rng() & 0x1ff));
            CK(hipMemcpy(gA, ra.data(), ra.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(gB, ra.data(), ra.size() * 2, hipMemcpyHostToDevice));
            launch_gemm_bf16(v, getenv("DBTR") != nullptr, vc, 1, 0, s1); CK(hipDeviceSynchronize());
            CK(hipMemcpy(c0.data(), gC, c0.size() * 4, hipMemcpyDeviceToHost));
            int badv = 0;
            for (int it = 0; it < nrep; ++it) {
                hipLaunchKernelGGL(k_mfma_neighbour, dim3(1024), dim3(256), 0, s2, 400, (unsigned*)gB);
                launch_gemm_bf16(v, getenv("DBTR") != nullptr, vc, 1, 0, s1); CK(hipDeviceSynchronize());
                CK(hipMemcpy(c1.data(), gC, c1.size() * 4, hipMemcpyDeviceToHost));
                size_t nd = 0; for (size_t i = 0; i < c0.size(); ++i) nd += c0[i] != c1[i];
                if (nd) ++badv;
            }
            printf("   VICTIM gemm_bf16 cfg %d: %d of %d runs differ from the solo run\n", vc, badv, nrep);
            return 0;
        }
        std::vector<float> h2(hS.size());
        int bad = 0;
        for (int it = 0; it < nrep; ++it) {
            CK(hipMemsetAsync(dS, 0xFF, hS.size() * 4, s1));
            if (getenv("POISON")) { static bool a_ = false; if (!a_) { hipFuncSetAttribute((const void*)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160000); a_ = true; } hipLaunchKernelGGL(k_lds_poison, dim3(1024), dim3(1024), 160000, s1, (unsigned*)gC); }
            else if (getenv("MFMAN")) hipLaunchKernelGGL(k_mfma_neighbour, dim3(atoi(getenv("MFMAN"))), dim3(256), 0, s2, 400, (unsigned*)gC);
            else if (getenv("HOLD")) hipLaunchKernelGGL(k_lds_holder, dim3(getenv("HOLDN") ? atoi(getenv("HOLDN")) : 200), dim3(256), 0, s2, atoi(getenv("HOLD")), 400, (unsigned*)gC);
            else if (getenv("DCFG")) { for (int k = 0; k < 4; ++k) launch_gemm_bf16(g, getenv("DBTR") != nullptr, atoi(getenv("DCFG")), 1, 0, s2); }
            else for (int k = 0; k < 4; ++k) launch_gemm_bf16(g, (it & 1) != 0, it % 10, 1, 0, s2);
            launch_wgrad_stream(p, s1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h2.data(), dS, h2.size() * 4, hipMemcpyDeviceToHost));
            size_t nd = 0; double md = 0;
            for (size_t i = 0; i < h2.size(); ++i) if (memcmp(&h2[i], &hS[i], 4) != 0) { ++nd; md = fmax(md, fabs((double)h2[i] - hS[i])); }
            if (nd) {
                ++bad; printf("   conc iter %d: %zu of %zu slab values differ (max %.3e)\n", it, nd, h2.size(), md);
                { size_t qh[4] = {0, 0, 0, 0}, other = 0; const size_t slab = (size_t)M * O; for (size_t i = 0; i < h2.size(); ++i) if (memcmp(&h2[i], &hS[i], 4) != 0) { const size_t w = i % slab; if (w < 2048) qh[w / 512]++; else other++; } printf("      probe words differing: q0 %zu q1 %zu q2 %zu q3 %zu other %zu\n", qh[0], qh[1], qh[2], qh[3], other); }
                if (bad <= 2) {
                    unsigned hreg[256]; CK(hipMemcpyFromSymbol(hreg, HIP_SYMBOL(mcrn::g_wgs_dbg), sizeof hreg));
                    printf("      LDS_ALLOC of the differing chunks:"); 
                    std::vector<int> perz(nslab, 0), perm(M, 0), pero(O, 0);
                    for (size_t i = 0; i < h2.size(); ++i) if (h2[i] != hS[i]) { perz[i / ((size_t)M * O)]++; perm[(i / O) % M]++; pero[i % O]++; }
                    { std::vector<int> pz(nslab, 0); for (size_t i = 0; i < h2.size(); ++i) if (h2[i] != hS[i]) pz[i / ((size_t)M * O)]++; for (int z = 0; z < nslab && z < 256; ++z) if (pz[z]) printf(" %d:%08x", z, hreg[z]); printf("\n      LDS_ALLOC histogram of all chunks:"); std::vector<unsigned> seen; for (int z = 0; z < nslab && z < 256; ++z) { bool f = false; for (unsigned u : seen) f |= u == hreg[z]; if (!f) { seen.push_back(hreg[z]); int c = 0, cb = 0; for (int y = 0; y < nslab && y < 256; ++y) if (hreg[y] == hreg[z]) { ++c; cb += pz[y] != 0; } printf(" %08x x%d(bad %d)", hreg[z], c, cb); } } printf("\n"); }
                    printf("      chunks:"); for (int z = 0; z < nslab; ++z) if (perz[z]) printf(" %d(%d)", z, perz[z]); printf("\n      rows m:");
                    for (int m = 0; m < M; ++m) if (perm[m]) printf(" %d", m); printf("\n      cols o:");
                    for (int o = 0; o < O; ++o) if (pero[o]) printf(" %d", o); printf("\n");
                    size_t shown = 0;
                    for (size_t i = 0; i < h2.size() && shown < 6; ++i) if (h2[i] != hS[i]) { printf("      [%zu] solo %.6f conc %.6f\n", i, hS[i], h2[i]); ++shown; }
                }
            }
        }
        printf("   CONC: %d of %d runs differ from the solo run\n", bad, nrep);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // rotate over NROT copies of the operands (env, default 6): every launch streams data that is in no cache, as in a train step
    const int nrot = getenv("NROT") ? atoi(getenv("NROT")) : 6;
    std::vector<float*> rx(nrot), ry(nrot);
    for (int i = 0; i < nrot; ++i) {
        CK(hipMalloc(&rx[i], hX.size() * 4)); CK(hipMalloc(&ry[i], hY.size() * 4));
        CK(hipMemcpy(rx[i], dX, hX.size() * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ry[i], dYd, hY.size() * 4, hipMemcpyDeviceToDevice));
    }
    for (int i = 0; i < 3; ++i) launch_wgrad_stream(p, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) { p.X = rx[i % nrot]; p.dY = ry[i % nrot]; launch_wgrad_stream(p, 0); }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = ((double)T * ZT + (double)T * R * O) * 4 + (double)nslab * M * O * 4;
    printf("   %.2f us/launch  %.2f TB/s (operands + slabs = %.1f MB)\n", 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12, bytes / 1e6);
    return 0;
}
