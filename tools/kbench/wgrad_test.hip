// Stand-alone correctness + timing harness for megacrn_amd/csrc/wgrad_stream.h (no torch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wgrad_test wgrad_test.hip
//   ./wgrad_test T R G Cp O cpt reps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <random>
#include "../../megacrn_amd/csrc/wgrad_stream.h"
using namespace mcrn;
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
    WgradP p; p.X = dX; p.step_stride = ZT; p.PS = PS; p.Cp = Cp; p.G = G; p.T = T; p.R = R; p.dY = dYd; p.O = O; p.slabs = dS;
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
