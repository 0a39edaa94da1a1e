// wgrad_stream_api.h - parameter block and host entry point of the streaming weight-gradient kernel (wgrad_stream.h,
// compiled in wgrad_stream_unit.hip).  engine.hip includes only this file.
#pragma once
#include <hip/hip_runtime.h>

namespace mcrn {

// slabs[z][m][o] = sum over the rows k of chunk z of  X(k, m) * dY[k][o]          (model/MegaCRN.py:26-28, backward:
//   k = (t, r): t < T, r < R                                                         dW = X^T dY over every step and row)
//   m = (g, c): g < G, c < Cp     X(k, m) = X[t * step_stride + g * PS + r * Cp + c]
//   chunk z = (t, j < cpt): rows [j * kch, min(R, (j + 1) * kch)) of step t ; nslab = T * cpt
struct WgradP {
    const float* X;
    long long step_stride, PS;
    int Cp, G, T;
    long long R;
    const float* dY;          // [T * R][O]
    int O;
    float* slabs;             // [T * cpt][G * Cp + ones][O]
    int cpt, kch;             // kch % 32 == 0
    int ones;                 // 1: X gets a virtual all-ones column, i.e. row G*Cp of every slab = column sums of dY (the bias
                              //    gradient, model/MegaCRN.py:28) - dY is in LDS anyway, a separate column-sum pass re-read it
    // MCRN_BF16, bf16-resident propagated planes (wp_stream.h): when Xb != null, the state channels c < H of the planes
    // g >= 1 are read from  Xb[t * xb_step + (g - 1) * xb_plane + r * H + c]  (bf16) instead of X; plane 0 and the input
    // channels [H, Cp) of every plane still come from X.  (16 readable bytes are assumed behind every 8-byte quad.)
    const unsigned short* Xb;
    long long xb_step, xb_plane;
    int H;
    // ... and, when Xc != null, their input channels come from the compact array  Xc[t * xc_step + (g - 1) * xc_plane + r * 4 + 0..3]
    // (k_scatter_compact: d <= 4 channels + zeros): the quad at channel H of a plane g >= 1 is one row of it, later quads are zero
    const float* Xc;
    long long xc_step, xc_plane;
};

// shapes the kernel takes: O <= 512, O % 4 == 0, Cp % 4 == 0, 16-byte aligned bases
bool wgrad_stream_ok(int G, int Cp, int O);
hipError_t launch_wgrad_stream(const WgradP& p, hipStream_t st);

}  // namespace mcrn
