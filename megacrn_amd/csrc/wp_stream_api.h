// wp_stream_api.h - parameter block and host entry points of the streaming weight-pool kernel (wp_stream.h, compiled in
// wp_stream_unit.hip).  engine.hip includes only this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace mcrn {

enum WpEpi { WP_GATE = 1, WP_UPDATE = 2 };

// out = epi( [P_0 | P_1 .. P_nbp | input channels of every plane] x W + b )        (model/MegaCRN.py:26-27, :43-47)
//   P_0[r][c]   = Z[r*Cp + c]                         c < H    fp32 state channels of plane 0
//   P_g[r][c]   = Pb[(g-1)*PSh + r*H + c]             g >= 1   bf16 planes written by the propagation GEMM (Pb != NULL)
//               = Z[g*PS + r*Cp + c]                  g >= 1   fp32 planes (Pb == NULL: the bf16x3 parity mode)
//   in_g[r][j]  = Z[g*PS + r*Cp + H + j]              j < d    raw inputs (g = 0) and their hoisted propagation (g >= 1)
struct WpP {
    const float* Z;           // plane set (plane stride PS, row stride Cp)
    const uint16_t* Pb;       // bf16 planes 1 .. nbp, [nbp][R][H] ; NULL: the state channels of planes 1 .. nbp are fp32, in Z
    long long PS, PSh, R;
    int Cp, H, d, nbp, O;
    const uint4* Wimg;        // launch_wp_img_build
    const float* bias;        // [O]
    int epi;                  // WP_GATE: O = 2H ; WP_UPDATE: O = H
    float* out;               // GATE: zr [R][2H] = sigmoid(.) ; UPDATE: hc [R][H] = tanh(.)
    float* out2;              // GATE: z*h (columns < H), UPDATE: h' = r*h + (1-r)*hc ; row stride out2_ld
    long long out2_ld;
    uint16_t* out2b;          // nullable: the same values as packed bf16 [R][H] = the [N][B*H] operand of the next propagation GEMM
    long long out2b_lo;       // > 0 (a bf16x3 session on the resident data flow, x3r): the rounding residuals as a second bf16 image that many elements behind
    const float* hsrc;        // UPDATE: previous state h[r*hsrc_ld + c]   (GATE reads it from plane 0 of Z)
    long long hsrc_ld;
    const float* zr;          // UPDATE: the gate call's `out`
    const float* Xc;          // nullable: compact input channels of planes 1 .. nbp, [nbp][R][4] (k_scatter_compact) - then Z is read
    long long xc_plane;       //           for plane 0 only; stride between planes in floats (= 4 R)
    int ncb;                  // set by the launcher: column blocks per row block (O / (32 NBF))
};

bool wp_stream_ok(int H, int d, int nbp, int O);
size_t wp_img_uint4(int H, int nbp, int O);                                  // uint4 elements of the weight image
// Wf: the prepared weights [(1 + nbp) * Cp][O] (k_wprep layout: row (g, c') = g*Cp + c')
// (R = rows of the launches that will use the image: the column-block width of the image depends on it)
// nbf_force > 0: that many 32-column fragments per column block (agcn_fused.h consumes blocks of 2)
hipError_t launch_wp_img_build(const float* Wf, int Cp, int H, int d, int nbp, int O, long long R, uint4* img, hipStream_t st, int nbf_force = 0);
hipError_t launch_wp_stream(const WpP& p, hipStream_t st);

}  // namespace mcrn
