// ops.h - the non-GEMM kernels of the MegaCRN hot path (gfx950, wave64).
// Everything here is HBM/L2-bound elementwise, row-softmax or reduction work.
// Internal activation layout is node-major: row r = n*B + b, planes Z[g][r][Cp].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcrn {

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// plane-0 packing
// ---------------------------------------------------------------------------------------------
// dst[t][r][col0 + j] = src[b*sb + t*st + n*sn + j]   for j < w, r = n*B + b, t < T
// (dst_t = per-step stride of the destination buffer).  Used for x / ycov input channels.
__global__ void k_fill_cols(float* __restrict__ dst, long long dst_t, int Cp, int col0, int w,
                            const float* __restrict__ src, long long sb, long long st, long long sn,
                            int B, int N, int T) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)T * N * B * w;
    if (i >= tot) return;
    int j = (int)(i % w);
    long long q = i / w;
    int b = (int)(q % B); q /= B;
    int n = (int)(q % N);
    int t = (int)(q / N);
    dst[t * dst_t + ((long long)n * B + b) * Cp + col0 + j] = src[b * sb + t * st + n * sn + j];
}

// dst[t][r][c] = 0 for c in [c0, c1)
__global__ void k_zero_cols(float* __restrict__ dst, long long dst_t, int Cp, int c0, int c1,
                            long long R, int T) {
    int w = c1 - c0;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)T * R * w;
    if (i >= tot) return;
    int j = (int)(i % w);
    long long q = i / w;
    long long r = q % R;
    int t = (int)(q / R);
    dst[t * dst_t + r * Cp + c0 + j] = 0.f;
}

// batch-major (B,N,w) <-> node-major columns [col0, col0+w) of a (R x ld) buffer
__global__ void k_bnc_to_rows(float* __restrict__ dst, long long ld, int col0, int w,
                              const float* __restrict__ src, int B, int N) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)B * N * w;
    if (i >= tot) return;
    int j = (int)(i % w);
    long long q = i / w;
    int n = (int)(q % N);
    int b = (int)(q / N);
    dst[((long long)n * B + b) * ld + col0 + j] = src[i];
}
__global__ void k_rows_to_bnc(float* __restrict__ dst, const float* __restrict__ src, long long ld,
                              int col0, int w, int B, int N, int accumulate) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)B * N * w;
    if (i >= tot) return;
    int j = (int)(i % w);
    long long q = i / w;
    int n = (int)(q % N);
    int b = (int)(q / N);
    float v = src[((long long)n * B + b) * ld + col0 + j];
    dst[i] = accumulate ? dst[i] + v : v;
}

// ---------------------------------------------------------------------------------------------
// AGCN weight layout:  reference rows k_global*C + c_ref  <->  internal rows (plane g, channel c')
//   c' <  H        -> c_ref = d + c'      (state channels; the reference concatenates [x | h])
//   c' in [H, H+d) -> c_ref = c' - H      (input channels)
//   c' >= H+d      -> zero padding
// plane 0 merges the two identity blocks; Wd is the d-grad variant with the Chebyshev recursion
// folded in for cheb_k = 3 (plane0 -= W_2 blocks, k=2 planes *= 2).
// ---------------------------------------------------------------------------------------------
__global__ void k_wprep(const float* __restrict__ W, float* __restrict__ Wf, float* __restrict__ Wd,
                        int d, int H, int Cp, int K, int O, int fold) {
    const int G = 2 * K - 1, C = d + H;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)G * Cp * O;
    if (i >= tot) return;
    int o = (int)(i % O);
    int q = (int)(i / O);
    int cp = q % Cp, g = q / Cp;
    float wf = 0.f, wd = 0.f;
    if (cp < C) {
        int cref = cp < H ? d + cp : cp - H;
        if (g == 0) {
            float w0 = W[((long long)(0 * K + 0) * C + cref) * O + o];
            float w1 = W[((long long)(1 * K + 0) * C + cref) * O + o];
            wf = w0 + w1;
            wd = wf;
            if (K == 3 && fold) {
                wd -= W[((long long)(0 * K + 2) * C + cref) * O + o];
                wd -= W[((long long)(1 * K + 2) * C + cref) * O + o];
            }
        } else {
            int s = (g - 1) / (K - 1), k = 1 + (g - 1) % (K - 1);
            wf = W[((long long)(s * K + k) * C + cref) * O + o];
            wd = (k == 2 && fold) ? 2.f * wf : wf;   // fold == 0 (MCRN_BF16): the planes are T_k(S) x, no recursion to fold
        }
    }
    Wf[i] = wf;
    Wd[i] = wd;
}

// dW (reference layout) = sum over slabs of dW' (internal layout), identity plane copied to both supports.
// 1024 threads = 64 consecutive outputs x 16 slab lanes (up to 256 slabs from the streaming weight-gradient kernel: one
// thread walking them serially made this a 20 us tail); lane partials are added in a fixed order.
__global__ __launch_bounds__(1024) void k_wunprep(float* __restrict__ dW, const float* __restrict__ slabs, int nslab,
                                                  long long slab, int d, int H, int Cp, int K, int O, float* __restrict__ dbias) {
    __shared__ float sh[16][64];
    const int C = d + H;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + el;
    const long long tot = (long long)2 * K * C * O;
    const long long all = tot + (dbias ? O : 0);                  // slab row G*Cp: column sums of dY = the bias gradient
    float v = 0.f;
    if (i < all) {
        long long srcoff;
        if (i < tot) {
            int o = (int)(i % O);
            int q = (int)(i / O);
            int cref = q % C, kg = q / C;
            int s = kg / K, k = kg % K;
            int g = (k == 0) ? 0 : 1 + s * (K - 1) + (k - 1);
            int cp = cref < d ? H + cref : cref - d;
            srcoff = ((long long)g * Cp + cp) * O + o;
        } else {
            const int G = 1 + 2 * (K - 1);
            srcoff = (long long)G * Cp * O + (i - tot);
        }
        const float* src = slabs + srcoff;
        for (int z = zl; z < nslab; z += 16) v += src[z * slab];
    }
    sh[zl][el] = v;
    __syncthreads();
    if (zl == 0 && i < all) {
        float t = 0.f;
#pragma unroll
        for (int z = 0; z < 16; ++z) t += sh[z][el];
        if (i < tot) dW[i] = t;
        else dbias[i - tot] = t;
    }
}

// The same reduction with FOUR consecutive outputs per thread (O % 4 == 0: four consecutive reference elements share their
// (support, k, channel) row, so they are four consecutive slab elements too): 16-byte loads, a wave reads 1 KB of a slab per
// instruction instead of 256 B - the 256-slab reduction of the N = 1843 decoder gates was running at ~270 GB/s
// (profiles/r3: 174 us x 4 per step on the helper stream).
__global__ __launch_bounds__(1024) void k_wunprep4(float* __restrict__ dW, const float* __restrict__ slabs, int nslab,
                                                   long long slab, int d, int H, int Cp, int K, int O, float* __restrict__ dbias) {
    __shared__ float4 sh[16][64];
    const int C = d + H;
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long long i = ((long long)blockIdx.x * 64 + el) * 4;
    const long long tot = (long long)2 * K * C * O;
    const long long all = tot + (dbias ? O : 0);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < all) {
        long long srcoff;
        if (i < tot) {
            int o = (int)(i % O);
            int q = (int)(i / O);
            int cref = q % C, kg = q / C;
            int s = kg / K, k = kg % K;
            int g = (k == 0) ? 0 : 1 + s * (K - 1) + (k - 1);
            int cp = cref < d ? H + cref : cref - d;
            srcoff = ((long long)g * Cp + cp) * O + o;
        } else {
            const int G = 1 + 2 * (K - 1);
            srcoff = (long long)G * Cp * O + (i - tot);
        }
        const float* src = slabs + srcoff;
        for (int z = zl; z < nslab; z += 16) {
            const float4 x = *reinterpret_cast<const float4*>(src + z * slab);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
    }
    sh[zl][el] = v;
    __syncthreads();
    if (zl == 0 && i < all) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int z = 0; z < 16; ++z) { const float4 x = sh[z][el]; t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w; }
        float* dst = i < tot ? dW + i : dbias + (i - tot);
        *reinterpret_cast<float4*>(dst) = t;
    }
}

// ---------------------------------------------------------------------------------------------
// adjacency: g = rowsoftmax(relu(L))  (model/MegaCRN.py:171-172), one wave per row
// ---------------------------------------------------------------------------------------------
// WPR waves per row (a 256-thread workgroup holds 4 / WPR rows): one wave per row leaves a large graph's launch at ~2
// waves per SIMD, each walking 29+ elements per lane through three dependent passes (N = 1843: 25 us forward, 63 us
// backward); with the whole workgroup on one row the same passes are 4x shorter and 4x as many waves hide the latency.
template <int WPR>
__device__ __forceinline__ float rows_reduce(float v, bool is_max, float* red, int wave) {
    v = is_max ? wave_max(v) : wave_sum(v);
    if (WPR == 1) return v;
    __syncthreads();                       // (the previous reduction's readers are done with `red`)
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < WPR; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
template <int WPR>
__global__ void k_relu_softmax_rows(const float* __restrict__ L, long long ldl, float* __restrict__ G,
                                    long long ldg, int N) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (4 / WPR) + wave / WPR;
    const int lane = (wave % WPR) * 64 + (threadIdx.x & 63);
    constexpr int STEP = 64 * WPR;
    if (WPR == 1 && row >= N) return;      // (WPR == 4: one row per workgroup, the grid is exactly N)
    const float* l = L + (long long)row * ldl;
    float mx = 0.f;  // relu output is >= 0
    for (int c = lane; c < N; c += STEP) mx = fmaxf(mx, l[c]);
    mx = rows_reduce<WPR>(mx, true, red, wave);
    float s = 0.f;
    for (int c = lane; c < N; c += STEP) s += expf(fmaxf(l[c], 0.f) - mx);
    s = rows_reduce<WPR>(s, false, red, wave);
    float inv = 1.f / s;
    float* g = G + (long long)row * ldg;
    for (int c = lane; c < N; c += STEP) g[c] = expf(fmaxf(l[c], 0.f) - mx) * inv;
}

// dL = g * (dS - sum(dS*g)) * [L > 0], dS = sum over slabs
template <int WPR>
__global__ void k_relu_softmax_rows_bwd(const float* __restrict__ L, long long ldl,
                                        const float* __restrict__ G, long long ldg,
                                        const float* __restrict__ dS, long long ldd, int nslab,
                                        long long slab, float* __restrict__ dL, long long ldo, int N) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (4 / WPR) + wave / WPR;
    const int lane = (wave % WPR) * 64 + (threadIdx.x & 63);
    constexpr int STEP = 64 * WPR;
    if (WPR == 1 && row >= N) return;
    const float* g = G + (long long)row * ldg;
    const float* l = L + (long long)row * ldl;
    const float* d = dS + (long long)row * ldd;
    float dot = 0.f;
    for (int c = lane; c < N; c += STEP) {
        float v = 0.f;
        for (int z = 0; z < nslab; ++z) v += d[z * slab + c];
        dot += v * g[c];
    }
    dot = rows_reduce<WPR>(dot, false, red, wave);
    float* o = dL + (long long)row * ldo;
    for (int c = lane; c < N; c += STEP) {
        float v = 0.f;
        for (int z = 0; z < nslab; ++z) v += d[z * slab + c];
        o[c] = l[c] > 0.f ? g[c] * (v - dot) : 0.f;
    }
}

// dst[i][j] = (add ? add[i][j] : 0) + src[j][i]   (N x N, LDS-tiled 32x32 transpose)
__global__ void k_transpose_add(float* __restrict__ dst, long long ldd, const float* __restrict__ src,
                                long long lds_, const float* __restrict__ add, long long lda, int N) {
    __shared__ float t[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int k = ty; k < 32; k += 8) {
        int r = bx + k, c = by + tx;                     // src[r][c], r in x-tile (becomes dst column)
        t[k][tx] = (r < N && c < N) ? src[(long long)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int i = by + k, j = bx + tx;                     // dst[i][j] = src[j][i]
        if (i < N && j < N) {
            float v = t[tx][k];
            if (add) v += add[(long long)i * lda + j];
            dst[(long long)i * ldd + j] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// memory head (model/MegaCRN.py:159-166), thread per row r = n*B + b
// LDS: Wq (H*D) | Mem (M*D) | per-thread scratch (D + M) x blockDim (thread fastest)
// ---------------------------------------------------------------------------------------------
__global__ void k_memory_fwd(const float* __restrict__ h, long long ldh, const float* __restrict__ Wq,
                             const float* __restrict__ Mem, int B, int N, int H, int M, int D,
                             float* __restrict__ q_rows, float* __restrict__ att_rows,
                             int* __restrict__ ind_rows,
                             float* __restrict__ s0, long long lds0,       // decoder state [h | value] (nullable)
                             float* __restrict__ val_bnc, float* __restrict__ q_bnc,
                             float* __restrict__ pos_bnc, float* __restrict__ neg_bnc,
                             int* __restrict__ ind_bnc) {
    extern __shared__ float sm[];
    float* sWq = sm;
    float* sMem = sWq + H * D;
    float* scr = sMem + M * D;
    const int nt = blockDim.x, t = threadIdx.x;
    for (int i = t; i < H * D; i += nt) sWq[i] = Wq[i];
    for (int i = t; i < M * D; i += nt) sMem[i] = Mem[i];
    __syncthreads();
    long long R = (long long)N * B;
    long long r = (long long)blockIdx.x * nt + t;
    if (r >= R) return;
    float* q = scr;              // q[d*nt + t]
    float* sc = scr + D * nt;    // sc[m*nt + t]
    const float* hr = h + r * ldh;
    for (int d = 0; d < D; ++d) q[d * nt + t] = 0.f;
    for (int c = 0; c < H; ++c) {
        float hv = hr[c];
        for (int d = 0; d < D; ++d) q[d * nt + t] += hv * sWq[c * D + d];
    }
    float mx = -3.4e38f;
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += q[d * nt + t] * sMem[m * D + d];
        sc[m * nt + t] = s;
        mx = fmaxf(mx, s);
    }
    float den = 0.f;
    for (int m = 0; m < M; ++m) { float e = expf(sc[m * nt + t] - mx); sc[m * nt + t] = e; den += e; }
    float inv = 1.f / den;
    int i0 = 0, i1 = -1; float b0 = -1.f, b1 = -1.f;
    for (int m = 0; m < M; ++m) {
        float a = sc[m * nt + t] * inv;
        sc[m * nt + t] = a;
        att_rows[r * M + m] = a;
        if (a > b0) { b1 = b0; i1 = i0; b0 = a; i0 = m; }
        else if (a > b1) { b1 = a; i1 = m; }
    }
    if (i1 < 0) i1 = i0;
    ind_rows[r * 2] = i0; ind_rows[r * 2 + 1] = i1;
    int n = (int)(r / B), b = (int)(r % B);
    long long o = ((long long)b * N + n) * D;
    if (ind_bnc) { ind_bnc[((long long)b * N + n) * 2] = i0; ind_bnc[((long long)b * N + n) * 2 + 1] = i1; }
    for (int d = 0; d < D; ++d) {
        float v = 0.f;
        for (int m = 0; m < M; ++m) v += sc[m * nt + t] * sMem[m * D + d];
        float qq = q[d * nt + t];
        q_rows[r * D + d] = qq;
        val_bnc[o + d] = v;
        q_bnc[o + d] = qq;
        pos_bnc[o + d] = sMem[i0 * D + d];
        neg_bnc[o + d] = sMem[i1 * D + d];
        if (s0) s0[r * lds0 + H + d] = v;
    }
    if (s0) for (int c = 0; c < H; ++c) s0[r * lds0 + c] = hr[c];
}

// memory head, row part after the GEMMs q = h Wq and sc = q Mem^T: softmax over M, top-2, value = att Mem,
// all outputs.  One wave per row: lanes over M for the softmax, lanes over D for the value / copies.
__global__ void k_memory_rows(const float* __restrict__ q_rows, const float* __restrict__ sc_rows,
                              const float* __restrict__ Mem, const float* __restrict__ h, long long ldh,
                              int B, int N, int H, int M, int D, float* __restrict__ att_rows,
                              int* __restrict__ ind_rows, float* __restrict__ s0, long long lds0,
                              float* __restrict__ val_bnc, float* __restrict__ q_bnc, float* __restrict__ pos_bnc,
                              float* __restrict__ neg_bnc, int* __restrict__ ind_bnc) {
    const long long R = (long long)N * B;
    const int wpb = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += (long long)gridDim.x * wpb) {
        if (M <= 64) {
            // M <= 64 (every configuration of the reference: mem_num = 20): lane m owns attention weight m - ONE exponential per
            // lane, the weights travel by lane shuffles (the general form below evaluates 3 M exponentials per lane: 79 us at
            // N = 1843, compute-bound).  Same expressions, same order of additions: bit-identical outputs.
            const float scv = lane < M ? sc_rows[r * M + lane] : -3.4e38f;
            const float mx = wave_max(scv);
            const float e = lane < M ? expf(scv - mx) : 0.f;
            float den = wave_sum(e);
            const float inv = 1.f / den;
            const float a_own = e * inv;
            if (lane < M) att_rows[r * M + lane] = a_own;
            int i0 = 0, i1 = -1; float b0 = -1.f, b1 = -1.f;
            for (int m = 0; m < M; ++m) {
                const float a = __shfl(a_own, m);
                if (a > b0) { b1 = b0; i1 = i0; b0 = a; i0 = m; }
                else if (a > b1) { b1 = a; i1 = m; }
            }
            if (i1 < 0) i1 = i0;
            const int n = (int)(r / B), b = (int)(r % B);
            const long long o = ((long long)b * N + n) * D;
            if (lane == 0) {
                ind_rows[r * 2] = i0; ind_rows[r * 2 + 1] = i1;
                if (ind_bnc) { ind_bnc[((long long)b * N + n) * 2] = i0; ind_bnc[((long long)b * N + n) * 2 + 1] = i1; }
            }
            for (int d0 = 0; d0 < D; d0 += 64) {                 // (uniform trip count: the shuffles need every lane)
                const int d = d0 + lane;
                const int dc = d < D ? d : D - 1;
                float v = 0.f;
                for (int m = 0; m < M; ++m) v += __shfl(a_own, m) * Mem[m * D + dc];
                if (d < D) {
                    const float qq = q_rows[r * D + d];
                    val_bnc[o + d] = v; q_bnc[o + d] = qq;
                    pos_bnc[o + d] = Mem[i0 * D + d]; neg_bnc[o + d] = Mem[i1 * D + d];
                    if (s0) s0[r * lds0 + H + d] = v;
                }
            }
            if (s0) for (int c = lane; c < H; c += 64) s0[r * lds0 + c] = h[r * ldh + c];
            continue;
        }
        // softmax over M (M <= 64 handled in one pass per lane; larger M loops)
        float mx = -3.4e38f;
        for (int m = lane; m < M; m += 64) mx = fmaxf(mx, sc_rows[r * M + m]);
        mx = wave_max(mx);
        float den = 0.f;
        for (int m = lane; m < M; m += 64) den += expf(sc_rows[r * M + m] - mx);
        den = wave_sum(den);
        const float inv = 1.f / den;
        // top-2 (largest first, lowest index on ties): every lane scans, M is small
        int i0 = 0, i1 = -1; float b0 = -1.f, b1 = -1.f;
        for (int m = 0; m < M; ++m) {
            const float a = expf(sc_rows[r * M + m] - mx) * inv;
            if (a > b0) { b1 = b0; i1 = i0; b0 = a; i0 = m; }
            else if (a > b1) { b1 = a; i1 = m; }
            if (lane == 0) att_rows[r * M + m] = a;
        }
        if (i1 < 0) i1 = i0;
        const int n = (int)(r / B), b = (int)(r % B);
        const long long o = ((long long)b * N + n) * D;
        if (lane == 0) {
            ind_rows[r * 2] = i0; ind_rows[r * 2 + 1] = i1;
            if (ind_bnc) { ind_bnc[((long long)b * N + n) * 2] = i0; ind_bnc[((long long)b * N + n) * 2 + 1] = i1; }
        }
        for (int d = lane; d < D; d += 64) {
            float v = 0.f;
            for (int m = 0; m < M; ++m) v += (expf(sc_rows[r * M + m] - mx) * inv) * Mem[m * D + d];
            const float qq = q_rows[r * D + d];
            val_bnc[o + d] = v; q_bnc[o + d] = qq;
            pos_bnc[o + d] = Mem[i0 * D + d]; neg_bnc[o + d] = Mem[i1 * D + d];
            if (s0) s0[r * lds0 + H + d] = v;
        }
        if (s0) for (int c = lane; c < H; c += 64) s0[r * lds0 + c] = h[r * ldh + c];
    }
}

// per row: datt = dval Mem^T ; dsc = att*(datt - sum(att*datt)) ; dq = dsc Mem + dq_ext
// dval = dval_rows[r*ldv + c0 + d] (nullable) + dval_bnc (nullable)
__global__ void k_memory_bwd_rows(const float* __restrict__ dval_rows, long long ldv, int c0,
                                  const float* __restrict__ dval_bnc, const float* __restrict__ dq_bnc,
                                  const float* __restrict__ att_rows, const float* __restrict__ Mem,
                                  int B, int N, int M, int D, float* __restrict__ dval_out,
                                  float* __restrict__ dsc_rows, float* __restrict__ dq_rows) {
    extern __shared__ float sm[];
    float* sMem = sm;
    float* scr = sMem + M * D;
    const int nt = blockDim.x, t = threadIdx.x;
    for (int i = t; i < M * D; i += nt) sMem[i] = Mem[i];
    __syncthreads();
    long long R = (long long)N * B;
    long long r = (long long)blockIdx.x * nt + t;
    if (r >= R) return;
    int n = (int)(r / B), b = (int)(r % B);
    long long o = ((long long)b * N + n) * D;
    float* dv = scr;             // dv[d*nt+t]
    float* da = scr + D * nt;    // da[m*nt+t]
    for (int d = 0; d < D; ++d) {
        float v = 0.f;
        if (dval_rows) v += dval_rows[r * ldv + c0 + d];
        if (dval_bnc) v += dval_bnc[o + d];
        dv[d * nt + t] = v;
        dval_out[r * D + d] = v;
    }
    float dot = 0.f;
    for (int m = 0; m < M; ++m) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += dv[d * nt + t] * sMem[m * D + d];
        da[m * nt + t] = s;
        dot += s * att_rows[r * M + m];
    }
    for (int m = 0; m < M; ++m) {
        float v = att_rows[r * M + m] * (da[m * nt + t] - dot);
        da[m * nt + t] = v;
        dsc_rows[r * M + m] = v;
    }
    for (int d = 0; d < D; ++d) {
        float v = dq_bnc ? dq_bnc[o + d] : 0.f;
        for (int m = 0; m < M; ++m) v += da[m * nt + t] * sMem[m * D + d];
        dq_rows[r * D + d] = v;
    }
}

// Same, one WAVE per row (M <= 64, D <= 256): lanes over d for the two contractions with Mem (staged in LDS once per
// workgroup), lanes over m for the softmax backward.  The thread-per-row kernel above ran 207 single-wave workgroups
// with 2*M*D serial multiply-adds per thread: 94 us at METR-LA.
__global__ __launch_bounds__(256) void k_memory_bwd_rows_w(const float* __restrict__ dval_rows, long long ldv, int c0,
                                                           const float* __restrict__ dval_bnc, const float* __restrict__ dq_bnc,
                                                           const float* __restrict__ att_rows, const float* __restrict__ Mem,
                                                           int B, int N, int M, int D, float* __restrict__ dval_out,
                                                           float* __restrict__ dsc_rows, float* __restrict__ dq_rows) {
    extern __shared__ float sMem[];
    for (int i = threadIdx.x; i < M * D; i += blockDim.x) sMem[i] = Mem[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const long long R = (long long)N * B;
    for (long long r = (long long)blockIdx.x * wpb + wave; r < R; r += (long long)gridDim.x * wpb) {
        const int n = (int)(r / B), b = (int)(r % B);
        const long long o = ((long long)b * N + n) * D;
        float dv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane + 64 * j;
            float v = 0.f;
            if (d < D) {
                if (dval_rows) v += dval_rows[r * ldv + c0 + d];
                if (dval_bnc) v += dval_bnc[o + d];
                dval_out[r * D + d] = v;
            }
            dv[j] = v;
        }
        float da = 0.f;                                   // lane m holds datt[m] = sum_d dv[d] Mem[m][d]
        for (int m = 0; m < M; ++m) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = lane + 64 * j;
                if (d < D) part += dv[j] * sMem[m * D + d];
            }
            const float sm = wave_sum(part);
            if (lane == m) da = sm;
        }
        const float att = lane < M ? att_rows[r * M + lane] : 0.f;
        const float dot = wave_sum(da * att);
        const float dsc = att * (da - dot);
        if (lane < M) dsc_rows[r * M + lane] = dsc;
        float dq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane + 64 * j;
            dq[j] = (dq_bnc && d < D) ? dq_bnc[o + d] : 0.f;
        }
        for (int m = 0; m < M; ++m) {
            const float w = __shfl(dsc, m, 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = lane + 64 * j;
                if (d < D) dq[j] += w * sMem[m * D + d];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane + 64 * j;
            if (d < D) dq_rows[r * D + d] = dq[j];
        }
    }
}

// dMem[ind[r][which]] += dsel[b,n,:]   (only when the caller did not detach pos/neg)
__global__ void k_memory_scatter(float* __restrict__ dMem, const int* __restrict__ ind_rows, int which,
                                 const float* __restrict__ dsel_bnc, int B, int N, int D) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)N * B * D;
    if (i >= tot) return;
    int d = (int)(i % D);
    long long r = i / D;
    int n = (int)(r / B), b = (int)(r % B);
    atomicAdd(&dMem[(long long)ind_rows[r * 2 + which] * D + d], dsel_bnc[((long long)b * N + n) * D + d]);
}

// ---------------------------------------------------------------------------------------------
// decoder projection (model/MegaCRN.py:186-191): one wave per row, lanes over channels
// go = h' Wp^T + bp -> out[b,t,n,:] ; next step's `go` input channels (teacher forcing folded in)
// ---------------------------------------------------------------------------------------------
__global__ void k_proj_fwd(const float* __restrict__ hn, long long ldh, const float* __restrict__ Wp,
                           const float* __restrict__ bp, int Hd, int od, int B, int N,
                           float* __restrict__ out_bt, long long out_sb, long long out_sn,   // out[b*sb + n*sn + j]
                           float* __restrict__ nextZ, float* __restrict__ nextY, long long ldn, int col0,
                           const float* __restrict__ lab_bt /* labels[:,t] or null */) {
    long long R = (long long)N * B;
    int wpb = blockDim.x >> 6;
    int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += (long long)gridDim.x * wpb) {
        int n = (int)(r / B), b = (int)(r % B);
        for (int j = 0; j < od; ++j) {
            float s = 0.f;
            for (int c = lane; c < Hd; c += 64) s += hn[r * ldh + c] * Wp[(long long)j * Hd + c];
            s = wave_sum(s);
            if (lane == 0) {
                float go = s + bp[j];
                long long o = b * out_sb + n * out_sn + j;
                out_bt[o] = go;
                if (nextZ) {
                    float nx = lab_bt ? lab_bt[o] : go;
                    nextZ[r * ldn + col0 + j] = nx;
                    nextY[r * ldn + col0 + j] = nx;
                }
            }
        }
    }
}

// dgo[r][j] = d_out[b,t,n,j] + (use_next ? dxin[r][j] : 0) ; dhn[r][c] = dstate[r][c] + sum_j dgo*Wp[j][c]
__global__ void k_proj_bwd(const float* __restrict__ dout_bt, long long out_sb, long long out_sn,
                           const float* __restrict__ dxin, long long ldx, int use_next,
                           const float* __restrict__ Wp, int Hd, int od, int B, int N,
                           const float* __restrict__ dstate, float* __restrict__ dhn,
                           float* __restrict__ dgo_rows) {
    long long R = (long long)N * B;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Hd) return;
    int c = (int)(i % Hd);
    long long r = i / Hd;
    int n = (int)(r / B), b = (int)(r % B);
    float acc = dstate ? dstate[i] : 0.f;
    for (int j = 0; j < od; ++j) {
        float g = dout_bt[b * out_sb + n * out_sn + j];
        if (use_next) g += dxin[r * ldx + j];
        acc += g * Wp[(long long)j * Hd + c];
        if (c == 0) dgo_rows[r * od + j] = g;
    }
    dhn[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// GRU backward elementwise (SURVEY.md A.3 "cell")
// ---------------------------------------------------------------------------------------------
// A: dr, dhc from dh' ; dU = dhc*(1-hc^2) ; dG[:,H:] = dr*r*(1-r) ; dacc = dh'*r
__global__ void k_cell_bwd_a(const float* __restrict__ dhn, const float* __restrict__ z0, long long ldz,
                             const float* __restrict__ zr, const float* __restrict__ hc, int H, long long R,
                             float* __restrict__ dU, float* __restrict__ dG, float* __restrict__ dacc) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * H) return;
    int c = (int)(i % H);
    long long r = i / H;
    float g = dhn[i];
    float h = z0[r * ldz + c];
    float rr = zr[r * 2 * H + H + c];
    float hv = hc[i];
    dU[i] = g * (1.f - rr) * (1.f - hv * hv);
    dG[r * 2 * H + H + c] = g * (h - hv) * rr * (1.f - rr);
    dacc[i] = g * rr;
}
// B: dzh = dY0[state part] ; dG[:, :H] = dzh*h*z*(1-z) ; dacc += dzh*z
// (dy0x: `nx` extra planes `xs` floats apart that hold further partial sums of plane 0: the second support's S^T
//  contribution of prop2_bwd_kernel, or the K splits 1 .. nx of the bf16 transposed propagation)
__global__ void k_cell_bwd_b(const float* __restrict__ dy0, const float* __restrict__ dy0x, int nx, long long xs, long long ldy,
                             const float* __restrict__ z0, long long ldz, const float* __restrict__ zr, int H,
                             long long R, float* __restrict__ dG, float* __restrict__ dacc) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * H) return;
    int c = (int)(i % H);
    long long r = i / H;
    float dzh = dy0[r * ldy + c];
    for (int e = 0; e < nx; ++e) dzh += dy0x[e * xs + r * ldy + c];
    float h = z0[r * ldz + c];
    float z = zr[r * 2 * H + c];
    dG[r * 2 * H + c] = dzh * h * z * (1.f - z);
    dacc[i] += dzh * z;
}
// C: dh_prev = dacc + dZ0[state] ; dxin = dY0[input] + dZ0[input]
// (xcols: the extra partial planes cover columns < xcols only - all Cp columns, or just the state channels when the
//  transposed propagation was hoisted)
__global__ void k_cell_bwd_c(const float* __restrict__ dz0, const float* __restrict__ dz0x, int nzx,
                             const float* __restrict__ dy0, const float* __restrict__ dy0x, int nyx, long long xs, int xcols, long long ld,
                             int H, int d, long long R, float* __restrict__ dacc, float* __restrict__ dxin) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int C = H + d;
    if (i >= R * C) return;
    int c = (int)(i % C);
    long long r = i / C;
    float a = dz0[r * ld + c];
    if (c < xcols) for (int e = 0; e < nzx; ++e) a += dz0x[e * xs + r * ld + c];
    if (c < H) dacc[r * H + c] += a;
    else {
        float b = dy0[r * ld + c];
        if (c < xcols) for (int e = 0; e < nyx; ++e) b += dy0x[e * xs + r * ld + c];
        dxin[r * d + (c - H)] = a + b;
    }
}

// C(step t+1) + [decoder: projection backward of step t] + A(step t) in ONE launch: the three are element-wise over
// the same (row, state channel) grid and adjacent in the BPTT loop, so interior steps save two launches per cell
// (the decoder-input gradient dxin[r][j] that the projection needs is recomputed from the four plane entries).
//   dh      = dacc + dZ0[state]                                    (C)
//   dgo_j   = d_out[b,t,n,j] + (use_next ? dY0[in j] + dZ0[in j] : 0) ; dh += sum_j dgo_j Wp[j][c]   (k_proj_bwd)
//   dU, dG[:,H:], dacc from dh and step t's saved z0 / zr / hc     (A)
__global__ void k_cell_bwd_ca(const float* __restrict__ dz0, const float* __restrict__ dz0x, int nzx,
                              const float* __restrict__ dy0, const float* __restrict__ dy0x, int nyx, long long xs, int xcols, long long ld,
                              const float* __restrict__ dout_bt, long long out_sb, long long out_sn, int use_next,
                              const float* __restrict__ Wp, int od, float* __restrict__ dgo_rows, int B,
                              const float* __restrict__ z0, long long ldz, const float* __restrict__ zr,
                              const float* __restrict__ hc, int H, long long R,
                              float* __restrict__ dU, float* __restrict__ dG, float* __restrict__ dacc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * H) return;
    const int c = (int)(i % H);
    const long long r = i / H;
    float a = dz0[r * ld + c];
    for (int e = 0; e < nzx; ++e) a += dz0x[e * xs + r * ld + c];
    float g = dacc[i] + a;
    if (Wp) {
        const int n = (int)(r / B), b = (int)(r % B);
        for (int j = 0; j < od; ++j) {
            float go = dout_bt[b * out_sb + n * out_sn + j];
            if (use_next) {
                float x = dz0[r * ld + H + j] + dy0[r * ld + H + j];
                if (H + j < xcols) {
                    for (int e = 0; e < nzx; ++e) x += dz0x[e * xs + r * ld + H + j];
                    for (int e = 0; e < nyx; ++e) x += dy0x[e * xs + r * ld + H + j];
                }
                go += x;
            }
            g += go * Wp[(long long)j * H + c];
            if (c == 0) dgo_rows[r * od + j] = go;
        }
    }
    const float h = z0[r * ldz + c];
    const float rr = zr[r * 2 * H + H + c];
    const float hv = hc[i];
    dU[i] = g * (1.f - rr) * (1.f - hv * hv);
    dG[r * 2 * H + H + c] = g * (h - hv) * rr * (1.f - rr);
    dacc[i] = g * rr;
}

// ---- float4 forms of B and CA (H % 4 == 0; every row stride here is a multiple of 4 floats and every base 16-byte aligned):
// one thread = 4 consecutive state channels of a row, 16-byte loads / stores, the per-row quantities (go symbol) once per quad
static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
static __device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__global__ void k_cell_bwd_b4(const float* __restrict__ dy0, const float* __restrict__ dy0x, int nx, long long xs, long long ldy,
                              const float* __restrict__ z0, long long ldz, const float* __restrict__ zr, int H,
                              long long R, float* __restrict__ dG, float* __restrict__ dacc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H4 = H >> 2;
    if (i >= R * H4) return;
    const int c = 4 * (int)(i % H4);
    const long long r = i / H4;
    float4 dzh = ld4(dy0 + r * ldy + c);
    for (int e = 0; e < nx; ++e) dzh = add4(dzh, ld4(dy0x + e * xs + r * ldy + c));
    const float4 h = ld4(z0 + r * ldz + c), z = ld4(zr + r * 2 * H + c), a = ld4(dacc + r * H + c);
    st4(dG + r * 2 * H + c, make_float4(dzh.x * h.x * z.x * (1.f - z.x), dzh.y * h.y * z.y * (1.f - z.y),
                                        dzh.z * h.z * z.z * (1.f - z.z), dzh.w * h.w * z.w * (1.f - z.w)));
    st4(dacc + r * H + c, make_float4(a.x + dzh.x * z.x, a.y + dzh.y * z.y, a.z + dzh.z * z.z, a.w + dzh.w * z.w));
}
__global__ void k_cell_bwd_ca4(const float* __restrict__ dz0, const float* __restrict__ dz0x, int nzx,
                               const float* __restrict__ dy0, const float* __restrict__ dy0x, int nyx, long long xs, int xcols, long long ld,
                               const float* __restrict__ dout_bt, long long out_sb, long long out_sn, int use_next,
                               const float* __restrict__ Wp, int od, float* __restrict__ dgo_rows, int B,
                               const float* __restrict__ z0, long long ldz, const float* __restrict__ zr,
                               const float* __restrict__ hc, int H, long long R,
                               float* __restrict__ dU, float* __restrict__ dG, float* __restrict__ dacc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H4 = H >> 2;
    if (i >= R * H4) return;
    const int c = 4 * (int)(i % H4);
    const long long r = i / H4;
    float4 a = ld4(dz0 + r * ld + c);
    for (int e = 0; e < nzx; ++e) a = add4(a, ld4(dz0x + e * xs + r * ld + c));
    float4 g = add4(ld4(dacc + r * H + c), a);
    if (Wp) {
        const int n = (int)(r / B), b = (int)(r % B);
        for (int j = 0; j < od; ++j) {
            float go = dout_bt[b * out_sb + n * out_sn + j];
            if (use_next) {
                float x = dz0[r * ld + H + j] + dy0[r * ld + H + j];
                if (H + j < xcols) {
                    for (int e = 0; e < nzx; ++e) x += dz0x[e * xs + r * ld + H + j];
                    for (int e = 0; e < nyx; ++e) x += dy0x[e * xs + r * ld + H + j];
                }
                go += x;
            }
            const float* __restrict__ w = Wp + (long long)j * H + c;       // (a caller's parameter: no alignment assumed)
            g.x += go * w[0]; g.y += go * w[1]; g.z += go * w[2]; g.w += go * w[3];
            if (c == 0) dgo_rows[r * od + j] = go;
        }
    }
    const float4 h = ld4(z0 + r * ldz + c), rr = ld4(zr + r * 2 * H + H + c), hv = ld4(hc + r * H + c);
    st4(dU + r * H + c, make_float4(g.x * (1.f - rr.x) * (1.f - hv.x * hv.x), g.y * (1.f - rr.y) * (1.f - hv.y * hv.y),
                                    g.z * (1.f - rr.z) * (1.f - hv.z * hv.z), g.w * (1.f - rr.w) * (1.f - hv.w * hv.w)));
    st4(dG + r * 2 * H + H + c, make_float4(g.x * (h.x - hv.x) * rr.x * (1.f - rr.x), g.y * (h.y - hv.y) * rr.y * (1.f - rr.y),
                                            g.z * (h.z - hv.z) * rr.z * (1.f - rr.z), g.w * (h.w - hv.w) * rr.w * (1.f - rr.w)));
    st4(dacc + r * H + c, make_float4(g.x * rr.x, g.y * rr.y, g.z * rr.z, g.w * rr.w));
}

// dst += a * src   (plane-wise step of the Chebyshev recursion backward for cheb_k > 3: d_{k-2} -= d_k)
__global__ void k_axpy(float* __restrict__ dst, const float* __restrict__ src, float a, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += a * src[i];
}

// ---------------------------------------------------------------------------------------------
// deterministic column sums:  out[c] (+)= sum_r w[r]*X[r*ld + c]   (w nullable), two stages
// ---------------------------------------------------------------------------------------------
// grid (ceil(C/64), nchunk): 256 threads = 64 columns x 4 row lanes; rows of the chunk are strided over
// the lanes, partial[chunk][c] written after an LDS reduction over the 4 lanes (fixed order).
__global__ void k_colsum_stage1(const float* __restrict__ X, long long ld, long long rows, int C,
                                int chunk, float* __restrict__ part) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long long r0 = (long long)blockIdx.y * chunk;
    const long long r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float s = 0.f;
    if (c < C)
        for (long long r = r0 + rl; r < r1; r += 4) s += X[r * ld + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) part[(long long)blockIdx.y * C + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// grid ceil(C/64), 1024 threads: 64 columns x 16 row lanes over the partial rows (hundreds of partials at T*N*B rows: the
// 4-lane version spent 47 us walking them serially), fixed summation order
__global__ __launch_bounds__(1024) void k_colsum_stage2(const float* __restrict__ part, int nblk, int C, float* __restrict__ out,
                                                        int accumulate) {
    __shared__ float sh[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int b = rl; b < nblk; b += 16) s += part[(long long)b * C + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// out[i] = sum_z slabs[z*slab + i].  1024 threads = 64 consecutive elements x 16 slab lanes (the split-K products of the
// tiny-output gradients leave up to 256 slabs of a few thousand elements: one thread per element walked them as a
// serial chain of 256 dependent loads, 49 us); lane partials are added in a fixed order.
__global__ __launch_bounds__(1024) void k_reduce_slabs(float* __restrict__ out, const float* __restrict__ slabs, int nslab,
                                                       long long slab, long long n, int accumulate) {
    __shared__ float sh[16][64];
    const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < n)
        for (int z = zl; z < nslab; z += 16) s += slabs[z * slab + i];
    sh[zl][el] = s;
    __syncthreads();
    if (zl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int z = 0; z < 16; ++z) t += sh[z][el];
        out[i] = accumulate ? out[i] + t : t;
    }
}

// In-place first stage of a wide slab reduction: slab y (y < groups) becomes the sum of slabs y, y+groups, ...
// Each slab y is read and written only by group y, in a fixed order (deterministic); the grid has `groups` times
// more threads than elements so the 64 adjacency-gradient slabs stream at HBM rate instead of through N waves.
__global__ void k_reduce_slabs_groups(float* __restrict__ slabs, int nslab, long long slab, long long n, int groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 4
    for (int z = y; z < nslab; z += groups) s += slabs[z * slab + i];
    slabs[y * slab + i] = s;
}

// strided 2-D copy:  dst[r*ldd + c] = src[r*lds + c]
__global__ void k_copy2d(float* __restrict__ dst, long long ldd, const float* __restrict__ src,
                         long long lds_, long long rows, int cols) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int c = (int)(i % cols);
    long long r = i / cols;
    dst[r * ldd + c] = src[r * lds_ + c];
}

// ---------------------------------------------------------------------------------------------
// trainer loss, forward + backward fused (model/traintest_MegaCRN.py:118-125, model/utils.py:126-133):
//   loss = masked_mae(inv(output), inv(labels)) + lamb * TripletMarginLoss(q, pos, neg) + lamb1 * MSE(q, pos)
// stage1: deterministic per-block partial sums {mask count, sum |diff|*mask, sum triplet, sum sq} and d_query
// stage2: one wave reduces the partials -> losses[0..3], scalars[0] = 1/mask count
// stage3: d_output = sign(diff) * mask * std / mask count
// inverse_transform is evaluated as two roundings (mul, then add) like the reference's two torch ops:
// a fused multiply-add would not map the standardised missing value back to exactly 0 and break the mask.
// ---------------------------------------------------------------------------------------------
// x*std + mean with TWO roundings (the asm barrier keeps hipcc from contracting it into one FMA)
__device__ __forceinline__ float inv_transform(float x, float stdv, float mean) {
    float t = x * stdv;
    asm volatile("" : "+v"(t));
    return t + mean;
}
__global__ void k_loss_stage1(const float* __restrict__ out, const float* __restrict__ lab, long long nout,
                              const float* __restrict__ q, const float* __restrict__ pos,
                              const float* __restrict__ neg, long long rows, int D, float mean, float stdv,
                              float lamb, float lamb1, float margin, float* __restrict__ part,
                              float* __restrict__ dq) {
    __shared__ float sh[4][4];
    float cnt = 0.f, sab = 0.f, trip = 0.f, sq = 0.f;
    const long long gs = (long long)gridDim.x * blockDim.x, g0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = g0; i < nout; i += gs) {
        const float yt = inv_transform(lab[i], stdv, mean);
        const float yp = inv_transform(out[i], stdv, mean);
        if (yt != 0.f) { cnt += 1.f; sab += fabsf(yp - yt); }
    }
    const float inv_rows = 1.f / (float)rows, inv_rd = 1.f / ((float)rows * (float)D);
    // triplet + MSE terms: one WAVE per (b, n) row, lanes over the D channels (coalesced; a thread per row walked its
    // three D-vectors serially: 56 us).  Row sums are wave-uniform, so only lane 0 adds them to its partials.
    const int lane = threadIdx.x & 63;
    const long long nwave = gs >> 6;
    for (long long r = g0 >> 6; r < rows; r += nwave) {
        const float* qr = q + r * D; const float* pr = pos + r * D; const float* nr = neg + r * D;
        float lp = 0.f, ln = 0.f, s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float a = qr[d], dp = a - pr[d] + 1e-6f, dn = a - nr[d] + 1e-6f, e = a - pr[d];
            lp += dp * dp; ln += dn * dn; s2 += e * e;
        }
        lp = sqrtf(wave_sum(lp)); ln = sqrtf(wave_sum(ln)); s2 = wave_sum(s2);
        const float v = lp - ln + margin;
        const bool act = v > 0.f;
        if (lane == 0) { if (act) trip += v; sq += s2; }
        const float cp = act ? lamb * inv_rows / lp : 0.f, cn = act ? lamb * inv_rows / ln : 0.f;
        for (int d = lane; d < D; d += 64) {
            const float a = qr[d];
            dq[r * D + d] = cp * (a - pr[d] + 1e-6f) - cn * (a - nr[d] + 1e-6f) + lamb1 * 2.f * inv_rd * (a - pr[d]);
        }
    }
    cnt = wave_sum(cnt); sab = wave_sum(sab); trip = wave_sum(trip); sq = wave_sum(sq);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = cnt; sh[w][1] = sab; sh[w][2] = trip; sh[w][3] = sq; }
    __syncthreads();
    if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void k_loss_stage2(const float* __restrict__ part, int nblk, long long nout, long long rows, int D,
                              float lamb, float lamb1, float* __restrict__ losses, float* __restrict__ scal) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 64)
        for (int j = 0; j < 4; ++j) a[j] += part[b * 4 + j];
    for (int j = 0; j < 4; ++j) a[j] = wave_sum(a[j]);
    if (threadIdx.x == 0) {
        const float l1 = a[0] > 0.f ? a[1] / a[0] : 0.f;        // mean(|d| * mask / mean(mask)); NaN -> 0 (:131)
        const float l2 = a[2] / (float)rows, l3 = a[3] / ((float)rows * (float)D);
        losses[0] = l1 + lamb * l2 + lamb1 * l3; losses[1] = l1; losses[2] = l2; losses[3] = l3;
        scal[0] = a[0] > 0.f ? 1.f / a[0] : 0.f;
    }
}
__global__ void k_loss_dout(const float* __restrict__ out, const float* __restrict__ lab, long long nout,
                            float mean, float stdv, const float* __restrict__ scal, float* __restrict__ dout) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout) return;
    const float yt = inv_transform(lab[i], stdv, mean);
    const float yp = inv_transform(out[i], stdv, mean);
    const float d = yp - yt;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dout[i] = yt != 0.f ? sg * stdv * scal[0] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// flat clip_grad_norm_ + Adam   (model/traintest_MegaCRN.py:104,129-130)
// ---------------------------------------------------------------------------------------------
__global__ void k_sumsq_stage1(const float* __restrict__ g, long long n, float scale, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = g[i] * scale;
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void k_sumsq_stage2(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    // single wave; deterministic order
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = sqrtf(s);
}
__global__ void k_clip_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2_sqrt, float max_norm, float scale,
                            const float* __restrict__ total_norm, float* __restrict__ total_norm_out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float tn = total_norm[0];
    if (i == 0 && total_norm_out) total_norm_out[0] = tn;   // (a 4-byte hipMemcpyAsync here cost ~100 us of queue idle per step)
    const float c = max_norm / (tn + 1e-6f);
    float coef = c > 1.f ? 1.f : c;        // torch.clamp(c, max=1): a NaN norm stays NaN like clip_grad_norm_ (fminf would drop it)
    float gi = g[i] * scale * coef;
    g[i] = gi;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}


// ---------------------------------------------------------------------------------------------
// bf16-resident operands of the large-graph path (MCRN_BF16, gemm_bf16.h)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bf16_rne(float f) {        // finite values
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// the bf16 "lo" word of a value whose "hi" word is h: bf16(f - float(h)).  hi + lo carries 16 mantissa bits: the operand pairs of the
// hi/lo products of gemm_bf16.h (Bf16GemmP::nterm = 3, the bf16x3 arithmetic on bf16-resident operands; round 5)
__device__ __forceinline__ unsigned bf16_lo(float f, unsigned h) { return bf16_rne(f - __uint_as_float(h << 16)); }
// two values -> packed hi word pair and (optionally) lo word pair
__device__ __forceinline__ void bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ha = bf16_rne(a), hb = bf16_rne(b);
    hi = ha | (hb << 16);
    lo = bf16_lo(a, ha) | (bf16_lo(b, hb) << 16);
}
// fp32 planes [np][N][ld] (plane stride ps_src) -> bf16 planes [np][Kp][ldp] with ZERO pad rows / columns (the operand
// contract of gemm_bf16.h), optionally also the plane centred over its rows (nodes):
//   dst = bf16(src) ; cen = bf16(src - colsum[col] * inv_rows).
// `cen` exists because the adjacency gradient dP x X^T only enters the row-softmax backward, which is blind to anything
// constant along a row of dS: the node-mean of X can be removed BEFORE rounding to bf16, and that mean is exactly what
// makes the softmax backward cancel catastrophically on large, nearly uniform supports.  8 elements per thread.
__global__ void k_plane_to_bf16(const float* __restrict__ src, long long ps_src, int N, int ld, int nvalid, int Kp, int ldp,
                                int np, uint4* __restrict__ dst, uint4* __restrict__ cen, const float* __restrict__ colsum,
                                float inv_rows, long long colsum_stride, long long dst_ps8 /* uint4 between output planes */,
                                long long lo8 /* > 0: the lo images are written lo8 uint4 behind dst / cen (bf16_lo) */) {
    const int c8n = ldp / 8;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_plane = (long long)Kp * c8n;
    if (i0 >= per_plane * np) return;
    const int pl = (int)(i0 / per_plane);
    const long long q = i0 - (long long)pl * per_plane;
    const long long i = (long long)pl * dst_ps8 + q;
    const int row = (int)(q / c8n), c0 = (int)(q - (long long)row * c8n) * 8;
    unsigned w[4] = {0u, 0u, 0u, 0u}, wc[4] = {0u, 0u, 0u, 0u}, wl[4] = {0u, 0u, 0u, 0u}, wcl[4] = {0u, 0u, 0u, 0u};
    if (row < N && c0 < ld) {                                    // ld % 8 == 0: a chunk is all-in or all-out
        const float* s = src + (long long)pl * ps_src + (long long)row * ld + c0;
        const float4 a = reinterpret_cast<const float4*>(s)[0], b = reinterpret_cast<const float4*>(s)[1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (c0 + 8 > nvalid)                                     // columns nvalid .. ld-1 of the source are not data
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c0 + j < nvalid ? v[j] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf16_pair(v[2 * j], v[2 * j + 1], w[j], wl[j]);
        if (cen) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* cs = colsum + (long long)pl * colsum_stride;
                const float m0 = cs[c0 + 2 * j] * inv_rows, m1 = cs[c0 + 2 * j + 1] * inv_rows;
                bf16_pair(v[2 * j] - m0, v[2 * j + 1] - m1, wc[j], wcl[j]);
            }
        }
    }
    if (dst) dst[i] = make_uint4(w[0], w[1], w[2], w[3]);
    if (cen) cen[i] = make_uint4(wc[0], wc[1], wc[2], wc[3]);
    if (lo8 > 0) {
        if (dst) dst[lo8 + i] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
        if (cen) cen[lo8 + i] = make_uint4(wcl[0], wcl[1], wcl[2], wcl[3]);
    }
}

// Hoisted propagation: channel block [col0, col0 + w) of the rows (n, b) of `T` plane sets, packed as the
// [k][n] bf16 operand of the propagation GEMM:
//   dst[n][(t*B + b)*w + j] = bf16(src[t*src_t + n*ld + b*Cp + col0 + j])      n < N, zero rows up to Kp, zero columns up to ldo
// w == H, T == 1: the state channels of one AGCN call (the only part of the input that changes from step to step);
// w == d, T == T_in/T_out: the input channels of every step of a stack, propagated ONCE before the recurrence.
// 8 output elements per thread; w % 8 == 0 (and 16-byte aligned sources) takes two float4 loads.
// Generalisations used by the hoisted backward: blockIdx.y = plane (source + y * src_y, destination + y * dst_y uint4);
// mu != null: the value is centred over the nodes first, x - mu[t * mu_t + b * Cp + col0 + j] * inv_rows (see k_plane_to_bf16);
// a chunk's columns land at dst column  coff + c  (ldo is the destination row length, ncw = 8 * ceil(T*B*w / 8) columns are
// written per row: the stack-wide operands interleave the gate and the update calls of a step).
__global__ void k_pack_cols_bf16(const float* __restrict__ src, long long src_t, int N, long long ld, int Cp, int col0, int w, int B,
                                 int T, int Kp, int ldo, uint4* __restrict__ dst, long long src_y, long long dst_y,
                                 const float* __restrict__ mu, long long mu_t, long long mu_y, float inv_rows, int coff,
                                 int ncw, int tmul, int toff, long long lo8 /* > 0: lo image lo8 uint4 behind dst */) {
    const int ncols = T * B * w;
    const int c8n = (ncw > 0 ? ncw : ldo) / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)Kp * c8n) return;
    const int row = (int)(i / c8n), c0 = (int)(i - (long long)row * c8n) * 8;
    src += (long long)blockIdx.y * src_y;
    if (mu) mu += (long long)blockIdx.y * mu_y;
    unsigned wd[4] = {0u, 0u, 0u, 0u}, wl[4] = {0u, 0u, 0u, 0u};
    long long dcol = coff + c0;                       // destination column of the chunk
    if (tmul != 1 && c0 < ncols) {                    // (t*B + b)*w + j  ->  ((tmul*t + toff)*B + b)*w + j : chunks never straddle a step when B*w % 8 == 0
        const int t = c0 / (B * w);
        dcol = coff + (long long)(tmul * t + toff) * B * w + (c0 - t * B * w);
    }
    if (row < N && c0 < ncols) {
        float v[8];
        if ((w & 7) == 0 && ((ld | Cp | col0 | src_t) & 3) == 0) {
            const int q = c0 / w, j = c0 - q * w, t = q / B, b = q - t * B;
            const float* sp = src + (long long)t * src_t + (long long)row * ld + (long long)b * Cp + col0 + j;
            const float4 a = reinterpret_cast<const float4*>(sp)[0], bb = reinterpret_cast<const float4*>(sp)[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
            if (mu) {
                const float* m = mu + (long long)t * mu_t + (long long)b * Cp + col0 + j;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] -= m[e] * inv_rows;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e;
                float x = 0.f;
                if (c < ncols) {
                    const int q = c / w, j = c - q * w, t = q / B, b = q - t * B;
                    x = src[(long long)t * src_t + (long long)row * ld + (long long)b * Cp + col0 + j];
                    if (mu) x -= mu[(long long)t * mu_t + (long long)b * Cp + col0 + j] * inv_rows;
                }
                v[e] = x;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bf16_pair(v[2 * e], v[2 * e + 1], wd[e], wl[e]);
    }
    const long long o = (long long)blockIdx.y * dst_y + ((long long)row * ldo + dcol) / 8;
    dst[o] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    if (lo8 > 0) dst[lo8 + o] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
}
// dst[n][b*Cp + col0 + j] += tmp[n][coff + b*w + j]   (j < wuse <= w): the go-symbol share of a propagated input gradient
// (tmp: the sum of `nsplit` split-K partial products `slab` floats apart, added in a fixed order)
__global__ void k_scatter_add_cols(const float* __restrict__ tmp, int ldt, int coff, int nsplit, long long slab, int N, int B, int w, int wuse,
                                   float* __restrict__ dst, long long ld, int Cp, int col0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * B * wuse) return;
    const int j = (int)(i % wuse);
    const long long q = i / wuse;
    const int b = (int)(q % B);
    const long long n = q / B;
    const float* __restrict__ src = tmp + n * ldt + coff + b * w + j;
    float v = src[0];
    for (int z = 1; z < nsplit; ++z) v += src[z * slab];
    dst[n * ld + (long long)b * Cp + col0 + j] += v;
}
// fp32 variant for the small graphs (bf16x3 mode: the hoisted product runs on the fused two-hop kernel of prop_small.h):
//   dst[n][(t*B + b)*w + j] = src[t*src_t + n*ld + b*Cp + col0 + j]      columns up to ldo are zero
__global__ void k_pack_cols_f32(const float* __restrict__ src, long long src_t, int N, long long ld, int Cp, int col0, int w, int B,
                                int T, int ldo, float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * ldo) return;
    const int n = (int)(i / ldo), c = (int)(i - (long long)n * ldo);
    float x = 0.f;
    if (c < T * B * w) {
        const int q = c / w, j = c - q * w, t = q / B, b = q - t * B;
        x = src[(long long)t * src_t + (long long)n * ld + (long long)b * Cp + col0 + j];
    }
    dst[i] = x;
}
// ... and the way back: the propagated input channels, tmp[(blk*N + n)][(t*B + b)*w + j] (fp32, row stride ldt; the sum of
// `nsplit` split-K partial products `slab` floats apart), into
// columns [col0, col0 + w) of planes 1 + blk of the plane sets of step t - of BOTH sets of a cell (gate input Z and
// candidate input Y carry the same input channels; model/MegaCRN.py:42,45)
__global__ void k_scatter_cols(const float* __restrict__ tmp, int ldt, int nsplit, long long slab, int nb, int N, int B, int w, int T,
                               float* __restrict__ Z, float* __restrict__ Y, long long dst_t, long long PS, long long ld, int Cp, int col0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)T * B * w;
    if (i >= per * N * nb) return;
    const long long rowi = i / per;
    const int c = (int)(i - rowi * per);
    const int blk = (int)(rowi / N), n = (int)(rowi - (long long)blk * N);
    const int q = c / w, j = c - q * w, t = q / B, b = q - t * B;
    float v = tmp[rowi * ldt + c];
    for (int z = 1; z < nsplit; ++z) v += tmp[z * slab + rowi * ldt + c];      // split-K partial products, fixed order
    const long long o = (long long)t * dst_t + (long long)(1 + blk) * PS + (long long)n * ld + (long long)b * Cp + col0 + j;
    Z[o] = v;
    if (Y) Y[o] = v;
}

// The same sums into the COMPACT input array of the bf16-resident mode (the fp32 planes 1 .. nb of a plane set hold nothing but
// their d <= 4 input channels there; scattering them at a Cp-float stride is a 4-byte write into a separate memory line each:
// 82 + 117 us per EXPY-TKY step):  Xp[t][blk][r = n*B + b][4]  = channels c0 .. c0 + w of that row, one 16-byte row per
// (step, plane, row), read by wp_stream (input group) and wgrad_stream (the quad at channel H).
// full: w covers all d channels of the stack -> the whole float4 is written (zeros behind d); else only channels c0 .. c0 + w.
__global__ void k_scatter_compact(const float* __restrict__ tmp, int ldt, int nsplit, long long slab, int nb, int N, int B, int w, int T,
                                  float* __restrict__ Xp, long long xp_t, int c0, int full) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // (t, blk, n, b): consecutive threads = consecutive rows
    const long long R = (long long)N * B;
    if (i >= (long long)T * nb * R) return;
    const int b = (int)(i % B);
    long long q = i / B;
    const int n = (int)(q % N); q /= N;
    const int blk = (int)(q % nb), t = (int)(q / nb);
    const long long rowi = (long long)blk * N + n;
    const float* __restrict__ src = tmp + rowi * ldt + ((long long)t * B + b) * w;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < w; ++j) {
        float a = src[j];
        for (int z = 1; z < nsplit; ++z) a += src[z * slab + j];               // split-K partial products, fixed order
        v[c0 + j] = a;
    }
    float* __restrict__ dst = Xp + (long long)t * xp_t + ((long long)blk * R + (long long)n * B + b) * 4;
    if (full) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int j = 0; j < w; ++j) dst[c0 + j] = v[c0 + j];
}

// colsum[c] = (N / nsamp) * sum over nsamp evenly spaced rows of X[row][c]: an ESTIMATE of the column sums over all N
// rows.  The centring above is exact for ANY vector subtracted from every row (it only has to be the same vector for
// all rows); what matters numerically is that the bulk of the common component is gone, so a 64-row sample replaces
// a full pass over the plane.
__global__ __launch_bounds__(256) void k_colsum_sample(const float* __restrict__ X, long long ld, int N, int C, int nsamp,
                                                       float* __restrict__ out, long long x_stride, long long out_stride) {
    __shared__ float sh[4][64];
    X += (long long)blockIdx.y * x_stride;                        // grid.y = plane (one AGCN call each)
    out += (long long)blockIdx.y * out_stride;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;       // 64 columns x 4 row lanes: independent loads in flight
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int i = rl; i < nsamp; i += 4) s += X[(long long)(((long long)i * N) / nsamp) * ld + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[c] = ((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl])) * ((float)N / (float)nsamp);
}

// zero the pad rows [N, Kp) of `np` bf16 planes [Kp][ldp] (plane stride PSb): producers that write such planes in place
// (d-grad in MCRN_BF16 mode) only touch the N data rows
__global__ void k_zero_pad_rows(uint4* __restrict__ planes, long long psb8, int N, int Kp, int ldp8, long long np) {
    const long long per = (long long)(Kp - N) * ldp8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * np) return;
    const long long pl = i / per, q = i - pl * per;
    planes[pl * psb8 + (long long)N * ldp8 + q] = make_uint4(0u, 0u, 0u, 0u);
}

// One block of the stacked adjacency operand of the propagation GEMM (model/MegaCRN.py:20-25 as ONE product):
//   transpose == 0:  dst[(r0 + i) * ldd + c0 + j] = bf16(S[i][j])        forward stack  [S1; T2(S1); S2; T2(S2)]
//   transpose == 1:  dst[(r0 + i) * ldd + c0 + j] = bf16(S[j][i])        backward stack [S1^T | T2^T | S2^T | T2^T]
// for i < N, j < Kp (zero for j >= N: the K padding of the GEMM's A operand).  32 x 32 tiles through LDS.
__global__ void k_stack_build(const float* __restrict__ S, long long lds_, int N, int Kp, int transpose,
                              uint16_t* __restrict__ dst, long long ldd, long long r0, long long c0, long long lo_off /* > 0: lo image */) {
    __shared__ float t[32][33];
    const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;       // output tile origin (i, j)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads
    for (int k = ty; k < 32; k += 8) {
        if (transpose) {           // need S[j][i]: read rows j = bj + k, columns i = bi + tx
            const int j = bj + k, i = bi + tx;
            t[k][tx] = (j < N && i < N) ? S[(long long)j * lds_ + i] : 0.f;
        } else {
            const int i = bi + k, j = bj + tx;
            t[k][tx] = (i < N && j < N) ? S[(long long)i * lds_ + j] : 0.f;
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int i = bi + k, j = bj + tx;
        if (i < N && j < Kp) {
            const float v = transpose ? t[tx][k] : t[k][tx];
            const unsigned h = bf16_rne(v);
            dst[(r0 + i) * ldd + c0 + j] = (uint16_t)h;
            if (lo_off > 0) dst[lo_off + (r0 + i) * ldd + c0 + j] = (uint16_t)bf16_lo(v, h);
        }
    }
}

// T2 = 2 S S - I was left as 2 S S by the GEMM: subtract the identity
__global__ void k_sub_eye(float* __restrict__ A, long long ld, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) A[(long long)i * ld + i] -= 1.f;
}

// A (+)= B [+ C]  [- I]   over an N x N matrix of row stride ld: folds the second K split(s) of an N^3 product into the first
// (fixed order: one writer per element), optionally with the "- I" of T2 = 2 S S - I
__global__ void k_fold_splits(float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cc, long long ld, int N,
                              int sub_eye) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    if (c >= N) return;
    float v = A[i] + B[i];
    if (Cc) v += Cc[i];
    if (sub_eye && r == c) v -= 1.f;
    A[i] = v;
}

// ---------------------------------------------------------------------------------------------
// evaluation metrics of the trainer, ONE launch per batch, accumulated on device, no host sync
// (model/traintest_MegaCRN.py:63-93 with model/utils.py:126-160):
//   per batch: loss = masked_mae + lamb*triplet + lamb1*mse ; masked MAE / MAPE / MSE over the whole batch and over
//   the single-step slices y[:, t] for up to three horizons t (the trainer's 3 / 6 / 12) ; the epoch result is the
//   mean over batches of these per-batch values (RMSE = sqrt of the mean MSE), so the kernel adds each batch's
//   values to `acc` and counts batches:
//     acc[0] = sum loss ; acc[1 + 3s + {0,1,2}] = sum of {mae, mape, mse} of slice s (0 = overall, 1..3 = horizons) ;
//     acc[13] = number of batches.
// masked_X(pred, true) = mean(X * mask / mean(mask)), NaN -> 0  ==  sum_{true != 0} X / count   (0 when count == 0).
// Every block leaves 18 partial sums; the last block to arrive (agent-scope release / ticket / acquire, guide G16)
// reduces them in a fixed order, so the result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
#define MCRN_EVAL_NQ 18
__global__ __launch_bounds__(256) void k_eval_metrics(const float* __restrict__ out, const float* __restrict__ lab,
                                                      long long nout, int T, long long inner /* N*od */, int h0, int h1,
                                                      int h2, const float* __restrict__ q, const float* __restrict__ pos,
                                                      const float* __restrict__ neg, long long rows, int D, float mean,
                                                      float stdv, float lamb, float lamb1, float margin,
                                                      float* __restrict__ part, unsigned* __restrict__ ticket,
                                                      float* __restrict__ acc) {
    __shared__ float sh[4][MCRN_EVAL_NQ];
    __shared__ int s_last;
    float a[MCRN_EVAL_NQ];
#pragma unroll
    for (int j = 0; j < MCRN_EVAL_NQ; ++j) a[j] = 0.f;
    const long long gs = (long long)gridDim.x * blockDim.x, g0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = g0; i < nout; i += gs) {
        const float yt = inv_transform(lab[i], stdv, mean);
        const float yp = inv_transform(out[i], stdv, mean);
        if (yt != 0.f) {
            const float d = yt - yp;
            const float ab = fabsf(yp - yt), ape = fabsf(d / yt), sq = d * d;
            const int t = (int)((i / inner) % T);
            a[0] += 1.f; a[1] += ab; a[2] += ape; a[3] += sq;
            if (t == h0) { a[4] += 1.f; a[5] += ab; a[6] += ape; a[7] += sq; }
            if (t == h1) { a[8] += 1.f; a[9] += ab; a[10] += ape; a[11] += sq; }
            if (t == h2) { a[12] += 1.f; a[13] += ab; a[14] += ape; a[15] += sq; }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long nwave = gs >> 6;
    for (long long r = g0 >> 6; r < rows; r += nwave) {     // triplet + MSE terms: one wave per (b, n) row
        const float* qr = q + r * D; const float* pr = pos + r * D; const float* nr = neg + r * D;
        float lp = 0.f, ln = 0.f, s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float x = qr[d], dp = x - pr[d] + 1e-6f, dn = x - nr[d] + 1e-6f, e = x - pr[d];
            lp += dp * dp; ln += dn * dn; s2 += e * e;
        }
        lp = sqrtf(wave_sum(lp)); ln = sqrtf(wave_sum(ln)); s2 = wave_sum(s2);
        const float v = lp - ln + margin;
        if (lane == 0) { if (v > 0.f) a[16] += v; a[17] += s2; }
    }
#pragma unroll
    for (int j = 0; j < MCRN_EVAL_NQ; ++j) a[j] = wave_sum(a[j]);
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < MCRN_EVAL_NQ; ++j) sh[w][j] = a[j];
    __syncthreads();
    if (threadIdx.x < MCRN_EVAL_NQ)
        part[(long long)blockIdx.x * MCRN_EVAL_NQ + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    // publish: the writers are all in wave 0; release at agent scope, then take a ticket
    if (w == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == gridDim.x - 1) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (w == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float r[MCRN_EVAL_NQ];
#pragma unroll
        for (int j = 0; j < MCRN_EVAL_NQ; ++j) r[j] = 0.f;
        for (int b = lane; b < (int)gridDim.x; b += 64)
#pragma unroll
            for (int j = 0; j < MCRN_EVAL_NQ; ++j) r[j] += part[(long long)b * MCRN_EVAL_NQ + j];
#pragma unroll
        for (int j = 0; j < MCRN_EVAL_NQ; ++j) r[j] = wave_sum(r[j]);
        if (lane == 0) {
            const float l1 = r[0] > 0.f ? r[1] / r[0] : 0.f;
            const float l2 = r[16] / (float)rows, l3 = r[17] / ((float)rows * (float)D);
            acc[0] += l1 + lamb * l2 + lamb1 * l3;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float c = r[4 * s];
                acc[1 + 3 * s + 0] += c > 0.f ? r[4 * s + 1] / c : 0.f;
                acc[1 + 3 * s + 1] += c > 0.f ? r[4 * s + 2] / c : 0.f;
                acc[1 + 3 * s + 2] += c > 0.f ? r[4 * s + 3] / c : 0.f;
            }
            acc[13] += 1.f;
            acc[14] = l1; acc[15] = l2; acc[16] = l3;      // last batch's loss terms (tests)
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
    }
}

}  // namespace mcrn
