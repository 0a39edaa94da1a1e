// prop_unit.hip - adjacency-stationary propagation / adjacency-gradient kernels for N <= 352 (prop_small.h)
#define MCRN_PROBE 1   // the tiled-GEMM launchers (and with them every tile instantiation) belong to tiled_unit.hip
#include "prop_small.h"
namespace mcrn { namespace ext {
hipError_t launch_prop_small(const PropP& p, int nbatch, hipStream_t st) { return ::mcrn::launch_prop_small(p, nbatch, st); }
hipError_t launch_prop2_fwd(const Prop2P& p, hipStream_t st) { return ::mcrn::launch_prop2_fwd(p, st); }
hipError_t launch_prop2_bwd(const Prop2P& p, hipStream_t st) { return ::mcrn::launch_prop2_bwd(p, st); }
hipError_t launch_ds_small(DsP p, int nslab, hipStream_t st, int nblk) { return ::mcrn::launch_ds_small(p, nslab, st, nblk); }
hipError_t launch_sfrag_multi(const float* const* S, const int* transpose, uint4* const* out, int n, long long ldS, int N, hipStream_t st) {
    SfragMultiP q;
    for (int i = 0; i < 8; ++i) { q.S[i] = i < n ? S[i] : nullptr; q.out[i] = i < n ? out[i] : nullptr; q.transpose[i] = i < n ? transpose[i] : 0; }
    q.ldS = ldS; q.N = N; q.NF = (N + 31) / 32; q.n = n;
    return ::mcrn::launch_sfrag_multi(q, st);
}
hipError_t launch_sfrag(const float* S, long long ldS, int N, int transpose, uint4* out, hipStream_t st) { return ::mcrn::launch_sfrag(S, ldS, N, transpose, out, st); }
#ifdef MCRN_TIMELINE
hipError_t timeline_copy(void* host_out, size_t bytes) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tl), bytes < sizeof(g_tl) ? bytes : sizeof(g_tl));
}
#endif
} }
