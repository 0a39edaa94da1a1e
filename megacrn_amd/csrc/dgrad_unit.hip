// dgrad_unit.hip - streaming d-grad (dgrad_stream.h)
#define MCRN_PROBE 1   // the tiled-GEMM launchers (and with them every tile instantiation) belong to tiled_unit.hip
#include "dgrad_stream.h"
namespace mcrn { namespace ext {
hipError_t launch_dgrad_stream(DgradP p, hipStream_t st) { return ::mcrn::launch_dgrad_stream(p, st); }
hipError_t launch_wfrag_build(const float* W, long long ld, int rows, int K, int KS, uint4* out, long long tot, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_wfrag_build, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, W, ld, rows, K, KS, out, tot);
    return hipGetLastError();
}
} }
