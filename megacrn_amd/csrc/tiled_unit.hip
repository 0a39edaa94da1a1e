// tiled_unit.hip - the tiled MFMA GEMMs (exact fp32 and bf16x3): every tile configuration x role is instantiated here
#include "gemm_f32.h"
#include "gemm_bf16x3.h"
namespace mcrn { namespace ext {
hipError_t launch_gemm_f32(GemmP p, bool akc, bool bkc, int max_split, hipStream_t st) { return ::mcrn::launch_gemm_f32(p, akc, bkc, max_split, st); }
hipError_t launch_gemm_x3(GemmP p, bool akc, bool bkc, int max_split, int role, hipStream_t st) { return ::mcrn::launch_gemm_x3(p, akc, bkc, max_split, role, st); }
hipError_t launch_bimg_build(const float* src, long long sk, long long sn, int K, int N, int npad, int kc_layout, uint4* img, hipStream_t st) {
    const long long n = (long long)((K + 31) / 32) * 4 * npad;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_bimg_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, sk, sn, K, N, npad, kc_layout, img);
    return hipGetLastError();
}
} }
