// Compile-only probe (not part of the library): instantiates the METR-LA-sized adjacency-stationary kernels so that
// register / scratch usage and the ISA can be inspected in seconds:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Rpass-analysis=kernel-resource-usage probe_prop.hip
#define MCRN_PROBE 1
#include <hip/hip_runtime.h>
#include "gemm_f32.h"
#include "gemm_bf16x3.h"
#include "prop_small.h"
#include "dgrad_stream.h"
namespace mcrn {
template __global__ void prop2_fwd_kernel<7, 2>(const Prop2P);
template __global__ void prop2_bwd_kernel<7, 2>(const Prop2P);
template __global__ void prop2_fwd_kernel<7, 3>(const Prop2P);
template __global__ void prop2_bwd_kernel<7, 3>(const Prop2P);
template __global__ void prop_small_kernel<7>(const PropP);
template __global__ void ds_small_kernel<7>(const DsP);
template __global__ void gemm_bf16x3_kernel<64, 64, 2, 2, true, false, ROLE_WP>(const GemmP);
template __global__ void gemm_bf16x3_kernel<64, 128, 2, 2, true, false, ROLE_WP>(const GemmP);
template __global__ void gemm_bf16x3_kernel<128, 128, 2, 2, true, true, ROLE_DGRAD>(const GemmP);
template __global__ void gemm_bf16x3_kernel<64, 64, 2, 2, false, false, ROLE_WGRAD>(const GemmP);
}
