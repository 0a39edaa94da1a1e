// gemm_bf16_unit.hip - translation unit of the bf16-resident GEMM kernels (gemm_bf16.h) of libmegacrn_hip.so: the plain forms
// (one MFMA per product) and the dispatcher; the hi/lo forms compile in gemm_bf16_x3_unit.hip
#define MCRN_BF16_PART 1
#include "gemm_bf16.h"
