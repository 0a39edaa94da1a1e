// gemm_bf16_unit.hip - translation unit of the bf16-resident GEMM kernels (gemm_bf16.h) of libmegacrn_hip.so
#include "gemm_bf16.h"
