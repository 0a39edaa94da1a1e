// gemm_bf16_x3_unit.hip - the hi/lo operand-pair forms of the bf16-resident GEMM (gemm_bf16.h, X3 = true: four images per K tile,
// three MFMA blocks per fragment pair): the 1e-4 arithmetic of the large graphs
#define MCRN_BF16_PART 2
#include "gemm_bf16.h"
