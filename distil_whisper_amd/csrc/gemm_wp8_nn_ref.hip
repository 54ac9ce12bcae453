// gemm_wp.h, row-major operands, operand DMA through the compiler builtin: the A/B reference for the inline-assembly DMA
// (dw_debug_set(0, v | 256)); not used by default
#include "gemm_wp.h"
int dw_gemm_wp8_nn_ref_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, false, 2, 4, false>(p, s); }
