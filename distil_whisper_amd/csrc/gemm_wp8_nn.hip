// gemm_wp.h: 8 waves (2 x 4), 128 x 64 per wave, row-major operands
#include "gemm_wp.h"
int dw_gemm_wp8_nn_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, false, 2, 4>(p, s); }
