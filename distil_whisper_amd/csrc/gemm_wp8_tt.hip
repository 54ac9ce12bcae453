// gemm_wp.h: 8 waves (2 x 4), 128 x 64 per wave, both operands k-major (dW = dY^T . X)
#include "gemm_wp.h"
int dw_gemm_wp8_tt_launch(const GemmP& p, hipStream_t s) { return launch_wp<true, true, 2, 4>(p, s); }
