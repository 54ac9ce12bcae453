// gemm_wp16.h, both operands k-major (weight-gradient GEMMs)
#include "gemm_wp16.h"
int dw_gemm_wp16_tt_launch(const GemmP& p, hipStream_t s) { return launch_wp16<true, true, 256>(p, s); }
