// gemm_wp16.h, row-major A, k-major B (dX GEMMs): 256-row and 320-row block tiles
#include "gemm_wp16.h"
int dw_gemm_wp16_nt_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, true, 256>(p, s); }
int dw_gemm_wp16_nt320_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, true, 320>(p, s); }
