// gemm_wp.h: 8 waves (2 x 4), 320 x 256 block tile (160 x 64 per wave), row-major A; B row-major (NN) or k-major (NT)
#include "gemm_wp.h"
int dw_gemm_wp8_nn320_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, false, 2, 4, true, 0, 320>(p, s); }
int dw_gemm_wp8_nt320_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, true, 2, 4, true, 0, 320>(p, s); }
