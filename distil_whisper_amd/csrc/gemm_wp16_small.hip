// gemm_wp16.h, 256-row tile, row-major A: the instantiations the small-M rule of gemm.hip launches (dw_debug_set key 25) -- the same
// code as gemm_wp16_nn.hip / gemm_wp16_nt.hip under a second symbol (TAG = 1), so that profiles keep the decoders' M = 32 x live
// positions launches apart from the step's M = 48 000 ones.
#include "gemm_wp16.h"
int dw_gemm_wp16_nn_small_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, false, 256, 0, 4, 1>(p, s); }
int dw_gemm_wp16_nt_small_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, true, 256, 0, 4, 1>(p, s); }
