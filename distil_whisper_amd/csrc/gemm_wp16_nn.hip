// gemm_wp16.h (v_mfma_f32_16x16x32_bf16 main loop), row-major operands: 256-row and 320-row block tiles
#include "gemm_wp16.h"
int dw_gemm_wp16_nn_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, false, 256>(p, s); }
int dw_gemm_wp16_nn320_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, false, 320>(p, s); }
int dw_gemm_wp16_nn_dbg_launch(const GemmP& p, int dbg, hipStream_t s) {      // main-loop ablations (results wrong by construction)
    if (dbg == 1) return launch_wp16<false, false, 256, 1>(p, s);
    if (dbg == 2) return launch_wp16<false, false, 256, 2>(p, s);
    return launch_wp16<false, false, 256, 3>(p, s);
}
