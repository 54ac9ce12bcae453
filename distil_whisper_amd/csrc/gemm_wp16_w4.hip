// gemm_wp16.h with FOUR waves per workgroup (one per SIMD, 128 x 128 output per wave) on the 256 x 256 block tile
#include "gemm_wp16.h"
int dw_gemm_wp16_nn_w4_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, false, 256, 0, 2>(p, s); }
int dw_gemm_wp16_nt_w4_launch(const GemmP& p, hipStream_t s) { return launch_wp16<false, true, 256, 0, 2>(p, s); }
int dw_gemm_wp16_nn_w4_dbg_launch(const GemmP& p, int dbg, hipStream_t s) {      // main-loop ablations (results wrong by construction)
    if (dbg == 1) return launch_wp16<false, false, 256, 1, 2>(p, s);
    if (dbg == 2) return launch_wp16<false, false, 256, 2, 2>(p, s);
    return launch_wp16<false, false, 256, 3, 2>(p, s);
}
