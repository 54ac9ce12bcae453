// 128 x 128 block tile with FOUR waves (2 x 2, 64 x 64 of output per wave), two workgroups per CU.  The 8-wave form of
// gemm_tile128.hip gives a wave 64 x 32: three LDS fragment reads per two MFMAs, the same LDS-bandwidth-bound ratio per flop
// that the 256-row kernels left behind in round 2; 64 x 64 per wave reads four fragments per four MFMAs (-33 % LDS bytes per
// flop).  Round 6 experiment for the teacher decoder's M ~ 4 100 launches (dw_debug_set key 24: 1 = this tile with the plain
// K loop, 2 = with the register double buffer of gemm_kernel.h VAR 2).
#include "gemm_kernel.h"

int dw_gemm_tile128w4_launch(const GemmP& p, int ta, int tb, int var, hipStream_t s) {
    if (var == 2) return launch_tile<128, 128, 2, 2, 2>(p, ta, tb, s);
    return launch_tile<128, 128, 2, 2, 0>(p, ta, tb, s);
}
