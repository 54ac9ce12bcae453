// 256 x 256 block tile, 16 waves (4 x 4, 64 x 64 per wave): the workhorse of the distillation step.
#include "gemm_kernel.h"

int dw_gemm_tile256_launch(const GemmP& p, int ta, int tb, hipStream_t s) {
    return launch_tile<256, 256, 4, 4, 0>(p, ta, tb, s);
}
