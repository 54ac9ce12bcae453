// 128 x 128 block tile, 8 waves (2 x 4), two workgroups per CU: outputs with fewer than two rounds of 256-tiles.
#include "gemm_kernel.h"

int dw_gemm_tile128_launch(const GemmP& p, int ta, int tb, hipStream_t s) {
    return launch_tile<128, 128, 2, 4, 0>(p, ta, tb, s);
}
