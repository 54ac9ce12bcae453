// gemm_wp.h: 8 waves (2 x 4), 128 x 256 block tile (64 x 64 per wave), row-major A; B row-major (NN) or k-major (NT).
// The software-pipelined loop for outputs with too few rows for two rounds of 256-row tiles (the decoders' M = 32 x live
// positions): twice the tiles of the 256-row kernel per row of CUs, the same one-barrier-per-K-tile loop with the operand DMA
// in a THREE-stage ring (gemm_wp.h NST = 3: a sub-step of this tile is four MFMAs, too short for the two-stage lead) -- the
// lock-step 128 x 128 kernel of gemm_kernel.h waits out every operand tile.
#include "gemm_wp.h"
int dw_gemm_wp8_nn128_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, false, 2, 4, true, 0, 128, 3>(p, s); }
int dw_gemm_wp8_nt128_launch(const GemmP& p, hipStream_t s) { return launch_wp<false, true, 2, 4, true, 0, 128, 3>(p, s); }
