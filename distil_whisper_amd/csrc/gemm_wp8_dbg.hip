// gemm_wp.h, row-major operands, main-loop ablations for profiling (dw_debug_set(0, v | 512 | 1024)): 512 = no fragment
// reads in the K loop (the MFMAs run on the fragments of the first K tile), 1024 = no operand DMA in the K loop.  Results
// are wrong by construction; tools/gemm_overhead.py uses them to price the main loop's memory operations.
#include "gemm_wp.h"
int dw_gemm_wp8_nn_dbg_launch(const GemmP& p, int dbg, hipStream_t s) {
    if (dbg == 1) return launch_wp<false, false, 2, 4, true, 1>(p, s);
    if (dbg == 2) return launch_wp<false, false, 2, 4, true, 2>(p, s);
    if (dbg == 4) return launch_wp<false, false, 2, 4, true, 4>(p, s);     // operand DMA always L2-warm (K advance dropped)
    return launch_wp<false, false, 2, 4, true, 3>(p, s);
}
