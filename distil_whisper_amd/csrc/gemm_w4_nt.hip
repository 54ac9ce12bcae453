// gemm_w4.h instantiated for one operand layout (one translation unit per layout: they compile in parallel)
#include "gemm_w4.h"
int dw_gemm_w4_nt_launch(const GemmP& p, hipStream_t s) { return launch_w4<false, true>(p, s); }
