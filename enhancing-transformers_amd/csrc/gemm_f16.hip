// gemm_f16.hip — the fp16 instantiation of the GEMM kernel family (gemm_kernels.h): v_mfma_f32_*_f16, v_cvt_pk_f16_f32 (round-to-nearest-even)
#include "gemm_kernels.h"

template void gemm_launch<F16>(const GemmArgs&, const GemmLaunch&, hipStream_t);
