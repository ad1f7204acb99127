// gemm_bf16.hip — the bf16 instantiation of the GEMM kernel family (gemm_kernels.h) + the bf16-only x3 split epilogues
#include "gemm_kernels.h"

template void gemm_launch<BF16>(const GemmArgs&, const GemmLaunch&, hipStream_t);

void gemm_split_launch_bf16(const GemmArgs& g, int regstaged, int dyn, int tanh_mode, unsigned wgs, hipStream_t s) {
  typedef void (*gemm_fn)(const GemmArgs);
  // [A-in-registers form (even number of K stages >= 6: every x3 call of the engine) | plain persistent form][schedule][mode]
  static const gemm_fn stable[2][2][2] = {
      {{gemm_w256p_kernel<BF16, false, false, EPI_BF16_SPLIT, false>, gemm_w256p_kernel<BF16, false, false, EPI_BF16_TANH_SPLIT, false>},
       {gemm_w256p_kernel<BF16, false, false, EPI_BF16_SPLIT, true>, gemm_w256p_kernel<BF16, false, false, EPI_BF16_TANH_SPLIT, true>}},
      {{gemm_w256r_kernel<BF16, false, EPI_BF16_SPLIT, false>, gemm_w256r_kernel<BF16, false, EPI_BF16_TANH_SPLIT, false>},
       {gemm_w256r_kernel<BF16, false, EPI_BF16_SPLIT, true>, gemm_w256r_kernel<BF16, false, EPI_BF16_TANH_SPLIT, true>}}};
  static const bool s_attr = [] {
    for (int r = 0; r < 2; ++r)
      for (int d = 0; d < 2; ++d)
        for (int e = 0; e < 2; ++e)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stable[r][d][e]), hipFuncAttributeMaxDynamicSharedMemorySize, W2P_LDS_BYTES);
    return true;
  }();
  (void)s_attr;
  hipLaunchKernelGGL(stable[regstaged][dyn][tanh_mode], dim3(wgs), dim3(256), (size_t)W2P_LDS_BYTES, s, g);
}
