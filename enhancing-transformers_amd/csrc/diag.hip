// diag.hip — measurement aids that are part of the C ABI (not of any product path).
#include "common.h"

// Holds `n_wg` CUs for `ms` milliseconds: one workgroup per CU (the whole LDS of a CU as its dynamic allocation, so nothing that needs LDS can share it),
// 256 threads asleep on the constant 100-MHz wall clock.  A stand-in for a collective's kernel (RCCL holds one workgroup per channel for the duration of
// an all-reduce) in the one-GPU contention experiment of tools/comm_contention.py.  Always terminates: the duration is clamped.
__global__ __launch_bounds__(256) void occupy_cus_kernel(long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[0];   // keeps the allocation alive in the compiler's eyes (sink is normally null)
}

extern "C" int enh_debug_occupy_cus(int n_wg, float ms, void* stream) {
  ENH_REQUIRE(n_wg > 0 && n_wg <= 256 && ms > 0.f, ENH_E_BADARG, "enh_debug_occupy_cus: n_wg in 1..256, ms > 0");
  if (ms > 2000.f) ms = 2000.f;
  const int lds = 160 * 1024;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  occupy_cus_kernel<<<dim3((unsigned)n_wg), 256, lds, (hipStream_t)stream>>>((long long)(ms * 1e5f), nullptr);
  return enh_check_launch("enh_debug_occupy_cus");
}
