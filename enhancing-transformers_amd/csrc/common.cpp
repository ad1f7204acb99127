// common.cpp — error state + ABI version for libenh_hip.so
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void enh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int enh_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    enh_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return ENH_E_HIP_BASE - (int)e;
  }
  return ENH_OK;
}

extern "C" const char* enh_last_error(void) { return g_err; }
extern "C" int enh_abi_version(void) { return ENH_ABI_VERSION; }

// CU budget (round 4): how many CUs the launches that size themselves by the CU count may count on — persistent GEMM grids, one-round split-K plans,
// the LayerNorm backward's one-workgroup-per-CU grid.  Data-parallel training runs RCCL's kernels beside the backward pass (enhancing/engine/ddp.py;
// reference main.py:54-57); a grid sized for ALL CUs then has workgroups that wait for a CU until another retires (2x on that launch, measured in
// profiles/r04_comm_contention.txt).  0 = every CU of the device.  Explicit state behind an explicit call: the library reads no environment.
// Device-bound state is keyed by the CURRENT device of the calling thread (round 6; until round 5 the CU count, the budget and the persistent GEMMs' tile-claim
// counters belonged to the first device the library was used on — fine for one process per GPU, a trap for a process that drives several devices through
// the C ABI): a budget set while device d is current applies to launches issued while d is current.
int enh_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < ENH_MAX_DEVICES ? dev : 0;
}
static int g_cu_budget[ENH_MAX_DEVICES] = {0};
static int g_device_cus[ENH_MAX_DEVICES] = {0};      // 0 = not asked yet (benign race: every thread writes the same value)
int enh_device_cus() {
  const int dev = enh_current_device();
  if (g_device_cus[dev] == 0) {
    int c = 256;
    (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
    g_device_cus[dev] = c > 0 ? c : 256;
  }
  return g_device_cus[dev];
}
int enh_cu_budget() {
  const int b = g_cu_budget[enh_current_device()], n = enh_device_cus();
  return b > 0 && b < n ? b : n;
}
extern "C" int enh_set_cu_budget(int n_cus) {
  ENH_REQUIRE(n_cus >= 0, ENH_E_BADARG, "enh_set_cu_budget: n_cus must be >= 0 (0 = all)");
  g_cu_budget[enh_current_device()] = n_cus;
  return ENH_OK;
}
extern "C" int enh_get_cu_budget(void) { return enh_cu_budget(); }

// Zero fill as a KERNEL.  hipMemsetAsync inside the library is not graph-safe on this stack: a captured 2-KiB memset node re-executed only half of its range
// on replay (found by the graph-replay bit-identity test of the two-optimizer step: garbage in half of a bias gradient from the second replay on;
// reproduced in isolation with tools/graph_probe2.py).  Every "out = 0 before the atomics / partial sums" of the library goes through this.
__global__ __launch_bounds__(256) void enh_zero_f32_kernel(float* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
int enh_zero_f32_launch(float* p, int64_t n, hipStream_t s) {
  if (n <= 0) return ENH_OK;
  enh_zero_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(p, n);
  return enh_check_launch("enh_zero_f32");
}
