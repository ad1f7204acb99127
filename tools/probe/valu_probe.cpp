// valu_probe — issue cost (shader cycles per wave64 instruction) of the VALU ops the attention softmax uses, alone and interleaved with
// v_mfma_f32_32x32x16_bf16, at 1 and 2 waves per SIMD.  Dev tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/valu_probe.cpp -o tools/probe/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// KIND: 0 v_exp_f32, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max3_f32, 5 v_pk_mul_f32, 6 v_pk_add_f32, 7 v_exp_f16 (none)
template <int KIND>
__device__ __forceinline__ void valu_op(float (&v)[16], int i) {
  float& a = v[i & 15];
  float& b = v[(i + 5) & 15];
  if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (KIND == 2) { f32x2 x = {v[(2 * i) & 15], v[(2 * i + 1) & 15]}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(x)); v[(2 * i) & 15] = x[0]; v[(2 * i + 1) & 15] = x[1]; }
  if (KIND == 3) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); a = __builtin_bit_cast(float, r); }
  if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (KIND == 5) { f32x2 x = {v[(2 * i) & 15], v[(2 * i + 1) & 15]}; asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(x)); v[(2 * i) & 15] = x[0]; v[(2 * i + 1) & 15] = x[1]; }
  if (KIND == 6) { f32x2 x = {v[(2 * i) & 15], v[(2 * i + 1) & 15]}; asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(x)); v[(2 * i) & 15] = x[0]; v[(2 * i + 1) & 15] = x[1]; }
}

// per iteration: NM MFMAs, each followed by NV VALU ops of KIND
template <int KIND, int NM, int NV, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe(int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  s16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
  fa[0] = (short)lane;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = -0.01f * (lane + i);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < (NM ? NM : 1); ++m) {
      if (NM) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb), __builtin_bit_cast(bf16x8, fa), acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NV; ++k) valu_op<KIND>(v, m * NV + k);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NM, int NV, int WPE>
static void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = 2000, grid = 256 * WPE;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<KIND, NM, NV, WPE><<<grid, 256>>>(iters, d_out, d_sink);
  CK(hipEventRecord(e0));
  probe<KIND, NM, NV, WPE><<<grid, 256>>>(iters, d_out, d_sink);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 4);
  CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
  double avg = 0; for (auto x : h) avg += (double)x; avg /= h.size();
  // s_memtime ticks at a constant 100 MHz; wall time / iterations gives ns per iteration, so report both
  const int nm = NM ? NM : 1;
  printf("%-18s waves/SIMD %d  MFMA/iter %d  VALU per MFMA %2d : %8.1f ns/iter  = %7.2f ns per (MFMA + %d VALU)   [counter %.0f]\n", name, WPE, NM, NV, ms * 1e6 / iters, ms * 1e6 / iters / nm, NV, avg / iters);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  CK(hipMalloc(&d_out, 8 * 4096)); CK(hipMalloc(&d_sink, 64));
#define ROW(K, name) \
  run<K, 0, 16, 1>(name, d_out, d_sink); run<K, 0, 16, 2>(name, d_out, d_sink); \
  run<K, 8, 0, 1>(name, d_out, d_sink); run<K, 8, 2, 1>(name, d_out, d_sink); run<K, 8, 4, 1>(name, d_out, d_sink); run<K, 8, 8, 1>(name, d_out, d_sink); \
  run<K, 8, 4, 2>(name, d_out, d_sink); run<K, 8, 8, 2>(name, d_out, d_sink);
  ROW(0, "v_exp_f32")
  ROW(1, "v_fma_f32")
  ROW(2, "v_pk_fma_f32")
  ROW(3, "v_cvt_pk_bf16_f32")
  ROW(4, "v_max3_f32")
  ROW(5, "v_pk_mul_f32")
  ROW(6, "v_pk_add_f32")
  return 0;
}
