// issue_probe — how many cycles of MFMA cover does one memory instruction need when there is ONE wave per SIMD (in-order issue)?
// Each wave runs: loop { NM x v_mfma_f32_32x32x16_bf16 (independent accumulators) ; one instruction of KIND } and reports shader cycles per
// iteration.  With no memory instruction an iteration costs NM*32 cycles; the excess is the exposed issue cost.  Dev tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/issue_probe.cpp -o tools/probe/issue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// KIND: 0 none, 1 global_load_lds x4 (64 lanes), 2 global_load_lds x4 issued as two half-wave instructions, 3 ds_read_b128, 4 global_load_dwordx4 -> VGPR,
//       5 global_load_lds under half exec only (one half per iteration), 6 global_load_dwordx4 + ds_write_b128 of the previous one
template <int KIND, int NM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const unsigned char* __restrict__ src, int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  s16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
  fa[0] = (short)lane; fb[1] = (short)(lane * 3);
  const unsigned char* g = src + ((size_t)blockIdx.x * 4 + wave) * 8192 + lane * 16;   // 8 KiB per wave, re-read every 8 iterations: L2 / L1 resident
  unsigned char* l = smem + wave * 8192;
  u32x4 keep = {0, 0, 0, 0}, prev = {0, 0, 0, 0};
  s16x8 rd = fa;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int slab = it & 7;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb), __builtin_bit_cast(bf16x8, fa), acc[m & 7], 0, 0, 0);
      if (KIND == 7) {   // staggered: wave w issues its load under MFMA w (wave-uniform branch)
        __builtin_amdgcn_sched_barrier(0);
        if (wave == (m & 3)) {
          __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + slab * 1024), (LDS_AS void*)(l + slab * 1024), 16, 0, 0);
          __builtin_amdgcn_s_waitcnt(0x4F78);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (KIND == 1) {
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + slab * 1024), (LDS_AS void*)(l + slab * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x4F78);   // vmcnt(24): bounded queue, never a drain
    } else if (KIND == 2) {
      if (lane < 32) __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + slab * 1024), (LDS_AS void*)(l + slab * 1024), 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (lane >= 32) __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + slab * 1024), (LDS_AS void*)(l + slab * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x8F70);   // vmcnt(32)
    } else if (KIND == 5) {
      if ((lane < 32) == ((it & 1) == 0)) __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + slab * 1024), (LDS_AS void*)(l + slab * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x4F78);
    } else if (KIND == 3) {   // ds_read_b128 through asm: no compiler-inserted wait, no consumer
      const unsigned addr = (unsigned)(uintptr_t)(LDS_AS const void*)(l + ((slab * 1024 + lane * 16) ^ ((lane >> 3) << 4)));
      asm volatile("ds_read_b128 %0, %1" : "+v"(keep) : "v"(addr) : "memory");
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else if (KIND == 4) {   // global_load_dwordx4 -> VGPR through asm
      const unsigned char* p = g + slab * 1024;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(keep) : "v"(p) : "memory");
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else if (KIND == 8) {   // ds_write_b128
      const unsigned addr = (unsigned)(uintptr_t)(LDS_AS const void*)(l + slab * 1024 + lane * 16);
      asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(prev) : "memory");
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    } else if (KIND == 6) {   // register staging: one global_load_dwordx4 and one ds_write_b128 per iteration, both un-waited
      const unsigned char* p = g + slab * 1024;
      const unsigned addr = (unsigned)(uintptr_t)(LDS_AS const void*)(l + slab * 1024 + lane * 16);
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(keep) : "v"(p) : "memory");
      asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(prev) : "memory");
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
  if (s == 123.456f) sink[0] = s + (float)keep[0] + (float)rd[0] + (float)prev[1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND, int NM>
static void run(const char* name, const unsigned char* src, unsigned long long* dout, float* sink, int nblk) {
  const int iters = 4000;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<KIND, NM>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  probe<KIND, NM><<<nblk, 256, 32768>>>(src, iters, dout, sink);
  probe<KIND, NM><<<nblk, 256, 32768>>>(src, iters, dout, sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(nblk);
  CK(hipMemcpy(h.data(), dout, nblk * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (auto v : h) s += (double)v;
  const double cyc = s / nblk / iters;
  printf("  %-44s NM=%d : %7.1f cycles/iter  (MFMA alone %4d)  exposed %6.1f\n", name, NM, cyc, NM * 32, cyc - NM * 32);
}

int main() {
  const int nblk = 256;
  unsigned char* src; unsigned long long* dout; float* sink;
  CK(hipMalloc(&src, (size_t)nblk * 4 * 8192)); CK(hipMemset(src, 1, (size_t)nblk * 4 * 8192));
  CK(hipMalloc(&dout, nblk * 8)); CK(hipMalloc(&sink, 64));
#define ROW(KIND, NAME) run<KIND, 1>(NAME, src, dout, sink, nblk); run<KIND, 2>(NAME, src, dout, sink, nblk); run<KIND, 4>(NAME, src, dout, sink, nblk);
  ROW(0, "no memory instruction")
  ROW(1, "global_load_lds dwordx4, 64 lanes")
  ROW(2, "global_load_lds dwordx4 as two half-wave instr")
  ROW(5, "global_load_lds dwordx4, 32 lanes per iteration")
  run<7, 4>("global_load_lds, wave w under MFMA w (staggered)", src, dout, sink, nblk);
  run<7, 8>("global_load_lds, 2 per 8 MFMAs, staggered", src, dout, sink, nblk);
  ROW(3, "ds_read_b128")
  ROW(4, "global_load_dwordx4 -> VGPR")
  ROW(8, "ds_write_b128")
  ROW(6, "global_load_dwordx4 + ds_write_b128 (reg staging)")
  return 0;
}
