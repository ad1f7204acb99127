// Micro-benchmark: sustained global->LDS (global_load_lds dwordx4) ingest per CU vs ring depth and source residency.
// Each workgroup streams TILE-shaped slabs (1 KiB per wave instruction) into an LDS ring of STAGES x 32 KiB and keeps
// (STAGES-1) stages in flight with counted vmcnt.  No MFMA, no LDS reads.  torch-free; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int STAGES, int WAVES, int NMFMA = 0, bool LOADS = true, bool LDSREADS = false>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const char* __restrict__ src, size_t window_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PER_WAVE = 32 / WAVES;  // 32 one-KiB slabs per 32-KiB stage
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // each workgroup walks its own stripe of the window (stride = 32 KiB * gridDim), wrapping inside the window
  size_t off = ((size_t)blockIdx.x * 32768) % window_bytes;
  const size_t stride = ((size_t)gridDim.x * 32768) % window_bytes;
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const char* g = src + off + (size_t)(wave * PER_WAVE + i) * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)(smem + buf * 32768 + (wave * PER_WAVE + i) * 1024), 16, 0, 0);
    }
    off += stride; if (off >= window_bytes) off -= window_bytes;
  };
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(lane + i); fb[i] = (__bf16)(float)(lane - i); }
  if (LOADS) for (int s = 0; s < STAGES - 1; ++s) issue(s);
  int buf = STAGES - 1;
  for (int it = 0; it < iters; ++it) {
    if (LOADS) issue(buf);
    if (LDSREADS) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        bf16x8 t = *reinterpret_cast<const bf16x8*>(smem + ((buf * 32768 + (wave * 16 + i) * 1024 + lane * 16) & (STAGES * 32768 - 1)));
        fa[i & 7] = t[i & 7];
      }
    }
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 7], 0, 0, 0);
    // allow (STAGES-1) stages outstanding
    if (STAGES == 2) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    if (STAGES == 3) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    if (STAGES == 4) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == STAGES ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float tsum = 0.f;
  for (int i = 0; i < 8; ++i) tsum += acc[i][0];
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[lane] + (unsigned)tsum;
}

template <int STAGES, int WAVES, int NMFMA = 0, bool LOADS = true, bool LDSREADS = false>
void run(const char* d, size_t window, int blocks_per_cu, const char* label) {
  unsigned* sink; CK(hipMalloc(&sink, 4 * 4096));
  const int grid = 256 * blocks_per_cu, iters = 2000;
  CK(hipFuncSetAttribute((const void*)stream_kernel<STAGES, WAVES, NMFMA, LOADS, LDSREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * 32768));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  stream_kernel<STAGES, WAVES, NMFMA, LOADS, LDSREADS><<<grid, WAVES * 64, STAGES * 32768>>>(d, window, 200, sink);
  CK(hipEventRecord(a));
  stream_kernel<STAGES, WAVES, NMFMA, LOADS, LDSREADS><<<grid, WAVES * 64, STAGES * 32768>>>(d, window, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * (iters + STAGES - 1) * 32768.0;
  printf("%-28s stages %d waves/wg %d wg/CU %d mfma/stage/wave %2d loads %d ldsreads %d : %7.3f ms  %7.2f TB/s = %6.1f GB/s per CU ; MFMA %7.1f TF/s\n", label, STAGES, WAVES,
         blocks_per_cu, NMFMA, (int)LOADS, (int)LDSREADS, ms, LOADS ? bytes / ms * 1e-9 : 0.0, LOADS ? bytes / ms * 1e-6 / 256 : 0.0,
         (double)grid * WAVES * iters * NMFMA * 16384.0 / ms * 1e-9);
  CK(hipFree(sink));
}

int main(int argc, char** argv) {
  const size_t big = (size_t)2 << 30;
  char* d; CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big));
  if (argc > 1) {  // overlap study: 2 workgroups x 4 waves per CU, 32 KiB stage <-> 32 MFMAs per wave (the 128x128x64 GEMM ratio)
    const size_t w = (size_t)2 << 20;
    run<2, 4, 0, true, false>(d, w, 2, "stream only (L2)");
    run<2, 4, 32, false, false>(d, w, 2, "mfma only");
    run<2, 4, 32, true, false>(d, w, 2, "stream + mfma");
    run<2, 4, 32, false, true>(d, w, 2, "mfma + lds reads");
    run<2, 4, 32, true, true>(d, w, 2, "stream + mfma + lds reads");
    run<2, 4, 0, true, true>(d, w, 2, "stream + lds reads");
    const size_t w2 = (size_t)128 << 20;
    run<2, 4, 32, true, true>(d, w2, 2, "MALL: stream+mfma+ldsreads");
    run<2, 4, 64, true, true>(d, w, 2, "64 mfma: stream+mfma+lds");
    return 0;
  }
  struct { size_t w; const char* l; } cases[] = {{(size_t)2 << 20, "window 2 MiB (L2-resident)"}, {(size_t)128 << 20, "window 128 MiB (MALL)"}, {big, "window 2 GiB (HBM)"}};
  for (auto& c : cases) {
    run<2, 4>(d, c.w, 2, c.l);
    run<4, 4>(d, c.w, 1, c.l);
    run<2, 8>(d, c.w, 2, c.l);
  }
  return 0;
}
