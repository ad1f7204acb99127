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

template <int STAGES, int WAVES>
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
  for (int s = 0; s < STAGES - 1; ++s) issue(s);
  int buf = STAGES - 1;
  for (int it = 0; it < iters; ++it) {
    issue(buf);
    // allow (STAGES-1) stages outstanding
    if (STAGES == 2) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    if (STAGES == 3) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    if (STAGES == 4) { if (PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == STAGES ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[lane];
}

template <int STAGES, int WAVES>
void run(const char* d, size_t window, int blocks_per_cu, const char* label) {
  unsigned* sink; CK(hipMalloc(&sink, 4 * 4096));
  const int grid = 256 * blocks_per_cu, iters = 2000;
  CK(hipFuncSetAttribute((const void*)stream_kernel<STAGES, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * 32768));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  stream_kernel<STAGES, WAVES><<<grid, WAVES * 64, STAGES * 32768>>>(d, window, 200, sink);
  CK(hipEventRecord(a));
  stream_kernel<STAGES, WAVES><<<grid, WAVES * 64, STAGES * 32768>>>(d, window, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * (iters + STAGES - 1) * 32768.0;
  printf("%-28s stages %d waves/wg %d wg/CU %d in-flight/CU %3d KiB : %7.2f TB/s aggregate = %6.1f GB/s per CU\n", label, STAGES, WAVES,
         blocks_per_cu, (STAGES - 1) * 32 * blocks_per_cu, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
  CK(hipFree(sink));
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* d; CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big));
  struct { size_t w; const char* l; } cases[] = {{(size_t)2 << 20, "window 2 MiB (L2-resident)"}, {(size_t)128 << 20, "window 128 MiB (MALL)"}, {big, "window 2 GiB (HBM)"}};
  for (auto& c : cases) {
    run<2, 4>(d, c.w, 2, c.l);
    run<3, 4>(d, c.w, 1, c.l);
    run<4, 4>(d, c.w, 1, c.l);
    run<2, 8>(d, c.w, 2, c.l);
    run<4, 8>(d, c.w, 1, c.l);
    run<2, 4>(d, c.w, 1, c.l);
  }
  return 0;
}
