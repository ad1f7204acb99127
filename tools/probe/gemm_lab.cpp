// gemm_lab — standalone (torch-free) laboratory for the bf16 GEMM main loop on MI355X.  Dev tool, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/gemm_lab.cpp -o tools/probe/gemm_lab -L enhancing-transformers_amd/lib -lenh_hip
// Runs C[M][N] = A[M][K] . B[N][K]^T (the forward "NT" role) with
//   * the product library's kernel (enh_gemm_bf16, whatever family it selects)               -> baseline in the same process
//   * "w4": 256x256 tile, FOUR waves (one per SIMD, 128x128 each, 256 accumulator registers), 4-slot ring of 32-deep K stages
//           (32 KiB each) filled by global_load_lds three stages ahead, one barrier per stage, ds_read / MFMA software-pipelined
//   * ablations of w4 (no loads / no LDS reads / MFMA only) to locate the bound
// and checks sampled outputs against an fp32 dot-product kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/enh_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __host__ inline uint16_t f2bf(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __host__ inline float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)(i * 2654435761u) ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f;  // uniform [-1, 1)
    p[i] = f2bf(u * scale);
  }
}

// sampled check: out[s] = sum_k A[r_s][k] * B[c_s][k]  (fp32 accumulate in fp64 order-insensitive enough for a 1e-3 check)
__global__ void ref_samples(const uint16_t* A, const uint16_t* B, int64_t K, int64_t lda, int64_t ldb, const int* rows, const int* cols, int ns, double* out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  double acc = 0.0;
  const uint16_t* a = A + (int64_t)rows[s] * lda;
  const uint16_t* b = B + (int64_t)cols[s] * ldb;
  for (int64_t k = 0; k < K; ++k) acc += (double)bf2f(a[k]) * (double)bf2f(b[k]);
  out[s] = acc;
}

// -------------------------------------------------------------------------------------------------------------------
// w4 kernel
// -------------------------------------------------------------------------------------------------------------------
// LDS: 4 slots x 32 KiB.  A slot holds one 32-deep K stage of the 256 x 256 tile as four sub-tiles [A rows 0-127 | A rows 128-255 |
// B cols 0-127 | B cols 128-255], each 128 rows x 64 B ("row32" image): the 16-B chunk c (0..3) of row r lives at
//     r*64 + ((c ^ ((r >> 2) & 3)) << 4)
// so that a ds_read_b128 lane group (16 lanes = rows {0-3,12-15,20-27} + 4*j, one chunk) covers all 16 slots of the 256-B bank row.
// global_load_lds writes lane-linearly (16 rows x 64 B per wave instruction): the XOR goes on the SOURCE chunk.
#define W4_SLOT 32768
#define W4_SUB 8192
__device__ __forceinline__ int row32_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

struct W4Args {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  int64_t M, N, K;
  float* c_f32; uint16_t* c_bf16; int64_t ldc;
  int nbm, nbn;
};

__device__ __forceinline__ void w4_tile_coords(const W4Args& a, int& tile_m, int& tile_n) {
  const int nwg = a.nbm * a.nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int per_group = 8 * a.nbn;
  const int grp = bid / per_group, within = bid - grp * per_group;
  const int rows = (a.nbm - grp * 8) < 8 ? (a.nbm - grp * 8) : 8;
  tile_m = grp * 8 + within % rows;
  tile_n = within / rows;
}

// FLAGS bit 0: skip global->LDS loads ; bit 1: skip LDS fragment reads ; bit 2: skip MFMAs ; bit 3: skip the epilogue stores
template <int FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(const W4Args args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool LOADS = !(FLAGS & 1), READS = !(FLAGS & 2), MFMA = !(FLAGS & 4), STORE = !(FLAGS & 8);
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n;
  w4_tile_coords(args, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;
  const int nst = (int)(args.K / 32);  // number of 32-deep stages

  // staging: wave w fills sub-tile w of every slot (w = 0,1: A rows w*128.. ; w = 2,3: B cols (w-2)*128..), 8 instructions of 16 rows
  const uint16_t* gsrc;
  int64_t gld;
  {
    const uint16_t* P = wave < 2 ? args.A : args.B;
    gld = wave < 2 ? args.lda : args.ldb;
    const int64_t x0 = wave < 2 ? m0 + wave * 128 : n0 + (wave - 2) * 128;
    const int r = lane >> 2, pc = lane & 3;           // row within the 16-row slab, physical chunk
    const int c = pc ^ ((r >> 2) & 3);                // logical chunk stored there (slab base rows are multiples of 16: (r>>2)&3 unaffected)
    gsrc = P + (x0 + r) * gld + c * 8;
  }
  const int64_t slab_step = 16 * gld;                 // elements between consecutive 16-row slabs
  unsigned char* const my_sub = smem + wave * W4_SUB;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { fa0[i] = fb0[i] = fa1[i] = fb1[i] = (s16x8){1, 2, 3, 4, 5, 6, 7, 8}; }

  const int l31 = lane & 31, hi = lane >> 5;
  // fragment (32 rows x 16 k) of sub-tile row-block i at k-step s (0,1): lane -> row i*32 + l31, chunk s*2 + hi
#define W4_ISSUE_ONE(SLOT, U)                                                                                                     \
  do {                                                                                                                            \
    if (LOADS)                                                                                                                    \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)(gsrc + (U) * slab_step), (LDS_AS void*)(my_sub + (SLOT) * W4_SLOT + (U) * 1024), 16, 0, 0); \
  } while (0)
#define W4_ISSUE_STAGE(SLOT)                                                                                                      \
  do {                                                                                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) W4_ISSUE_ONE(SLOT, u_);                                                      \
    gsrc += 32;                                                                                                                   \
  } while (0)
#define W4_READ(FA, FB, SLOT, S)                                                                                                  \
  do {                                                                                                                            \
    if (READS) {                                                                                                                  \
      const unsigned char* sa_ = smem + (SLOT) * W4_SLOT + wm * W4_SUB;                                                           \
      const unsigned char* sb_ = smem + (SLOT) * W4_SLOT + (2 + wn) * W4_SUB;                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) FA[i_] = *reinterpret_cast<const s16x8*>(sa_ + row32_off(i_ * 32 + l31, (S) * 2 + hi)); \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) FB[j_] = *reinterpret_cast<const s16x8*>(sb_ + row32_off(j_ * 32 + l31, (S) * 2 + hi)); \
    }                                                                                                                             \
  } while (0)
#define W4_MMA(FA, FB)                                                                                                            \
  do {                                                                                                                            \
    if (MFMA) {                                                                                                                   \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                          \
          acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FB[j_]), __builtin_bit_cast(bf16x8, FA[i_]), acc[i_][j_], 0, 0, 0); \
    }                                                                                                                             \
  } while (0)

  // one MFMA of the 4 x 4 block: index q = i*4 + j
#define W4_MM(Q, FA, FB)                                                                                                          \
  do {                                                                                                                            \
    if (MFMA) acc[(Q) >> 2][(Q) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FB[(Q) & 3]), __builtin_bit_cast(bf16x8, FA[(Q) >> 2]), acc[(Q) >> 2][(Q) & 3], 0, 0, 0); \
  } while (0)
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // first half of a stage: MFMAs on k-step 0 (F0) with the reads of k-step 1 (F1) issued under the first MFMAs
#define W4_HALF_A(SLOT)                                                                                                           \
  do {                                                                                                                            \
    W4_MM(0, fa0, fb0); W4_MM(1, fa0, fb0);                                                                                       \
    W4_FENCE();                                                                                                                   \
    W4_READ(fa1, fb1, SLOT, 1);                                                                                                   \
    W4_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 2; q_ < 16; ++q_) W4_MM(q_, fa0, fb0);                                                        \
    W4_FENCE();                                                                                                                   \
  } while (0)
  // second half: MFMAs on k-step 1 (F1); the refill of the vacated slot is spread one load per MFMA, then the reads of the next
  // stage's k-step 0 (F0)
#define W4_HALF_B(SLOT, NEXT_SLOT, ISSUE, READ_NEXT)                                                                              \
  do {                                                                                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                                            \
      W4_MM(q_, fa1, fb1);                                                                                                        \
      if (ISSUE) { W4_ISSUE_ONE(SLOT, q_); }                                                                                      \
      W4_FENCE();                                                                                                                 \
    }                                                                                                                             \
    if (ISSUE) gsrc += 32;                                                                                                        \
    W4_MM(8, fa1, fb1); W4_MM(9, fa1, fb1);                                                                                       \
    W4_FENCE();                                                                                                                   \
    if (READ_NEXT) { W4_READ(fa0, fb0, NEXT_SLOT, 0); }                                                                           \
    W4_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 10; q_ < 16; ++q_) W4_MM(q_, fa1, fb1);                                                       \
    W4_FENCE();                                                                                                                   \
  } while (0)

  // prologue: stages 0..3 fill the four slots; only stage 0 has to have landed (requires nst >= 4)
  W4_ISSUE_STAGE(0); W4_ISSUE_STAGE(1); W4_ISSUE_STAGE(2); W4_ISSUE_STAGE(3);
  __builtin_amdgcn_s_waitcnt(0x4F78);       // vmcnt(24)
  __builtin_amdgcn_s_barrier();
  W4_READ(fa0, fb0, 0, 0);
  W4_FENCE();

  int j = 0;
  for (; j + 4 < nst; ++j) {               // steady state: stages j+1 .. j+3 in flight, stage j+4 issued in the second half
    const int slot = j & 3;
    W4_HALF_A(slot);
    __builtin_amdgcn_s_waitcnt(0x4070);    // vmcnt(16): my loads of stage j+1 have landed ; lgkmcnt(0): my reads of this slot are complete
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    W4_HALF_B(slot, (j + 1) & 3, true, true);
  }
  // tail: the last four stages, nothing left to issue
  {
    W4_HALF_A(j & 3);
    __builtin_amdgcn_s_waitcnt(0x4070);    // vmcnt(16)
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    W4_HALF_B(j & 3, (j + 1) & 3, false, true);
    ++j;
    W4_HALF_A(j & 3);
    __builtin_amdgcn_s_waitcnt(0x0078);    // vmcnt(8)
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    W4_HALF_B(j & 3, (j + 1) & 3, false, true);
    ++j;
    W4_HALF_A(j & 3);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    W4_HALF_B(j & 3, (j + 1) & 3, false, true);
    ++j;
    W4_HALF_A(j & 3);
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0)
    W4_FENCE();
    W4_HALF_B(j & 3, 0, false, false);
  }
#undef W4_MM
#undef W4_FENCE
#undef W4_HALF_A
#undef W4_HALF_B
#undef W4_ISSUE_ONE
#undef W4_ISSUE_STAGE
#undef W4_READ
#undef W4_MMA

  // epilogue (swapped operands: acc[i][j][r] = C[m0 + wm*128 + i*32 + (lane&31)][n0 + wn*128 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)])
  if (STORE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wm * 128 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = n0 + wn * 128 + j * 32 + 8 * g4 + 4 * hi;
          const float v0 = acc[i][j][g4 * 4 + 0], v1 = acc[i][j][g4 * 4 + 1], v2 = acc[i][j][g4 * 4 + 2], v3 = acc[i][j][g4 * 4 + 3];
          if (args.c_f32) { const f32x4 o = {v0, v1, v2, v3}; *reinterpret_cast<f32x4*>(args.c_f32 + m * args.ldc + n) = o; }
          if (args.c_bf16) {
            const u32x2 o = {(uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16), (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16)};
            *reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n) = o;
          }
        }
      }
    }
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1234.5678f) args.c_f32[0] = s;   // keeps the accumulators live
  }
}

template <int FLAGS>
static void launch_w4(const W4Args& a, hipStream_t s) {
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * W4_SLOT)); set = true; }
  gemm_w4_kernel<FLAGS><<<dim3(a.nbm * a.nbn), 256, 4 * W4_SLOT, s>>>(a);
}

static double time_ms(void (*fn)(void*), void* ctx, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn(ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) fn(ctx);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

struct Ctx { W4Args a; int variant; };
static void run_variant(void* p) {
  Ctx* c = (Ctx*)p;
  switch (c->variant) {
    case 0: launch_w4<0>(c->a, 0); break;
    case 1: launch_w4<1>(c->a, 0); break;
    case 2: launch_w4<2>(c->a, 0); break;
    case 3: launch_w4<3>(c->a, 0); break;
    case 11: launch_w4<11>(c->a, 0); break;   // MFMA only, no epilogue
    case 8: launch_w4<8>(c->a, 0); break;     // full main loop, no epilogue
    case 100: {
      int rc = enh_gemm_bf16((const enh_bf16*)c->a.A, c->a.lda, 0, (const enh_bf16*)c->a.B, c->a.ldb, 0, c->a.M, c->a.N, c->a.K, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0,
                             c->a.c_f32, (enh_bf16*)c->a.c_bf16, c->a.ldc, nullptr);
      if (rc) { printf("enh_gemm_bf16 rc=%d %s\n", rc, enh_last_error()); exit(1); }
    } break;
  }
}

int main(int argc, char** argv) {
  int64_t M = argc > 1 ? atoll(argv[1]) : 4096, N = argc > 2 ? atoll(argv[2]) : 4096, K = argc > 3 ? atoll(argv[3]) : 4096;
  const bool f32out = argc > 4 && atoi(argv[4]);
  const int iters = argc > 5 ? atoi(argv[5]) : 10;
  uint16_t *A, *B, *C16 = nullptr; float* C32 = nullptr;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2));
  if (f32out) CK(hipMalloc(&C32, (size_t)M * N * 4)); else CK(hipMalloc(&C16, (size_t)M * N * 2));
  fill_kernel<<<2048, 256>>>(A, (size_t)M * K, 0x1234567u, 1.0f);
  fill_kernel<<<2048, 256>>>(B, (size_t)N * K, 0x89abcdeu, 1.0f);
  CK(hipDeviceSynchronize());
  Ctx c;
  c.a.A = A; c.a.lda = K; c.a.B = B; c.a.ldb = K; c.a.M = M; c.a.N = N; c.a.K = K; c.a.c_f32 = C32; c.a.c_bf16 = C16; c.a.ldc = N;
  c.a.nbm = (int)(M / 256); c.a.nbn = (int)(N / 256);
  const double fl = 2.0 * M * N * K;
  // samples for the check
  const int ns = 4096;
  std::vector<int> hr(ns), hc(ns);
  uint32_t x = 12345;
  for (int i = 0; i < ns; ++i) { x = x * 1664525u + 1013904223u; hr[i] = (int)((x >> 8) % M); x = x * 1664525u + 1013904223u; hc[i] = (int)((x >> 8) % N); }
  hr[0] = 0; hc[0] = 0; hr[1] = (int)M - 1; hc[1] = (int)N - 1; hr[2] = (int)M - 1; hc[2] = 0; hr[3] = 0; hc[3] = (int)N - 1;
  int *dr, *dc; double* dref;
  CK(hipMalloc(&dr, ns * 4)); CK(hipMalloc(&dc, ns * 4)); CK(hipMalloc(&dref, ns * 8));
  CK(hipMemcpy(dr, hr.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), ns * 4, hipMemcpyHostToDevice));
  ref_samples<<<(ns + 63) / 64, 64>>>(A, B, K, K, K, dr, dc, ns, dref);
  std::vector<double> href(ns);
  CK(hipMemcpy(href.data(), dref, ns * 8, hipMemcpyDeviceToHost));
  auto check = [&](const char* name) {
    double num = 0, den = 0;
    std::vector<float> row32(1); std::vector<uint16_t> row16(1);
    for (int i = 0; i < ns; ++i) {
      double got;
      if (f32out) { float v; CK(hipMemcpy(&v, C32 + (size_t)hr[i] * N + hc[i], 4, hipMemcpyDeviceToHost)); got = v; }
      else { uint16_t v; CK(hipMemcpy(&v, C16 + (size_t)hr[i] * N + hc[i], 2, hipMemcpyDeviceToHost)); got = bf2f(v); }
      num += (got - href[i]) * (got - href[i]); den += href[i] * href[i];
    }
    printf("    check %-10s rel err %.3e over %d samples\n", name, sqrt(num / den), ns);
  };
  struct V { int id; const char* name; bool checkable; };
  const V vs[] = {{100, "library", true}, {0, "w4", true}, {1, "w4-noload", false}, {2, "w4-noread", false}, {3, "w4-noload-noread", false},
                  {8, "w4-noepi", false}, {11, "w4-mfma-only", false}};
  printf("M=%lld N=%lld K=%lld out=%s\n", (long long)M, (long long)N, (long long)K, f32out ? "f32" : "bf16");
  for (int rep = 0; rep < 2; ++rep)
    for (const V& v : vs) {
      if (argc > 6 && !strstr(argv[6], v.name) ) continue;
      c.variant = v.id;
      if (f32out) CK(hipMemset(C32, 0xff, (size_t)M * N * 4)); else CK(hipMemset(C16, 0xff, (size_t)M * N * 2));
      const double ms = time_ms(run_variant, &c, iters);
      printf("  %-18s %8.3f ms  %7.1f TF/s\n", v.name, ms, fl / ms / 1e9);
      if (v.checkable && rep == 0) check(v.name);
    }
  return 0;
}
