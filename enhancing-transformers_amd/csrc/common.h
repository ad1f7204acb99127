// common.h — shared device/host helpers for libenh_hip.so (gfx950 only; no compatibility layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/enh_hip.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
// cache-policy bits of the LDS-DMA staging requests.  sc1 (agent scope: no allocation in the CU's vector L1, which no tile ever hits) makes the split-K
// weight gradient 6.5 % faster IN ISOLATION (tools/wgrad_lab.py: none 535 us, sc0 512-525, sc1 503, sc0 + sc1 503, nt 570, sc1 + nt 552) and changes nothing
// in the training step (same-box interleaved A/B of two libraries, tools/gpu_session.sh lib-ab; round 4: tools/r4_sc1_ab.sh: 618.4 vs 619.3 images/s, adversarial step 207.3 vs 207.7): inside
// the step the operands come straight from their producers.  Left at the default.
#ifdef ENH_GLDS_AUX_OVERRIDE
#define ENH_GLDS_AUX ENH_GLDS_AUX_OVERRIDE
#else
#define ENH_GLDS_AUX 0
#endif

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;

void enh_set_error(const char* fmt, ...);
int enh_check_launch(const char* what);
int enh_zero_f32_launch(float* p, int64_t n, hipStream_t s);   // p[0..n) = 0 as a kernel launch (graph-safe; see common.cpp)
#define ENH_MAX_DEVICES 64
int enh_current_device();   // hipGetDevice of the calling thread, clamped to [0, ENH_MAX_DEVICES)
int enh_device_cus();   // CUs of the current device (256 on MI355X)
int enh_cu_budget();    // CUs a launch sized by the CU count may count on (enh_set_cu_budget; = enh_device_cus() unless a budget is set)
int enh_colsum_reduce_launch(const float* part, int chunks, int64_t N, float* out, int accumulate, hipStream_t s);   // elementwise.hip

#define ENH_REQUIRE(cond, code, ...)                  \
  do {                                                \
    if (!(cond)) {                                    \
      enh_set_error(__VA_ARGS__);                     \
      return (code);                                  \
    }                                                 \
  } while (0)

// round-to-nearest-even f32 -> bf16 bits (NaN kept quiet)
__device__ __host__ inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __host__ inline float bf16_bits_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two f32 -> packed bf16x2, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---- 16-bit operand types -------------------------------------------------------------------------------------------------------------------
// The MFMA operands of the product path are 16-bit floats in one of two formats, chosen per CALL by the `dtype` argument of the C ABI (ENH_DT_BF16 /
// ENH_DT_F16, include/enh_hip.h) and per KERNEL by a type tag: every kernel that reads or writes 16-bit operands is a template over OT = BF16 | F16.  The
// formats differ in exactly three places — the MFMA opcode, the f32 -> 16-bit packing (round-to-nearest-even in both: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32;
// never the round-toward-zero v_cvt_pkrtz) and the 16-bit -> f32 widening — so LDS images, swizzles, LDS-DMA and the transpose reads are shared code.
// fp16 (the reference's --use_amp dtype, main.py:25,52: Lightning precision=16) has 11 significand bits against bf16's 8: ~8x smaller operand rounding at
// the same MFMA rate, for a range (max 65504, normals down to 6.1e-5) that the O(1) operands behind a LayerNorm / tanh / softmax never leave; gradients
// need a loss scale (engine/stage1.py).
struct BF16 { static constexpr int id = 0; };
struct F16 { static constexpr int id = 1; };
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// two f32 -> packed 16-bit pair (element 0 in the low half), round-to-nearest-even
template <typename OT>
__device__ inline uint32_t pack2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  if constexpr (OT::id == 0) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// the low / high element of a packed pair, and a single element, widened to f32 (exact)
template <typename OT>
__device__ inline float unpack_lo(uint32_t u) {
  if constexpr (OT::id == 0) return __builtin_bit_cast(float, u << 16);
  else return (float)__builtin_bit_cast(f16x2_t, u)[0];
}
template <typename OT>
__device__ inline float unpack_hi(uint32_t u) {
  if constexpr (OT::id == 0) return __builtin_bit_cast(float, u & 0xffff0000u);
  else return (float)__builtin_bit_cast(f16x2_t, u)[1];
}
template <typename OT>
__device__ inline float unpack1(uint16_t h) {
  if constexpr (OT::id == 0) return __builtin_bit_cast(float, (uint32_t)h << 16);
  else return (float)__builtin_bit_cast(_Float16, h);
}
template <typename OT>
__device__ inline uint16_t pack1(float f) {
  if constexpr (OT::id == 0) return f32_to_bf16_bits(f);
  else return __builtin_bit_cast(uint16_t, (_Float16)f);
}
// D = A B + C on the matrix cores, f32 accumulation; operands as raw 16-bit lanes (s16x8 = 8 contraction elements per lane)
template <typename OT>
__device__ __forceinline__ f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
  if constexpr (OT::id == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <typename OT>
__device__ __forceinline__ f32x4 mfma16(const s16x8& a, const s16x8& b, const f32x4& c) {
  if constexpr (OT::id == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// run CALL with `OT` bound to the operand type tag of a C-ABI dtype value (validated by the caller with ENH_REQUIRE_DT)
#define ENH_DT_DISPATCH(dtype, CALL)                    \
  do {                                                   \
    if ((dtype) == ENH_DT_F16) { typedef F16 OT; CALL; } \
    else { typedef BF16 OT; CALL; }                      \
  } while (0)
#define ENH_REQUIRE_DT(dtype, what) ENH_REQUIRE((dtype) == ENH_DT_BF16 || (dtype) == ENH_DT_F16, ENH_E_BADARG, what ": dtype must be ENH_DT_BF16 (0) or ENH_DT_F16 (1), got %d", (int)(dtype))

// tanh for the x3 path, ONE function for the stand-alone split kernel (x3.hip split3) and the fused GEMM epilogue (gemm_tiles.h EPI_BF16_TANH_SPLIT), so the
// two forms of the x3 forward produce the same bits whatever the batch size selects: the path keeps ~2^-17 relative per element (hi + lo), which
// 1 - 2 / (exp(2x) + 1) alone loses below |x| ~ 0.1 (absolute error ~1e-7 against a small result) — small arguments take the odd series
// x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315) (next term 62 x^9/2835: < 3e-9 relative at 0.12).  exp2 / rcp on the transcendental unit.
__device__ __forceinline__ float tanh_x3(float x) {
  const float x2 = x * x;
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
  const float small = x * (1.f + x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * -0.05396825397f)));
  return x2 < 0.0144f ? small : big;
}

// LDS transpose read (ds_read_b64_tr_b16).  Verified on MI355X (profiles/hw_probe_r01.txt):
// within each 16-lane group, result(lane i, elem j) = the 16-bit element (i % 4) of the 8 bytes that lane
// (j*4 + i/4) of the same group addressed.
__device__ inline s16x4 lds_tr_read_b64(const void* lds_ptr) {
  bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_t*)lds_ptr);
  return __builtin_bit_cast(s16x4, v);
}

// Same instruction issued through inline asm: the compiler's wait-count pass treats the tr-read INTRINSIC as a potential
// reader of in-flight LDS-DMA data and inserts s_waitcnt vmcnt(0) in front of it, which destroys the load pipeline of
// kernels that keep global_load_lds in flight across their fragment reads.  With the asm form nothing is inserted: the
// caller must (a) have waited (vmcnt + barrier) for the data this read needs and (b) wait lgkmcnt before using the result
// (cdna_hip_programming.md §5.7 item 1) — the t256 GEMM schedule does both explicitly.
__device__ inline void lds_tr_read_b64_asm(unsigned long long& out, const void* lds_ptr) {
  const unsigned addr = (unsigned)(uintptr_t)(LDS_AS const void*)lds_ptr;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(out) : "v"(addr) : "memory");
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Fixed-order reduction of `rows` partial rows: returns, for thread (col = threadIdx.x & 15, grp = threadIdx.x >> 4) of a 256-thread workgroup handling
// columns n0 .. n0+15, the sum over all rows of part[r*stride + n] — valid in the threads with grp == 0.  Rows are split into 16 contiguous groups
// summed concurrently (ascending inside a group), the 16 group sums are then added in ascending order: the same bits on every run, and 16x less serial
// latency than one thread walking all rows (a 512-row walk by one thread per column cost ~100 us per call).
__device__ inline float fixed_order_rowsum16(const float* __restrict__ part, int rows, int64_t stride, int64_t n, bool n_ok, float* s_red /* [16][16] */) {
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int per = (rows + 15) / 16;
  const int r0 = grp * per, r1 = r0 + per < rows ? r0 + per : rows;
  float a = 0.f;
  if (n_ok)
    for (int r = r0; r < r1; ++r) a += part[(int64_t)r * stride + n];
  s_red[grp * 16 + col] = a;
  __syncthreads();
  float t = 0.f;
  if (grp == 0)
    for (int g = 0; g < 16; ++g) t += s_red[g * 16 + col];
  return t;
}
