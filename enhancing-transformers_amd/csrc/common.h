// common.h — shared device/host helpers for libenh_hip.so (gfx950 only; no compatibility layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/enh_hip.h"

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

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
