// gemm_tiles.h — building blocks shared by the dense GEMM kernels (gemm.hip) and the implicit-GEMM convolutions (conv_igemm.hip):
// swizzled LDS tile images, register staging, MFMA fragment reads (plain and transpose-read), the fused epilogue and the split-K second pass.
#pragma once
#include "common.h"

#ifndef ENH_NT_EPILOGUE
#define ENH_NT_EPILOGUE 0  // tried 1 (non-temporal C stores, to keep L2 for the operand slices): 442 -> 407 img/s, because the NEXT kernel
                           // (LayerNorm, attention, the following GEMM) finds C in L2 / Infinity Cache when it is stored normally
#endif
#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_TILE_BYTES 16384  // one operand tile (either layout)

// ---- LDS layouts -------------------------------------------------------------------------------
// "row" layout  (operand stored [rows][K]):   128 rows x 128 B ; 16-B chunk c (0..7) of row r lives at
//     r*128 + ((c ^ ((r>>1)&7)) << 4)
// "kmaj" layout (operand stored [K][cols]):    64 k-rows x 256 B ; 32-B chunk q (0..7) of k-row k lives at
//     k*256 + ((q ^ ((k&3) | (((k>>3)&1)<<2))) << 5)      (8-byte pieces inside a chunk stay in order)
__device__ __forceinline__ int lds_row_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int lds_kmaj_off(int k, int q) { return k * 256 + ((q ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 5); }

// ---- 32x32x16 fragments of the 256 x 256 kernels (gemm.hip w256 family, conv_igemm.hip conv_igemm_w256_kernel) ------------------------------
// second contraction-major image ("kmaj2": chunk q ^ (2*(k&3) | (k>>2)&1)): keeps the 32-column transpose reads of this fragment shape conflict-free
__device__ __forceinline__ int lds_kmaj2_off(int k, int q) { return k * 256 + ((q ^ (((k & 3) << 1) | ((k >> 2) & 1))) << 5); }
// 32x32x16 operand fragment: index = base + (lane & 31), k = s*16 + (lane>>5)*8 + 0..7
template <bool TR>
__device__ __forceinline__ s16x8 frag32(const unsigned char* tile, int base, int s, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  if (!TR) {
    return *reinterpret_cast<const s16x8*>(tile + lds_row_off(base + l31, s * 2 + hi));
  } else {
    const int G = lane >> 4, s16 = lane & 15;
    const int kr = s * 16 + hi * 8 + (s16 >> 2);
    const int q = (base >> 4) + (G & 1);
    typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
    u64x2 o;
    unsigned long long lo, up;
    lds_tr_read_b64_asm(lo, tile + lds_kmaj2_off(kr, q) + (s16 & 3) * 8);
    lds_tr_read_b64_asm(up, tile + lds_kmaj2_off(kr + 4, q) + (s16 & 3) * 8);
    o[0] = lo; o[1] = up;
    return __builtin_bit_cast(s16x8, o);
  }
}


// accumulator -> vector register AT THIS POINT of the instruction stream (left to the register allocator, an epilogue over 256 accumulators copies all of
// them out at its top: 256 live registers and spills)
__device__ __forceinline__ float acc_read(float a) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
  return x;
}

struct GemmArgs {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  int64_t M, N, K;
  int64_t k_per_split;  // multiple of G_BK
  const float* bias; int act; const uint16_t* aux; int64_t ldaux;
  const float* res; int64_t ldres; int64_t res_rows;
  int accumulate;       // 1: += C_old ; 2: split-K partial -> f32 atomicAdd into c_f32 ; 3: split-K partial -> workspace slab (two-pass, deterministic)
  float* ws;            // split-K workspace [splits][M][N] f32 (accumulate == 3)
  float* c_f32; uint16_t* c_bf16; int64_t ldc;
  int nbm, nbn;
  int splits;  // number of K-slices (1 = no split-K; otherwise a multiple of 8)
  float* colpart;  // tanh' mode of the persistent kernel: per-128-row partial column sums of the bf16 result, [M / 128][N] (or null)
  unsigned int* tile_ctr;  // persistent kernels, dynamic schedule: the launch's 8 per-XCD tile counters (zero at launch, reset by the kernel itself)
  // x3 split epilogues of the persistent kernel (EPI_BF16_SPLIT / EPI_BF16_TANH_SPLIT, enh_gemm_bf16_split): the hi plane goes to c_bf16 (ldc) and,
  // where non-null, to c2 / c3 as well; the lo plane bf16(v - hi) to clo
  uint16_t* c2; int64_t ldc2; uint16_t* c3; int64_t ldc3; uint16_t* clo; int64_t ldlo;
  int grp_rows;            // tile order of the non-split forms: row panels per group (8 = the shipped order); col_fast: columns fastest inside a group
  int col_fast;            // (enh_debug_gemm_order: what the 32 workgroups of an XCD have in flight together is a grp_rows x 32/grp_rows patch, or ~32/nbn rows x nbn)
};

// global -> registers: one 128 x 64 (row layout) or 64 x 128 (kmaj layout) bf16 operand tile, 4 x 16 B per thread
template <bool TR>
__device__ __forceinline__ void tile_gload(u32x4 (&r)[4], const uint16_t* __restrict__ P, int64_t ld, int64_t x0,
                                           int64_t X, int64_t k0, int64_t k_end, int t) {
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  if (!TR) {
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = x0 + r0 + 32 * i, kk = k0 + c * 8;
      r[i] = (row < X && kk < k_end) ? *reinterpret_cast<const u32x4*>(P + row * ld + kk) : zero4;
    }
  } else {
    const int c = t & 15, r0 = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t kk = k0 + r0 + 16 * i, col = x0 + c * 8;
      r[i] = (kk < k_end && col < X) ? *reinterpret_cast<const u32x4*>(P + kk * ld + col) : zero4;
    }
  }
}
// registers -> LDS (swizzled image)
template <bool TR>
__device__ __forceinline__ void tile_sstore(const u32x4 (&r)[4], unsigned char* tile, int t) {
  if (!TR) {
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tile + lds_row_off(r0 + 32 * i, c)) = r[i];
  } else {
    const int c = t & 15, r0 = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tile + lds_kmaj_off(r0 + 16 * i, c >> 1) + ((c & 1) << 4)) = r[i];
  }
}
// MFMA 16x16x32 operand fragment: lane (lg, l16) gets tile index base + l16, k = ks*32 + lg*8 + 0..7
template <bool TR>
__device__ __forceinline__ s16x8 tile_frag(const unsigned char* tile, int base, int ks, int lg, int l16) {
  if (!TR) {
    return *reinterpret_cast<const s16x8*>(tile + lds_row_off(base + l16, ks * 4 + lg));
  } else {
    // loader role of this lane inside its 16-lane group: k-row (l16>>2), 4 columns starting at (l16&3)*4
    const int kr = ks * 32 + lg * 8 + (l16 >> 2);
    const int q = base >> 4;  // 32-byte chunk = 16 columns
    const s16x4 lo = lds_tr_read_b64(tile + lds_kmaj_off(kr, q) + (l16 & 3) * 8);
    const s16x4 hi = lds_tr_read_b64(tile + lds_kmaj_off(kr + 4, q) + (l16 & 3) * 8);
    s16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  }
}

// ---- epilogue -----------------------------------------------------------------------------------------------
// One body for every kernel: 4 consecutive output columns of one row.  The fused options (bias / tanh / tanh' / residual / accumulate / f32 and
// bf16 stores / split-K partials) are RUNTIME arguments of the C ABI, but a kernel whose epilogue tests them per element pays for it: with one
// wave per SIMD nothing hides the instruction fetch after each (wave-uniform) branch — the 256 x 256 kernel lost 19 us per tile, more than its
// K loop at K = 768.  So the combinations the training step uses are compile-time MODES selected once per kernel; anything else takes the
// generic (branchy) mode.
enum { EPI_GENERIC = 0, EPI_BF16, EPI_BF16_BIAS_TANH, EPI_BF16_DTANH, EPI_F32_BIAS_RES, EPI_F32, EPI_WS, EPI_ATOMIC,
       EPI_BF16_SPLIT, EPI_BF16_TANH_SPLIT,   // persistent kernel only, chosen by enh_gemm_bf16_split (never by epi_mode): v | tanh(v + bias) -> hi / lo bf16 planes
       EPI_NMODES };

__host__ __device__ __forceinline__ int epi_mode(const GemmArgs& a) {
  if (a.accumulate == 3) return EPI_WS;
  if (a.accumulate == 2) return EPI_ATOMIC;
  if (a.accumulate == 0) {
    const bool only16 = a.c_bf16 && !a.c_f32, only32 = a.c_f32 && !a.c_bf16;
    if (only16 && !a.bias && a.act == ENH_ACT_NONE && !a.res) return EPI_BF16;
    if (only16 && a.bias && a.act == ENH_ACT_TANH && !a.res) return EPI_BF16_BIAS_TANH;
    if (only16 && !a.bias && a.act == ENH_ACT_DTANH && !a.res) return EPI_BF16_DTANH;
    if (only32 && a.bias && a.act == ENH_ACT_NONE && a.res) return EPI_F32_BIAS_RES;
    if (only32 && !a.bias && a.act == ENH_ACT_NONE && !a.res) return EPI_F32;
  }
  return EPI_GENERIC;
}

// The operands an epilogue READS (residual / position row, saved tanh output, previous C) are fetched by epi_load for a whole group of
// elements BEFORE any of them is consumed: a load issued and awaited per element exposes the full memory latency 64 times per wave.
struct EpiIn { float4 res; float4 old; uint2 aux; };

template <int MODE>
__device__ __forceinline__ EpiIn epi_load(const GemmArgs& args, int64_t m, int64_t n) {
  constexpr bool G = MODE == EPI_GENERIC;
  EpiIn in;
  in.res = make_float4(0.f, 0.f, 0.f, 0.f); in.old = in.res; in.aux = make_uint2(0u, 0u);
  if (MODE == EPI_BF16_DTANH || (G && args.act == ENH_ACT_DTANH)) in.aux = *reinterpret_cast<const uint2*>(args.aux + m * args.ldaux + n);
  if (MODE == EPI_F32_BIAS_RES || (G && args.res)) {
    const int64_t mr = args.res_rows == args.M ? m : m % args.res_rows;   // residual stream (res_rows = M) or position table (row mod n_tokens)
    in.res = *reinterpret_cast<const float4*>(args.res + mr * args.ldres + n);
  }
  if (G && args.accumulate == 1 && args.c_f32) in.old = *reinterpret_cast<const float4*>(args.c_f32 + m * args.ldc + n);
  return in;
}

// bias[n..n+3] for the modes that add one.  Loaded by the CALLER, once per column group and before the first store of the epilogue: CDNA4's vmcnt
// counts stores too and retires in order, so a load issued after a store cannot be waited for without also waiting for that store's acknowledgement
// — a bias load inside the per-element body serialised the whole epilogue on its own stores.
template <int MODE>
__device__ __forceinline__ float4 epi_bias(const GemmArgs& args, int64_t n) {
  if (MODE == EPI_BF16_BIAS_TANH || MODE == EPI_F32_BIAS_RES) return *reinterpret_cast<const float4*>(args.bias + n);
  return make_float4(0.f, 0.f, 0.f, 0.f);   // (the generic mode fetches its bias inside epi4, behind its runtime flags)
}

// the arithmetic of an epilogue on one group of 4 columns (bias, activation, residual, accumulate) — everything but the stores
template <int MODE, typename OT = BF16>
__device__ __forceinline__ void epi_value(const GemmArgs& args, float (&v)[4], const EpiIn& in, const float4& b4, int64_t n) {
  constexpr bool G = MODE == EPI_GENERIC;
  if (MODE == EPI_F32_BIAS_RES || MODE == EPI_BF16_TANH_SPLIT) { v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
  if (MODE == EPI_BF16_TANH_SPLIT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = tanh_x3(v[r]);      // common.h: the same function as the stand-alone split3 kernel (same bits in both forms)
  }
  if (G && args.bias) {
    const float4 bg = *reinterpret_cast<const float4*>(args.bias + n);
    v[0] += bg.x; v[1] += bg.y; v[2] += bg.z; v[3] += bg.w;
  }
  if (MODE == EPI_BF16_BIAS_TANH) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1) on the transcendental unit (v_exp_f32 + v_rcp_f32, ~6 instructions): absolute error ~1e-7, invisible after the
    // bf16 rounding of this mode's output.  libm's tanhf (~40 instructions) cost 0.55 ms of the 1.41 ms fc1 forward GEMM (402 M elements per launch).
    // Round 6: the vector half of it in PACKED f32 (v_pk_fma_f32 / v_pk_add_f32, two columns per instruction; the bias folded into the exponent's fma:
    // exp2(c*acc + c*bias), the c*bias pair is loop-invariant per column group) — per element 1.5 full-rate + 2 transcendental issues instead of 4.5 + 2.
    // The one-wave-per-SIMD kernels cannot hide this under MFMAs (no second wave, no spare accumulator set): it is 256 elements per lane per 256 x 256 tile.
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    const float kC = 2.8853900817779268f;
    const f32x2_ c2 = {kC, kC}, one2 = {1.f, 1.f}, mtwo2 = {-2.f, -2.f};
    const f32x2_ bc01 = {b4.x * kC, b4.y * kC}, bc23 = {b4.z * kC, b4.w * kC};
    f32x2_ a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
    a01 = a01 * c2 + bc01; a23 = a23 * c2 + bc23;
    f32x2_ t01 = {__builtin_amdgcn_exp2f(a01[0]), __builtin_amdgcn_exp2f(a01[1])}, t23 = {__builtin_amdgcn_exp2f(a23[0]), __builtin_amdgcn_exp2f(a23[1])};
    t01 = t01 + one2; t23 = t23 + one2;
    f32x2_ r01 = {__builtin_amdgcn_rcpf(t01[0]), __builtin_amdgcn_rcpf(t01[1])}, r23 = {__builtin_amdgcn_rcpf(t23[0]), __builtin_amdgcn_rcpf(t23[1])};
    r01 = r01 * mtwo2 + one2; r23 = r23 * mtwo2 + one2;
    v[0] = r01[0]; v[1] = r01[1]; v[2] = r23[0]; v[3] = r23[1];
  } else if (G && args.act == ENH_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
  } else if (MODE == EPI_BF16_DTANH || (G && args.act == ENH_ACT_DTANH)) {
    const float h0 = unpack_lo<OT>(in.aux.x), h1 = unpack_hi<OT>(in.aux.x);
    const float h2 = unpack_lo<OT>(in.aux.y), h3 = unpack_hi<OT>(in.aux.y);
    v[0] *= 1.f - h0 * h0; v[1] *= 1.f - h1 * h1; v[2] *= 1.f - h2 * h2; v[3] *= 1.f - h3 * h3;
  }
  if (MODE == EPI_F32_BIAS_RES || (G && args.res)) { v[0] += in.res.x; v[1] += in.res.y; v[2] += in.res.z; v[3] += in.res.w; }
  if (G && args.accumulate == 1 && args.c_f32) { v[0] += in.old.x; v[1] += in.old.y; v[2] += in.old.z; v[3] += in.old.w; }
}

template <int MODE, typename OT = BF16>
__device__ __forceinline__ void epi4(const GemmArgs& args, float (&v)[4], const EpiIn& in, const float4& b4, int64_t m, int64_t n, int split) {
  constexpr bool G = MODE == EPI_GENERIC;
  if (MODE == EPI_WS || (G && args.accumulate == 3)) {   // split-K partial -> workspace slab [split][M][N] (reduced by splitk_reduce_kernel in a fixed order)
    const f32x4 o_ = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(args.ws + ((int64_t)split * args.M + m) * args.N + n) = o_;
    return;
  }
  float* cp = (G ? args.c_f32 != nullptr : (MODE == EPI_F32_BIAS_RES || MODE == EPI_F32 || MODE == EPI_ATOMIC)) ? args.c_f32 + m * args.ldc + n : nullptr;
  if (MODE == EPI_ATOMIC || (G && args.accumulate == 2)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
    return;
  }
  epi_value<MODE, OT>(args, v, in, b4, n);
  if (cp) { const f32x4 o_ = {v[0], v[1], v[2], v[3]}; if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<f32x4*>(cp)); else *reinterpret_cast<f32x4*>(cp) = o_; }
  if (MODE == EPI_BF16 || MODE == EPI_BF16_BIAS_TANH || MODE == EPI_BF16_DTANH || (G && args.c_bf16)) {
    const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
    if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n)); else *reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n) = o_;
  }
}

// run LOOPS<MODE>(...) with the mode chosen once (wave-uniform switch)
#define EPI_DISPATCH(CALL)                                                     \
  do {                                                                         \
    switch (epi_mode(args)) {                                                  \
      case EPI_BF16: { constexpr int EM = EPI_BF16; CALL; } break;             \
      case EPI_BF16_BIAS_TANH: { constexpr int EM = EPI_BF16_BIAS_TANH; CALL; } break; \
      case EPI_BF16_DTANH: { constexpr int EM = EPI_BF16_DTANH; CALL; } break; \
      case EPI_F32_BIAS_RES: { constexpr int EM = EPI_F32_BIAS_RES; CALL; } break; \
      case EPI_F32: { constexpr int EM = EPI_F32; CALL; } break;               \
      case EPI_WS: { constexpr int EM = EPI_WS; CALL; } break;                 \
      case EPI_ATOMIC: { constexpr int EM = EPI_ATOMIC; CALL; } break;         \
      default: { constexpr int EM = EPI_GENERIC; CALL; } break;                \
    }                                                                          \
  } while (0)

// 16x16 accumulator layout (pipe2 / fallback): lane (lg, l16) holds C[m = .. + l16][n = .. + lg*4 + 0..3] (MFMA issued with swapped operands)
template <int MODE, typename OT = BF16>
__device__ __forceinline__ void gemm_epilogue_loops(const GemmArgs& args, f32x4 (&acc)[4][4], int64_t m0, int64_t n0, int wm, int wn, int lg, int l16, int split) {
  float4 b4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
    b4[j] = n < args.N ? epi_bias<MODE>(args, n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + l16;
    if (m >= args.M) continue;
    EpiIn in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n < args.N) in[j] = epi_load<MODE>(args, m, n);  // N % 4 == 0: the 4 columns are in or out together
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= args.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epi4<MODE, OT>(args, v, in[j], b4[j], m, n, split);
    }
  }
}
template <typename OT = BF16>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& args, f32x4 (&acc)[4][4], int64_t m0, int64_t n0, int wm, int wn, int lg, int l16, int split) {
  EPI_DISPATCH((gemm_epilogue_loops<EM, OT>(args, acc, m0, n0, wm, wn, lg, l16, split)));
}

// tile scheduling shared by both kernels
__device__ __forceinline__ void gemm_tile_coords_of(const GemmArgs& args, int block, int& split, int& tile_m, int& tile_n) {
  // (1) XCD-aware: workgroup b runs on XCD b % 8 -> give each XCD a contiguous run of tile slots;
  // (2) grouped order inside the run: 8 row-panels x all column tiles, row-fastest, so the ~64 tiles an XCD has in
  //     flight form an ~8 x 8 patch whose A and B panels (8 x 196 KB each at K = 768) both stay in its 4 MiB L2.
  const int nwg = args.nbm * args.nbn;  // tiles per K-split
  if (args.splits > 1) {
    // split-K (weight gradients): the (K slice, row, column) space is ONE line — the shorter of the two tile dimensions fastest — and each XCD takes a
    // contiguous run of it (W/8 +- 1 workgroups: one round).  The tiles of a K slice that share an operand slice (the column tiles of a row share
    // A's, the rows share B's) then run on the SAME XCD, i.e. behind one L2: with the slices spread over all XCDs (round 2) every slice was fetched
    // by ~3 XCDs and the launch moved 2.7 GB through the fabric for 0.9 GB of operands (PMC) — at 6.3 TB/s that WAS its run time.  Same-box A/B
    // (tools/wgrad_lab.py, B = 128): qkv 0.436 -> 0.417 ms, out 0.179 -> 0.160, fc1 0.578 -> 0.539, fc2 0.561 -> 0.537; bit-identical sums.
    // (Round 2's "pin each K slice to XCD (slice mod 8)" was slower because 9 or 28 slices do not divide by 8: two rounds on some XCDs.)
    const int total = nwg * args.splits;
    const int q = total >> 3, r = total & 7, xcd = block & 7, pos = block >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    split = lin / nwg;
    const int tl = lin - split * nwg;
    if (args.nbm < args.nbn) { tile_n = tl / args.nbm; tile_m = tl - tile_n * args.nbm; }   // rows fastest
    else { tile_m = tl / args.nbn; tile_n = tl - tile_m * args.nbn; }                       // columns fastest
    return;
  }
  split = block / nwg;
  int bid = block - split * nwg;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int gr = args.grp_rows;
  const int per_group = gr * args.nbn;
  const int grp = bid / per_group, within = bid - grp * per_group;
  const int rows = (args.nbm - grp * gr) < gr ? (args.nbm - grp * gr) : gr;
  if (args.col_fast) { tile_m = grp * gr + within / args.nbn; tile_n = within - (within / args.nbn) * args.nbn; }
  else { tile_m = grp * gr + within % rows; tile_n = within / rows; }
}
__device__ __forceinline__ void gemm_tile_coords(const GemmArgs& args, int& split, int& tile_m, int& tile_n) {
  gemm_tile_coords_of(args, (int)blockIdx.x, split, tile_m, tile_n);
}


#ifndef W2_LDS_STORE
#define W2_LDS_STORE 1   // lab switch: bf16 outputs through an LDS transpose (whole-row 16-byte stores) vs 8-byte stores straight from the accumulator layout
#endif

// epilogue for the swapped 32x32 accumulator layout: acc[i][j][r] = C[mw + i*32 + (lane&31)][nw + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
template <int MODE, int NJ, bool BOUNDS, typename OT = BF16>
__device__ __forceinline__ void gemm_epilogue32_loops(const GemmArgs& args, f32x16 (&acc)[4][NJ], int64_t mw, int64_t nw, int lane, int split, float* wave_bias,
                                                      unsigned char* stage, unsigned char* stage32) {
  const int l31 = lane & 31, hi = lane >> 5;
  // Row-blocks whose inputs (saved tanh output / residual / old C) are requested TOGETHER before the first is consumed.  With one wave per SIMD every
  // group costs one full memory round trip (~2-4 us under load) during which nothing else runs, so the group is made as large as the registers allow:
  // the accumulators live in AGPRs and the K loop's fragment registers are dead here.  8 bytes per element group: all four row-blocks (128 VGPRs);
  // 16 bytes (f32 residual): two.
  constexpr int GI = MODE == EPI_BF16_DTANH ? 4 : (MODE == EPI_F32_BIAS_RES ? 2 : 1);
  if (MODE == EPI_GENERIC) {   // runtime-flag mode: one row-block at a time, the bias fetched where it is used (the round-1 code shape: no spills)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = mw + i * 32 + l31;
      if (BOUNDS && m >= args.M) continue;
      EpiIn in[NJ][4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
          if (!BOUNDS || n < args.N) in[j][g4] = epi_load<MODE>(args, m, n);
        }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
          if (BOUNDS && n >= args.N) continue;
          float v[4] = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
          epi4<MODE, OT>(args, v, in[j][g4], epi_bias<MODE>(args, n), m, n, split);
        }
      }
    }
    return;
  }
  // The bias of the wave's NJ*32 columns goes through a wave-private LDS strip: fetched once and BEFORE any store (see epi_bias), and read back with
  // ds_read_b128 — LDS traffic is counted by lgkmcnt, so unlike a global load it can be waited for without waiting for the stores in flight, and
  // it costs no registers across the row-blocks.
  constexpr bool HAS_BIAS = MODE == EPI_BF16_BIAS_TANH || MODE == EPI_F32_BIAS_RES;
  if (HAS_BIAS && lane < NJ * 8) {
    const int64_t n = nw + lane * 4;
    const float4 bv = (!BOUNDS || n < args.N) ? *reinterpret_cast<const float4*>(args.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(wave_bias + lane * 4) = bv;
  }
#if W2_LDS_STORE
  // bf16 outputs leave through a wave-private 8-KiB LDS tile: per lane the accumulator layout only offers 8-byte pieces 256 B apart (a wave store
  // touches 32 rows x 16 B; a pure fill kernel writes HBM at 7.4 TB/s, this pattern at 3.1-3.9), so each 32 x 128 block is transposed through LDS
  // (XOR-swizzled 16-byte chunks: the 8-byte writes and the 16-byte reads are both conflict-free) and leaves as 16 bytes per lane, whole 256-byte row
  // segments per instruction.  `stage` overlays the K loop's slots: one barrier first.
  constexpr bool STAGED = (MODE == EPI_BF16 || MODE == EPI_BF16_BIAS_TANH || MODE == EPI_BF16_DTANH) && NJ == 4 && !BOUNDS;
  if (STAGED) {
    EpiIn in[GI][NJ][4];
    if (MODE == EPI_BF16_DTANH) {
#pragma unroll
      for (int ii = 0; ii < GI; ++ii)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) in[ii][j][g4] = epi_load<MODE>(args, mw + ii * 32 + l31, nw + j * 32 + 8 * g4 + 4 * hi);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's last fragment reads are done
    __builtin_amdgcn_s_barrier();         // ... and so are everybody else's: the slots may be overwritten
    const int rrow = lane >> 4, rc16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float v[4] = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
          const float4 b4 = HAS_BIAS ? *reinterpret_cast<const float4*>(wave_bias + j * 32 + 8 * g4 + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
          epi_value<MODE, OT>(args, v, in[MODE == EPI_BF16_DTANH ? i : 0][j][g4], b4, nw + j * 32 + 8 * g4 + 4 * hi);
          const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
          *reinterpret_cast<u32x2*>(stage + l31 * 256 + (((j * 4 + g4) ^ (l31 & 15)) << 4) + hi * 8) = o_;
        }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int row = p * 4 + rrow;
        const u32x4 w = *reinterpret_cast<const u32x4*>(stage + row * 256 + ((rc16 ^ (row & 15)) << 4));
        *reinterpret_cast<u32x4*>(args.c_bf16 + (mw + i * 32 + row) * args.ldc + nw + rc16 * 8) = w;
      }
    }
    return;
  }
  // f32 outputs (C, or a split-K partial slab): the same transpose with a 16-KiB tile per wave — the accumulator layout gives 16 bytes per lane 512 B
  // apart (32 rows x 32 B per instruction); through LDS a store instruction covers two whole 512-byte row segments
  constexpr bool STAGED32 = (MODE == EPI_F32_BIAS_RES || MODE == EPI_F32 || MODE == EPI_WS) && NJ == 4 && !BOUNDS;
  if (STAGED32) {
    unsigned char* st32 = stage32;   // 16 KiB per wave
    float* dst = MODE == EPI_WS ? args.ws + ((int64_t)split * args.M + mw) * args.N + nw : args.c_f32 + mw * args.ldc + nw;
    const int64_t ldd = MODE == EPI_WS ? args.N : args.ldc;
    const int rrow = lane >> 5, rc = lane & 31;
    bool synced = false;
    constexpr int GS = 1;   // row-blocks of residual requested together in this form (two = 128 registers on top of the read-back temporaries: spills)
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += GS) {
      EpiIn in[GS][NJ][4];
      if (MODE == EPI_F32_BIAS_RES) {
#pragma unroll
        for (int ii = 0; ii < GS; ++ii)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) in[ii][j][g4] = epi_load<MODE>(args, mw + (i0 + ii) * 32 + l31, nw + j * 32 + 8 * g4 + 4 * hi);
      }
      if (!synced) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        synced = true;
      }
#pragma unroll
      for (int ii = 0; ii < GS; ++ii) {
        const int i = i0 + ii;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float v[4] = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
            const float4 b4 = HAS_BIAS ? *reinterpret_cast<const float4*>(wave_bias + j * 32 + 8 * g4 + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE != EPI_WS) epi_value<MODE, OT>(args, v, in[ii][j][g4], b4, nw + j * 32 + 8 * g4 + 4 * hi);
            const f32x4 o_ = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(st32 + l31 * 512 + (((j * 8 + g4 * 2 + hi) ^ l31) << 4)) = o_;
          }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const int row = p * 2 + rrow;
          const f32x4 w = *reinterpret_cast<const f32x4*>(st32 + row * 512 + ((rc ^ row) << 4));
          *reinterpret_cast<f32x4*>(dst + (int64_t)(i * 32 + row) * ldd + rc * 4) = w;
          if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four read / store pairs in flight at a time: hoisting all 16 reads costs 64 registers
        }
      }
    }
    return;
  }
#endif
#pragma unroll
  for (int i0 = 0; i0 < 4; i0 += GI) {
    EpiIn in[GI][NJ][4];
#pragma unroll
    for (int ii = 0; ii < GI; ++ii) {
      const int64_t m = mw + (i0 + ii) * 32 + l31;
      if (BOUNDS && m >= args.M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
          if (!BOUNDS || n < args.N) in[ii][j][g4] = epi_load<MODE>(args, m, n);
        }
    }
#pragma unroll
    for (int ii = 0; ii < GI; ++ii) {
      const int i = i0 + ii;
      const int64_t m = mw + i * 32 + l31;
      if (BOUNDS && m >= args.M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
          if (BOUNDS && n >= args.N) continue;
          float v[4] = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
          const float4 b4 = HAS_BIAS ? *reinterpret_cast<const float4*>(wave_bias + j * 32 + 8 * g4 + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
          epi4<MODE, OT>(args, v, in[ii][j][g4], b4, m, n, split);
        }
      }
    }
  }
}

// the same for a 16-bit output (the token-gradient GEMMs write their result in the operand format): sum in f32, one RNE pack
template <typename OT>
static __global__ __launch_bounds__(256) void splitk_reduce16_kernel(const float* __restrict__ ws, int splits, int64_t MN, int64_t N, uint16_t* __restrict__ c, int64_t ldc) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= MN) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + i4);
  for (int k = 1; k < splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (int64_t)k * MN + i4);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  const int64_t m = i4 / N, n = i4 - m * N;
  const u32x2 o_ = {pack2<OT>(s[0], s[1]), pack2<OT>(s[2], s[3])};
  *reinterpret_cast<u32x2*>(c + m * ldc + n) = o_;
}

// split-K second pass: C[m][n] (+)= sum over the splits of the partial slabs, in a fixed order (deterministic, no atomics).  The forward kind of split
// (gemm.hip gemm_splittable == 2: few tiles, long K, f32 output) carries its epilogue here: + bias[n] + res[m mod res_rows][n] (either may be null).
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t MN, int64_t N, float* __restrict__ c, int64_t ldc, int accumulate,
                                                                   const float* __restrict__ bias = nullptr, const float* __restrict__ res = nullptr, int64_t ldres = 0,
                                                                   int64_t res_rows = 0) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= MN) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + i4);
  for (int k = 1; k < splits; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (int64_t)k * MN + i4);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  const int64_t m = i4 / N, n = i4 - m * N;
  if (bias) {       // the order of gemm_tiles.h epi_value: bias, residual, previous C
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
    s[0] += b[0]; s[1] += b[1]; s[2] += b[2]; s[3] += b[3];
  }
  if (res) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + (res_rows == MN / N ? m : m % res_rows) * ldres + n);
    s[0] += r[0]; s[1] += r[1]; s[2] += r[2]; s[3] += r[3];
  }
  float* cp = c + m * ldc + n;
  if (accumulate) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(cp);
    s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
  }
  *reinterpret_cast<f32x4*>(cp) = s;
}
