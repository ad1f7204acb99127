// gemm.hip — bf16 MFMA GEMM with fused epilogues for gfx950 (CDNA4), wave64.
//
// One kernel family serves every dense contraction of the ViT towers (reference
// enhancing/modules/stage1/layers.py:99-101,118,120,169,204 and vitvqgan.py:38-39) in all three roles:
//   forward  y  = x  W^T        : A [M][K] row-major,            B = W  [N][K]            (trans_a=0, trans_b=0)
//   dgrad    dx = dy W          : A = dy [M][N_out] row-major,   B = W  stored [K=N_out][N=K_in] (trans_b=1)
//   wgrad    dW = dy^T x        : A = dy stored [K=tokens][M=N_out] (trans_a=1), B = x stored [K=tokens][N] (trans_b=1)
// so no transposed copies of activations or weights are ever written to HBM: operands whose contraction
// index is the slow storage index are staged as-is and read from LDS with the hardware transpose read
// ds_read_b64_tr_b16 (semantics verified on MI355X, profiles/hw_probe_r01.txt).
//
// Kernel families (enh_gemm_bf16_variant() reports the per-shape choice; ENH_GEMM_KERNEL = reg | pipe2 | t256 overrides):
//   gemm_bf16_pipe2_kernel  128x128x64 tile, 4 waves (2x2, each 4x4 v_mfma_f32_16x16x32_bf16), two 32-KiB LDS stages filled by
//                           global_load_lds, K-loop software-pipelined around one mid-iteration barrier; 2 workgroups per CU.  Default.
//   gemm_bf16_t256_kernel   256x256x64 tile, 8 waves (2x4, each 4x2 v_mfma_f32_32x32x16_bf16), two 64-KiB stages, staggered two-group
//                           schedule; chosen for long-K forward / dgrad shapes (half the staged bytes per flop).
//   gemm_bf16_kernel        register-staged 128x128x64 fallback for K not a multiple of 64 (zero-fills partial tiles).
// All LDS images are XOR-swizzled so that staging writes, ds_read_b128 fragments and the transpose reads are bank-conflict free under
// the gfx950 bank model (MI355X_MICROARCH.md §LDS; checked by tools/lds_bank_check.py; SQ_LDS_BANK_CONFLICT = 0 measured).
// The MFMA is issued with swapped operands (D = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE output columns of one row: the
// epilogue reads bias / residual / aux and writes C with 16-byte (f32) or 8-byte (bf16) accesses.  Workgroup ids are remapped so each XCD
// (private L2) walks a contiguous, grouped run of tiles.  What bounds these kernels (L2 misses, not structure): DESIGN.md §3.1.
#include <stdlib.h>
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

struct GemmArgs {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  int64_t M, N, K;
  int64_t k_per_split;  // multiple of G_BK
  const float* bias; int act; const uint16_t* aux; int64_t ldaux;
  const float* res; int64_t ldres; int64_t res_rows;
  int accumulate;       // 1: += C_old ; 2: split-K partial -> f32 atomicAdd into c_f32
  float* c_f32; uint16_t* c_bf16; int64_t ldc;
  int nbm, nbn;
  int splits;  // number of K-slices (1 = no split-K; otherwise a multiple of 8)
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

// ---- epilogue: lane (lg, l16) holds C[m = .. + l16][n = .. + lg*4 + 0..3] (MFMA issued with swapped operands) ----
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& args, f32x4 (&acc)[4][4], int64_t m0, int64_t n0, int wm, int wn,
                                              int lg, int l16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + l16;
    if (m >= args.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= args.N) continue;  // N % 4 == 0: the 4 columns are in or out together
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      float* cp = args.c_f32 ? args.c_f32 + m * args.ldc + n : nullptr;
      if (args.accumulate == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
        continue;
      }
      if (args.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(args.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (args.act == ENH_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      } else if (args.act == ENH_ACT_DTANH) {
        const uint2 a2 = *reinterpret_cast<const uint2*>(args.aux + m * args.ldaux + n);
        const float h0 = bf16_bits_to_f32((uint16_t)(a2.x & 0xffffu)), h1 = bf16_bits_to_f32((uint16_t)(a2.x >> 16));
        const float h2 = bf16_bits_to_f32((uint16_t)(a2.y & 0xffffu)), h3 = bf16_bits_to_f32((uint16_t)(a2.y >> 16));
        v[0] *= 1.f - h0 * h0; v[1] *= 1.f - h1 * h1; v[2] *= 1.f - h2 * h2; v[3] *= 1.f - h3 * h3;
      }
      if (args.res) {
        const float4 r4 = *reinterpret_cast<const float4*>(args.res + (m % args.res_rows) * args.ldres + n);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
      if (args.accumulate == 1 && cp) {
        const float4 o4 = *reinterpret_cast<const float4*>(cp);
        v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
      }
      if (cp) { const f32x4 o_ = {v[0], v[1], v[2], v[3]}; if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<f32x4*>(cp)); else *reinterpret_cast<f32x4*>(cp) = o_; }
      if (args.c_bf16) { const u32x2 o_ = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n)); else *reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n) = o_; }
    }
  }
}

// tile scheduling shared by both kernels
__device__ __forceinline__ void gemm_tile_coords(const GemmArgs& args, int& split, int& tile_m, int& tile_n) {
  // (1) XCD-aware: workgroup b runs on XCD b % 8 -> give each XCD a contiguous run of tile slots;
  // (2) grouped order inside the run: 8 row-panels x all column tiles, row-fastest, so the ~64 tiles an XCD has in
  //     flight form an ~8 x 8 patch whose A and B panels (8 x 196 KB each at K = 768) both stay in its 4 MiB L2.
  const int nwg = args.nbm * args.nbn;  // tiles per K-split
  // (tried: pinning each K-slice of a split-K launch to one XCD halves the fabric traffic PMC reports, but runs 10-15 % slower —
  //  the duplicated fetches were Infinity-Cache hits, and spreading a slice over all XCDs gives more channel parallelism)
  split = blockIdx.x / nwg;
  int bid = blockIdx.x - split * nwg;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int per_group = 8 * args.nbn;
  const int grp = bid / per_group, within = bid - grp * per_group;
  const int rows = (args.nbm - grp * 8) < 8 ? (args.nbm - grp * 8) : 8;
  tile_m = grp * 8 + within % rows;
  tile_n = within / rows;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile | B tile]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;

  int split, tile_m, tile_n;
  gemm_tile_coords(args, split, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nk = (int)((k_end - k_begin + G_BK - 1) / G_BK);

  u32x4 ra[4], rb[4];

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    tile_gload<TA>(ra, args.A, args.lda, m0, args.M, k_begin, k_end, t);
    tile_gload<TB>(rb, args.B, args.ldb, n0, args.N, k_begin, k_end, t);
    tile_sstore<TA>(ra, smem, t);
    tile_sstore<TB>(rb, smem + G_TILE_BYTES, t);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) {
      const int64_t k0 = k_begin + (int64_t)(kt + 1) * G_BK;
      tile_gload<TA>(ra, args.A, args.lda, m0, args.M, k0, k_end, t);
      tile_gload<TB>(rb, args.B, args.ldb, n0, args.N, k0, k_end, t);
    }
    const unsigned char* sa = smem + stage * (2 * G_TILE_BYTES);
    const unsigned char* sb = sa + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = tile_frag<TA>(sa, wm * 64 + i * 16, ks, lg, l16);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = tile_frag<TB>(sb, wn * 64 + j * 16, ks, lg, l16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      unsigned char* na = smem + (stage ^ 1) * (2 * G_TILE_BYTES);
      tile_sstore<TA>(ra, na, t);
      tile_sstore<TB>(rb, na + G_TILE_BYTES, t);
    }
    __syncthreads();
  }

  gemm_epilogue(args, acc, m0, n0, wm, wn, lg, l16);
}


// =================================================================================================
// direct-to-LDS staging: global_load_lds (16 B / lane) writes the swizzled LDS image itself — the LDS destination of a wave
// instruction is lane-linear (base + lane*16 B, verified in profiles/hw_probe_r01.txt), so the XOR swizzle is applied to each lane's
// SOURCE address instead.  No staging VGPRs, no ds_write pass.  Requires every K-slice to be a multiple of 64 (no zero-fill is
// possible); out-of-range rows / columns are clamped to the last valid one — their products land in outputs that are never stored.
// (Earlier variants — a 2-buffer kernel with __syncthreads() drains and a 3-stage 256x128 kernel — lost the A/B comparisons recorded
// in profiles/r01_gemm_ablation.txt and were removed.)
// =================================================================================================
template <bool TR>
__device__ __forceinline__ const uint16_t* glds_src_ptr(const uint16_t* __restrict__ P, int64_t ld, int64_t x0, int64_t X,
                                                        int64_t k_begin, int slab, int lane) {
  if (!TR) {
    const int r = slab * 8 + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((r >> 1) & 7);
    int64_t row = x0 + r;
    if (row > X - 1) row = X - 1;
    return P + row * ld + k_begin + c * 8;
  } else {
    const int k = slab * 4 + (lane >> 4), pp = lane & 15;
    const int q = (pp >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2));
    int64_t col = x0 + q * 16 + (pp & 1) * 8;
    if (col > X - 8) col = X - 8;
    return P + (k_begin + k) * ld + col;
  }
}

// =================================================================================================
// "pipe2": 128 x 128 x 64 tile, 4 waves, two LDS stages, direct-to-LDS loads — with the K-loop software-pipelined
// around ONE mid-iteration barrier:
//     read F1 = fragments (kt, k 32..63)            | LDS latency of F1 hides under ...
//     16 MFMAs on F0 = fragments (kt, k 0..31)      | ... these MFMAs
//     lgkmcnt(0) ; vmcnt(0) ; s_barrier             <- every wave now holds ALL of stage kt in registers, and its
//                                                      share of stage kt+1 (issued one full iteration ago) has landed
//     global_load_lds stage kt+2 -> the buffer of stage kt   (free: nobody reads it any more)
//     read F0 = fragments (kt+1, k 0..31)           | latency hides under ...
//     16 MFMAs on F1                                | ... these MFMAs
// so loads get a whole iteration to arrive with only two 32-KiB buffers (two workgroups per CU), and no ds_read
// latency is exposed in steady state.  Raw s_barrier + explicit waits: __syncthreads() would drain differently.
// =================================================================================================
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_pipe2_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile | B tile]
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;
  int split, tile_m, tile_n;
  gemm_tile_coords(args, split, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nk = (int)((k_end - k_begin) / G_BK);

  const uint16_t* src[8];
  int64_t step[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    src[i] = glds_src_ptr<TA>(args.A, args.lda, m0, args.M, k_begin, wave * 4 + i, lane);
    src[4 + i] = glds_src_ptr<TB>(args.B, args.ldb, n0, args.N, k_begin, wave * 4 + i, lane);
    step[i] = TA ? (int64_t)G_BK * args.lda : (int64_t)G_BK;
    step[4 + i] = TB ? (int64_t)G_BK * args.ldb : (int64_t)G_BK;
  }
#define P2_ISSUE(BUF)                                                                                                    \
  do {                                                                                                                   \
    unsigned char* base_ = smem + (BUF) * (2 * G_TILE_BYTES);                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                   \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[i_],                                                      \
                                       (LDS_AS void*)(base_ + (i_ >> 2) * G_TILE_BYTES + (wave * 4 + (i_ & 3)) * 1024), 16, 0, 0); \
      src[i_] += step[i_];                                                                                               \
    }                                                                                                                    \
  } while (0)
#define P2_READ(FA, FB, BUF, KS)                                                                                         \
  do {                                                                                                                   \
    const unsigned char* sa_ = smem + (BUF) * (2 * G_TILE_BYTES);                                                        \
    const unsigned char* sb_ = sa_ + G_TILE_BYTES;                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) FA[i_] = tile_frag<TA>(sa_, wm * 64 + i_ * 16, KS, lg, l16);        \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) FB[j_] = tile_frag<TB>(sb_, wn * 64 + j_ * 16, KS, lg, l16);        \
  } while (0)
#define P2_MMA(FA, FB)                                                                                                   \
  do {                                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                     \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                   \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, FB[j_]), __builtin_bit_cast(bf16x8, FA[i_]), acc[i_][j_], 0, 0, 0); \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];

  if (nk > 0) {
    P2_ISSUE(0);
    if (nk > 1) {
      P2_ISSUE(1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage 0 landed (stage 1's 8 loads may be outstanding)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    P2_READ(fa0, fb0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): same state on both edges into the loop header
  }
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    P2_READ(fa1, fb1, buf, 1);
    __builtin_amdgcn_sched_barrier(0);
    P2_MMA(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): F1 in registers, my share of stage kt+1 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) P2_READ(fa0, fb0, buf ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    // second half: the 8 global_load_lds of stage kt+2 are spread one per two MFMAs instead of issued as a burst — a
    // burst is back-pressured by the texture addresser (~64 B/clk/CU) and the in-order wave cannot issue MFMAs meanwhile
    const bool more = kt + 2 < nk;
    unsigned char* nbase = smem + buf * (2 * G_TILE_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int ld = i * 2 + jj;  // load slot 0..7
        acc[i][jj * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb1[jj * 2]), __builtin_bit_cast(bf16x8, fa1[i]), acc[i][jj * 2], 0, 0, 0);
        acc[i][jj * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb1[jj * 2 + 1]), __builtin_bit_cast(bf16x8, fa1[i]), acc[i][jj * 2 + 1], 0, 0, 0);
        if (more) {
          __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[ld], (LDS_AS void*)(nbase + (ld >> 2) * G_TILE_BYTES + (wave * 4 + (ld & 3)) * 1024), 16, 0, 0);
          src[ld] += step[ld];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: next F0 has arrived under the MFMAs above (builtin, so the
                                         // compiler's wait-count pass sees it and adds no conservative wait at the loop top)
    buf ^= 1;
  }
#undef P2_ISSUE
#undef P2_READ
#undef P2_MMA
  gemm_epilogue(args, acc, m0, n0, wm, wn, lg, l16);
}


// =================================================================================================
// "t256": 256 x 256 x 64 workgroup tile, 8 waves (2 x 4), each 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16.
// Motivation (measured, profiles/r01_gemm_ablation.txt): the 128 x 128 tile is bound by global->LDS ingest
// (~30 B/clk/CU sustained; 64 FLOP per staged byte needs 64 B/clk/CU at MFMA peak).  256 x 256 doubles the
// intensity to 128 FLOP/B.  A stage = four 16-KiB sub-tiles [A rows 0-127 | A rows 128-255 | B 0-127 | B 128-255];
// two stages = 128 KiB LDS, one workgroup per CU.  Schedule = pipe2 with the K-step split into four k16 sub-steps and
// two alternating fragment sets (6 fragments each): the reads of sub-step s+1 are issued before the 8 MFMAs of
// sub-step s; after sub-step 2 every wave holds the rest of stage kt in registers -> lgkmcnt(0) + vmcnt(0) + barrier,
// global_load_lds of stage kt+2 into the freed buffer, first fragments of stage kt+1, then sub-step 3's MFMAs.
// The contraction-major ("kmaj") sub-tiles use a second swizzle (q ^ (2*(k&3) | (k>>2)&1)) that keeps the 32-column
// transpose reads of this fragment shape conflict-free.
// =================================================================================================
#define G4_BM 256
#define G4_BN 256
#define G4_STAGE_BYTES (4 * G_TILE_BYTES)
__device__ __forceinline__ int lds_kmaj2_off(int k, int q) { return k * 256 + ((q ^ (((k & 3) << 1) | ((k >> 2) & 1))) << 5); }

template <bool TR>
__device__ __forceinline__ const uint16_t* glds_src_ptr2(const uint16_t* __restrict__ P, int64_t ld, int64_t x0, int64_t X,
                                                         int64_t k_begin, int slab, int lane) {
  if (!TR) return glds_src_ptr<false>(P, ld, x0, X, k_begin, slab, lane);
  const int k = slab * 4 + (lane >> 4), pp = lane & 15;
  const int q = (pp >> 1) ^ (((k & 3) << 1) | ((k >> 2) & 1));
  int64_t col = x0 + q * 16 + (pp & 1) * 8;
  if (col > X - 8) col = X - 8;
  return P + (k_begin + k) * ld + col;
}
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

// epilogue for the swapped 32x32 accumulator layout: acc[i][j][r] = C[m0 + i*32 + (lane&31)][n0 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
__device__ __forceinline__ void gemm_epilogue32(const GemmArgs& args, f32x16 (&acc)[4][2], int64_t mw, int64_t nw, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = mw + i * 32 + l31;
    if (m >= args.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
        if (n >= args.N) continue;
        float v[4] = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
        float* cp = args.c_f32 ? args.c_f32 + m * args.ldc + n : nullptr;
        if (args.accumulate == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
          continue;
        }
        if (args.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(args.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (args.act == ENH_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        } else if (args.act == ENH_ACT_DTANH) {
          const uint2 a2 = *reinterpret_cast<const uint2*>(args.aux + m * args.ldaux + n);
          const float h0 = bf16_bits_to_f32((uint16_t)(a2.x & 0xffffu)), h1 = bf16_bits_to_f32((uint16_t)(a2.x >> 16));
          const float h2 = bf16_bits_to_f32((uint16_t)(a2.y & 0xffffu)), h3 = bf16_bits_to_f32((uint16_t)(a2.y >> 16));
          v[0] *= 1.f - h0 * h0; v[1] *= 1.f - h1 * h1; v[2] *= 1.f - h2 * h2; v[3] *= 1.f - h3 * h3;
        }
        if (args.res) {
          const float4 r4 = *reinterpret_cast<const float4*>(args.res + (m % args.res_rows) * args.ldres + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (args.accumulate == 1 && cp) {
          const float4 o4 = *reinterpret_cast<const float4*>(cp);
          v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
        }
        if (cp) { const f32x4 o_ = {v[0], v[1], v[2], v[3]}; if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<f32x4*>(cp)); else *reinterpret_cast<f32x4*>(cp) = o_; }
        if (args.c_bf16) { const u32x2 o_ = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; if (ENH_NT_EPILOGUE) __builtin_nontemporal_store(o_, reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n)); else *reinterpret_cast<u32x2*>(args.c_bf16 + m * args.ldc + n) = o_; }
      }
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A0 | A1 | B0 | B1]
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, each 128 (M) x 64 (N)
  int split, tile_m, tile_n;
  gemm_tile_coords(args, split, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * G4_BM, n0 = (int64_t)tile_n * G4_BN;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nk = (int)((k_end - k_begin) / G_BK);

  const uint16_t* src[8];
  int lds_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = wave * 8 + i, sub = g >> 4, slab = g & 15;
    lds_off[i] = sub * G_TILE_BYTES + slab * 1024;
    src[i] = sub < 2 ? glds_src_ptr2<TA>(args.A, args.lda, m0 + sub * 128, args.M, k_begin, slab, lane)
                     : glds_src_ptr2<TB>(args.B, args.ldb, n0 + (sub - 2) * 128, args.N, k_begin, slab, lane);
  }
  const int64_t step_a = TA ? (int64_t)G_BK * args.lda : (int64_t)G_BK;
  const int64_t step_b = TB ? (int64_t)G_BK * args.ldb : (int64_t)G_BK;
  // waves 0-3 stage A slabs (sub-tiles 0,1), waves 4-7 stage B slabs (sub-tiles 2,3): g>>4 = wave>>1 ... per-wave uniform
  const int64_t my_step = wave < 4 ? step_a : step_b;
#define T4_ISSUE(BUF)                                                                                                    \
  do {                                                                                                                   \
    unsigned char* base_ = smem + (BUF) * G4_STAGE_BYTES;                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                   \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[i_], (LDS_AS void*)(base_ + lds_off[i_]), 16, 0, 0);      \
      src[i_] += my_step;                                                                                                \
    }                                                                                                                    \
  } while (0)
#define T4_READ(FA, FB, BUF, S)                                                                                          \
  do {                                                                                                                   \
    const unsigned char* sa_ = smem + (BUF) * G4_STAGE_BYTES + wm * G_TILE_BYTES;                                        \
    const unsigned char* sb_ = smem + (BUF) * G4_STAGE_BYTES + (2 + (wn >> 1)) * G_TILE_BYTES;                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) FA[i_] = frag32<TA>(sa_, i_ * 32, S, lane);                         \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) FB[j_] = frag32<TB>(sb_, (wn & 1) * 64 + j_ * 32, S, lane);         \
  } while (0)
#define T4_MMA(FA, FB)                                                                                                   \
  do {                                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                     \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                   \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FB[j_]), __builtin_bit_cast(bf16x8, FA[i_]), acc[i_][j_], 0, 0, 0); \
  } while (0)
#define T4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define T4_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F) /* lgkmcnt(0) */

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fae[4], fbe[2], fao[4], fbo[2];

  // ---- staggered two-group schedule ----------------------------------------------------------------------------
  // Each k16 sub-step is split into a READ segment (fragment ds_reads of the NEXT sub-step + a share of the
  // global_load_lds traffic) and an MMA segment (8 MFMAs), every segment closed by s_barrier.  Waves 4-7 (the second
  // wave of every SIMD) run ONE barrier behind waves 0-3, so on each SIMD one wave is always inside an MMA segment
  // while its partner issues LDS / VMEM work: the matrix pipe is fed continuously and read / load issue is hidden.
  //   stage kt lives in buffer kt & 1; its sub-step-0 fragments are read in R3 of step kt-1, sub-steps 1..3 in R0..R2;
  //   stage kt+2 is loaded into the same buffer: first half issued in R3(kt) — by then both groups have completed
  //   their last reads of stage kt (lgkmcnt(0) before the barrier closing R2) — second half in R0(kt+1); every wave
  //   waits vmcnt(0) at the end of R2(kt+1), i.e. before the barriers that precede any R3(kt+1) read of that stage.
  const bool late = wave >= 4;  // wave-uniform (readfirstlane above)
#define T4_BAR() __builtin_amdgcn_s_barrier()
#define T4_HALF(BUF, H)                                                                                                  \
  do {                                                                                                                   \
    unsigned char* base_ = smem + (BUF) * G4_STAGE_BYTES;                                                                \
    _Pragma("unroll") for (int i_ = (H) * 4; i_ < (H) * 4 + 4; ++i_) {                                                   \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[i_], (LDS_AS void*)(base_ + lds_off[i_]), 16, 0, 0);      \
      src[i_] += my_step;                                                                                                \
    }                                                                                                                    \
  } while (0)
#define T4_MMA_SEG(FA, FB)                                                                                               \
  do {                                                                                                                   \
    T4_WAIT_LDS();                                                                                                       \
    T4_FENCE();                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    T4_MMA(FA, FB);                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
    T4_FENCE();                                                                                                          \
    T4_BAR();                                                                                                            \
  } while (0)

  if (nk > 0) {
    T4_ISSUE(0);
    if (nk > 1) T4_ISSUE(1);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    T4_BAR();
    T4_READ(fae, fbe, 0, 0);
    if (late) T4_BAR();  // stagger: the second wave of every SIMD runs one barrier behind
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // ---- sub-step 0 ----
      T4_READ(fao, fbo, buf, 1);
      if (kt >= 1 && kt + 1 < nk) T4_HALF(buf ^ 1, 1);   // second half of stage kt+1 (first half went out in R3(kt-1))
      T4_FENCE(); T4_BAR();
      T4_MMA_SEG(fae, fbe);
      // ---- sub-step 1 ----
      T4_READ(fae, fbe, buf, 2);
      T4_FENCE(); T4_BAR();
      T4_MMA_SEG(fao, fbo);
      // ---- sub-step 2 ----
      T4_READ(fao, fbo, buf, 3);
      T4_FENCE();
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): my share of stage kt+1 has landed ; lgkmcnt(0): my reads of stage kt are done
      T4_BAR();
      T4_MMA_SEG(fae, fbe);
      // ---- sub-step 3 ----
      if (kt + 1 < nk) T4_READ(fae, fbe, buf ^ 1, 0);
      if (kt + 2 < nk) T4_HALF(buf, 0);                   // first half of stage kt+2 into the buffer stage kt just vacated
      T4_FENCE(); T4_BAR();
      T4_MMA_SEG(fao, fbo);
      buf ^= 1;
    }
    if (!late) T4_BAR();  // barrier counts must match across the workgroup
  }
#undef T4_BAR
#undef T4_HALF
#undef T4_MMA_SEG
#undef T4_ISSUE
#undef T4_READ
#undef T4_MMA
#undef T4_FENCE
#undef T4_WAIT_LDS
  gemm_epilogue32(args, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// =================================================================================================
// "p8" (EXPERIMENTAL — written after round 1's GPU budget was spent: compiles for gfx950, NOT yet run; reachable only with
// ENH_GEMM_KERNEL=8phase; tests/test_ops_gpu.py::test_gemm_p8_* are skipped unless ENH_TEST_EXPERIMENTAL=1).
//
// Why: t256 above waits vmcnt(0) once per K-tile — with two 64-KiB K-tile buffers nothing newer is in flight at that point, so the
// global->LDS queue drains every K-tile and a load gets <= 0.75 K-tile (~0.7 us) to arrive.  cdna_hip_programming.md ("The 256^2
// 8-phase template") measures that drain as the whole difference between ~900 and ~1320 TF/s.  The fix is not more LDS but FINER
// SLOTS: the eight 16-KiB half-tiles {stage 0,1} x {A0, A1, B0, B1} are freed and refilled one at a time.  For a half-tile to be
// released before its K-tile is finished, one phase must consume it completely — so a half-tile is not a contiguous 128-row block
// but, for every wave, the rows of ONE of its quadrant operands:
//      A half h = rows  wm*128 + h*64 + (0..63)  for wm = 0,1          B half h = cols  wn*64 + h*32 + (0..31)  for wn = 0..3
// and a phase computes one 64x32 quadrant of the wave's 128x64 tile over the whole K-tile (8 MFMAs 32x32x16):
//      P0: read A0, B0 (12 ds_read_b128) ; q(0,0)        P2: read A1 (8) ; q(1,1)  (B1 still in registers)
//      P1: read B1 (4)                   ; q(0,1)        P3: no read     ; q(1,0)  (B0 still in registers)
// Last reads: A0, B0 in P0, B1 in P1, A1 in P2.  A slot may be refilled two phases after its last read (both staggered wave groups
// have retired their reads and passed a barrier by then), so every phase issues exactly ONE half-tile (2 global_load_lds per wave):
//      P0(kt): B1(kt+1)     P1(kt): A1(kt+1)     P2(kt): A0(kt+2)     P3(kt): B0(kt+2)          -> 5-6 phases (1.25-1.5 K-tiles) of flight
// and the waits are counted: vmcnt(6) = "everything except the three most recent half-tiles has landed", placed in P0, P2, P3 before
// the phase's first barrier; the data each one retires is first read two phases later (one barrier more than strictly required).
//      wait@P0(kt) retires A1(kt) [read P2(kt)]   wait@P2(kt) retires A0,B0(kt+1) [read P0(kt+1)]   wait@P3(kt) retires B1(kt+1) [read P1(kt+1)]
// In the last two K-tiles some issues are skipped, so the count no longer covers the needed half-tile: vmcnt(0) there.
// Same stagger as t256: waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while the other
// issues its reads / loads.  Same LDS images per slot (row / kmaj2 swizzles), same fragment readers, same epilogue.
// =================================================================================================
__device__ __forceinline__ int p8_row_a(int h, int r) { return (r >> 6) * 128 + h * 64 + (r & 63); }   // slot row -> tile row
__device__ __forceinline__ int p8_row_b(int h, int r) { return (r >> 5) * 64 + h * 32 + (r & 31); }

template <bool TR, bool IS_A>
__device__ __forceinline__ const uint16_t* p8_src_ptr(const uint16_t* __restrict__ P, int64_t ld, int64_t x0, int64_t X,
                                                      int64_t k_begin, int h, int slab, int lane) {
  if (!TR) {  // slab = 8 slot rows x 128 B
    const int r = slab * 8 + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((r >> 1) & 7);
    int64_t row = x0 + (IS_A ? p8_row_a(h, r) : p8_row_b(h, r));
    if (row > X - 1) row = X - 1;
    return P + row * ld + k_begin + c * 8;
  } else {    // slab = 4 k-rows x 256 B; a lane's 16 bytes = 8 consecutive slot columns (never straddling a 32-column group)
    const int k = slab * 4 + (lane >> 4), pp = lane & 15;
    const int q = (pp >> 1) ^ (((k & 3) << 1) | ((k >> 2) & 1));
    const int cs = q * 16 + (pp & 1) * 8;
    int64_t col = x0 + (IS_A ? p8_row_a(h, cs) : p8_row_b(h, cs));
    if (col > X - 8) col = X - 8;
    return P + (k_begin + k) * ld + col;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A0 | A1 | B0 | B1], 16 KiB each
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, each 128 (M) x 64 (N)
  int split, tile_m, tile_n;
  gemm_tile_coords(args, split, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * G4_BM, n0 = (int64_t)tile_n * G4_BN;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nk = (int)((k_end - k_begin) / G_BK);

  // every wave stages 2 of the 16 one-KiB slabs of EVERY half-tile kind (0 = A0, 1 = A1, 2 = B0, 3 = B1)
  const uint16_t* src[8];
  int lds_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kind = i >> 1, slab = wave * 2 + (i & 1);
    lds_off[i] = kind * G_TILE_BYTES + slab * 1024;
    src[i] = kind < 2 ? p8_src_ptr<TA, true>(args.A, args.lda, m0, args.M, k_begin, kind, slab, lane)
                      : p8_src_ptr<TB, false>(args.B, args.ldb, n0, args.N, k_begin, kind - 2, slab, lane);
  }
  const int64_t step_a = TA ? (int64_t)G_BK * args.lda : (int64_t)G_BK;
  const int64_t step_b = TB ? (int64_t)G_BK * args.ldb : (int64_t)G_BK;

#define P8_ISSUE(ST, KIND)                                                                                                          \
  do {                                                                                                                              \
    unsigned char* base_ = smem + (ST) * G4_STAGE_BYTES;                                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                                                              \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[(KIND) * 2 + u_], (LDS_AS void*)(base_ + lds_off[(KIND) * 2 + u_]), 16, 0, 0); \
      src[(KIND) * 2 + u_] += ((KIND) < 2 ? step_a : step_b);                                                                       \
    }                                                                                                                               \
  } while (0)
#define P8_READ_A(ST, IH)                                                                                                           \
  do {                                                                                                                              \
    const unsigned char* sa_ = smem + (ST) * G4_STAGE_BYTES + (IH) * G_TILE_BYTES;                                                  \
    _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_)                                                                             \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) fa[ib_][s_] = frag32<TA>(sa_, wm * 64 + ib_ * 32, s_, lane);                 \
  } while (0)
#define P8_READ_B(FB, ST, J)                                                                                                        \
  do {                                                                                                                              \
    const unsigned char* sb_ = smem + (ST) * G4_STAGE_BYTES + (2 + (J)) * G_TILE_BYTES;                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) FB[s_] = frag32<TB>(sb_, wn * 32, s_, lane);                                   \
  } while (0)
  // one quadrant: two accumulators alternate, so consecutive MFMAs never depend on each other
#define P8_MMA(IH, J, FB)                                                                                                           \
  do {                                                                                                                              \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                                \
      _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_)                                                                           \
        acc[(IH) * 2 + ib_][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FB[s_]), __builtin_bit_cast(bf16x8, fa[ib_][s_]), acc[(IH) * 2 + ib_][J], 0, 0, 0); \
  } while (0)
#define P8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P8_BAR() __builtin_amdgcn_s_barrier()
#define P8_WAIT_VM(STEADY)                                                                                                          \
  do {                                                                                                                              \
    if (STEADY) __builtin_amdgcn_s_waitcnt(0x0F76); /* vmcnt(6): all but the three newest half-tiles (2 loads each) have landed */ \
    else __builtin_amdgcn_s_waitcnt(0x0F70);        /* vmcnt(0) */                                                                  \
  } while (0)
  // second half of a phase: close the read / issue segment, then the MFMA segment
#define P8_MMA_SEG(IH, J, FB)                                                                                                       \
  do {                                                                                                                              \
    P8_FENCE(); P8_BAR();                                                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): this phase's fragments are in registers */                                   \
    P8_FENCE();                                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                                  \
    P8_MMA(IH, J, FB);                                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                                  \
    P8_FENCE(); P8_BAR();                                                                                                           \
  } while (0)

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa[2][4], fb0[4], fb1[4];

  if (nk > 0) {
    // prologue: stages 0 and 1 in the steady-state issue order (A0, B0, B1, A1); only stage 0 has to have landed before the loop —
    // stage 1's eight loads stay in flight and are retired by the counted waits of K-tile 0 like any later stage
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st < nk) { P8_ISSUE(st, 0); P8_ISSUE(st, 2); P8_ISSUE(st, 3); P8_ISSUE(st, 1); }
    }
    if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0078);  // vmcnt(8) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0)
    P8_BAR();
    const bool late = wave >= 4;  // wave-uniform
    if (late) P8_BAR();           // stagger: the second wave of every SIMD runs one barrier behind
    for (int kt = 0; kt < nk; ++kt) {
      const int st = kt & 1;
      const bool steady = kt + 2 < nk;       // all four issues of this K-tile and of the previous one exist
      const bool next1 = kt >= 1 && kt + 1 < nk;
      // ---- P0: q(0,0) ----
      P8_READ_B(fb0, st, 0);
      P8_FENCE();
      P8_READ_A(st, 0);
      if (next1) P8_ISSUE(st ^ 1, 3);        // B1(kt+1): its slot was last read in P1(kt-1)
      P8_WAIT_VM(steady);                    // retires A1(kt), read in P2
      P8_MMA_SEG(0, 0, fb0);
      // ---- P1: q(0,1) ----
      P8_READ_B(fb1, st, 1);
      if (next1) P8_ISSUE(st ^ 1, 1);        // A1(kt+1): its slot was last read in P2(kt-1)
      P8_MMA_SEG(0, 1, fb1);
      // ---- P2: q(1,1) ----
      P8_READ_A(st, 1);
      if (steady) P8_ISSUE(st, 0);           // A0(kt+2): A0(kt) was last read in P0
      P8_WAIT_VM(steady);                    // retires A0(kt+1), B0(kt+1), read in P0(kt+1)
      P8_MMA_SEG(1, 1, fb1);
      // ---- P3: q(1,0) ----
      if (steady) P8_ISSUE(st, 2);           // B0(kt+2): B0(kt) was last read in P0
      P8_WAIT_VM(steady);                    // retires B1(kt+1), read in P1(kt+1)
      P8_MMA_SEG(1, 0, fb0);
    }
    if (!late) P8_BAR();  // barrier counts must match across the workgroup
  }
#undef P8_ISSUE
#undef P8_READ_A
#undef P8_READ_B
#undef P8_MMA
#undef P8_FENCE
#undef P8_BAR
#undef P8_WAIT_VM
#undef P8_MMA_SEG
  gemm_epilogue32(args, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// =================================================================================================
// "p8p" (EXPERIMENTAL, never run — ENH_GEMM_KERNEL=9persist): the p8 schedule made persistent.  One workgroup per CU walks the
// tiles b, b + gridDim.x, ... ; the half-tile issue stream of p8 simply CONTINUES across the tile boundary (each of the four kinds
// counts its own K-tiles and re-derives its source pointers when it wraps), so the load queue never drains and the epilogue of one
// tile runs while the first two K-tiles of the next are in flight — what a 12-K-tile problem (K = 768: 42 % of the training step)
// needs from a 256x256 tile.  Synchronisation is p8's with a longer K sequence (tools/p8_schedule_check.py covers it as nk = the
// workgroup's total K-tile count); gfx9 also counts the epilogue's stores / loads in vmcnt, which only makes the counted waits
// conservative right after an epilogue.  No split-K (the launcher falls back to p8 then).  The macro block is a copy of p8's on
// purpose: the two kernels are to be validated and tuned independently.
// =================================================================================================
__device__ __forceinline__ void gemm_tile_coords_v(const GemmArgs& args, int vbid, int& tile_m, int& tile_n) {
  const int nwg = args.nbm * args.nbn;
  int bid = vbid;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int per_group = 8 * args.nbn;
  const int grp = bid / per_group, within = bid - grp * per_group;
  const int rows = (args.nbm - grp * 8) < 8 ? (args.nbm - grp * 8) : 8;
  tile_m = grp * 8 + within % rows;
  tile_n = within / rows;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_bf16_p8p_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A0 | A1 | B0 | B1], 16 KiB each
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = args.nbm * args.nbn;
  const int G = (int)gridDim.x;  // a multiple of 8 whenever ntiles > G (launcher), so a workgroup never leaves its XCD's run of tiles
  const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
  const int nk = (int)(args.K / G_BK);
  const int gtot = my_tiles * nk;  // K-tiles this workgroup will consume, across all of its tiles

  const uint16_t* src[8];
  int lds_off[8];
  int ktc[4] = {0, 0, 0, 0};  // per half-tile kind: K-tile (within its tile) of the NEXT issue ...
  int tjc[4] = {0, 0, 0, 0};  // ... and which of this workgroup's tiles that issue belongs to
#pragma unroll
  for (int i = 0; i < 8; ++i) lds_off[i] = (i >> 1) * G_TILE_BYTES + (wave * 2 + (i & 1)) * 1024;
  const int64_t step_a = TA ? (int64_t)G_BK * args.lda : (int64_t)G_BK;
  const int64_t step_b = TB ? (int64_t)G_BK * args.ldb : (int64_t)G_BK;

#define P9_SETPTR(KIND, J)                                                                                                          \
  do {                                                                                                                              \
    int tm_, tn_;                                                                                                                   \
    gemm_tile_coords_v(args, (int)blockIdx.x + (J) * G, tm_, tn_);                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                                                              \
      if ((KIND) < 2) src[(KIND) * 2 + u_] = p8_src_ptr<TA, true>(args.A, args.lda, (int64_t)tm_ * G4_BM, args.M, 0, (KIND), wave * 2 + u_, lane);        \
      else src[(KIND) * 2 + u_] = p8_src_ptr<TB, false>(args.B, args.ldb, (int64_t)tn_ * G4_BN, args.N, 0, (KIND) - 2, wave * 2 + u_, lane);              \
    }                                                                                                                               \
  } while (0)
#define P9_ISSUE(ST, KIND)                                                                                                          \
  do {                                                                                                                              \
    unsigned char* base_ = smem + (ST) * G4_STAGE_BYTES;                                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_)                                                                                \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[(KIND) * 2 + u_], (LDS_AS void*)(base_ + lds_off[(KIND) * 2 + u_]), 16, 0, 0); \
    if (++ktc[KIND] == nk) {              /* this kind has issued the last K-tile of its tile: move to my next tile */                \
      ktc[KIND] = 0;                                                                                                                \
      ++tjc[KIND];                                                                                                                  \
      if (tjc[KIND] < my_tiles) P9_SETPTR(KIND, tjc[KIND]);                                                                         \
    } else {                                                                                                                        \
      _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) src[(KIND) * 2 + u_] += ((KIND) < 2 ? step_a : step_b);                     \
    }                                                                                                                               \
  } while (0)
#define P9_READ_A(ST, IH)                                                                                                           \
  do {                                                                                                                              \
    const unsigned char* sa_ = smem + (ST) * G4_STAGE_BYTES + (IH) * G_TILE_BYTES;                                                  \
    _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_)                                                                             \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) fa[ib_][s_] = frag32<TA>(sa_, wm * 64 + ib_ * 32, s_, lane);                 \
  } while (0)
#define P9_READ_B(FB, ST, J)                                                                                                        \
  do {                                                                                                                              \
    const unsigned char* sb_ = smem + (ST) * G4_STAGE_BYTES + (2 + (J)) * G_TILE_BYTES;                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) FB[s_] = frag32<TB>(sb_, wn * 32, s_, lane);                                   \
  } while (0)
#define P9_MMA(IH, J, FB)                                                                                                           \
  do {                                                                                                                              \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                                \
      _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_)                                                                           \
        acc[(IH) * 2 + ib_][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FB[s_]), __builtin_bit_cast(bf16x8, fa[ib_][s_]), acc[(IH) * 2 + ib_][J], 0, 0, 0); \
  } while (0)
#define P9_FENCE() __builtin_amdgcn_sched_barrier(0)
#define P9_BAR() __builtin_amdgcn_s_barrier()
#define P9_WAIT_VM(STEADY)                                                                                                          \
  do {                                                                                                                              \
    if (STEADY) __builtin_amdgcn_s_waitcnt(0x0F76); /* vmcnt(6) */                                                                  \
    else __builtin_amdgcn_s_waitcnt(0x0F70);        /* vmcnt(0) */                                                                  \
  } while (0)
#define P9_MMA_SEG(IH, J, FB)                                                                                                       \
  do {                                                                                                                              \
    P9_FENCE(); P9_BAR();                                                                                                           \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                                            \
    P9_FENCE();                                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                                  \
    P9_MMA(IH, J, FB);                                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                                  \
    P9_FENCE(); P9_BAR();                                                                                                           \
  } while (0)
#define P9_ZERO_ACC()                                                                                                               \
  do {                                                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                                \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.f;                                                    \
  } while (0)

  f32x16 acc[4][2];
  P9_ZERO_ACC();
  s16x8 fa[2][4], fb0[4], fb1[4];

  if (gtot > 0) {
    P9_SETPTR(0, 0); P9_SETPTR(1, 0); P9_SETPTR(2, 0); P9_SETPTR(3, 0);
    // prologue: versions 0 and 1 of the stream, in its steady-state order; only version 0 has to have landed
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      if (v < gtot) { P9_ISSUE(v, 0); P9_ISSUE(v, 2); P9_ISSUE(v, 3); P9_ISSUE(v, 1); }
    }
    if (gtot > 1) __builtin_amdgcn_s_waitcnt(0x0078);  // vmcnt(8) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0) lgkmcnt(0)
    P9_BAR();
    const bool late = wave >= 4;
    if (late) P9_BAR();
    int kt_in = 0, tile_j = 0;  // position of the K-tile being consumed
    for (int g = 0; g < gtot; ++g) {
      const int st = g & 1;
      const bool steady = g + 2 < gtot;
      const bool next1 = g >= 1 && g + 1 < gtot;
      // ---- P0: q(0,0) ----
      P9_READ_B(fb0, st, 0);
      P9_FENCE();
      P9_READ_A(st, 0);
      if (next1) P9_ISSUE(st ^ 1, 3);
      P9_WAIT_VM(steady);
      P9_MMA_SEG(0, 0, fb0);
      // ---- P1: q(0,1) ----
      P9_READ_B(fb1, st, 1);
      if (next1) P9_ISSUE(st ^ 1, 1);
      P9_MMA_SEG(0, 1, fb1);
      // ---- P2: q(1,1) ----
      P9_READ_A(st, 1);
      if (steady) P9_ISSUE(st, 0);
      P9_WAIT_VM(steady);
      P9_MMA_SEG(1, 1, fb1);
      // ---- P3: q(1,0) ----
      if (steady) P9_ISSUE(st, 2);
      P9_WAIT_VM(steady);
      P9_MMA_SEG(1, 0, fb0);
      if (++kt_in == nk) {  // the tile is complete: store it (loads of my next tile are already in flight), start the next one
        int tm, tn;
        gemm_tile_coords_v(args, (int)blockIdx.x + tile_j * G, tm, tn);
        gemm_epilogue32(args, acc, (int64_t)tm * G4_BM + wm * 128, (int64_t)tn * G4_BN + wn * 64, lane);
        P9_ZERO_ACC();
        kt_in = 0;
        ++tile_j;
      }
    }
    if (!late) P9_BAR();
  }
#undef P9_SETPTR
#undef P9_ISSUE
#undef P9_READ_A
#undef P9_READ_B
#undef P9_MMA
#undef P9_FENCE
#undef P9_BAR
#undef P9_WAIT_VM
#undef P9_MMA_SEG
#undef P9_ZERO_ACC
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// kernel family: 0 = register-staged (any K % 8), 3 = pipe2, 4 = t256, 5 = p8, 6 = p8p (experimental, only by ENH_GEMM_KERNEL=8phase / 9persist)
static int gemm_family(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K) {
  static const int kernel_sel = [] {  // ENH_GEMM_KERNEL = reg | pipe2 | t256 ; unset = per-shape choice
    const char* e = getenv("ENH_GEMM_KERNEL");
    if (!e) return -1;
    if (e[0] == 'r') return 0;
    if (e[0] == 'p') return 3;
    if (e[0] == 't') return 4;
    if (e[0] == '8') return 5;
    if (e[0] == '9') return 6;
    return -1;
  }();
  const bool k64 = K % G_BK == 0 && (!trans_a || M >= 8) && (!trans_b || N >= 8);
  // per-shape choice (measured on MI355X, profiles/r01_gemm_ablation.txt): the 256x256 tile wins when the K loop is long
  // and the A operand is row-major (fc2 forward, dgrad of qkv / fc1); everything else runs the 128x128 pipe2 kernel.
  int family = !k64 ? 0 : (kernel_sel >= 0 ? kernel_sel : ((K >= 2048 && !trans_a) ? 4 : 3));
  if (family >= 4 && (M < 256 || N < 256)) family = 3;
  return family;
}

extern "C" const char* enh_gemm_bf16_variant(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K) {
  static const char* names[7] = {"gemm_bf16_kernel", "", "", "gemm_bf16_pipe2_kernel", "gemm_bf16_t256_kernel", "gemm_bf16_p8_kernel", "gemm_bf16_p8p_kernel"};
  return names[gemm_family(trans_a, trans_b, M, N, K)];
}

extern "C" int enh_gemm_bf16(const enh_bf16* A, int64_t lda, int trans_a, const enh_bf16* B, int64_t ldb, int trans_b,
                             int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_bf16* aux,
                             int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                             float* c_f32, enh_bf16* c_bf16, int64_t ldc, void* stream) {
  ENH_REQUIRE(A && B && (c_f32 || c_bf16), ENH_E_BADARG, "enh_gemm_bf16: null pointer");
  ENH_REQUIRE(M > 0 && N > 0 && K > 0, ENH_E_BADARG, "enh_gemm_bf16: M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  ENH_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B), ENH_E_SHAPE,
              "enh_gemm_bf16: K, lda, ldb must be multiples of 8 and A, B 16-byte aligned (K=%lld lda=%lld ldb=%lld)", (long long)K, (long long)lda, (long long)ldb);
  ENH_REQUIRE(N % 4 == 0 && ldc % 4 == 0, ENH_E_SHAPE, "enh_gemm_bf16: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", (long long)N, (long long)ldc);
  ENH_REQUIRE(!trans_a || M % 8 == 0, ENH_E_SHAPE, "enh_gemm_bf16: trans_a needs M %% 8 == 0");
  ENH_REQUIRE(!trans_b || N % 8 == 0, ENH_E_SHAPE, "enh_gemm_bf16: trans_b needs N %% 8 == 0");
  ENH_REQUIRE(act == ENH_ACT_NONE || act == ENH_ACT_TANH || (act == ENH_ACT_DTANH && aux && ldaux % 4 == 0), ENH_E_BADARG, "enh_gemm_bf16: bad act/aux");
  ENH_REQUIRE(!res || (res_rows > 0 && ldres % 4 == 0), ENH_E_BADARG, "enh_gemm_bf16: res needs res_rows > 0 and ldres %% 4 == 0");
  ENH_REQUIRE(accumulate == 0 || (accumulate == 1 && c_f32), ENH_E_BADARG, "enh_gemm_bf16: accumulate needs an f32 output");
  ENH_REQUIRE((!c_f32 || aligned16(c_f32)) && (!c_bf16 || (reinterpret_cast<uintptr_t>(c_bf16) & 7u) == 0), ENH_E_SHAPE, "enh_gemm_bf16: output alignment");

  const int family = gemm_family(trans_a, trans_b, M, N, K);
  const int bm = family >= 4 ? G4_BM : G_BM;
  const int bn = family >= 4 ? G4_BN : G_BN;

  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.act = act; g.aux = aux; g.ldaux = ldaux; g.res = res; g.ldres = ldres; g.res_rows = res_rows;
  g.accumulate = accumulate; g.c_f32 = c_f32; g.c_bf16 = c_bf16; g.ldc = ldc;
  g.nbm = (int)((M + bm - 1) / bm);
  g.nbn = (int)((N + bn - 1) / bn);
  const int64_t tiles = (int64_t)g.nbm * g.nbn;
  ENH_REQUIRE(tiles < (1ll << 30), ENH_E_SHAPE, "enh_gemm_bf16: grid too large");
  // split-K (f32 atomics into a pre-initialised C) when a weight-gradient-shaped problem cannot fill 256 CUs
  const int64_t ksteps = (K + G_BK - 1) / G_BK;
  const int64_t fill = family >= 4 ? 256 : 512;  // resident workgroup slots
  int splits = 1;
  if (accumulate == 1 && c_f32 && !c_bf16 && !bias && act == ENH_ACT_NONE && !res && tiles < fill / 2 && K >= 2048) {
    // measured (profiles/r01_gemm_ablation.txt): the f32-atomic epilogue makes every extra K-slice expensive and a ragged
    // last round is worse still -> the largest split count whose workgroups fit ONE round of resident slots
    // (2 workgroups per CU for the 128x128 kernels, 1 for t256)
    const int64_t slots = family >= 4 ? 256 : 512;
    int64_t want = slots / tiles;
    if (want > ksteps / 8) want = ksteps / 8;
    if (want > 64) want = 64;
    if (want >= 2) splits = (int)want;
  }
  g.k_per_split = ((ksteps + splits - 1) / splits) * G_BK;
  splits = (int)((K + g.k_per_split - 1) / g.k_per_split);
  g.splits = splits;
  if (splits > 1) g.accumulate = 2;
  int launch_family = family;
  if (family == 6 && splits > 1) launch_family = 5;  // the persistent kernel has no split-K
  // persistent: one workgroup per CU (256); fewer tiles than that -> one workgroup per tile
  const dim3 grid(launch_family == 6 ? (unsigned)(tiles < 256 ? tiles : 256) : (unsigned)(tiles * splits));
  hipStream_t s = (hipStream_t)stream;
  static const bool attr_set = [] {
    const int b2 = 4 * G_TILE_BYTES;
#define SET_ATTR(K_, B_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K_), hipFuncAttributeMaxDynamicSharedMemorySize, B_)
    SET_ATTR((gemm_bf16_kernel<false, false>), b2); SET_ATTR((gemm_bf16_kernel<false, true>), b2);
    SET_ATTR((gemm_bf16_kernel<true, false>), b2); SET_ATTR((gemm_bf16_kernel<true, true>), b2);
    SET_ATTR((gemm_bf16_pipe2_kernel<false, false>), b2); SET_ATTR((gemm_bf16_pipe2_kernel<false, true>), b2);
    SET_ATTR((gemm_bf16_pipe2_kernel<true, false>), b2); SET_ATTR((gemm_bf16_pipe2_kernel<true, true>), b2);
    SET_ATTR((gemm_bf16_t256_kernel<false, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_t256_kernel<false, true>), 2 * G4_STAGE_BYTES);
    SET_ATTR((gemm_bf16_t256_kernel<true, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_t256_kernel<true, true>), 2 * G4_STAGE_BYTES);
    SET_ATTR((gemm_bf16_p8_kernel<false, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_p8_kernel<false, true>), 2 * G4_STAGE_BYTES);
    SET_ATTR((gemm_bf16_p8_kernel<true, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_p8_kernel<true, true>), 2 * G4_STAGE_BYTES);
    SET_ATTR((gemm_bf16_p8p_kernel<false, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_p8p_kernel<false, true>), 2 * G4_STAGE_BYTES);
    SET_ATTR((gemm_bf16_p8p_kernel<true, false>), 2 * G4_STAGE_BYTES); SET_ATTR((gemm_bf16_p8p_kernel<true, true>), 2 * G4_STAGE_BYTES);
#undef SET_ATTR
    return true;
  }();
  (void)attr_set;
  const size_t lds2 = 4 * G_TILE_BYTES;
#define LAUNCH(KERN, THREADS, LDS)                                                          \
  do {                                                                                      \
    if (!trans_a && !trans_b) KERN<false, false><<<grid, THREADS, LDS, s>>>(g);            \
    else if (!trans_a && trans_b) KERN<false, true><<<grid, THREADS, LDS, s>>>(g);         \
    else if (trans_a && !trans_b) KERN<true, false><<<grid, THREADS, LDS, s>>>(g);         \
    else KERN<true, true><<<grid, THREADS, LDS, s>>>(g);                                   \
  } while (0)
  if (launch_family == 6) LAUNCH(gemm_bf16_p8p_kernel, 512, (size_t)(2 * G4_STAGE_BYTES));
  else if (launch_family == 5) LAUNCH(gemm_bf16_p8_kernel, 512, (size_t)(2 * G4_STAGE_BYTES));
  else if (family == 4) LAUNCH(gemm_bf16_t256_kernel, 512, (size_t)(2 * G4_STAGE_BYTES));
  else if (family == 3) LAUNCH(gemm_bf16_pipe2_kernel, 256, lds2);
  else LAUNCH(gemm_bf16_kernel, 256, lds2);
#undef LAUNCH
  return enh_check_launch("enh_gemm_bf16");
}
