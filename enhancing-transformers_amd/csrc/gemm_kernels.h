// gemm_kernels.h — the 16-bit-operand MFMA GEMM kernels with fused epilogues for gfx950 (CDNA4), wave64: every kernel is a template over the operand
// type tag OT = BF16 | F16 (common.h); gemm_bf16.hip / gemm_f16.hip instantiate gemm_launch<OT> (two translation units: they compile in parallel),
// gemm.hip holds the planner and the C ABI.
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
// Kernel families (enh_gemm_bf16_variant() reports the per-shape choice; enh_gemm_set_kernel() overrides it):
//   gemm_w256_kernel   256x256x64 tile, 4 waves (2x2, each 4x4 v_mfma_f32_32x32x16_bf16 = 128x128, one wave per SIMD), round 2; every shape
//                           with M, N multiples of 256 that fills the chip (with split-K if needed).
//   gemm_pipe2_kernel  128x128x64 tile, 4 waves (2x2, each 4x4 v_mfma_f32_16x16x32_bf16), two 32-KiB LDS stages filled by
//                           global_load_lds, K-loop software-pipelined around one mid-iteration barrier; 2 workgroups per CU.  Default.
//   gemm_kernel        register-staged 128x128x64 fallback for K not a multiple of 64 (zero-fills partial tiles).
// All LDS images are XOR-swizzled so that staging writes, ds_read_b128 fragments and the transpose reads are bank-conflict free under
// the gfx950 bank model (MI355X_MICROARCH.md §LDS; checked by tools/lds_bank_check.py; SQ_LDS_BANK_CONFLICT = 0 measured).
// The MFMA is issued with swapped operands (D = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE output columns of one row: the
// epilogue reads bias / residual / aux and writes C with 16-byte (f32) or 8-byte (bf16) accesses.  Workgroup ids are remapped so each XCD
// (private L2) walks a contiguous, grouped run of tiles.  What bounds these kernels (L2 misses, not structure): DESIGN.md §3.1.
#pragma once
#include <stdlib.h>
#include "common.h"

#include "gemm_launch.h"

template <typename OT, bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs args) {
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
          acc[i][j] = mfma16<OT>(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      unsigned char* na = smem + (stage ^ 1) * (2 * G_TILE_BYTES);
      tile_sstore<TA>(ra, na, t);
      tile_sstore<TB>(rb, na + G_TILE_BYTES, t);
    }
    __syncthreads();
  }

  gemm_epilogue<OT>(args, acc, m0, n0, wm, wn, lg, l16, split);
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
template <typename OT, bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_pipe2_kernel(const GemmArgs args) {
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
                                       (LDS_AS void*)(base_ + (i_ >> 2) * G_TILE_BYTES + (wave * 4 + (i_ & 3)) * 1024), 16, 0, ENH_GLDS_AUX); \
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
        acc[i_][j_] = mfma16<OT>(FB[j_], FA[i_], acc[i_][j_]); \
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
        acc[i][jj * 2] = mfma16<OT>(fb1[jj * 2], fa1[i], acc[i][jj * 2]);
        acc[i][jj * 2 + 1] = mfma16<OT>(fb1[jj * 2 + 1], fa1[i], acc[i][jj * 2 + 1]);
        if (more) {
          __builtin_amdgcn_global_load_lds((const GLB_AS void*)src[ld], (LDS_AS void*)(nbase + (ld >> 2) * G_TILE_BYTES + (wave * 4 + (ld & 3)) * 1024), 16, 0, ENH_GLDS_AUX);
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
  gemm_epilogue<OT>(args, acc, m0, n0, wm, wn, lg, l16, split);
}


// =================================================================================================
// helpers of the 256 x 256 kernel: 32x32x16 fragment readers over the "row" image and over a second contraction-major image ("kmaj2":
// chunk q ^ (2*(k&3) | (k>>2)&1)) that keeps the 32-column transpose reads of that fragment shape conflict-free.  (The 8-wave "t256" kernel these
// were written for in round 1 lost to the 4-wave w256 on every shape — fc2 forward 940 vs 957, dgrad 958/991 vs 995/1017 TF/s — and was removed
// together with the two 8-phase variants that had never run; numbers in profiles/r02_gemm_lab.txt.)
// =================================================================================================
#define G4_STAGE_BYTES (4 * G_TILE_BYTES)

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
// (gemm_epilogue32_loops, the epilogue of the swapped 32x32 accumulator layout: gemm_tiles.h)

// =================================================================================================
// "w256": 256 x 256 x 64 workgroup tile, FOUR waves (2 x 2) of 128 x 128 — one wave per SIMD, 256 accumulator registers (AGPRs) + ~170 VGPRs.
// Round-2 design, measured step by step in tools/probe/gemm_lab.cpp (profiles/r02_gemm_lab.txt):
//   * one wave per SIMD reads each LDS byte once per 128 x 128 sub-tile: 32 fragment reads per 64 MFMAs (t256's 128 x 64 waves need 48), and
//     there is no second wave group to keep in phase — ONE barrier per K stage instead of eight;
//   * an in-order wave stalls the matrix pipe whenever an instruction takes longer to issue than the ~28 cycles of cover one MFMA gives, so
//     nothing is issued in bursts: fragment reads go one per MFMA under the first 8 MFMAs of every k16 step (all four waves hit the one LDS
//     at once: a burst of 32 reads costs ~128 cycles), global_load_lds one per two MFMAs (texture addresser ~64 B/clk per CU);
//     measured MFMA utilisation inside the K loop: 96 % without loads, 90 % with L2-resident operands, 70-80 % streaming from HBM;
//   * operands are staged as WHOLE 128-byte lines (64-deep K stages): fetching each line as two 64-byte halves one stage apart (a 4-slot
//     ring of 32-deep stages, which would allow a deeper prefetch) costs 7-11 % utilisation on HBM-streamed operands, while one stage less
//     of prefetch depth costs only 1-2 %;
//   * two 64-KiB slots [A0 | A1 | B0 | B1] (row / kmaj2 images as t256).  The barrier sits after the reads of the last k-step: the slot is
//     then free and the loads of stage j+2 are spread over the next 32 MFMAs; every load gets 32-64 MFMAs (1-2 K-steps x 4) to land and the
//     wait at the next barrier is vmcnt(0) with nothing newer in flight — a count, not a drain.
// Shapes: M, N multiples of 256, every K slice a multiple of 64 with at least two stages; everything else runs pipe2 / the fallback.
// =================================================================================================
#define W2_SLOT (4 * G_TILE_BYTES)
#define W2_BIAS_BYTES 2048   // behind the two slots: 128 f32 bias values per wave for the epilogue (see gemm_epilogue32_loops)
// staging source of the lane for slab parity p (slabs 2u + p): !TR: 8 rows x 128 B per slab ; TR: 4 k-rows x 256 B per slab
template <bool TR>
__device__ __forceinline__ const uint16_t* w256_src(const uint16_t* __restrict__ P, int64_t ld, int64_t x0, int64_t k_begin, int p, int lane) {
  if (!TR) {
    const int r = p * 8 + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((r >> 1) & 7);             // (slab*8 + r) >> 1 & 7 depends on the slab only through its parity
    return P + (x0 + r) * ld + k_begin + c * 8;
  } else {
    const int k = p * 4 + (lane >> 4), pp = lane & 15;
    const int q = (pp >> 1) ^ (((k & 3) << 1) | ((k >> 2) & 1));
    return P + (k_begin + k) * ld + x0 + q * 16 + (pp & 1) * 8;
  }
}

// EPI: the epilogue mode is a template parameter of THIS kernel (chosen on the host): an in-kernel 8-way switch over unrolled epilogues made the
// code 10x larger and the whole kernel ~10 % slower (measured, same main loop)
// LAB (measurement only, wrong results): 1 = fragments fetched with plain ds_read_b128 from the same tiles (same LDS bytes, no transpose reads),
// 2 = no fragment reads at all (the loop's MFMA + LDS-DMA ceiling), 3 = neither fragment reads nor staging requests (MFMAs + barriers), 4 = all reads,
// every second staging request, 5 = all reads, no staging requests
template <typename OT, bool TA, bool TB, int EPI, int LAB>
__device__ __forceinline__ void gemm_w256_body(const GemmArgs& args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 slots][A0 | A1 | B0 | B1], 16 KiB each
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, each 128 (M) x 128 (N)
  int split, tile_m, tile_n;
  gemm_tile_coords(args, split, tile_m, tile_n);
  const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nst = (int)((k_end - k_begin) / G_BK);   // >= 2 (launcher)

  // staging: wave w fills sub-tile w of every slot (0, 1: A halves ; 2, 3: B halves): 16 one-KiB slabs per stage, two source patterns
  const bool stage_a = wave < 2;   // wave-uniform
  const bool my_tr = stage_a ? TA : TB;
  const int64_t my_ld = stage_a ? args.lda : args.ldb;
  const int64_t pair_step = (my_tr ? 8 : 16) * my_ld;              // elements between slabs u and u + 2
  const int64_t stage_step = my_tr ? (int64_t)G_BK * my_ld : (int64_t)G_BK;
  const uint16_t* gsrc_e = stage_a ? w256_src<TA>(args.A, args.lda, m0 + wave * 128, k_begin, 0, lane) : w256_src<TB>(args.B, args.ldb, n0 + (wave - 2) * 128, k_begin, 0, lane);
  const uint16_t* gsrc_o = stage_a ? w256_src<TA>(args.A, args.lda, m0 + wave * 128, k_begin, 1, lane) : w256_src<TB>(args.B, args.ldb, n0 + (wave - 2) * 128, k_begin, 1, lane);
  unsigned char* const my_sub = smem + wave * G_TILE_BYTES;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  if (LAB == 2 || LAB == 3) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { fa0[u] = (s16x8){1, 1, 1, 1, 1, 1, 1, 1}; fb0[u] = fa0[u]; fa1[u] = fa0[u]; fb1[u] = fa0[u]; }
  }

#define W2_ISSUE_ONE(SLOT, U)                                                                                                     \
  do {                                                                                                                            \
    if (LAB == 3 || LAB == 5 || (LAB == 4 && ((U) & 1))) break;                                                                   \
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)((((U) & 1) ? gsrc_o : gsrc_e) + ((U) >> 1) * pair_step),                \
                                     (LDS_AS void*)(my_sub + (SLOT) * W2_SLOT + (U) * 1024), 16, 0,                               \
                                     LAB == 6 ? 1 : (LAB == 7 ? 2 : (LAB == 8 ? 16 : (LAB == 9 ? 17 : ENH_GLDS_AUX)))); /* lab 6-9: sc0 / nt / sc1 / sc0 sc1 */ \
  } while (0)
#define W2_ADVANCE() do { gsrc_e += stage_step; gsrc_o += stage_step; } while (0)
  // fragment u of k16-step S from slot SLOT: u = 0..3 the wave's A row-blocks, 4..7 its B column-blocks.  Transposed operands are read with the
  // asm transpose read (the compiler's wait-count pass knows nothing about them: explicit lgkmcnt(0) at every k-step boundary below)
#define W2_READ_ONE(FA, FB, SLOT, S, U)                                                                                           \
  do {                                                                                                                            \
    if (LAB == 2 || LAB == 3) break;                                                                                              \
    if ((U) < 4) FA[(U) & 3] = frag32<TA && LAB == 0>(smem + (SLOT) * W2_SLOT + wm * G_TILE_BYTES, ((U) & 3) * 32, S, lane);      \
    else FB[(U) & 3] = frag32<TB && LAB == 0>(smem + (SLOT) * W2_SLOT + (2 + wn) * G_TILE_BYTES, ((U) & 3) * 32, S, lane);        \
  } while (0)
#define W2_MM(Q, FA, FB)                                                                                                          \
  acc[(Q) >> 2][(Q) & 3] = mfma32<OT>(FB[(Q) & 3], FA[(Q) >> 2], acc[(Q) >> 2][(Q) & 3])
#define W2_MMZ(Q, FA, FB)                                                                                                         \
  acc[(Q) >> 2][(Q) & 3] = mfma32<OT>(FB[(Q) & 3], FA[(Q) >> 2], zero16)
#define W2_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one k16 step: 16 MFMAs on (FA, FB); under MFMAs 0-7 one fragment read each (k-step RS of slot RSLOT into RA / RB); under every odd MFMA one
  // global_load_lds (pieces G0 .. G0+7 into slot GSLOT).  The step opens with lgkmcnt(0): its fragments were read >= 8 MFMAs ago.
#define W2_KSTEP(FA, FB, RA, RB, RSLOT, RS, DO_READ, GSLOT, G0, DO_ISSUE)                                                         \
  do {                                                                                                                            \
    if (TA || TB) __builtin_amdgcn_s_waitcnt(0xC07F); /* asm transpose reads are invisible to the compiler's wait-count pass */      \
    W2_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      W2_MM(q_, FA, FB);                                                                                                          \
      if ((DO_READ) && q_ < 8) { W2_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                            \
      if ((DO_ISSUE) && (q_ & 1)) { W2_ISSUE_ONE(GSLOT, (G0) + (q_ >> 1)); }                                                      \
      W2_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)

  // the first k16 step of a tile in the persistent kernel: C operand = 0 instead of cleared accumulators; reads k-step RS, requests nothing
#define W2_KSTEP_Z(FA, FB, RA, RB, RSLOT, RS)                                                                                     \
  do {                                                                                                                            \
    if (TA || TB) __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    W2_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      W2_MMZ(q_, FA, FB);                                                                                                         \
      if (q_ < 8) { W2_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                                         \
      W2_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)

  // prologue: stage 0 -> slot 0 completely; pieces 0-7 of stage 1 -> slot 1 (pieces 8-15 follow under the first k-step)
  {
#pragma unroll
  for (int u = 0; u < 16; ++u) W2_ISSUE_ONE(0, u);
  W2_ADVANCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2_ISSUE_ONE(1, u);
  __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): stage 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2_READ_ONE(fa0, fb0, 0, 0, u);
  W2_FENCE();

  // invariant at the top of iteration j: the source pointers are at stage j+1, whose pieces 0-7 are already issued into slot (j+1)&1
  int j = 0;
  for (; j + 2 < nst; ++j) {
    const int slot = j & 1;
    W2_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage j+1
    W2_ADVANCE();
    W2_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    W2_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): stage j+1 landed (nothing newer outstanding) ; lgkmcnt(0): this slot is read out
    __builtin_amdgcn_s_barrier();
    W2_FENCE();
    W2_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, slot, 0, true);      // + pieces 0-7 of stage j+2 into the slot just vacated
  }
  {  // tail: stages nst-2 and nst-1
    const int slot = j & 1;
    W2_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage nst-1
    W2_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    W2_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    W2_FENCE();
    W2_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, 0, 0, false);
    W2_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 1, true, 0, 0, false);
    W2_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 2, true, 0, 0, false);
    W2_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 3, true, 0, 0, false);
    W2_KSTEP(fa1, fb1, fa0, fb0, 0, 0, false, 0, 0, false);
  }
  }
  gemm_epilogue32_loops<EPI, 4, false, OT>(args, acc, m0 + wm * 128, n0 + wn * 128, lane, split,
                                       reinterpret_cast<float*>(smem + 2 * W2_SLOT) + wave * 128, smem + wave * 8192, smem + wave * 16384);
}

template <typename OT, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w256_kernel(const GemmArgs args) {
  gemm_w256_body<OT, TA, TB, EPI, 0>(args);
}
// the measurement-only forms of the split-K weight-gradient loop (enh_debug_gemm_lab)
template <int LAB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w256_lab_kernel(const GemmArgs args) {
  gemm_w256_body<BF16, true, true, EPI_WS, LAB>(args);
}

// =================================================================================================
// "w256p": the w256 main loop as a PERSISTENT kernel (round 3; laboratory notes: profiles/r03_gemm_persistent_lab.txt).  One workgroup per CU walks
// its tiles (virtual block ids b, b + grid, ...: the same XCD-grouped order as w256), and the K stages of consecutive tiles form ONE load stream: the
// next tile's stage 0 is requested under the last stage's MFMAs and has landed before the epilogue starts; its stage 1 is requested before the first
// store.  No launch gap, no cold prologue per tile, and the stores drain under the next tile's first stage instead of in front of a workgroup exit.
// vmcnt counts loads and stores in one counter and a wait can only name how many of the youngest operations may remain, so the epilogue is arranged so
// that nothing it READS is waited for behind one of its stores:
//   * the saved tanh output is requested for all eight blocks before the first store (as whole row segments, brought into the accumulator layout
//     through a second LDS tile); the f32 residual two row-blocks ahead, in the output's layout (the later row-blocks' requests follow earlier
//     stores: their waits are the one place where a load is awaited with stores in flight); the bias a K loop early;
//   * outputs leave through a wave-private 4-KiB LDS tile that does NOT overlay the K slots (32 rows x 128 B per block: whole 128-byte lines per row),
//     so no barrier separates the K loop from the epilogue and the next tile's operands are already in LDS while the stores drain;
//   * accumulators are copied out with explicit v_accvgpr_read at the point of use and every lane-derived address is recomputed per tile: left to the
//     register allocator the epilogue held all 256 accumulators in vector registers and spilled — and a scratch reload is a vector-memory load, i.e.
//     a vmcnt wait on the next tile's requests.
// Accumulators are not cleared: the first k16 step of a tile multiplies into a zero C operand.
// LDS: [2 slots, 128 KiB][bias strips, 2 KiB][4 x 4 KiB store tiles] = 146 KiB (tanh' mode: [2 slots][4 input tiles][4 store tiles] = 160 KiB).
// Forward / input-gradient roles only (A stored [M][K], no split-K).
// =================================================================================================
#define W2P_STAGE_BYTES 4096
#define W2P_LDS_BYTES (2 * W2_SLOT + W2_BIAS_BYTES + 4 * W2P_STAGE_BYTES)
#define W2P_LDS_BYTES_DTANH (2 * W2_SLOT + 8 * W2P_STAGE_BYTES)   // 160 KiB: the whole LDS of a CU

// accumulator -> vector register AT THIS POINT of the instruction stream (the register allocator otherwise copies all 256 accumulators out at the top
// of the epilogue: 256 live registers, spills, and scratch reloads are vector-memory operations that wait on the next tile's requests)
// (acc_read: gemm_tiles.h)

// Cache policy of the persistent epilogues' result stores (round 5, tools/gpu_session.sh lib-ab -> profiles/r05_cache_policy_ab.txt, same-box in-step A/B):
// with the non-temporal hint on EVERY mode's stores the step gains 0.7 % — all of it in the two modes whose tile also READS a row-contiguous operand
// (bias + residual f32: 0.433 -> 0.419 ms; tanh': 0.762 -> 0.740 ms), while the plain bf16 / bias + tanh modes lose 0.5 %.  So: nt where it pays.
// ENH_P_NT_STORE = 1 forces it everywhere, 0 nowhere (lab).  ENH_A_NT (lab): the streamed A operand requested non-temporal — measured slower
// (qkv forward 0.408 -> 0.424 ms), left off.
#ifndef ENH_P_NT_STORE
#define ENH_P_NT_STORE -1
#endif
#ifndef ENH_A_NT
#define ENH_A_NT 0
#endif
template <bool NT, typename T>
__device__ __forceinline__ void p_store(T* ptr, const T& v) {
  if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
}
// ENH_P_NT_LOAD: the epilogue's read-once row-contiguous operands (saved tanh output, residual stream) are requested non-temporal as well: +0.24 % on the step
// (bias + residual 0.4286 -> 0.4238 ms, tanh' 0.7412 -> 0.7325; same-box A/B, profiles/r05_cache_policy_ab.txt); 0 = plain loads (lab)
#ifndef ENH_P_NT_LOAD
#define ENH_P_NT_LOAD 1
#endif
template <typename T>
__device__ __forceinline__ T p_load(const T* ptr) {
  if (ENH_P_NT_LOAD) return __builtin_nontemporal_load(ptr);
  return *ptr;
}
template <int MODE, typename OT>
__device__ __forceinline__ void gemm_epilogue_p(const GemmArgs& args, f32x16 (&acc)[4][4], int64_t mw, int64_t nw, int lane_in, float* wave_bias, unsigned char* st,
                                                unsigned char* at, const float4& bias4) {
  // everything lane-derived is recomputed per tile: hoisted out of the persistent loop, the ~40 loop-invariant addresses would be carried through the
  // K loop and spilled (scratch reloads are vector-memory operations: they would put vmcnt waits on the next tile's requests into the epilogue)
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr bool HAS_BIAS = MODE == EPI_BF16_BIAS_TANH || MODE == EPI_F32_BIAS_RES || MODE == EPI_BF16_TANH_SPLIT;
  constexpr bool SPLIT = MODE == EPI_BF16_SPLIT || MODE == EPI_BF16_TANH_SPLIT;   // x3 producers: hi = bf16(v) and lo = bf16(v - hi) planes (csrc/x3.hip split2 / split3, fused)
  constexpr bool OUT16 = MODE == EPI_BF16 || MODE == EPI_BF16_BIAS_TANH || MODE == EPI_BF16_DTANH || SPLIT;
  constexpr bool NT_OUT = ENH_P_NT_STORE < 0 ? (MODE == EPI_F32_BIAS_RES || MODE == EPI_BF16_DTANH) : (ENH_P_NT_STORE != 0);
  // the tile's bias values were requested by the caller a K loop ago (bias4, lanes 0-31): requested here, the wait for them would also be a wait for
  // the next tile's operand requests, which are older (vmcnt retires in order)
  if (HAS_BIAS && lane < 32) *reinterpret_cast<float4*>(wave_bias + lane * 4) = bias4;
  const int rrow = lane >> 3, rc = lane & 7;
  if (OUT16) {
    // The saved tanh output (tanh' mode) is read the way the output is written: whole 128-byte row segments, 16 bytes per lane (32 x 64 block = four
    // loads), all eight blocks of the wave's tile requested before the first store (128 registers; the K loop's fragment registers are dead here),
    // and brought into the accumulator layout through a second wave-private LDS tile.  Read in the accumulator layout directly (8 bytes per lane,
    // 32 rows x 16 B per instruction) the same bytes cost the L1 eight times the line accesses: ~4 us of a 32-us tile.
    u32x4 a4[8][4];
    const unsigned aux_off = (unsigned)rrow * (unsigned)args.ldaux + (unsigned)rc * 8u;
    if (MODE == EPI_BF16_DTANH) {
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int p = 0; p < 4; ++p)
          a4[b][p] = p_load(reinterpret_cast<const u32x4*>(args.aux + ((mw + (b >> 1) * 32 + p * 8) * args.ldaux + nw + (b & 1) * 64) + aux_off));
    }
    const int wsw = (l31 >> 1) & 7;
    const unsigned out_off = (unsigned)rrow * (unsigned)args.ldc + (unsigned)rc * 8u;
    uint2 hq[2][8];   // the block's saved values in the accumulator layout; the next block's are fetched from LDS while this one is computed
    float cs[2][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};   // column sums (tanh' mode with args.colpart)
    if (MODE == EPI_BF16_DTANH) {
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(at + (p * 8 + rrow) * 128 + ((rc ^ (((p * 8 + rrow) >> 1) & 7)) << 4)) = a4[0][p];
#pragma unroll
      for (int c = 0; c < 8; ++c) hq[0][c] = *reinterpret_cast<const uint2*>(at + l31 * 128 + ((c ^ wsw) << 4) + hi * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        const int b = i * 2 + jh;
        if (MODE == EPI_BF16_DTANH && b + 1 < 8) {
#pragma unroll
          for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(at + (p * 8 + rrow) * 128 + ((rc ^ (((p * 8 + rrow) >> 1) & 7)) << 4)) = a4[b + 1][p];
#pragma unroll
          for (int c = 0; c < 8; ++c) hq[(b + 1) & 1][c] = *reinterpret_cast<const uint2*>(at + l31 * 128 + ((c ^ wsw) << 4) + hi * 8);
        }
        u32x2 lo_[SPLIT ? 8 : 1];   // SPLIT: the block's lo plane, parked in registers while the hi plane goes through the store tile
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int j = jh * 2 + j2;
            float v[4] = {acc_read(acc[i][j][g4 * 4 + 0]), acc_read(acc[i][j][g4 * 4 + 1]), acc_read(acc[i][j][g4 * 4 + 2]), acc_read(acc[i][j][g4 * 4 + 3])};
            const float4 b4 = HAS_BIAS ? *reinterpret_cast<const float4*>(wave_bias + j * 32 + 8 * g4 + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
            EpiIn in;
            in.aux = hq[MODE == EPI_BF16_DTANH ? (b & 1) : 0][MODE == EPI_BF16_DTANH ? j2 * 4 + g4 : 0];
            epi_value<MODE, OT>(args, v, in, b4, nw + j * 32 + 8 * g4 + 4 * hi);
            const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
            *reinterpret_cast<u32x2*>(st + l31 * 128 + (((j2 * 4 + g4) ^ wsw) << 4) + hi * 8) = o_;
            if (SPLIT) {   // lo = bf16(v - float(hi)), the definition of csrc/x3.hip (bitwise: the subtraction is exact in f32)
              const float l0 = v[0] - __builtin_bit_cast(float, o_.x << 16), l1 = v[1] - __builtin_bit_cast(float, o_.x & 0xffff0000u);
              const float l2 = v[2] - __builtin_bit_cast(float, o_.y << 16), l3 = v[3] - __builtin_bit_cast(float, o_.y & 0xffff0000u);
              lo_[j2 * 4 + g4] = (u32x2){pack_bf16x2(l0, l1), pack_bf16x2(l2, l3)};
            }
          }
        u32x4 w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row = p * 8 + rrow;
          w[p] = *reinterpret_cast<const u32x4*>(st + row * 128 + ((rc ^ ((row >> 1) & 7)) << 4));
        }
        if (SPLIT) {   // the lo plane through the same tile (a wave's LDS operations execute in order: these writes follow the reads above)
          u32x4 wl[4];
#pragma unroll
          for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x2*>(st + l31 * 128 + ((c ^ wsw) << 4) + hi * 8) = lo_[c];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int row = p * 8 + rrow;
            wl[p] = *reinterpret_cast<const u32x4*>(st + row * 128 + ((rc ^ ((row >> 1) & 7)) << 4));
          }
          const unsigned lo_off = (unsigned)rrow * (unsigned)args.ldlo + (unsigned)rc * 8u;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            *reinterpret_cast<u32x4*>(args.clo + ((mw + i * 32 + p * 8) * args.ldlo + nw + jh * 64) + lo_off) = wl[p];
          if (args.c2) {     // (wave-uniform) the second / third copy of the hi plane: the x3 row [hi | lo | hi] and the plane the backward reads
            const unsigned o2 = (unsigned)rrow * (unsigned)args.ldc2 + (unsigned)rc * 8u;
#pragma unroll
            for (int p = 0; p < 4; ++p)
              *reinterpret_cast<u32x4*>(args.c2 + ((mw + i * 32 + p * 8) * args.ldc2 + nw + jh * 64) + o2) = w[p];
          }
          if (args.c3) {
            const unsigned o3 = (unsigned)rrow * (unsigned)args.ldc3 + (unsigned)rc * 8u;
#pragma unroll
            for (int p = 0; p < 4; ++p)
              *reinterpret_cast<u32x4*>(args.c3 + ((mw + i * 32 + p * 8) * args.ldc3 + nw + jh * 64) + o3) = w[p];
          }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
          p_store<NT_OUT>(reinterpret_cast<u32x4*>(args.c_bf16 + ((mw + i * 32 + p * 8) * args.ldc + nw + jh * 64) + out_off), w[p]);   // uniform base + 32-bit lane offset
        if (MODE == EPI_BF16_DTANH && args.colpart) {   // column sums of what was just stored (the ROUNDED values): this lane's 8 columns of 4 rows
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              cs[jh][2 * e] += unpack_lo<OT>(w[p][e]);
              cs[jh][2 * e + 1] += unpack_hi<OT>(w[p][e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one block at a time: the scheduler otherwise reads all 256 accumulators first
      }
    if (MODE == EPI_BF16_DTANH && args.colpart) {
      // lane (rrow, rc) holds, per column half, the sums of its 8 columns over rows rrow, rrow + 8, ... of the wave's 128: the 8 row classes meet in
      // the (now idle) input tile [8][128] f32, lane l adds columns 2l, 2l + 1 in the fixed order 0..7 and writes the wave's partial row
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        const f32x4 lo = {cs[jh][0], cs[jh][1], cs[jh][2], cs[jh][3]}, up = {cs[jh][4], cs[jh][5], cs[jh][6], cs[jh][7]};
        *reinterpret_cast<f32x4*>(at + rrow * 512 + (jh * 64 + rc * 8) * 4) = lo;
        *reinterpret_cast<f32x4*>(at + rrow * 512 + (jh * 64 + rc * 8) * 4 + 16) = up;
      }
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float2 v2 = *reinterpret_cast<const float2*>(at + r * 512 + lane * 8);
        s0 += v2.x; s1 += v2.y;
      }
      *reinterpret_cast<float2*>(args.colpart + (mw >> 7) * args.N + nw + lane * 2) = make_float2(s0, s1);
    }
  } else {
    // f32 outputs: 32 x 32 blocks.  The residual stream is read in the OUTPUT's layout (whole 128-byte row segments, 16 bytes per lane) and added after
    // the transposition, two row-blocks (128 registers) ahead: the first two before any store, the others as their registers come free.
    float* dst = args.c_f32 + mw * args.ldc + nw;
    const unsigned out_off = (unsigned)rrow * (unsigned)args.ldc + (unsigned)rc * 4u;
    const unsigned res_off = (unsigned)rrow * (unsigned)args.ldres + (unsigned)rc * 4u;   // res_rows == M here (launcher): no row wrap
    f32x4 rr[2][4][4];
    if (MODE == EPI_F32_BIAS_RES) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 4; ++p)
            rr[i][j][p] = p_load(reinterpret_cast<const f32x4*>(args.res + ((mw + i * 32 + p * 8) * args.ldres + nw + j * 32) + res_off));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 o_ = {acc[i][j][g4 * 4 + 0], acc[i][j][g4 * 4 + 1], acc[i][j][g4 * 4 + 2], acc[i][j][g4 * 4 + 3]};
          *reinterpret_cast<f32x4*>(st + l31 * 128 + (((g4 * 2 + hi) ^ (l31 & 7)) << 4)) = o_;
        }
        f32x4 w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row = p * 8 + rrow;
          w[p] = *reinterpret_cast<const f32x4*>(st + row * 128 + ((rc ^ (row & 7)) << 4));
        }
        if (MODE == EPI_F32_BIAS_RES) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(wave_bias + j * 32 + rc * 4);
#pragma unroll
          for (int p = 0; p < 4; ++p) w[p] = (w[p] + b4) + rr[i & 1][j][p];   // (acc + bias) + residual, the order of every other kernel family
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
          p_store<NT_OUT>(reinterpret_cast<f32x4*>(dst + ((int64_t)(i * 32 + p * 8) * args.ldc + j * 32) + out_off), w[p]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == EPI_F32_BIAS_RES && i + 2 < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 4; ++p)
            rr[i & 1][j][p] = p_load(reinterpret_cast<const f32x4*>(args.res + ((mw + (i + 2) * 32 + p * 8) * args.ldres + nw + j * 32) + res_off));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ---- dynamic tile schedule of the persistent kernels (round 4; measured in profiles/r04_comm_contention.txt) -------------------------------------
// With a static partition (workgroup b walks tiles b, b + grid, ...) and grid = CU count, every workgroup MUST get a CU at once: one wave per SIMD with
// all 512 registers and 146-160 KiB of LDS shares a CU with nothing.  A collective's kernel (RCCL under data-parallel training, engine/ddp.py) that
// holds k CUs when the GEMM is dispatched leaves k workgroups waiting for another to retire — the launch takes up to twice as long.  DYN: tiles are
// CLAIMED instead — one queue per XCD (workgroups of XCD x take virtual tiles x, x + 8, x + 16, ... in order, i.e. exactly the XCD-grouped walk of
// the static form, so operand slices still meet in one L2), an atomic counter per queue.  A workgroup that starts late finds its queue empty and
// exits; the ones that got a CU absorb its share: the launch slows by k / CUs, not 2x.  The claim for the NEXT tile is issued at the top of a tile
// (thread 0; it is the oldest vector-memory operation of everything the K loop then counts, so no counted wait changes), handed to the other waves
// through an LDS word behind the first K-stage barrier, and consumed nst - 2 stages later.  The counters reset themselves: every workgroup makes
// exactly one failing claim, so the claim that returns (tiles of the queue) + (workgroups of the XCD) - 1 is the launch's last and stores 0.
// the mailbox word is read and written with explicit LDS instructions: through a (volatile) generic pointer the compiler emits FLAT accesses, which count in
// vmcnt as well and drew an s_waitcnt vmcnt(0) — a drain of the whole operand pipeline — at every tile (found in the ISA)
__device__ __forceinline__ void lds_store_u32(const void* lds_ptr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(uintptr_t)(LDS_AS const void*)lds_ptr), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_load_u32_sync(const void* lds_ptr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(LDS_AS const void*)lds_ptr) : "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// (gemm.o is compiled with -amdgpu-atomic-optimizer-strategy=None: the optimizer turns a one-lane atomic with a uniform address into a wave reduction that
// needs its result AT ONCE, i.e. s_waitcnt vmcnt(0) right behind the atomic)
__device__ __forceinline__ unsigned tile_claim(unsigned* ctr) { return __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tile_claim_retire(unsigned* ctr, unsigned f, int ntiles, int xcd) {
  const unsigned n_q = ntiles > xcd ? (unsigned)(ntiles - xcd + 7) >> 3 : 0u, w_q = ((unsigned)gridDim.x - (unsigned)xcd + 7u) >> 3;
  if (f == n_q + w_q - 1u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename OT, bool TA, bool TB, int EPI, bool DYN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w256p_kernel(const GemmArgs args) {
  constexpr int LAB = 0;   // (the shared K-step macros name the one-tile kernel's laboratory switch)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = args.nbm * args.nbn;
  const int nst = (int)(args.K / G_BK);   // >= 3 (launcher); no split-K in this kernel
  const int xcd = (int)blockIdx.x & 7;
  unsigned* const ctr = DYN ? args.tile_ctr + xcd : nullptr;

  const bool stage_a = wave < 2;
  const bool my_tr = stage_a ? TA : TB;
  const int64_t my_ld = stage_a ? args.lda : args.ldb;
  const int64_t pair_step = (my_tr ? 8 : 16) * my_ld;
  const int64_t stage_step = my_tr ? (int64_t)G_BK * my_ld : (int64_t)G_BK;
  const int64_t x_step = my_tr ? 1 : my_ld;                    // elements per unit of the wave's row / column origin
  const int xw = stage_a ? wave * 128 : (wave - 2) * 128;
  // lane pointers of the two slab parities at origin 0, K offset 0; a tile adds its (wave-uniform) origin
  const uint16_t* const base_e = stage_a ? w256_src<TA>(args.A, args.lda, 0, 0, 0, lane) : w256_src<TB>(args.B, args.ldb, 0, 0, 0, lane);
  const uint16_t* const base_o = stage_a ? w256_src<TA>(args.A, args.lda, 0, 0, 1, lane) : w256_src<TB>(args.B, args.ldb, 0, 0, 1, lane);
  unsigned char* const my_sub = smem + wave * G_TILE_BYTES;
  // behind the two slots: bias strips (2 KiB) + four store tiles; the tanh' mode has no bias and puts four more tiles (the saved tanh output on its
  // way into the accumulator layout) in front of the store tiles: 160 KiB in all
  float* const wave_bias = reinterpret_cast<float*>(smem + 2 * W2_SLOT) + wave * 128;
  unsigned char* const at = smem + 2 * W2_SLOT + wave * W2P_STAGE_BYTES;
  unsigned char* const st = smem + 2 * W2_SLOT + (EPI == EPI_BF16_DTANH ? 4 * W2P_STAGE_BYTES : W2_BIAS_BYTES) + wave * W2P_STAGE_BYTES;
  // mailbox of the dynamic schedule: the first word of wave 0's store tile (idle from the end of an epilogue to the next one)
  const unsigned char* const s_next = smem + 2 * W2_SLOT + (EPI == EPI_BF16_DTANH ? 4 * W2P_STAGE_BYTES : W2_BIAS_BYTES);

  int vt = (int)blockIdx.x, split_, tile_m, tile_n;
  if (DYN) {
    if (t == 0) {
      const unsigned f = tile_claim(ctr);
      tile_claim_retire(ctr, f, ntiles, xcd);
      lds_store_u32(s_next, f);
    }
    __syncthreads();
    vt = xcd + 8 * (int)lds_load_u32_sync(s_next);
    if (vt >= ntiles) return;     // a workgroup that started late: its queue is empty
    __syncthreads();
  }
  gemm_tile_coords_of(args, vt, split_, tile_m, tile_n);
  const uint16_t *gsrc_e, *gsrc_o;
  {
    const int64_t x0 = (stage_a ? (int64_t)tile_m : (int64_t)tile_n) * 256 + xw;
    gsrc_e = base_e + x0 * x_step; gsrc_o = base_o + x0 * x_step;
  }
  f32x16 acc[4][4];
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // prologue of the workgroup: stages 0 and 1 of its first tile
#pragma unroll
  for (int u = 0; u < 16; ++u) W2_ISSUE_ONE(0, u);
  W2_ADVANCE();
#pragma unroll
  for (int u = 0; u < 16; ++u) W2_ISSUE_ONE(1, u);
  W2_ADVANCE();
  __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16): stage 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2_READ_ONE(fa0, fb0, 0, 0, u);
  W2_FENCE();

  int par = 0;   // slot of the current tile's stage 0
  for (;;) {
    // invariant: stage 0 of this tile is in slot par and its first fragments in fa0 / fb0; stage 1 is completely requested into slot par ^ 1;
    // the source pointers are at stage 2
    const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;
    int vnext = vt + (int)gridDim.x;
    bool last_tile = vnext >= ntiles;
    if (last_tile) vnext = vt;   // the last tile re-requests its own first stages (never consumed; drained before the kernel ends)
    unsigned fnext = 0;
    if (DYN && t == 0) fnext = tile_claim(ctr);     // the next tile of this XCD's queue: in flight under the first K stage
    {
      W2_KSTEP_Z(fa0, fb0, fa1, fb1, par, 1);
      W2_KSTEP(fa1, fb1, fa0, fb0, par, 2, true, 0, 0, false);
      W2_KSTEP(fa0, fb0, fa1, fb1, par, 3, true, 0, 0, false);
      __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): stage 1 landed (and the previous tile's stores are acknowledged, and the claim has returned)
      if (DYN) {
        if (t == 0) { tile_claim_retire(ctr, fnext, ntiles, xcd); lds_store_u32(s_next, fnext); }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the mailbox write is performed before the barrier lets anybody read it
      }
      __builtin_amdgcn_s_barrier();
      W2_FENCE();
      if (DYN) {
        vnext = xcd + 8 * (int)lds_load_u32_sync(s_next);
        last_tile = vnext >= ntiles;
        if (last_tile) vnext = vt;
      }
      W2_KSTEP(fa1, fb1, fa0, fb0, par ^ 1, 0, true, par, 0, true);      // + pieces 0-7 of stage 2
    }
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);   // this tile's bias, for the epilogue's LDS strip: requested now, consumed a K loop later
    if ((EPI == EPI_BF16_BIAS_TANH || EPI == EPI_F32_BIAS_RES || EPI == EPI_BF16_TANH_SPLIT) && lane < 32) bias4 = *reinterpret_cast<const float4*>(args.bias + n0 + wn * 128 + lane * 4);
    for (int j = 1; j < nst; ++j) {
      const int slot = par ^ (j & 1);
      W2_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage j+1
      if (j == nst - 2) {   // stage j+2 is the NEXT tile's stage 0
        gemm_tile_coords_of(args, vnext, split_, tile_m, tile_n);
        const int64_t x0 = (stage_a ? (int64_t)tile_m : (int64_t)tile_n) * 256 + xw;
        gsrc_e = base_e + x0 * x_step; gsrc_o = base_o + x0 * x_step;
      } else {
        W2_ADVANCE();
      }
      W2_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
      W2_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
      __builtin_amdgcn_s_waitcnt(0x0070);
      __builtin_amdgcn_s_barrier();
      W2_FENCE();
      W2_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, slot, 0, true);      // + pieces 0-7 of stage j+2 into the slot just vacated
    }
    {   // the rest of the next tile's stage 1, before any store of this tile
      const int slot = par ^ ((nst - 1) & 1);
#pragma unroll
      for (int u = 8; u < 16; ++u) W2_ISSUE_ONE(slot, u);
      W2_ADVANCE();
    }
    gemm_epilogue_p<EPI, OT>(args, acc, m0 + wm * 128, n0 + wn * 128, lane, wave_bias, st, at, bias4);
    W2_FENCE();
    if (last_tile) break;
    vt = vnext;
    par ^= nst & 1;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the dummy requests write LDS: they must have landed before the workgroup's LDS is released
}
// =================================================================================================
// "w256r": w256p with the A operand REGISTER-STAGED (round 3).  The two-slot LDS ring gives a request at most ~3/4 of a K stage (~0.8 us) to land
// — enough for the L2-resident weights, not for the activation rows that stream from HBM while the previous tile's stores drain
// (profiles/r03_gemm_persistent_lab.txt: with A resident in L2 the same kernel is 6-15 % faster, and only when it also stores).  LDS has no room for
// a third slot, the register file does (the main loop needs ~100 of 256 vector registers): under the first k-step of stage j every wave writes its
// 8 KiB share of A(j+1) from registers into the slot stage j+1 will read and re-uses the registers at once for its share of A(j+3).  B still arrives by
// LDS-DMA, each wave's 8 KiB share requested under the last k-step of stage j for stage j+2.  vmcnt retires in order, so A(j+3) has to be complete
// when B(j+2) — requested after it — is awaited at the end of stage j+1: 1.75 stages after its request instead of 0.5-0.75 (a deeper register
// pipeline would not be allowed to stay in flight any longer: two sets are all the scheme can use).
// Per stage and wave: 8 global loads (A) + 8 LDS-DMA (B) — the same 16 vector-memory operations as w256 — plus 8 ds_write_b128.
// Waits: B(j+1) is awaited with vmcnt(8) (the 8 A loads of this stage are the only younger operations; everything older, the A registers about to be
// written included, is then complete) — except in a tile's first stage, where the previous tile's stores are younger too and are NOT waited for.
// Needs an even number of K stages, at least 6.
// =================================================================================================
#define W2R_D 2
template <typename OT, bool TB, int EPI, bool DYN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w256r_kernel(const GemmArgs args) {
  constexpr int LAB = 0;   // (the shared K-step macros name the one-tile kernel's laboratory switch)
  constexpr bool TA = false;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = args.nbm * args.nbn;
  const int nst = (int)(args.K / G_BK);   // even, >= 6 (launcher)
  const int sub = wave >> 1, half = wave & 1;   // this wave stages slabs half*8 .. half*8+7 of sub-tile `sub` of BOTH operands
  const int xcd = (int)blockIdx.x & 7;
  unsigned* const ctr = DYN ? args.tile_ctr + xcd : nullptr;     // dynamic tile schedule: see gemm_w256p_kernel
  const unsigned char* const s_next = smem + 2 * W2_SLOT + (EPI == EPI_BF16_DTANH ? 4 * W2P_STAGE_BYTES : W2_BIAS_BYTES);

  const uint16_t* const baseA_e = w256_src<false>(args.A, args.lda, sub * 128 + half * 64, 0, 0, lane);
  const uint16_t* const baseA_o = w256_src<false>(args.A, args.lda, sub * 128 + half * 64, 0, 1, lane);
  const uint16_t* const baseB_e = TB ? w256_src<true>(args.B, args.ldb, sub * 128, half * 32, 0, lane) : w256_src<false>(args.B, args.ldb, sub * 128 + half * 64, 0, 0, lane);
  const uint16_t* const baseB_o = TB ? w256_src<true>(args.B, args.ldb, sub * 128, half * 32, 1, lane) : w256_src<false>(args.B, args.ldb, sub * 128 + half * 64, 0, 1, lane);
  const int64_t pairA = 16 * args.lda, pairB = (TB ? 8 : 16) * args.ldb;
  const int64_t stageB = TB ? (int64_t)G_BK * args.ldb : (int64_t)G_BK;
  const int64_t xB_step = TB ? 1 : args.ldb;
  const int a_lds = sub * G_TILE_BYTES + half * 8192;
  const int b_lds = (2 + sub) * G_TILE_BYTES + half * 8192;
  float* const wave_bias = reinterpret_cast<float*>(smem + 2 * W2_SLOT) + wave * 128;
  unsigned char* const at = smem + 2 * W2_SLOT + wave * W2P_STAGE_BYTES;
  unsigned char* const st = smem + 2 * W2_SLOT + (EPI == EPI_BF16_DTANH ? 4 * W2P_STAGE_BYTES : W2_BIAS_BYTES) + wave * W2P_STAGE_BYTES;

  int vt = (int)blockIdx.x, split_, tile_m, tile_n;
  if (DYN) {
    if (t == 0) {
      const unsigned f = tile_claim(ctr);
      tile_claim_retire(ctr, f, ntiles, xcd);
      lds_store_u32(s_next, f);
    }
    __syncthreads();
    vt = xcd + 8 * (int)lds_load_u32_sync(s_next);
    if (vt >= ntiles) return;     // a workgroup that started late: its queue is empty
    __syncthreads();
  }
  gemm_tile_coords_of(args, vt, split_, tile_m, tile_n);
  const uint16_t *gA_e = baseA_e + (int64_t)tile_m * 256 * args.lda, *gA_o = baseA_o + (int64_t)tile_m * 256 * args.lda;
  const uint16_t *gB_e = baseB_e + (int64_t)tile_n * 256 * xB_step, *gB_o = baseB_o + (int64_t)tile_n * 256 * xB_step;

  f32x16 acc[4][4];
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  u32x4 ra[W2R_D][8];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // vmcnt(8 + the epilogue's stores), lgkmcnt(0): 32 stores for bf16 outputs -> vmcnt(40); 64 for f32 -> more than the counter holds (63): B(1) is
  // then complete by the time the 8 loads behind the stores have issued at all
  // (the x3 split epilogues store 2-4 planes: 64-128 stores, the counter-saturation argument of the f32 modes)
  constexpr int EPI_STORES_WAIT = (EPI == EPI_F32 || EPI == EPI_F32_BIAS_RES || EPI == EPI_BF16_SPLIT || EPI == EPI_BF16_TANH_SPLIT) ? 0xC07F : 0x8078;

#define W2R_APTR(U, KOFF) ((((U) & 1) ? gA_o : gA_e) + ((U) >> 1) * pairA + (KOFF))
#define W2R_BPTR(U, KOFF) ((((U) & 1) ? gB_o : gB_e) + ((U) >> 1) * pairB + (KOFF))
#define W2R_A_LOAD(SET, U, KOFF) ra[SET][U] = ENH_A_NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W2R_APTR(U, KOFF))) : *reinterpret_cast<const u32x4*>(W2R_APTR(U, KOFF))
#define W2R_A_WRITE(SLOT, SET, U) *reinterpret_cast<u32x4*>(smem + (SLOT) * W2_SLOT + a_lds + (U) * 1024 + lane * 16) = ra[SET][U]
#define W2R_A_DMA(SLOT, U, KOFF) __builtin_amdgcn_global_load_lds((const GLB_AS void*)W2R_APTR(U, KOFF), (LDS_AS void*)(smem + (SLOT) * W2_SLOT + a_lds + (U) * 1024), 16, 0, ENH_A_NT ? 2 : ENH_GLDS_AUX)
#define W2R_B_DMA(SLOT, U, KOFF) __builtin_amdgcn_global_load_lds((const GLB_AS void*)W2R_BPTR(U, KOFF), (LDS_AS void*)(smem + (SLOT) * W2_SLOT + b_lds + (U) * 1024), 16, 0, ENH_GLDS_AUX)
  // k16 step 0 of a stage: + fragment reads of k-step 1 ; under every odd MFMA: the A share of the next stage leaves its registers for LDS and the
  // registers are re-used at once for the share three stages further on
#define W2R_K0(FA, FB, RA, RB, RSLOT, ZERO, WSLOT, SET)                                                                           \
  do {                                                                                                                            \
    if (TA || TB) __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    W2_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      if (ZERO) { W2_MMZ(q_, FA, FB); } else { W2_MM(q_, FA, FB); }                                                               \
      if (q_ < 8) { W2_READ_ONE(RA, RB, RSLOT, 1, q_); }                                                                          \
      if (q_ & 1) { W2R_A_WRITE(WSLOT, SET, q_ >> 1); W2R_A_LOAD(SET, q_ >> 1, 0); }                                              \
      W2_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)
#define W2R_K12(FA, FB, RA, RB, RSLOT, RS)                                                                                        \
  do {                                                                                                                            \
    if (TA || TB) __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    W2_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      W2_MM(q_, FA, FB);                                                                                                          \
      if (q_ < 8) { W2_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                                         \
      W2_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)
#define W2R_K3(FA, FB, RA, RB, RSLOT, BSLOT)                                                                                      \
  do {                                                                                                                            \
    if (TA || TB) __builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
    W2_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      W2_MM(q_, FA, FB);                                                                                                          \
      if (q_ < 8) { W2_READ_ONE(RA, RB, RSLOT, 0, q_); }                                                                          \
      if (q_ & 1) { W2R_B_DMA(BSLOT, q_ >> 1, 0); }                                                                               \
      W2_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)
  // one K stage (local index j): SET = (j + 1) % 2 is the register set that holds A(j+1)
#define W2R_STAGE(ZERO, SET, WAIT, HOOK)                                                                                               \
  do {                                                                                                                            \
    const int slot_ = j & 1;                                                                                                      \
    if (j == nst - 1 - W2R_D) { gA_e = baseA_e + offA_next; gA_o = baseA_o + offA_next; }   /* A(j+3) is the next tile's stage 0 */ \
    W2R_K0(fa0, fb0, fa1, fb1, slot_, ZERO, slot_ ^ 1, SET);                                                                      \
    gA_e += G_BK; gA_o += G_BK;                                                                                                   \
    W2R_K12(fa1, fb1, fa0, fb0, slot_, 2);                                                                                        \
    W2R_K12(fa0, fb0, fa1, fb1, slot_, 3);                                                                                        \
    /* B(j+1) landed (only this stage's 8 A loads are younger; after an epilogue its stores are younger too and stay in flight) ; A(j+1) written */ \
    __builtin_amdgcn_s_waitcnt(WAIT);                                                                                             \
    HOOK;                                                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                                 \
    W2_FENCE();                                                                                                                   \
    if (j == nst - 2) { gB_e = baseB_e + offB_next; gB_o = baseB_o + offB_next; }           /* B(j+2) is the next tile's stage 0 */ \
    W2R_K3(fa1, fb1, fa0, fb0, slot_ ^ 1, slot_);                                                                                 \
    gB_e += stageB; gB_o += stageB;                                                                                               \
    ++j;                                                                                                                          \
  } while (0)

  // prologue of the workgroup: A(0) by DMA -> slot 0 | A(1) -> set 1 | B(0) -> slot 0 | A(2) -> set 0 | B(1) -> slot 1   (in THIS order: the
  // wait below leaves the last 16 in flight; fenced, because the scheduler clusters the two register groups by address otherwise)
#pragma unroll
  for (int u = 0; u < 8; ++u) W2R_A_DMA(0, u, 0);
  W2_FENCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2R_A_LOAD(1, u, G_BK);
  W2_FENCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2R_B_DMA(0, u, 0);
  W2_FENCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2R_A_LOAD(0, u, 2 * G_BK);
  W2_FENCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2R_B_DMA(1, u, stageB);
  W2_FENCE();
  gA_e += 3 * G_BK; gA_o += 3 * G_BK;
  gB_e += 2 * stageB; gB_o += 2 * stageB;
  __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16): stage 0 landed (set 0 and B(1) are younger)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 8; ++u) W2_READ_ONE(fa0, fb0, 0, 0, u);
  W2_FENCE();

  // The tile loop is rotated — its body runs stages 1 .. nst-1, the epilogue and the NEXT tile's stage 0 — so that the two forms of stage 0's wait
  // (vmcnt(8) in the workgroup's first tile, vmcnt(8 + stores) after an epilogue) sit on separate paths: with one stage-0 body and a runtime flag the
  // compiler has to assume the permissive wait on the path from the prologue and puts a vmcnt(0) in front of the next stage's register writes.  For the
  // same reason the loop has ONE exit, at its top (a counted loop, no break): the last tile runs a stage 0 of its own re-requested operands for nothing
  // (~1 us per launch) — a mid-loop exit left a never-taken edge from the exit path back into the loop, and with it the same vmcnt(0).
  const int my_tiles = __builtin_amdgcn_readfirstlane((ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
  int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;
  int vnext = vt + (int)gridDim.x;
  if (vnext >= ntiles) vnext = vt;   // the last tile re-requests its own first stages (never consumed; drained before the kernel ends)
  if (!DYN) gemm_tile_coords_of(args, vnext, split_, tile_m, tile_n);
  int64_t offA_next = (int64_t)tile_m * 256 * args.lda, offB_next = (int64_t)tile_n * 256 * xB_step;
  int j = 0;
  W2R_STAGE(true, 1, 0x0078, (void)0);
  // DYN: the loop runs while the tile in hand is real; the claim for the next one is issued at the top of the body (older than everything stage 1
  // counts), returns under stage 1 and is handed round behind stage 1's barrier — two stages before stage nst - 3 needs the next tile's A origin.
  bool more = true;
  for (int it = 0; DYN ? more : it < my_tiles; ++it) {
    unsigned fnext = 0;
    if (DYN && t == 0) fnext = tile_claim(ctr);
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);   // this tile's bias, for the epilogue's LDS strip: requested now, consumed a K loop later
    if ((EPI == EPI_BF16_BIAS_TANH || EPI == EPI_F32_BIAS_RES || EPI == EPI_BF16_TANH_SPLIT) && lane < 32) bias4 = *reinterpret_cast<const float4*>(args.bias + n0 + wn * 128 + lane * 4);
    if (DYN) {
      W2R_STAGE(false, 0, 0x0078, do { if (t == 0) { tile_claim_retire(ctr, fnext, ntiles, xcd); lds_store_u32(s_next, fnext); } __builtin_amdgcn_s_waitcnt(0xC07F); } while (0));
      vnext = xcd + 8 * (int)lds_load_u32_sync(s_next);
      more = vnext < ntiles;
      if (!more) vnext = vt;
      gemm_tile_coords_of(args, vnext, split_, tile_m, tile_n);
      offA_next = (int64_t)tile_m * 256 * args.lda; offB_next = (int64_t)tile_n * 256 * xB_step;
    } else {
      W2R_STAGE(false, 0, 0x0078, (void)0);
    }
    while (j < nst) {
      W2R_STAGE(false, 1, 0x0078, (void)0);
      W2R_STAGE(false, 0, 0x0078, (void)0);
    }
    gemm_epilogue_p<EPI, OT>(args, acc, m0 + wm * 128, n0 + wn * 128, lane, wave_bias, st, at, bias4);   // (the next wait counts this epilogue's stores)
    W2_FENCE();
    vt = vnext;
    m0 = (int64_t)tile_m * 256; n0 = (int64_t)tile_n * 256;
    if (!DYN) {
      vnext = vt + (int)gridDim.x;
      if (vnext >= ntiles) vnext = vt;
      gemm_tile_coords_of(args, vnext, split_, tile_m, tile_n);
      offA_next = (int64_t)tile_m * 256 * args.lda; offB_next = (int64_t)tile_n * 256 * xB_step;
    }
    j = 0;
    W2R_STAGE(true, 1, EPI_STORES_WAIT, (void)0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the dummy requests write LDS / registers: landed before the workgroup's resources are released
#undef W2R_APTR
#undef W2R_BPTR
#undef W2R_A_LOAD
#undef W2R_A_WRITE
#undef W2R_A_DMA
#undef W2R_B_DMA
#undef W2R_K0
#undef W2R_K12
#undef W2R_K3
#undef W2R_STAGE
}
#undef W2_ISSUE_ONE
#undef W2_ADVANCE
#undef W2_READ_ONE
#undef W2_MM
#undef W2_MMZ
#undef W2_FENCE
#undef W2_KSTEP
#undef W2_KSTEP_Z


// =================================================================================================
// launcher: the kernel tables of one operand type
// =================================================================================================
template <typename OT>
void gemm_launch(const GemmArgs& g, const GemmLaunch& L, hipStream_t s) {
  typedef void (*gemm_fn)(const GemmArgs);
  const dim3 grid(L.grid);
  const int layout = (L.trans_a ? 2 : 0) + (L.trans_b ? 1 : 0);
  if (L.family != 7) {
    static const gemm_fn small[2][4] = {
        {gemm_kernel<OT, false, false>, gemm_kernel<OT, false, true>, gemm_kernel<OT, true, false>, gemm_kernel<OT, true, true>},
        {gemm_pipe2_kernel<OT, false, false>, gemm_pipe2_kernel<OT, false, true>, gemm_pipe2_kernel<OT, true, false>, gemm_pipe2_kernel<OT, true, true>}};
    static const bool attr_set = [] {
      for (int f = 0; f < 2; ++f)
        for (int l = 0; l < 4; ++l) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small[f][l]), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G_TILE_BYTES);
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL(small[L.family == 3 ? 1 : 0][layout], grid, dim3(256), (size_t)(4 * G_TILE_BYTES), s, g);
    return;
  }
#define W2_ROW(TA_, TB_) {gemm_w256_kernel<OT, TA_, TB_, EPI_GENERIC>, gemm_w256_kernel<OT, TA_, TB_, EPI_BF16>, gemm_w256_kernel<OT, TA_, TB_, EPI_BF16_BIAS_TANH>, \
                          gemm_w256_kernel<OT, TA_, TB_, EPI_BF16_DTANH>, gemm_w256_kernel<OT, TA_, TB_, EPI_F32_BIAS_RES>, gemm_w256_kernel<OT, TA_, TB_, EPI_F32>,   \
                          gemm_w256_kernel<OT, TA_, TB_, EPI_WS>, gemm_w256_kernel<OT, TA_, TB_, EPI_ATOMIC>}
  static const gemm_fn table[4][EPI_NMODES] = {W2_ROW(false, false), W2_ROW(false, true), W2_ROW(true, false), W2_ROW(true, true)};
#undef W2_ROW
#define W2P_ROW(TB_, D_) {nullptr, gemm_w256p_kernel<OT, false, TB_, EPI_BF16, D_>, gemm_w256p_kernel<OT, false, TB_, EPI_BF16_BIAS_TANH, D_>, gemm_w256p_kernel<OT, false, TB_, EPI_BF16_DTANH, D_>, \
                          gemm_w256p_kernel<OT, false, TB_, EPI_F32_BIAS_RES, D_>, gemm_w256p_kernel<OT, false, TB_, EPI_F32, D_>, nullptr, nullptr}
  static const gemm_fn ptable[4][EPI_NMODES] = {W2P_ROW(false, false), W2P_ROW(true, false), W2P_ROW(false, true), W2P_ROW(true, true)};   // [2 * DYN + TB]
#undef W2P_ROW
#define W2R_ROW(TB_, D_) {nullptr, gemm_w256r_kernel<OT, TB_, EPI_BF16, D_>, gemm_w256r_kernel<OT, TB_, EPI_BF16_BIAS_TANH, D_>, nullptr, nullptr, gemm_w256r_kernel<OT, TB_, EPI_F32, D_>, nullptr, nullptr}
  static const gemm_fn rtable[4][EPI_NMODES] = {W2R_ROW(false, false), W2R_ROW(true, false), W2R_ROW(false, true), W2R_ROW(true, true)};
#undef W2R_ROW
  static const bool w2_attr = [] {
    for (int l = 0; l < 4; ++l)
      for (int e = 0; e < EPI_NMODES; ++e) {
        if (table[l][e]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(table[l][e]), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W2_SLOT + W2_BIAS_BYTES);
        if (ptable[l][e]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ptable[l][e]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    e == EPI_BF16_DTANH ? W2P_LDS_BYTES_DTANH : W2P_LDS_BYTES);
        if (rtable[l][e]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rtable[l][e]), hipFuncAttributeMaxDynamicSharedMemorySize, W2P_LDS_BYTES);
      }
    return true;
  }();
  (void)w2_attr;
  if constexpr (OT::id == 0) if (L.lab) {   // measurement only (enh_debug_gemm_lab): bf16 forms of the split-K weight-gradient loop
#define W2_LAB_GO(L_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w256_lab_kernel<L_>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W2_SLOT + W2_BIAS_BYTES); \
                           gemm_w256_lab_kernel<L_><<<grid, 256, 2 * W2_SLOT + W2_BIAS_BYTES, s>>>(g); } while (0)
    switch (L.lab) {
      case 1: W2_LAB_GO(1); break; case 2: W2_LAB_GO(2); break; case 3: W2_LAB_GO(3); break; case 4: W2_LAB_GO(4); break; case 5: W2_LAB_GO(5); break;
      case 6: W2_LAB_GO(6); break; case 7: W2_LAB_GO(7); break; case 8: W2_LAB_GO(8); break; default: W2_LAB_GO(9); break;
    }
#undef W2_LAB_GO
    return;
  }
  const int prow = 2 * L.dyn + (L.trans_b ? 1 : 0);
  if (L.form == 2) hipLaunchKernelGGL(rtable[prow][L.mode], grid, dim3(256), (size_t)W2P_LDS_BYTES, s, g);
  else if (L.form == 1) hipLaunchKernelGGL(ptable[prow][L.mode], grid, dim3(256), (size_t)(L.mode == EPI_BF16_DTANH ? W2P_LDS_BYTES_DTANH : W2P_LDS_BYTES), s, g);
  else hipLaunchKernelGGL(table[layout][L.mode], grid, dim3(256), (size_t)(2 * W2_SLOT + W2_BIAS_BYTES), s, g);
}
