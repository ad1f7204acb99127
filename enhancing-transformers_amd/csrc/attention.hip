// attention.hip — fused (flash-style) multi-head attention forward / backward for gfx950, d_head = 64.
//
// Replaces Attention.forward between to_qkv and to_out (reference enhancing/modules/stage1/layers.py:123-130):
// chunk(3) + 'b n (h d) -> b h n d' + softmax(q k^T * 64^-0.5) v + 'b h n d -> b n (h d)'.  The kernels read the
// packed [B, N, 3*H*64] bf16 QKV GEMM output directly and write the [B, N, H*64] layout to_out consumes, so
// neither einops rearrange nor the N x N score matrix (50 MB fp32 per image per layer at base) ever touches HBM;
// only the row log-sum-exp is saved for backward.
//
// MFMA: v_mfma_f32_32x32x16_bf16 with the product computed "swapped" (S^T = K Q^T): each lane then owns ONE
// query column, so the online-softmax row statistics are lane-local (one cross-half exchange) and the
// probabilities feed the second MFMA straight from registers (the k-slot permutation of the C layout is applied
// to the other operand instead).  Operands contracted over their slow storage index (V in P V, K in dS K,
// dO / Q in the dK / dV products) are read from LDS with ds_read_b64_tr_b16.
//
// Forward / dQ: workgroup = 128 queries (4 waves x 32), K/V streamed in 64-key tiles through a 2-stage LDS ring.
// dK/dV: workgroup = 128 keys (4 waves x 32), Q / dO streamed in 64-query tiles.  dQ and dK/dV are separate
// kernels (S is recomputed twice) so that no atomics are needed and the result is deterministic.
#include "attention_common.h"

// =================================================================================================
// forward
// =================================================================================================
template <typename OT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, int B, int N, int H, float scale_log2,
                                                          uint16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ATT_TILE_BYTES];  // [stage][K | V]
  int blk, head;
  if (!att_block_coords((N + 127) / 128, B * H, blk, head)) return;
  attn_fwd_exact<OT>(qkv, B, N, H, scale_log2, out, lse, smem, blk, head);
}

// Round 5 — the forward for PRE-SCALED q (the training path's convention: the products are log2-domain scores), with the vector stream cut where the
// round-4 anatomy says the kernel's time is (profiles/r04_attention_lab.txt: it runs at the speed of its vector instructions):
//   * -m_ref rides in the MFMA C operand of the first S product (a lane owns ONE query column, so -m_ref is a per-lane constant in a 16-register block,
//     rewritten only when the reference is raised — a handful of tiles per row): the exponential reads the accumulator directly, the 32 multiply-subtracts
//     per tile are gone (what the dQ / dK/dV kernels do with -lse);
//   * the row sum runs on float pairs (16 v_pk_add_f32 per tile instead of 32 adds), still exact f32: lse keeps its 5e-8.  (Summing the PACKED bf16
//     numerators with v_dot2c_f32_bf16 against (1, 1) — also 16 instructions — measured the same time and moved lse to 2e-5: not adopted.)
// Same skeleton, LDS images and results layout as attn_fwd_exact; the exact running-reference semantics are kept (no fallback path).
__device__ __forceinline__ float dot2_ones(uint32_t pk, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, 0x3f803f80u), acc, false);
}
template <typename OT>
__global__ __launch_bounds__(256, 2) void attn_fwd_pre_kernel(const uint16_t* __restrict__ qkv, int B, int N, int H, uint16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ATT_TILE_BYTES];  // [stage][K | V]
  int blk, head;
  if (!att_block_coords((N + 127) / 128, B * H, blk, head)) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = head / H, h = head - b * H;
  const int q0 = blk * 128 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;
  const bool active = q0 < N;
  const int qrow = active ? q0 + l31 : l31;
  s16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)qrow * RS + ds * 16 + hi * 8);
  f32x16 o[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
  float m_ref = 0.f;                   // the first tile's scores are taken against 0 and re-based below (kt == 0)
  f32x2 l2[2] = {{0.f, 0.f}, {0.f, 0.f}};

  const int nt = N / 64;
  // K / V tiles by LDS-DMA (round 5): no staging registers, no ds_write pass; the next tile is requested at the top of a tile into the stage the last barrier freed
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ko0 = att_dma_lane_off((int)RS, lane, 0), ko1 = att_dma_lane_off((int)RS, lane, 1);
  att_dma_tile(Kp, RS, 0, smem[0][0], wave_u, ko0, ko1);
  att_dma_tile(Vp, RS, 0, smem[0][1], wave_u, ko0, ko1);
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) att_pin(qf[ds]);
  ATT_LOOP_ENTRY();
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nt) {
      att_dma_tile(Kp, RS, (kt + 1) * 64, smem[st ^ 1][0], wave_u, ko0, ko1);
      att_dma_tile(Vp, RS, (kt + 1) * 64, smem[st ^ 1][1], wave_u, ko0, ko1);
    }
    const unsigned char* kt_ = smem[st][0];
    const unsigned char* vt_ = smem[st][1];
    // (Round 5: issuing the tile's fragment reads ahead of their use in a fenced order — all eight K fragments at once, one V fragment behind every S
    // product, counted lgkmcnt waits instead of the compiler's read / wait(0) / MFMA chains — needs 168 registers (three waves per SIMD instead of four)
    // and measured no faster than the round-4 kernel; the compiler's serial form at four waves is the fastest of the three: profiles/r05_attention_lab.txt §6.)
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) s[kb] = MFMA32(att_frag_row(kt_, kb * 32, ds, l31, hi), qf[ds], ds == 0 ? negm : s[kb]);   // S^T[key][q] - m_ref[q]
    // four independent v_max3 chains of depth 4 (one dependent chain of 16 leaves the in-order wave waiting on its own previous instruction)
    float mq[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x16& sv = s[c >> 1];
      const int r0 = (c & 1) * 8;
      mq[c] = max3(sv[r0], sv[r0 + 1], sv[r0 + 2]);
      mq[c] = max3(mq[c], sv[r0 + 3], sv[r0 + 4]);
      mq[c] = max3(mq[c], sv[r0 + 5], sv[r0 + 6]);
    }
    float mx = max3(mq[0], mq[1], s[0][7]);
    mx = max3(mx, mq[2], s[0][15]);
    mx = max3(mx, mq[3], s[1][7]);
    mx = xhalf_max(__builtin_fmaxf(mx, s[1][15]));       // this tile's row maximum RELATIVE to the reference
    if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > 8.0f) != 0) {   // wave-uniform: the reference is raised on the first tile and when outgrown by 2^8
      const float delta = kt == 0 ? mx : __builtin_fmaxf(mx, 0.f);
      if (kt != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l2[0] *= alpha; l2[1] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
      m_ref += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[0][r] -= delta; s[1][r] -= delta; negm[r] = -m_ref; }
    }
    // ---- numerators, packed; row sum of the packed values; O^T[d][q] += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        float p8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = __builtin_amdgcn_exp2f(s[kb][c2 * 8 + j]);
        const u32x4 pk = {pack2<OT>(p8[0], p8[1]), pack2<OT>(p8[2], p8[3]), pack2<OT>(p8[4], p8[5]), pack2<OT>(p8[6], p8[7])};
#pragma unroll
        for (int j = 0; j < 4; ++j) l2[j & 1] += (f32x2){p8[2 * j], p8[2 * j + 1]};      // exact f32 row sum, two lanes of one v_pk_add_f32 (lse stays exact to f32), two chains
        const s16x8 pb = __builtin_bit_cast(s16x8, pk);
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(att_frag_tr(vt_, kb * 32 + 16 * c2, db, lane), pb, o[db]);
      }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's share of the next tile has landed
    __syncthreads();
  }
  const float l_part = (l2[0][0] + l2[1][0]) + (l2[0][1] + l2[1][1]);
  const float l = l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.0f / l;
  if (!active) return;
  uint16_t* op = out + ((int64_t)b * N + q0 + l31) * (H * ATT_D) + h * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      u32x2 w = {pack2<OT>(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv), pack2<OT>(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv)};
      *reinterpret_cast<u32x2*>(op + d0) = w;
    }
  if (hi == 0) lse[((int64_t)b * H + h) * N + q0 + l31] = (m_ref + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
}

// =================================================================================================
// backward: dQ  (same skeleton as forward; K tile is read both as rows and transposed)
// =================================================================================================
// MODE 0: the round-2 arithmetic.  MODE 1: -delta enters as the C operand of the first dP product (a lane owns ONE query column, so -delta_q is a
// per-lane constant kept in a 16-register block; D = A B + C with D != C): 32 subtractions per tile gone.  MODE 2 (q pre-scaled by scale*log2e, i.e.
// the products are already log2-domain scores): -lse enters the S product the same way and the exponential reads the accumulator directly — no
// vector arithmetic left but exp, the P o dP' multiply and the bf16 packing.  The kernels are vector-ISSUE bound (profiles/r03_attention_lab.txt).
template <int MODE, typename OT>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o,
                                                             const float* __restrict__ lse, float* __restrict__ delta, int B, int N,
                                                             int H, float scale, float scale_log2, uint16_t* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ATT_TILE_BYTES];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int blk, head;
  if (!att_block_coords((N + 127) / 128, B * H, blk, head)) return;
  const int b = head / H, h = head - b * H;
  const int q0 = blk * 128 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;
  const uint16_t* dOp = d_o + (int64_t)b * N * (H * ATT_D) + h * ATT_D;

  const bool active = q0 < N;
  const int qrow = active ? q0 + l31 : l31;
  s16x8 qf[4], dof[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qf[ds] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)qrow * RS + ds * 16 + hi * 8);
    dof[ds] = *reinterpret_cast<const s16x8*>(dOp + (int64_t)qrow * (H * ATT_D) + ds * 16 + hi * 8);
  }
  const float lse_q = lse[((int64_t)b * H + h) * N + qrow] * 1.4426950408889634f;
  // delta[q] = sum_d dO[q][d] * O[q][d]: a lane already holds half of its query's dO row as MFMA fragments, so the row dot product is 32 products per
  // lane and one cross-half exchange here — and is WRITTEN for the dK/dV kernel that runs next (this replaced a separate pass over O and dO per layer)
  float dpart = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    const s16x8 of = *reinterpret_cast<const s16x8*>(o + ((int64_t)b * N + qrow) * (H * ATT_D) + h * ATT_D + ds * 16 + hi * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) dpart += unpack1<OT>((uint16_t)of[k]) * unpack1<OT>((uint16_t)dof[ds][k]);
  }
  const float del_q = dpart + __shfl_xor(dpart, 32, 64);
  if (active && hi == 0) delta[((int64_t)b * H + h) * N + qrow] = del_q;
  f32x16 negd, negl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { negd[r] = MODE >= 1 ? -del_q : 0.f; negl[r] = MODE == 2 ? -lse_q : 0.f; }

  f32x16 dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const int nt = N / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // K / V tiles by LDS-DMA (see attn_fwd_pre_kernel)
  const unsigned ko0 = att_dma_lane_off((int)RS, lane, 0), ko1 = att_dma_lane_off((int)RS, lane, 1);
  att_dma_tile(Kp, RS, 0, smem[0][0], wave_u, ko0, ko1);
  att_dma_tile(Vp, RS, 0, smem[0][1], wave_u, ko0, ko1);
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) { att_pin(qf[ds]); att_pin(dof[ds]); }
  ATT_LOOP_ENTRY();
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nt) {
      att_dma_tile(Kp, RS, (kt + 1) * 64, smem[st ^ 1][0], wave_u, ko0, ko1);
      att_dma_tile(Vp, RS, (kt + 1) * 64, smem[st ^ 1][1], wave_u, ko0, ko1);
    }
    const unsigned char* kt_ = smem[st][0];
    const unsigned char* vt_ = smem[st][1];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        s = MFMA32(att_frag_row(kt_, kb * 32, ds, l31, hi), qf[ds], ds == 0 ? negl : s);     // S^T[key][q]  (- lse[q] in MODE 2)
        dp = MFMA32(att_frag_row(vt_, kb * 32, ds, l31, hi), dof[ds], ds == 0 ? negd : dp);  // dP^T[key][q] = V dO^T  (- delta[q] in MODE >= 1)
      }
      float dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = MODE == 2 ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse_q);
        dsv[r] = MODE >= 1 ? pr * dp[r] : pr * (dp[r] - del_q);     // (the factor `scale` of dS is applied once to the finished dQ)
      }
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const s16x8 dsb = pack8<OT>(&dsv[c2 * 8]);
#pragma unroll
        for (int db = 0; db < 2; ++db) dq[db] = MFMA32(att_frag_tr(kt_, kb * 32 + 16 * c2, db, lane), dsb, dq[db]);  // dQ^T[d][q] += K^T dS^T
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's share of the next tile has landed
    __syncthreads();
  }
  if (!active) return;
  uint16_t* op = dqkv + ((int64_t)b * N + q0 + l31) * RS + h * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      u32x2 w = {pack2<OT>(dq[db][g4 * 4 + 0] * scale, dq[db][g4 * 4 + 1] * scale), pack2<OT>(dq[db][g4 * 4 + 2] * scale, dq[db][g4 * 4 + 3] * scale)};
      *reinterpret_cast<u32x2*>(op + d0) = w;
    }
}

// =================================================================================================
// backward: dK, dV  (workgroup owns 128 keys; Q / dO tiles stream through LDS)
// =================================================================================================
// CINIT: -delta (and, with PRE — q pre-scaled by scale*log2e — also -lse) enter as the C operands of the first dP / S products: the statistics are
// loaded from LDS straight into the accumulator registers (the same four 16-byte reads per block as before, no extra registers), which removes the
// per-element subtraction (and the scale-and-subtract before the exponential).  kscale: the factor of the finished dK (scale, or ln 2 with PRE).
template <bool CINIT, bool PRE, typename OT>
#ifndef ATT_DKV_DMA
#define ATT_DKV_DMA 1
#endif
#ifndef ATT_DKV_OCC
#define ATT_DKV_OCC 2
#endif
// (round 5: forcing three waves per SIMD here — __launch_bounds__(256, 3), 168 registers — spills 18 registers into the tile loop and costs +20 % on the backward;
// four waves on the forward, 38 spills, doubles its time: profiles/r05_attention_lab.txt §5.  The occupancy these kernels have is the one their live set allows.)
__global__ __launch_bounds__(256, ATT_DKV_OCC) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta, int B, int N,
                                                              int H, float scale, float scale_log2, uint16_t* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ATT_TILE_BYTES];  // [stage][Q | dO]
  __shared__ __attribute__((aligned(16))) float s_stat[2][2][64];                   // [stage][lse*log2e | delta]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int blk, head;
  if (!att_block_coords((N + 127) / 128, B * H, blk, head)) return;
  const int b = head / H, h = head - b * H;
  const int key0 = blk * 128 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const int64_t OS = (int64_t)H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;
  const uint16_t* dOp = d_o + (int64_t)b * N * OS + h * ATT_D;
  const float* lsep = lse + ((int64_t)b * H + h) * N;
  const float* delp = delta + ((int64_t)b * H + h) * N;

  const bool active = key0 < N;
  const int krow = active ? key0 + l31 : l31;
  s16x8 kf[4], vf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    kf[ds] = *reinterpret_cast<const s16x8*>(Kp + (int64_t)krow * RS + ds * 16 + hi * 8);
    vf[ds] = *reinterpret_cast<const s16x8*>(Vp + (int64_t)krow * RS + ds * 16 + hi * 8);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const int nt = N / 64;
#if ATT_DKV_DMA
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned qo0 = att_dma_lane_off((int)RS, lane, 0), qo1 = att_dma_lane_off((int)RS, lane, 1);
  const unsigned do0 = att_dma_lane_off((int)OS, lane, 0), do1 = att_dma_lane_off((int)OS, lane, 1);
  float rstat = 0.f;
  att_dma_tile(Qp, RS, 0, smem[0][0], wave_u, qo0, qo1);
  att_dma_tile(dOp, OS, 0, smem[0][1], wave_u, do0, do1);
#else
  u32x4 rq[2], rd[2];
  float rstat = 0.f;
  att_gload(rq, Qp, RS, 0, t);
  att_gload(rd, dOp, OS, 0, t);
#endif
  // statistics of the 64 queries of a tile: threads 0-63 fetch lse (kept in the log2 domain), 64-127 delta — through ONE select-addressed load in the
  // straight-line code.  The former `if (t < 64) .. else if (t < 128) ..` put each load in its own divergent block, and the wait-count pass closed
  // that block with s_waitcnt vmcnt(0): every iteration waited for the Q / dO prefetch issued just before it (found in the ISA, round 3).
  const float* statp = ((t & 64) ? delp : lsep) + (t & 63);
  const float stat_mul = (t & 64) ? (CINIT ? -1.0f : 1.0f) : ((CINIT && PRE) ? -1.4426950408889634f : 1.4426950408889634f);   // stored negated where they are C operands
  rstat = statp[0];
#if !ATT_DKV_DMA
  att_sstore(rq, smem[0][0], t);
  att_sstore(rd, smem[0][1], t);
#endif
  if (t < 128) s_stat[0][t >> 6][t & 63] = rstat * stat_mul;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) { att_pin(kf[ds]); att_pin(vf[ds]); }
  ATT_LOOP_ENTRY();
  __syncthreads();
  for (int qt = 0; qt < nt; ++qt) {
    const int st = qt & 1;
    if (qt + 1 < nt) {
#if ATT_DKV_DMA
      att_dma_tile(Qp, RS, (qt + 1) * 64, smem[st ^ 1][0], wave_u, qo0, qo1);      // the other stage is free since the barrier that closed tile qt - 1
      att_dma_tile(dOp, OS, (qt + 1) * 64, smem[st ^ 1][1], wave_u, do0, do1);
#else
      att_gload(rq, Qp, RS, (qt + 1) * 64, t);
      att_gload(rd, dOp, OS, (qt + 1) * 64, t);
#endif
      rstat = statp[(qt + 1) * 64];                 // (scaled when it is stored, after the tile's arithmetic: nothing here waits for the load)
    }
    const unsigned char* qt_ = smem[st][0];
    const unsigned char* dot_ = smem[st][1];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp, lrow;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {           // the statistics of the 16 query rows this lane holds (rows 8 g4 + 4 hi + 0..3 of the block)
        const int row0 = qb * 32 + 8 * g4 + 4 * hi;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&s_stat[st][0][row0]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&s_stat[st][1][row0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lrow[g4 * 4 + k] = l4[k]; dp[g4 * 4 + k] = d4[k]; }
      }
      const f32x16 drow = dp;
      if (!(CINIT && PRE)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
      } else {
        s = lrow;                               // -lse (log2 domain) as the C operand
      }
      if (!CINIT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        s = MFMA32(att_frag_row(qt_, qb * 32, ds, l31, hi), kf[ds], s);      // S[q][key]  (- lse[q])
        dp = MFMA32(att_frag_row(dot_, qb * 32, ds, l31, hi), vf[ds], dp);   // dP[q][key] = dO V^T  (- delta[q])
      }
      float pv[16], dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = (CINIT && PRE) ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * scale_log2 - lrow[r]);
        dsv[r] = CINIT ? pv[r] * dp[r] : pv[r] * (dp[r] - drow[r]);      // the factor of dS is applied once to the finished dK
      }
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const s16x8 pa = pack8<OT>(&pv[c2 * 8]);
        const s16x8 dsa = pack8<OT>(&dsv[c2 * 8]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv[db] = MFMA32(att_frag_tr(dot_, qb * 32 + 16 * c2, db, lane), pa, dv[db]);  // dV^T[d][key] += dO^T P
          dk[db] = MFMA32(att_frag_tr(qt_, qb * 32 + 16 * c2, db, lane), dsa, dk[db]);  // dK^T[d][key] += Q^T dS
        }
      }
    }
    if (qt + 1 < nt) {
#if !ATT_DKV_DMA
      att_sstore(rq, smem[st ^ 1][0], t);
      att_sstore(rd, smem[st ^ 1][1], t);
#endif
      if (t < 128) s_stat[st ^ 1][t >> 6][t & 63] = rstat * stat_mul;
    }
#if ATT_DKV_DMA
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's share of the next tile has landed in LDS
#endif
    __syncthreads();
  }
  if (!active) return;
  // D^T[d][key]: lane (key = key0 + l31, hi) holds d = db*32 + 8*(r>>2) + 4*hi + (r&3): four consecutive d per register group -> 8-byte stores
  uint16_t* dkp = dqkv + ((int64_t)b * N + key0 + l31) * RS + H * ATT_D + h * ATT_D;
  uint16_t* dvp = dkp + H * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      const u32x2 wk = {pack2<OT>(dk[db][g4 * 4 + 0] * scale, dk[db][g4 * 4 + 1] * scale), pack2<OT>(dk[db][g4 * 4 + 2] * scale, dk[db][g4 * 4 + 3] * scale)};
      const u32x2 wv = {pack2<OT>(dv[db][g4 * 4 + 0], dv[db][g4 * 4 + 1]), pack2<OT>(dv[db][g4 * 4 + 2], dv[db][g4 * 4 + 3])};
      *reinterpret_cast<u32x2*>(dkp + d0) = wk;
      *reinterpret_cast<u32x2*>(dvp + d0) = wv;
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
// kernel family per pass (explicit state behind an explicit call, as enh_gemm_set_kernel); 0 = the library's choice:
//   forward: 1 round-2 kernel | 5 round-2 skeleton with -m_ref as the MFMA C operand, packed row sum, K / V by LDS-DMA (round 5; pre-scaled q only, else family 1)
//   dQ     : 1 round-2 arithmetic | 3 -delta (and, pre-scaled q, -lse) as MFMA C operands
//   dK/dV  : 1 round-2 arithmetic | 2 -delta (and, pre-scaled q, -lse) as MFMA C operands
// (forward 2 / 3, dQ 2: the software-pipelined round-3 kernels; forward 4, dK/dV 3: the eight-wave antiphase kernels of round 4 — all measured slower and deleted;
//  their lab notes stay in profiles/r03_attention_lab.txt, r04_attention_lab.txt.)
static int g_att_fwd = 0, g_att_dq = 0, g_att_dkv = 0;
#define ATT_DEFAULT_FWD 5
#define ATT_DEFAULT_DQ 3
#define ATT_DEFAULT_DKV 2

extern "C" int enh_attention_set_kernel(int fwd, int dq, int dkv) {
  ENH_REQUIRE((fwd == 0 || fwd == 1 || fwd == 5) && (dq == 0 || dq == 1 || dq == 3) && dkv >= 0 && dkv <= 2, ENH_E_BADARG,
              "enh_attention_set_kernel: fwd in {0, 1, 5}, dq in {0, 1, 3}, dkv in 0..2");
  g_att_fwd = fwd; g_att_dq = dq; g_att_dkv = dkv;
  return ENH_OK;
}

#define ATT_LOG2E 1.4426950408889634f
#define ATT_LN2 0.6931471805599453f

extern "C" int enh_attention_forward(const enh_h16* qkv, int B, int N, int H, float scale, int q_prescaled, enh_h16* out, float* lse, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_attention_forward");
  ENH_REQUIRE(qkv && out && lse, ENH_E_BADARG, "enh_attention_forward: null pointer");
  ENH_REQUIRE(B > 0 && H > 0 && N > 0 && N % 64 == 0, ENH_E_SHAPE, "enh_attention_forward: need N %% 64 == 0 (B=%d N=%d H=%d)", B, N, H);
  ENH_REQUIRE(scale > 0.f, ENH_E_BADARG, "enh_attention_forward: scale must be positive");
  const int64_t nblk = (N + 127) / 128, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));  // 1-D: see att_block_coords
  int fam = g_att_fwd ? g_att_fwd : ATT_DEFAULT_FWD;
  if (fam == 5 && !q_prescaled) fam = 1;                         // -m_ref as a C operand needs log2-domain products
  const float sl2 = q_prescaled ? 1.0f : scale * ATT_LOG2E;       // pre-scaled q: the products are log2-domain scores already
  if (fam == 5) ENH_DT_DISPATCH(dtype, (attn_fwd_pre_kernel<OT><<<grid, 256, 0, (hipStream_t)stream>>>(qkv, B, N, H, out, lse)));
  else ENH_DT_DISPATCH(dtype, (attn_fwd_kernel<OT><<<grid, 256, 0, (hipStream_t)stream>>>(qkv, B, N, H, sl2, out, lse)));
  return enh_check_launch("enh_attention_forward");
}

extern "C" int enh_attention_backward(const enh_h16* qkv, const enh_h16* out, const enh_h16* dout, const float* lse, int B, int N,
                                      int H, float scale, int q_prescaled, enh_h16* dqkv, float* delta_ws, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_attention_backward");
  ENH_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws, ENH_E_BADARG, "enh_attention_backward: null pointer");
  ENH_REQUIRE(B > 0 && H > 0 && N > 0 && N % 64 == 0, ENH_E_SHAPE, "enh_attention_backward: need N %% 64 == 0 (B=%d N=%d H=%d)", B, N, H);
  ENH_REQUIRE(scale > 0.f, ENH_E_BADARG, "enh_attention_backward: scale must be positive");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = (N + 127) / 128, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));
  const bool pre = q_prescaled != 0;
  const float sl2 = pre ? 1.0f : scale * ATT_LOG2E;
  // dQ is the gradient with respect to the UNSCALED q in both conventions (what the projection's weight / input gradients need): factor `scale`.
  // dK is formed from the q tile as stored: with pre-scaled q' = q * scale * log2e the factor is scale / (scale * log2e) = ln 2.
  const float kscale = pre ? ATT_LN2 : scale;
  // (either dQ kernel also writes delta_ws = rowsum(dO * O) for the dK/dV kernel that follows)
  const int fq = g_att_dq ? g_att_dq : ATT_DEFAULT_DQ;
  const int fk = g_att_dkv ? g_att_dkv : ATT_DEFAULT_DKV;
  if (fq == 1) ENH_DT_DISPATCH(dtype, (attn_bwd_dq_kernel<0, OT><<<grid, 256, 0, s>>>(qkv, out, dout, lse, delta_ws, B, N, H, scale, sl2, dqkv)));
  else if (pre) ENH_DT_DISPATCH(dtype, (attn_bwd_dq_kernel<2, OT><<<grid, 256, 0, s>>>(qkv, out, dout, lse, delta_ws, B, N, H, scale, sl2, dqkv)));
  else ENH_DT_DISPATCH(dtype, (attn_bwd_dq_kernel<1, OT><<<grid, 256, 0, s>>>(qkv, out, dout, lse, delta_ws, B, N, H, scale, sl2, dqkv)));
  if (fk == 1) ENH_DT_DISPATCH(dtype, (attn_bwd_dkv_kernel<false, false, OT><<<grid, 256, 0, s>>>(qkv, dout, lse, delta_ws, B, N, H, kscale, sl2, dqkv)));
  else if (pre) ENH_DT_DISPATCH(dtype, (attn_bwd_dkv_kernel<true, true, OT><<<grid, 256, 0, s>>>(qkv, dout, lse, delta_ws, B, N, H, kscale, sl2, dqkv)));
  else ENH_DT_DISPATCH(dtype, (attn_bwd_dkv_kernel<true, false, OT><<<grid, 256, 0, s>>>(qkv, dout, lse, delta_ws, B, N, H, kscale, sl2, dqkv)));
  return enh_check_launch("enh_attention_backward");
}
