// attention_v2.hip — round-3 attention kernels: the data layout, MFMA shapes and LDS images of attention.hip (helpers shared through attention_common.h),
// re-scheduled as SOFTWARE-PIPELINED, MFMA-PACED instruction streams.  Reference semantics: enhancing/modules/stage1/layers.py:123-130.
//
// Why (profiles/r02_attention_lab.txt, profiles/r03_attention_lab.txt): the round-1/2 kernels ran at the SUM of their MFMA and vector time (forward: 512 + 940
// cycles per 64-key tile and wave, measured 1390) because inside one wave every product waited for the softmax that waited for the previous product.  Here
// each loop iteration carries INDEPENDENT strands that belong to different tiles, e.g. forward iteration t:
//     S(t+1) = K(t+1) Q^T  [8 MFMA]   |   numerators of tile t  [vector]   |   O += V(t-1)^T P(t-1),  l += 1^T P(t-1)  [8 + 4 MFMA]
// and the source is written as slices of { one MFMA ; ~5 vector ops ; at most two LDS reads } separated by scheduling fences
// (__builtin_amdgcn_sched_barrier(0)), as the w256 GEMM (gemm.hip) is, so that the compiler keeps the interleave.  The vector work was cut to what the
// exponentials need:
//   * the running maximum is only a REFERENCE m_ref: tile t is exponentiated against the m_ref of the tiles before it, and O / l (and the numerators still
//     waiting for their product) are rescaled only when a row's tile maximum exceeds m_ref by more than 2^ATT_THR — a wave-uniform branch taken on the first
//     tile and then almost never.  Numerators stay below 2^ATT_THR, bf16's RELATIVE precision does not depend on that scale, and lse = m_ref + log2(l) is
//     exact either way (cdna_hip_programming.md T13; the ordering hazard it describes is handled by rescaling the pending numerators too);
//   * the row sum l comes from the matrix pipe, l^T[.][q] += ones[.][k] P^T[k][q]: four more MFMAs per tile instead of 32 dependent vector adds, and the
//     normaliser is then the sum of exactly the bf16 numerators that entered P V;
//   * no per-tile multiply of the O accumulator, no serial max / sum chains on the critical path (the maximum of tile t+1 is taken under tile t-1's P V).
// K runs two tiles ahead in a 2-slot LDS ring, V one tile ahead in a 3-slot ring; one barrier per tile; tiles are staged through registers, the global
// loads issued at the top of an iteration and written to LDS at its end.
#include "attention_common.h"

#define A2_FENCE() __builtin_amdgcn_sched_barrier(0)
#define ATT_THR 64.0f
// Pure vector arithmetic has no chain to the scheduling fences: instruction selection emits it where its RESULT is first needed (the numerators of a
// slice all sank below the last MFMA of the iteration).  An empty asm that "modifies" the value pins its computation to this point of the stream.
#define A2_PIN1(a) asm volatile("" : "+v"(a))
#define A2_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))

__device__ __forceinline__ f32x16 f32x16_zero() {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}
// =================================================================================================
// forward
// =================================================================================================
// One pipelined iteration (see the file header).  SC: scores of the current tile (consumed), SN: scores of the next tile (produced), PC: packed numerators
// of the current tile (produced), PP: those of the previous tile (consumed by the P V products).  PV: whether a previous tile exists (compile time).
// PRE: q arrives pre-scaled by scale * log2(e) (the products ARE log2-domain scores) and -m_ref enters as the C operand of each tile's first S product
// (a lane owns one query column: a 16-register block holding -m_ref), so the exponential reads the accumulator directly — no multiply-add per score.
template <bool PV, bool ONES, bool PRE>
__device__ __forceinline__ void fwd2_body(const unsigned char* kbuf, const unsigned char* vbuf, const int lane, const float c, const float m_ref,
                                          const f32x16& negm, const f32x16 (&SC)[2], f32x16 (&SN)[2], s16x8 (&PC)[4], const s16x8 (&PP)[4], f32x16 (&o)[2],
                                          f32x16& lacc, float (&lsum)[2], const s16x8 (&qf)[4], const s16x8& ones, float& mx_next) {
  const int l31 = lane & 31, hi = lane >> 5;
  s16x8 kf[3];
  s16x8 vf[2][2];
  float e[8];
  float mx = -__builtin_inff();
  kf[0] = att_frag_row(kbuf, 0, 0, l31, hi);
  kf[1] = att_frag_row(kbuf, 32, 0, l31, hi);
  A2_FENCE();
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    // ---- the MFMA of this slice: 0-7 the next tile's scores, 8-19 the previous tile's P V (+ row sums) ----
    if (k < 8) {
      const int kb = k & 1, ds = k >> 1;                  // the two 32-key accumulators alternate: no MFMA waits for the one issued just before it
      SN[kb] = (ds == 0) ? (PRE ? MFMA32(kf[k % 3], qf[ds], negm) : MFMA32(kf[k % 3], qf[ds], f32x16_zero())) : MFMA32(kf[k % 3], qf[ds], SN[kb]);
      if (k + 2 < 8) kf[(k + 2) % 3] = att_frag_row(kbuf, ((k + 2) & 1) * 32, (k + 2) >> 1, l31, hi);
    } else if (PV) {
      const int i = (k - 8) / 3, part = (k - 8) % 3;       // P slice i (16 keys); part 0 / 1: d-block 0 / 1, part 2: the row sum
      if (part < 2) o[part] = MFMA32(vf[i & 1][part], PP[i], o[part]);
      else if (ONES) lacc = MFMA32(ones, PP[i], lacc);
    }
    // ---- V^T fragments of P slice j, three slices before its first product ----
    if (PV && k >= 5 && (k - 5) / 3 < 4 && (k - 5) % 3 < 2) {
      const int j = (k - 5) / 3, db = (k - 5) % 3;
      vf[j & 1][db] = att_frag_tr(vbuf, (j >> 1) * 32 + 16 * (j & 1), db, lane);
    }
    // ---- vector work: numerators of the current tile — P slice k / 5, parts 0-3 two exponentials each, part 4 the packing ----
    {
      const int i = k / 5, part = k % 5, kb = i >> 1, r0 = (i & 1) * 8 + 2 * part;
      if (part < 4) {
        e[2 * part] = PRE ? __builtin_amdgcn_exp2f(SC[kb][r0]) : __builtin_amdgcn_exp2f(__builtin_fmaf(SC[kb][r0], c, -m_ref));
        e[2 * part + 1] = PRE ? __builtin_amdgcn_exp2f(SC[kb][r0 + 1]) : __builtin_amdgcn_exp2f(__builtin_fmaf(SC[kb][r0 + 1], c, -m_ref));
        if (!ONES) { lsum[0] += e[2 * part]; lsum[1] += e[2 * part + 1]; A2_PIN2(lsum[0], lsum[1]); }
        A2_PIN2(e[2 * part], e[2 * part + 1]);
      } else {
        u32x4 u = {pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7])};
        A2_PIN1(u);
        PC[i] = __builtin_bit_cast(s16x8, u);
      }
    }
    // ---- maximum of the NEXT tile's scores (complete once the product of slice 7 has retired) ----
    if (k >= 12) {
      const int g = k - 12, kb = g >> 2, r0 = (g & 3) * 4;
      mx = max3(max3(mx, SN[kb][r0], SN[kb][r0 + 1]), SN[kb][r0 + 2], SN[kb][r0 + 3]);
      A2_PIN1(mx);
    }
    A2_FENCE();
  }
  mx_next = mx;
}

// the last tile's P V (nothing left to overlap it with)
template <bool ONES>
__device__ __forceinline__ void fwd2_tail(const unsigned char* vbuf, const int lane, const s16x8 (&PP)[4], f32x16 (&o)[2], f32x16& lacc, const s16x8& ones) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const s16x8 v0 = att_frag_tr(vbuf, (i >> 1) * 32 + 16 * (i & 1), 0, lane);
    const s16x8 v1 = att_frag_tr(vbuf, (i >> 1) * 32 + 16 * (i & 1), 1, lane);
    o[0] = MFMA32(v0, PP[i], o[0]);
    o[1] = MFMA32(v1, PP[i], o[1]);
    if (ONES) lacc = MFMA32(ones, PP[i], lacc);
  }
}

// The pipelined kernel never rescales: the reference m_ref is the exact maximum of the FIRST tile, and a later tile may exceed it by up to 2^ATT_THR
// (fp32 / bf16 share an 8-bit exponent: numerators up to 2^64, row sums up to 2^74 are far from overflow, and the relative precision of every term is
// what it would be against the true maximum).  A workgroup in which some row grows beyond that — logits spanning more than 64 octaves, never seen
// outside the spiked-score tests — raises a flag in LDS, leaves the fast loop at the next barrier and recomputes its 128 queries with the exact
// round-2 loop (attn_fwd_exact).  So the hot loop carries no rescale code at all (in-branch writes to O / the score tiles cost 75 spilled registers).
template <bool ONES, bool PRE>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const uint16_t* __restrict__ qkv, int B, int N, int H, float scale_log2,
                                                           uint16_t* __restrict__ out, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[5][ATT_TILE_BYTES];   // K ring: slots 0, 1 ; V ring: slots 2, 3, 4
  __shared__ int s_bad;
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
  const bool active = q0 < N;   // N % 64 == 0: a wave's 32 queries are all in or all out
  const int qrow = active ? q0 + l31 : l31;
  s16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)qrow * RS + ds * 16 + hi * 8);
  const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const s16x8 ones = __builtin_bit_cast(s16x8, ones_u);

  f32x16 o[2] = {f32x16_zero(), f32x16_zero()};
  f32x16 lacc = f32x16_zero();
  float lsum[2] = {0.f, 0.f};
  f32x16 sA[2], sB[2];
  s16x8 pA[4], pB[4];
  const int nt = N / 64;
  if (t == 0) s_bad = 0;

  // prologue: K(0), V(0), K(1) -> LDS ; S(0) ; m_ref = the first tile's exact maximum
  u32x4 rk[2], rv[2];
  att_gload(rk, Kp, RS, 0, t);
  att_gload(rv, Vp, RS, 0, t);
  att_sstore(rk, smem[0], t);
  att_sstore(rv, smem[2], t);
  if (nt > 1) {
    att_gload(rk, Kp, RS, 64, t);
    att_sstore(rk, smem[1], t);
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) att_pin(qf[ds]);
  ATT_LOOP_ENTRY();
  __syncthreads();
  float m_ref;
  {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        sA[kb] = (ds == 0) ? MFMA32(att_frag_row(smem[0], kb * 32, ds, l31, hi), qf[ds], f32x16_zero()) : MFMA32(att_frag_row(smem[0], kb * 32, ds, l31, hi), qf[ds], sA[kb]);
    float mx = sA[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = max3(mx, sA[kb][r], sA[kb][r + 1]);
    m_ref = PRE ? xhalf_max(mx) : xhalf_max(mx) * scale_log2;
    if (PRE) {         // the loop's tiles get -m_ref through the C operand; this first one is shifted here
#pragma unroll
      for (int r = 0; r < 16; ++r) { sA[0][r] -= m_ref; sA[1][r] -= m_ref; }
    }
  }
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = PRE ? -m_ref : 0.f;

  // iteration kt: global loads of K(kt+2) and V(kt+1) at the top, their LDS stores at the bottom (slots free since the barrier that closed kt-1:
  // K(kt) was last read by the S products of iteration kt-1, V(kt-2) by its P V products)
  int vslot = 0;   // slot of V(kt) in the 3-ring (index into smem[2..4])
  int bad = 0;     // the flag as read after the previous barrier (tested one iteration late: the read never stalls the loop)
#define FWD2_ITER(SC_, SN_, PC_, PP_, FIRST_)                                                                                      \
  do {                                                                                                                             \
    const bool ldk = kt + 2 < nt, ldv = kt + 1 < nt;                                                                               \
    const int vnext = vslot == 2 ? 0 : vslot + 1, vprev = vslot == 0 ? 2 : vslot - 1;                                              \
    if (ldk) att_gload(rk, Kp, RS, (kt + 2) * 64, t);                                                                              \
    if (ldv) att_gload(rv, Vp, RS, (kt + 1) * 64, t);                                                                              \
    const int bad_now = *reinterpret_cast<volatile int*>(&s_bad);                                                                  \
    A2_FENCE();                                                                                                                    \
    float mxn;                                                                                                                     \
    if (FIRST_) fwd2_body<false, ONES, PRE>(smem[(kt + 1) & 1], smem[2 + vprev], lane, scale_log2, m_ref, negm, SC_, SN_, PC_, PP_, o, lacc, lsum, qf, ones, mxn); \
    else fwd2_body<true, ONES, PRE>(smem[(kt + 1) & 1], smem[2 + vprev], lane, scale_log2, m_ref, negm, SC_, SN_, PC_, PP_, o, lacc, lsum, qf, ones, mxn);         \
    A2_FENCE();                                                                                                                    \
    if (ldk) att_sstore(rk, smem[kt & 1], t);                                                                                      \
    if (ldv) att_sstore(rv, smem[2 + vnext], t);                                                                                   \
    vslot = vnext;                                                                                                                 \
    if (ldv) {   /* reference check for tile kt+1 (its scores are in SN_) */                                                       \
      const float tn = xhalf_max(mxn);                                                                                             \
      const float grow = PRE ? tn : tn * scale_log2 - m_ref;                        /* PRE: SN_ is already relative to m_ref */    \
      if (__builtin_amdgcn_ballot_w64(grow > ATT_THR) != 0ull && lane == 0) *reinterpret_cast<volatile int*>(&s_bad) = 1;          \
    }                                                                                                                              \
    bad |= bad_now;                                                                                                                \
    __syncthreads();                                                                                                               \
  } while (0)

  int kt = 0;
  FWD2_ITER(sA, sB, pA, pB, true);
  for (kt = 1; kt + 1 < nt && !bad; kt += 2) {
    FWD2_ITER(sB, sA, pB, pA, false);
    ++kt;
    FWD2_ITER(sA, sB, pA, pB, false);
    --kt;
  }
  if (!bad) {
    if (kt < nt) {            // nt even: one more iteration (an odd tile index: buffers B)
      FWD2_ITER(sB, sA, pB, pA, false);
      fwd2_tail<ONES>(smem[2 + (vslot == 0 ? 2 : vslot - 1)], lane, pB, o, lacc, ones);
    } else {
      fwd2_tail<ONES>(smem[2 + (vslot == 0 ? 2 : vslot - 1)], lane, pA, o, lacc, ones);
    }
  }
#undef FWD2_ITER
  // every wave reads the flag after the same barrier: the decision is workgroup-uniform (flags raised in the last iterations included)
  if (bad | *reinterpret_cast<volatile int*>(&s_bad)) {
    __syncthreads();
    attn_fwd_exact(qkv, B, N, H, scale_log2, out, lse, reinterpret_cast<unsigned char (*)[2][ATT_TILE_BYTES]>(&smem[0][0]), blk, head);
    return;
  }

  float l;
  if (ONES) l = lacc[0];
  else l = xhalf_sum(lsum[0] + lsum[1]);
  const float inv = 1.0f / l;
  if (!active) return;
  uint16_t* op = out + ((int64_t)b * N + q0 + l31) * (H * ATT_D) + h * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      u32x2 w = {pack_bf16x2(o[db][g4 * 4 + 0] * inv, o[db][g4 * 4 + 1] * inv), pack_bf16x2(o[db][g4 * 4 + 2] * inv, o[db][g4 * 4 + 3] * inv)};
      *reinterpret_cast<u32x2*>(op + d0) = w;
    }
  if (hi == 0) lse[((int64_t)b * H + h) * N + q0 + l31] = (m_ref + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
}

// =================================================================================================
// backward: dQ  — pipelined over 32-KEY BLOCKS (half tiles): iteration j carries
//     S(j+1) = K Q^T, dP'(j+1) = V dO^T - delta  [8 MFMA]   |   dS(j) = P(j) o dP'(j), packed  [vector]   |   dQ += K(j-1)^T dS(j-1)  [4 MFMA]
// delta enters as the C operand of the first dP product (a lane owns one query column, so -delta_q is a per-lane constant held in one 16-register block:
// D = A B + C with D != C), which removes the per-element subtraction; the softmax scale of dS is applied once to the finished dQ (exact for d = 64).
// K tiles live in a 3-slot ring (tile t-1 is still read transposed while tile t+1 is read by rows), V in a 2-slot ring; a tile is fetched into registers
// one tile ahead and written to LDS under the next tile's first block; one barrier per tile.
// =================================================================================================
template <bool PREV, bool PRE>
__device__ __forceinline__ void dq2_block(const unsigned char* krow, const unsigned char* vrow, const int rb_next, const unsigned char* kprev, const int rb_prev,
                                          const int lane, const float c, const float lse2, const f32x16& negl, const f32x16& negd, const f32x16& SC, const f32x16& DC,
                                          f32x16& SN, f32x16& DN, s16x8 (&dsC)[2], const s16x8 (&dsP)[2], f32x16 (&dq)[2], const s16x8 (&qf)[4],
                                          const s16x8 (&dof)[4]) {
  const int l31 = lane & 31, hi = lane >> 5;
  s16x8 fr[3];          // row fragments, rolling: slot k uses fr[k % 3]
  s16x8 kt[4];          // K^T fragments of the previous block: (c2, db) = (f >> 1, f & 1)
  unsigned dsw[8];
  fr[0] = att_frag_row(krow, rb_next, 0, l31, hi);
  fr[1] = att_frag_row(vrow, rb_next, 0, l31, hi);
  A2_FENCE();
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if (k < 8) {
      const int ds = k >> 1;
      if ((k & 1) == 0) SN = (ds == 0) ? (PRE ? MFMA32(fr[k % 3], qf[ds], negl) : MFMA32(fr[k % 3], qf[ds], f32x16_zero())) : MFMA32(fr[k % 3], qf[ds], SN);     // S^T[key][q] (- lse[q])
      else DN = (ds == 0) ? MFMA32(fr[k % 3], dof[ds], negd) : MFMA32(fr[k % 3], dof[ds], DN);                       // dP^T[key][q] - delta[q]
      if (k + 2 < 8) fr[(k + 2) % 3] = att_frag_row(((k + 2) & 1) ? vrow : krow, rb_next, (k + 2) >> 1, l31, hi);
    } else if (PREV) {
      const int f = k - 8;
      dq[f & 1] = MFMA32(kt[f], dsP[f >> 1], dq[f & 1]);                                                             // dQ^T[d][q] += K^T dS^T
    }
    if (PREV && k >= 4 && k < 8) kt[k - 4] = att_frag_tr(kprev, rb_prev + 16 * ((k - 4) >> 1), (k - 4) & 1, lane);
    // vector work: pair g of the current block at slices 0, 1, 3, 4, 6, 7, 9, 10
    if (k % 3 != 2) {
      const int g = (k / 3) * 2 + (k % 3), r = 2 * g;
      const float p0 = PRE ? __builtin_amdgcn_exp2f(SC[r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(SC[r], c, -lse2));
      const float p1 = PRE ? __builtin_amdgcn_exp2f(SC[r + 1]) : __builtin_amdgcn_exp2f(__builtin_fmaf(SC[r + 1], c, -lse2));
      dsw[g] = pack_bf16x2(p0 * DC[r], p1 * DC[r + 1]);
      A2_PIN1(dsw[g]);
    }
    A2_FENCE();
  }
  const u32x4 u0 = {dsw[0], dsw[1], dsw[2], dsw[3]}, u1 = {dsw[4], dsw[5], dsw[6], dsw[7]};
  dsC[0] = __builtin_bit_cast(s16x8, u0);
  dsC[1] = __builtin_bit_cast(s16x8, u1);
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o,
                                                              const float* __restrict__ lse, float* __restrict__ delta, int B, int N,
                                                              int H, float scale, float scale_log2, uint16_t* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[5][ATT_TILE_BYTES];   // K ring: slots 0-2 ; V ring: slots 3, 4
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
  const float lse2 = lse[((int64_t)b * H + h) * N + qrow] * 1.4426950408889634f;
  // delta[q] = sum_d dO[q][d] O[q][d] from the dO fragments already held (written for the dK/dV kernel)
  float dpart = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    const s16x8 of = *reinterpret_cast<const s16x8*>(o + ((int64_t)b * N + qrow) * (H * ATT_D) + h * ATT_D + ds * 16 + hi * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) dpart += bf16_bits_to_f32((uint16_t)of[k]) * bf16_bits_to_f32((uint16_t)dof[ds][k]);
  }
  const float del_q = xhalf_sum(dpart);
  if (active && hi == 0) delta[((int64_t)b * H + h) * N + qrow] = del_q;
  f32x16 negd, negl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { negd[r] = -del_q; negl[r] = PRE ? -lse2 : 0.f; }

  f32x16 dq[2] = {f32x16_zero(), f32x16_zero()};
  f32x16 sA, dA, sB, dB;
  s16x8 dsA[2], dsB[2];
  const int nt = N / 64;
  u32x4 rk[2], rv[2];
  att_gload(rk, Kp, RS, 0, t);
  att_gload(rv, Vp, RS, 0, t);
  att_sstore(rk, smem[0], t);
  att_sstore(rv, smem[3], t);
  if (nt > 1) {
    att_gload(rk, Kp, RS, 64, t);
    att_gload(rv, Vp, RS, 64, t);
    att_sstore(rk, smem[1], t);
    att_sstore(rv, smem[4], t);
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) { att_pin(qf[ds]); att_pin(dof[ds]); }
  ATT_LOOP_ENTRY();
  __syncthreads();
  // block (0, 0)
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    sA = (ds == 0) ? MFMA32(att_frag_row(smem[0], 0, ds, l31, hi), qf[ds], negl) : MFMA32(att_frag_row(smem[0], 0, ds, l31, hi), qf[ds], sA);
    dA = (ds == 0) ? MFMA32(att_frag_row(smem[3], 0, ds, l31, hi), dof[ds], negd) : MFMA32(att_frag_row(smem[3], 0, ds, l31, hi), dof[ds], dA);
  }
  int ks = 0;   // K ring slot of tile kt ; the V slot is kt & 1
  for (int kt = 0; kt < nt; ++kt) {
    const int ksn = ks == 2 ? 0 : ks + 1, ksp = ks == 0 ? 2 : ks - 1;
    const unsigned char* kcur = smem[ks];
    const unsigned char* vcur = smem[3 + (kt & 1)];
    // X: current block (kt, 0) in A ; produces (kt, 1) into B ; dQ of the previous tile's second block (its dS is in dsB)
    if (kt == 0) dq2_block<false, PRE>(kcur, vcur, 32, smem[ksp], 32, lane, scale_log2, lse2, negl, negd, sA, dA, sB, dB, dsA, dsB, dq, qf, dof);
    else dq2_block<true, PRE>(kcur, vcur, 32, smem[ksp], 32, lane, scale_log2, lse2, negl, negd, sA, dA, sB, dB, dsA, dsB, dq, qf, dof);
    A2_FENCE();
    if (kt >= 1 && kt + 1 < nt) {          // tile kt+1 (fetched during the previous tile's second block) -> LDS ; slots free since the last barrier
      att_sstore(rk, smem[ksn], t);
      att_sstore(rv, smem[3 + ((kt + 1) & 1)], t);
    }
    __syncthreads();
    if (kt + 2 < nt) {
      att_gload(rk, Kp, RS, (kt + 2) * 64, t);
      att_gload(rv, Vp, RS, (kt + 2) * 64, t);
    }
    A2_FENCE();
    // Y: current block (kt, 1) in B ; produces (kt+1, 0) into A (stale data after the last tile: never used) ; dQ of block (kt, 0) (dS in dsA)
    dq2_block<true, PRE>(smem[ksn], smem[3 + ((kt + 1) & 1)], 0, kcur, 0, lane, scale_log2, lse2, negl, negd, sB, dB, sA, dA, dsB, dsA, dq, qf, dof);
    A2_FENCE();
    ks = ksn;
  }
  {   // dQ of the last block (nt-1, 1): its tile sits in the slot before ks
    const unsigned char* klast = smem[ks == 0 ? 2 : ks - 1];
#pragma unroll
    for (int f = 0; f < 4; ++f) dq[f & 1] = MFMA32(att_frag_tr(klast, 32 + 16 * (f >> 1), f & 1, lane), dsB[f >> 1], dq[f & 1]);
  }
  if (!active) return;
  uint16_t* op = dqkv + ((int64_t)b * N + q0 + l31) * RS + h * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      u32x2 w = {pack_bf16x2(dq[db][g4 * 4 + 0] * scale, dq[db][g4 * 4 + 1] * scale), pack_bf16x2(dq[db][g4 * 4 + 2] * scale, dq[db][g4 * 4 + 3] * scale)};
      *reinterpret_cast<u32x2*>(op + d0) = w;
    }
}

// launchers used by attention.hip's C ABI (enh_attention_set_kernel selects the family)
void attn_fwd2_launch(const uint16_t* qkv, int B, int N, int H, float scale_log2, uint16_t* out, float* lse, bool ones, bool pre, dim3 grid, hipStream_t s) {
  if (ones) {
    if (pre) attn_fwd2_kernel<true, true><<<grid, 256, 0, s>>>(qkv, B, N, H, scale_log2, out, lse);
    else attn_fwd2_kernel<true, false><<<grid, 256, 0, s>>>(qkv, B, N, H, scale_log2, out, lse);
  } else {
    if (pre) attn_fwd2_kernel<false, true><<<grid, 256, 0, s>>>(qkv, B, N, H, scale_log2, out, lse);
    else attn_fwd2_kernel<false, false><<<grid, 256, 0, s>>>(qkv, B, N, H, scale_log2, out, lse);
  }
}
void attn_bwd_dq2_launch(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, float* delta, int B, int N, int H, float scale,
                         float scale_log2, uint16_t* dqkv, bool pre, dim3 grid, hipStream_t s) {
  if (pre) attn_bwd_dq2_kernel<true><<<grid, 256, 0, s>>>(qkv, o, d_o, lse, delta, B, N, H, scale, scale_log2, dqkv);
  else attn_bwd_dq2_kernel<false><<<grid, 256, 0, s>>>(qkv, o, d_o, lse, delta, B, N, H, scale, scale_log2, dqkv);
}
