// attention_common.h — helpers shared by the attention kernels (attention.hip: round-1/2 kernels; attention_v2.hip: the software-pipelined round-3 kernels):
// the swizzled LDS image of a [64][64] bf16 tile, register staging, MFMA operand fragments (row and transpose-read) and the workgroup -> (block, head) map.
#pragma once
#include "common.h"

#define ATT_D 64
#define ATT_TILE_BYTES 8192  // 64 rows x 64 bf16

// LDS image of a [64 rows][64 d] bf16 tile: 16-B chunk c (0..7) of row r at r*128 + ((c ^ f(r)) << 4) with
// f(r) = ((r>>1)&1)<<2 | (r>>2)&3 : conflict-free for ds_write_b128 (staging), the 32-row ds_read_b128
// fragments and the 4-row x 64-B transpose reads (gfx950 bank model; tools/lds_bank_check.py).
__device__ __forceinline__ int att_off(int r, int c) { return r * 128 + ((c ^ ((((r >> 1) & 1) << 2) | ((r >> 2) & 3))) << 4); }

// rows row0 .. row0+63 of a [*][rs] bf16 matrix, 2 x 16 B per thread.  The address is split into a WAVE-UNIFORM part (base + row0 * rs: scalar registers,
// advanced by the scalar unit) and a 32-bit per-lane byte offset that does not depend on the tile (hoisted out of the loop): the loads then use the
// scalar-base + vector-offset form and cost no vector instructions per tile (the 64-bit per-lane address arithmetic was 13 of dK/dV's ~170 vector
// instructions per tile, and the kernels' time follows that count: profiles/r03_attention_lab.txt).
__device__ __forceinline__ void att_gload(u32x4 (&r)[2], const uint16_t* __restrict__ base, int64_t rs, int row0, int t) {
  const int c = t & 7, r0 = t >> 3;
  const unsigned char* ub = reinterpret_cast<const unsigned char*>(base + (int64_t)row0 * rs);
  const unsigned lane_off = (unsigned)(r0 * (int)rs + c * 8) * 2u;
#pragma unroll
  for (int i = 0; i < 2; ++i) r[i] = *reinterpret_cast<const u32x4*>(ub + (size_t)(unsigned)(i * 32 * (int)rs * 2) + lane_off);
}
__device__ __forceinline__ void att_sstore(const u32x4 (&r)[2], unsigned char* tile, int t) {
  const int c = t & 7, r0 = t >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(tile + att_off(r0 + 32 * i, c)) = r[i];
}
// The same tile image written by LDS-DMA (global_load_lds, 16 B per lane): no staging registers, no ds_write pass.  The LDS destination of a wave
// instruction is lane-linear (base + lane * 16 B: 8 rows x 128 B), so the swizzle is applied to the lane's SOURCE address: slot p = lane & 7 of row r must
// receive chunk c = p ^ f(r).  Wave w of the four stages rows 16 w .. 16 w + 15 (two instructions per operand and tile).  The caller waits vmcnt(0) before the
// barrier that publishes the tile.
__device__ __forceinline__ unsigned att_dma_lane_off(int rs_elems, int lane, int i) {   // byte offset of this lane's 16 source bytes inside 8-row group i (0 | 1) of a wave's 16 rows
  const int rr = lane >> 3, p = lane & 7, r = rr + 8 * i;      // f(r) of att_off depends on r & 15 only (bits 1..3), and the wave's row base is a multiple of 16
  const int c = p ^ ((((r >> 1) & 1) << 2) | ((r >> 2) & 3));
  return (unsigned)(rr * rs_elems + c * 8) * 2u;
}
__device__ __forceinline__ void att_dma_tile(const uint16_t* __restrict__ base, int64_t rs, int row0, unsigned char* tile, int wave, unsigned lane_off0, unsigned lane_off1) {
  const unsigned char* ub = reinterpret_cast<const unsigned char*>(base + (int64_t)(row0 + wave * 16) * rs);
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)(ub + lane_off0), (LDS_AS void*)(tile + (wave * 16) * 128), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)(ub + (size_t)(unsigned)(8 * (int)rs * 2) + lane_off1), (LDS_AS void*)(tile + (wave * 16 + 8) * 128), 16, 0, 0);
}
// 32x32x16 operand fragment, rows = tile rows rb + (lane&31), k = d: ds*16 + hi*8 + 0..7
__device__ __forceinline__ s16x8 att_frag_row(const unsigned char* tile, int rb, int ds, int l31, int hi) {
  return *reinterpret_cast<const s16x8*>(tile + att_off(rb + l31, ds * 2 + hi));
}
// 32x32x16 operand fragment contracted over TILE ROWS: index = column cb*32 + (lane&31); k-slot (hi, j) is tile
// row rbase + 8*(j>>2) + 4*hi + (j&3)  — exactly the rows a lane holds in accumulator registers 8*c2 + j of a
// 32x32 C tile whose row block starts at rbase - 16*c2 (so P / dS go from registers to the next MFMA unmoved).
__device__ __forceinline__ s16x8 att_frag_tr(const unsigned char* tile, int rbase, int cb, int lane) {
  const int G = lane >> 4, s = lane & 15;
  const int row = rbase + 4 * (G >> 1) + (s >> 2);
  const int cch = cb * 4 + (G & 1) * 2 + ((s & 3) >> 1);
  const int sub = (s & 1) * 8;
  const s16x4 lo = lds_tr_read_b64(tile + att_off(row, cch) + sub);
  const s16x4 hi = lds_tr_read_b64(tile + att_off(row + 8, cch) + sub);
  s16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}
template <typename OT>
__device__ __forceinline__ s16x8 pack8(const float* p) {
  u32x4 u = {pack2<OT>(p[0], p[1]), pack2<OT>(p[2], p[3]), pack2<OT>(p[4], p[5]), pack2<OT>(p[6], p[7])};
  return __builtin_bit_cast(s16x8, u);
}
// Every global load of a kernel's prologue (Q / dO / K / V fragments, statistics) is waited for HERE, before the tile loop: the compiler's wait-count
// pass merges the loop-entry state into the loop body, so a fragment load still pending at entry made it wait, in EVERY iteration, for the oldest of
// the tile prefetch loads issued a few instructions earlier (s_waitcnt vmcnt(3) .. vmcnt(0) in front of the first MFMAs: the whole L2 latency exposed
// once per key tile — found in round 3 by reading the ISA).
#define ATT_LOOP_ENTRY() do { __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_sched_barrier(0); } while (0)
// pins a fragment loaded in the prologue: the value must be IN its registers at this point (IR-level sinking otherwise moves the load into the loop
// preheader, behind ATT_LOOP_ENTRY, and the pending-at-entry state is back)
__device__ __forceinline__ void att_pin(s16x8& f) {
  u32x4 u = __builtin_bit_cast(u32x4, f);
  asm volatile("" : "+v"(u));
  f = __builtin_bit_cast(s16x8, u);
}
// max(a, b, c) in ONE instruction (v_max3_f32).  Written as nested maxima the COMPILER sees: the attention objects are built with -fno-honor-nans, so no
// operand is canonicalised first (v_max_f32 x, x, x: seven instructions for a 4-way maximum, cdna_hip_programming.md "Fused attention" pitfalls) and the
// backend fuses the pair into v_max3_f32.  Round 4 used inline asm for this — and inline asm is invisible to the hazard recognizer: the first v_max3
// behind the S products read the MFMA's destination registers BEFORE the matrix pipe had written them (no s_nop in between), i.e. a stale maximum.  Any
// reference maximum gives a valid softmax, so every parity test passed; but the result bits changed from launch to launch (tools/attn_det_probe.py,
// found in round 5 by the graph-replay bit-identity test).  tests/test_isa_lint.py now refuses inline-asm VALU in the attention kernels' hot loops.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// combine a per-lane value with the other half-wave's (lane ^ 32) through one v_permlane32_swap (no LDS round trip).  Verified semantics
// (profiles/hw_probe_r01.txt P4): with both operands = v, every lane receives (v[lane & 31], v[(lane & 31) + 32]).
__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

#define ATT_FENCE() __builtin_amdgcn_sched_barrier(0)
// (the kernels are templates over the operand type tag OT = BF16 | F16, common.h: `OT` must name it where this macro is used)
#define MFMA32(a, b, c) mfma32<OT>((a), (b), (c))

// Workgroup -> (block-within-head, head) mapping.  Hardware places workgroup L on XCD L % 8 (each XCD has a private L2), and the nblk
// workgroups of one (batch, head) all stream the SAME K/V (or Q/dO) — so they are given ids that are congruent mod 8 and adjacent in
// dispatch order: the head's 256 KiB of K/V is then fetched into ONE L2 and re-used there, instead of once per XCD (8x the fabric traffic).
__device__ __forceinline__ bool att_block_coords(int nblk, int n_heads_total, int& blk, int& head) {
  const int L = blockIdx.x;
  const int xcd = L & 7, r = L >> 3;
  head = (r / nblk) * 8 + xcd;
  blk = r % nblk;
  return head < n_heads_total;
}


// The round-1/2 forward pass of one 128-query block (exact running maximum, O rescaled every tile): the body of attn_fwd_kernel, and the FALLBACK of the
// pipelined kernel of attention_v2.hip for a workgroup whose scores outgrow its fixed reference.  smem: [2][2][ATT_TILE_BYTES] ([stage][K | V]).
template <typename OT>
__device__ __forceinline__ void attn_fwd_exact(const uint16_t* __restrict__ qkv, int B, int N, int H, float scale_log2, uint16_t* __restrict__ out,
                                               float* __restrict__ lse, unsigned char (*smem)[2][ATT_TILE_BYTES], int blk, int head) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = head / H, h = head - b * H;
  const int q0 = blk * 128 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;

  const bool active = q0 < N;  // N % 64 == 0: a wave's 32 queries are all in or all out
  const int qrow = active ? q0 + l31 : l31;
  s16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)qrow * RS + ds * 16 + hi * 8);

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -__builtin_inff(), l_part = 0.f;

  const int nt = N / 64;
  u32x4 rk[2], rv[2];
  att_gload(rk, Kp, RS, 0, t);
  att_gload(rv, Vp, RS, 0, t);
  att_sstore(rk, smem[0][0], t);
  att_sstore(rv, smem[0][1], t);
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) att_pin(qf[ds]);
  ATT_LOOP_ENTRY();
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nt) {
      att_gload(rk, Kp, RS, (kt + 1) * 64, t);
      att_gload(rv, Vp, RS, (kt + 1) * 64, t);
    }
    const unsigned char* kt_ = smem[st][0];
    const unsigned char* vt_ = smem[st][1];
    // ---- S^T[key][q] = K Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) s[kb] = MFMA32(att_frag_row(kt_, kb * 32, ds, l31, hi), qf[ds], s[kb]);
    }
    // ---- online softmax for this lane's query column ----
    // Round 4 (the kernel runs at the speed of its VECTOR instruction stream, profiles/r04_attention_lab.txt): the row maximum through v_max3 (16
    // instructions instead of 32 v_max + canonicalisations) and one v_permlane32_swap instead of an LDS round trip; and the running maximum is a
    // REFERENCE that is raised — O and l rescaled, a wave-uniform branch — only on the first tile and when some row's tile maximum exceeds it by more
    // than 2^8: the 32 multiplies of O per tile are gone in all but a handful of tiles (numerators stay below 2^8; bf16's relative precision does not
    // depend on that scale, lse = m + log2(l) is exact either way; cdna_hip_programming.md T13: no product is pending across the rescale here).
    float mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
    mx = max3(mx, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, s[1][r], s[1][r + 1]);
    mx = xhalf_max(__builtin_fmaxf(mx, s[1][15]));
    const float mt = mx * scale_log2;
    if (kt == 0 || __builtin_amdgcn_ballot_w64(mt - m_run > 8.0f) != 0) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_part *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    float p[2][16];
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] * scale_log2 - m_run);
        psum += p[kb][r];
      }
    l_part += psum;
    // ---- O^T[d][q] += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const s16x8 pb = pack8<OT>(&p[kb][c2 * 8]);
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = MFMA32(att_frag_tr(vt_, kb * 32 + 16 * c2, db, lane), pb, o[db]);
      }
    if (kt + 1 < nt) {
      att_sstore(rk, smem[st ^ 1][0], t);
      att_sstore(rv, smem[st ^ 1][1], t);
    }
    __syncthreads();
  }
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
  if (hi == 0) lse[((int64_t)b * H + h) * N + q0 + l31] = (m_run + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
}

