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

__device__ __forceinline__ void att_gload(u32x4 (&r)[2], const uint16_t* __restrict__ base, int64_t rs, int row0, int t) {
  const int c = t & 7, r0 = t >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) r[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + r0 + 32 * i) * rs + c * 8);
}
__device__ __forceinline__ void att_sstore(const u32x4 (&r)[2], unsigned char* tile, int t) {
  const int c = t & 7, r0 = t >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(tile + att_off(r0 + 32 * i, c)) = r[i];
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
__device__ __forceinline__ s16x8 pack8_bf16(const float* p) {
  u32x4 u = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
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
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

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

