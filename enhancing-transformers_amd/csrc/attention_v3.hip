// attention_v3.hip — round-4 attention forward: EIGHT waves per workgroup in two groups that run in ANTIPHASE (the structure of
// MI355X_MICROARCH.md "Two waves per SIMD").  Reference semantics: enhancing/modules/stage1/layers.py:123-130; data layout, MFMA shapes and LDS images are
// those of attention.hip (attention_common.h).
//
// Why.  The round-2/3 kernels (four waves per workgroup, two or three workgroups per CU) spend, per SIMD, about the SUM of their matrix time
// (32 clk x #MFMA) and their vector time (profiles/r03_attention_lab.txt §3): inside a wave every product waits for the softmax that waits for the product
// before it, and the waves that share a SIMD belong to different workgroups and drift through those phases at random.  Re-ordering inside a wave
// (attention_v2.hip) cannot help.  Here the two waves that share a SIMD are PARTNERS: every wave alternates a pure vector segment (the softmax of
// tile i: row maximum, 32 exponentials, row sum, packing; plus its share of the tile staging) with a pure matrix segment (O += V(i)^T P(i) and
// S(i+1) = K(i+1) Q^T: 16 MFMAs and their fragment reads), the partner is half a period behind, and an s_barrier separates the segments — so a SIMD
// always has one wave feeding the matrix pipe while the other issues vector instructions:
//
//     half-step        2i              2i+1            2i+2            2i+3
//     group A      softmax(i)     PV(i), S(i+1)    softmax(i+1)    PV(i+1), S(i+2)
//     group B     PV(i-1), S(i)    softmax(i)      PV(i), S(i+1)    softmax(i+1)
//
// The vector segment is cut to what the exponentials need (as attention_v2.hip): q arrives pre-scaled by scale * log2(e) and -m_ref rides in the MFMA C
// operand, so a score is exponentiated straight from the accumulator; m_ref is a REFERENCE maximum that is only raised (O, l rescaled, a wave-uniform
// branch) when a row's tile maximum exceeds it by more than 2^8 — on the first tile and then almost never.
// LDS: K and V each in a 2-slot ring of 8-KiB tiles (32 KiB).  The matrix segment touches no memory: its sixteen operand fragments (V(i), K(i+1)) are
// fetched in the VECTOR segment in front of it (half-step 2i for group A, 2i+1 for B), so nothing in it waits for LDS (the first version read them in
// the matrix segment itself and ran 13 % slower than the four-wave kernel: with one wave per SIMD in that segment nobody covers a fragment's latency).
// Group A writes K(i+2) in its vector segment 2i (from registers loaded one period earlier) into the slot whose tile K(i) was last fetched in half-step
// 2i-1; group B writes V(i+1) in half-step 2i+1 into the slot of V(i-1), last fetched in 2i-1: every slot is rewritten at least one barrier after its
// last reader and read at least one barrier after its writer.
// Workgroup = 256 queries of one (image, head); needs N % 256 == 0 (the launcher falls back to the four-wave kernel otherwise).
#include "attention_common.h"

#define A3_THR 8.0f
// segment boundary: this wave's LDS traffic is performed, then all eight waves meet.  (Raw barrier: __syncthreads() would also wait for the tile
// prefetch that is meant to stay in flight across the boundary.)  The empty asm keeps IR-level passes from moving LDS accesses across it.
#define A3_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_s_waitcnt(0xC07F);       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");           \
  } while (0)


// Which of the two groups a wave joins is decided by WHERE the hardware put it: the two waves that share a SIMD must be in different groups, or the
// antiphase is between SIMDs instead of inside each (measured with the static rule "waves 0-3 / 4-7": every half-step took 2 x 16 MFMAs — both
// waves of a SIMD were in their matrix segment together, profiles/r04_attention_lab.txt).  Each wave reads its SIMD id (HW_ID bits 5:4) and takes a
// ticket from a per-SIMD LDS counter: ticket 0 -> group A, 1 -> group B, rank inside the group = SIMD id.  If the placement is not two waves on each
// of four SIMDs the static rule is used (same results, no antiphase).
__device__ __forceinline__ void a3_join_groups(int* s_cnt /* [4] LDS, zeroed here */, int t, int wave, int& grp, int& rank) {
  if (t < 4) s_cnt[t] = 0;
  __syncthreads();
  const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));      // HW_REG_HW_ID, offset 4, size 2
  int ticket = 0;
  if ((t & 63) == 0) ticket = atomicAdd(&s_cnt[simd], 1);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  __syncthreads();
  const bool ok = s_cnt[0] == 2 && s_cnt[1] == 2 && s_cnt[2] == 2 && s_cnt[3] == 2;
  grp = ok ? ticket : wave >> 2;
  rank = ok ? simd : wave & 3;
}

#define A3_PIN1(x) asm volatile("" : "+v"(x))
#define A3_PIN16(x) asm volatile("" : "+v"(x))
#define A3_PIN4(x) do { u32x4 u_ = __builtin_bit_cast(u32x4, (x)); asm volatile("" : "+v"(u_)); (x) = __builtin_bit_cast(s16x8, u_); } while (0)

// fragments of the NEXT matrix segment, read in the vector segment in front of it (the tiles they come from were written at least one barrier earlier):
// V^T of tile i for O += V^T P(i), K rows of tile i+1 for S(i+1) — 16 fragments, 64 registers; the matrix segment itself touches no memory
template <bool KNEXT>
__device__ __forceinline__ void a3_fetch(const unsigned char* kbuf, const unsigned char* vbuf, int lane, s16x8 (&vf)[4][2], s16x8 (&kf)[4][2]) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int db = 0; db < 2; ++db) vf[i][db] = att_frag_tr(vbuf, (i >> 1) * 32 + 16 * (i & 1), db, lane);
  if (KNEXT) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kf[ds][kb] = att_frag_row(kbuf, kb * 32, ds, l31, hi);
  }
}
__device__ __forceinline__ void a3_fetch_k(const unsigned char* kbuf, int lane, s16x8 (&kf)[4][2]) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) kf[ds][kb] = att_frag_row(kbuf, kb * 32, ds, l31, hi);
}

// matrix segment: O += V^T P (8 MFMAs, skipped for the segment in front of tile 0) and the next tile's scores S = K Q^T + (-m_ref) (8 MFMAs), all operands
// in registers: sixteen MFMAs back to back
template <bool PV, bool SNEXT, bool PRE>
__device__ __forceinline__ void a3_matrix(const s16x8 (&vf)[4][2], const s16x8 (&kf)[4][2], const s16x8 (&qf)[4], const s16x8 (&p)[4], const f32x16& negm,
                                          f32x16 (&o)[2], f32x16 (&s)[2]) {
  if (PV) {
#pragma unroll
    for (int i = 0; i < 4; ++i)            // P slice i = keys 16 i .. 16 i + 15 of the tile
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db] = MFMA32(vf[i][db], p[i], o[db]);
  }
  if (SNEXT) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {      // the two 32-key accumulators alternate: no MFMA waits for the one issued just before it
        if (ds == 0) {
          if (PRE) s[kb] = MFMA32(kf[ds][kb], qf[ds], negm);
          else {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            s[kb] = MFMA32(kf[ds][kb], qf[ds], z);
          }
        } else s[kb] = MFMA32(kf[ds][kb], qf[ds], s[kb]);
      }
  }
}

// vector segment: numerators of the tile whose scores are in s (relative to m_ref when PRE), row statistics, packing
template <bool PRE>
__device__ __forceinline__ void a3_softmax(bool first, float sl2, f32x16 (&s)[2], s16x8 (&p)[4], f32x16& negm, float& m_ref, float& l_part, f32x16 (&o)[2]) {
  // row maximum of the tile (this lane's 32 keys, then the other half-wave's)
  float mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
  mx = max3(mx, s[0][15], s[1][0]);
#pragma unroll
  for (int r = 1; r < 15; r += 2) mx = max3(mx, s[1][r], s[1][r + 1]);
  mx = __builtin_fmaxf(mx, s[1][15]);
  mx = xhalf_max(mx);
  if (!PRE) mx = mx * sl2 - m_ref;          // scores arrive raw: log2-domain and relative to the reference from here on
  // raise the reference only when some row outgrew it by more than 2^THR (always on the first tile: the reference starts at 0, not at the maximum)
  if (first || __builtin_amdgcn_ballot_w64(mx > A3_THR) != 0) {
    const float d = first ? mx : __builtin_fmaxf(mx, 0.f);
    const float alpha = __builtin_amdgcn_exp2f(-d);
    m_ref += d;
    l_part *= alpha;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    if (PRE) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] -= d;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
    }
  }
  float e[32];
  float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      e[kb * 16 + r] = PRE ? __builtin_amdgcn_exp2f(s[kb][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sl2, -m_ref));
      e[kb * 16 + r + 1] = PRE ? __builtin_amdgcn_exp2f(s[kb][r + 1]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r + 1], sl2, -m_ref));
      ls0 += e[kb * 16 + r];
      ls1 += e[kb * 16 + r + 1];
    }
  l_part += ls0 + ls1;
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = pack8_bf16(&e[i * 8]);
}

// TRACE (measurement aid, enh_debug_attention_fwd3_trace): waves 0 and 4 of workgroup 0 record s_memtime at the four edges of every period
// (vector segment start | at the first barrier | released | at the second barrier) into trace[wave >> 2][i][4]
template <bool PRE, bool TRACE>
__global__ __launch_bounds__(512, 2) void attn_fwd3_kernel(const uint16_t* __restrict__ qkv, int B, int N, int H, float sl2, uint16_t* __restrict__ out,
                                                           float* __restrict__ lse, unsigned long long* __restrict__ trace) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ATT_TILE_BYTES];  // [K | V][slot]
  int blk, head;
  if (!att_block_coords(N / 256, B * H, blk, head)) return;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  __shared__ int s_cnt[4];
  int grp, rank;
  a3_join_groups(s_cnt, t, wave, grp, rank);
  const int tg = rank * 64 + lane;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = head / H, h = head - b * H;
  const int q0 = blk * 256 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;
  const uint16_t* Sp = grp == 0 ? Kp : Vp;          // the operand this group stages (A: K, B: V)
  unsigned char(*sbuf)[ATT_TILE_BYTES] = smem[grp];
  const int nt = N / 64;

  s16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)(q0 + l31) * RS + ds * 16 + hi * 8);
  // staging: group A keeps K two tiles ahead of the softmax (K(i+2) goes to LDS in vector segment i), group B keeps V one tile ahead — so that in
  // vector segment i BOTH operands of the following matrix segment (K(i+1), V(i)) are already in LDS and their fragments can be fetched there
  const int ahead = grp == 0 ? 2 : 1;
  u32x4 rs[2];
  att_gload(rs, Sp, RS, 0, tg);                      // tile 0 of this group's operand -> slot 0
  att_sstore(rs, sbuf[0], tg);
  if (grp == 0 && nt > 1) {                          // K(1) -> slot 1
    att_gload(rs, Sp, RS, 64, tg);
    att_sstore(rs, sbuf[1], tg);
  }
  if (ahead < nt) att_gload(rs, Sp, RS, ahead * 64, tg);      // K(2) / V(1) stay in registers until the group's first vector segment
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) att_pin(qf[ds]);

  f32x16 o[2], s[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; s[0][r] = 0.f; s[1][r] = 0.f; }
  s16x8 p[4], vf[4][2], kf[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p[i] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
    vf[i][0] = p[i]; vf[i][1] = p[i]; kf[i][0] = p[i]; kf[i][1] = p[i];
  }
  float m_ref = 0.f, l_part = 0.f;
  A3_BARRIER();                                        // K(0), K(1), V(0) are in LDS
  // Both groups run the SAME instruction stream  [ vector(i) | barrier | matrix(i) | barrier ]  — group B one barrier behind group A (its lead-in barrier
  // here, group A's trailing one after the loop), which is what puts the partners of a SIMD in antiphase.  No group-dependent branch inside the loop:
  // with one, the compiler merged the two variants through register copies of the in-flight tile prefetch and waited for it in the matrix segment.
  a3_fetch_k(smem[0][0], lane, kf);
  a3_matrix<false, true, PRE>(vf, kf, qf, p, negm, o, s);      // S(0), both groups at once (the only un-phased segment)
  A3_PIN16(s[0]); A3_PIN16(s[1]);
  A3_BARRIER();                                        // everybody has read K(0): its slot may take K(2)
  if (grp == 1) A3_BARRIER();
  const bool tr_on = TRACE && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0;
#define A3_STAMP(K) do { if (TRACE && tr_on) trace[((wave >> 2) * nt + i) * 4 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
  for (int i = 0; i < nt; ++i) {
    const int cur = i & 1, nxt = cur ^ 1;
    A3_STAMP(0);
    // ---- vector segment: the fragments of the matrix segment that follows (V(i), K(i+1): written at least one barrier ago), the softmax of tile i,
    //      this group's staged tile from registers to LDS, the next one requested ----
    a3_fetch<true>(smem[0][nxt], smem[1][cur], lane, vf, kf);       // (last tile: the K slot holds an old tile and the scores made from it are never used —
                                                                    //  cheaper than a second code variant whose register assignment the loop has to reconcile)
    a3_softmax<PRE>(i == 0, sl2, s, p, negm, m_ref, l_part, o);
    if (i + ahead < nt) att_sstore(rs, sbuf[(i + ahead) & 1], tg);
    if (i + ahead + 1 < nt) att_gload(rs, Sp, RS, (i + ahead + 1) * 64, tg);
    A3_PIN4(p[0]); A3_PIN4(p[1]); A3_PIN4(p[2]); A3_PIN4(p[3]);      // (pure arithmetic otherwise sinks to its first use: into the matrix segment)
    A3_PIN1(l_part); A3_PIN1(m_ref); A3_PIN16(negm); A3_PIN16(o[0]); A3_PIN16(o[1]);
    A3_STAMP(1);
    A3_BARRIER();
    A3_STAMP(2);
    // ---- matrix segment: O += V(i)^T P(i), S(i+1) = K(i+1) Q^T - m_ref: sixteen MFMAs on registers ----
    a3_matrix<true, true, PRE>(vf, kf, qf, p, negm, o, s);
    A3_PIN16(o[0]); A3_PIN16(o[1]); A3_PIN16(s[0]); A3_PIN16(s[1]);
    A3_STAMP(3);
    A3_BARRIER();
  }
#undef A3_STAMP
  if (grp == 0) A3_BARRIER();

  const float l = xhalf_sum(l_part);
  const float inv = 1.0f / l;
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

void attn_fwd3_launch(const uint16_t* qkv, int B, int N, int H, float scale_log2, uint16_t* out, float* lse, bool pre, hipStream_t s) {
  const int64_t nblk = N / 256, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));
  if (pre) attn_fwd3_kernel<true, false><<<grid, 512, 0, s>>>(qkv, B, N, H, scale_log2, out, lse, nullptr);
  else attn_fwd3_kernel<false, false><<<grid, 512, 0, s>>>(qkv, B, N, H, scale_log2, out, lse, nullptr);
}

extern "C" int enh_debug_attention_fwd3_trace(const enh_bf16* qkv, int B, int N, int H, enh_bf16* out, float* lse, unsigned long long* trace, void* stream) {
  ENH_REQUIRE(qkv && out && lse && trace && N % 256 == 0, ENH_E_BADARG, "enh_debug_attention_fwd3_trace: bad arguments");
  const int64_t nblk = N / 256, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));
  attn_fwd3_kernel<true, true><<<grid, 512, 0, (hipStream_t)stream>>>(qkv, B, N, H, 1.0f, out, lse, trace);
  return enh_check_launch("enh_debug_attention_fwd3_trace");
}

// =================================================================================================
// backward: dK, dV — the same antiphase structure.  Workgroup = 256 keys of one (image, head): eight waves x 32 keys, K / V fragments of the wave's keys in
// registers for the whole sweep; Q / dO stream through LDS in 32-QUERY blocks (three-slot rings of 4-KiB tiles), and a wave alternates
//     vector segment i :  P = exp2(S(i)) , dS = P o dP(i) , both packed to bf16                                  [16 exponentials, 16 products, 16 packs]
//     matrix segment i :  dV += dO(i)^T P , dK += Q(i)^T dS  (8 MFMAs) ; S(i+1) = Q(i+1) K^T - lse , dP(i+1) = dO(i+1) V^T - delta  (8 MFMAs)
// with its SIMD partner half a period behind.  The four-wave kernel of attention.hip runs this work at two waves per SIMD and takes, per wave and 64-query
// tile, the SUM of its matrix time (32 MFMAs = 1024 clk) and its vector time (~1100 clk): 4400 wave cycles measured (profiles/r03_attention_lab.txt §3).
// 32-query phases (instead of the 64-query tiles of the other kernels) keep the scores that cross a barrier at 32 registers per product.
// Pre-scaled q only (the products are log2-domain scores; -lse and -delta ride in the MFMA C operand): the launcher keeps the four-wave kernel otherwise.
// Staging: group A writes Q(i+2) and the statistics of block i+2, group B writes dO(i+2), each in its vector segment i, into the slot of block i-1, whose
// last readers were the vector segments i-1; three slots per ring.
// =================================================================================================
#define A3Q_TILE_BYTES 4096   // 32 queries x 64 d, bf16
// one 32-row tile through 256 threads: 16 B per thread
__device__ __forceinline__ u32x4 a3q_gload(const uint16_t* __restrict__ base, int64_t rs, int row0, int tg) {
  const int c = tg & 7, r = tg >> 3;
  return *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + r) * rs + c * 8);
}
__device__ __forceinline__ void a3q_sstore(const u32x4& v, unsigned char* tile, int tg) {
  *reinterpret_cast<u32x4*>(tile + att_off(tg >> 3, tg & 7)) = v;
}

__global__ __launch_bounds__(512, 2) void attn_bwd_dkv3_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, int B, int N, int H, float kscale, uint16_t* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][3][A3Q_TILE_BYTES];   // [Q | dO][slot]
  __shared__ __attribute__((aligned(16))) float s_stat[3][2][32];                     // [slot][-lse * log2e | -delta]
  int blk, head;
  if (!att_block_coords(N / 256, B * H, blk, head)) return;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  __shared__ int s_cnt[4];
  int grp, rank;
  a3_join_groups(s_cnt, t, wave, grp, rank);
  const int tg = rank * 64 + lane;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = head / H, h = head - b * H;
  const int key0 = blk * 256 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D, OS = (int64_t)H * ATT_D;
  const uint16_t* Qp = qkv + (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kp = Qp + H * ATT_D;
  const uint16_t* Vp = Kp + H * ATT_D;
  const uint16_t* dOp = d_o + (int64_t)b * N * OS + h * ATT_D;
  const float* lsep = lse + ((int64_t)b * H + h) * N;
  const float* delp = delta + ((int64_t)b * H + h) * N;
  // this group's staged operand: A streams Q (row stride RS) and the statistics, B streams dO (row stride OS); both keep their operand TWO blocks ahead
  // (block i+2 goes to LDS in vector segment i), so that in vector segment i every fragment of the matrix segment that follows — the transposed
  // fragments of block i and the row fragments and statistics of block i+1 — is already in LDS and is fetched THERE: the matrix segment is sixteen MFMAs on
  // registers.  (Fetched inside the matrix segment they cost it their LDS latency with nobody to cover it: 1.0 ms instead of 0.89 for the four-wave kernel.)
  const uint16_t* Sp = grp == 0 ? Qp : dOp;
  const int64_t Ss = grp == 0 ? RS : OS;
  unsigned char(*sbuf)[A3Q_TILE_BYTES] = smem[grp];
  const int nt = N / 32;
  // statistics: threads 0-31 of group A fetch lse (stored as -lse * log2e), 32-63 delta (stored negated)
  const bool stat_thr = grp == 0 && tg < 64;
  const float* statp = (tg & 32) ? delp : lsep;
  const float stat_mul = (tg & 32) ? -1.0f : -1.4426950408889634f;

  s16x8 kf[4], vf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    kf[ds] = *reinterpret_cast<const s16x8*>(Kp + (int64_t)(key0 + l31) * RS + ds * 16 + hi * 8);
    vf[ds] = *reinterpret_cast<const s16x8*>(Vp + (int64_t)(key0 + l31) * RS + ds * 16 + hi * 8);
  }
  // prologue: blocks 0 and 1 of both operands (+ statistics) -> LDS; block 2 stays in registers for the first vector segment
  u32x4 rs = a3q_gload(Sp, Ss, 0, tg);
  a3q_sstore(rs, sbuf[0], tg);
  float rstat = 0.f;
  if (stat_thr) s_stat[0][tg >> 5][tg & 31] = statp[tg & 31] * stat_mul;
  if (nt > 1) {
    rs = a3q_gload(Sp, Ss, 32, tg);
    a3q_sstore(rs, sbuf[1], tg);
    if (stat_thr) s_stat[1][tg >> 5][tg & 31] = statp[32 + (tg & 31)] * stat_mul;
  }
  if (nt > 2) {
    rs = a3q_gload(Sp, Ss, 64, tg);
    if (stat_thr) rstat = statp[64 + (tg & 31)];
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) { att_pin(kf[ds]); att_pin(vf[ds]); }

  f32x16 dk[2], dv[2], s, dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
  s16x8 pa[2], dsa[2], qt[2][2], dt[2][2], qr[4], dr[4];      // packed P / dS ; transposed fragments of Q / dO (block i) ; row fragments (block i+1)
  pa[0] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0}; pa[1] = pa[0]; dsa[0] = pa[0]; dsa[1] = pa[0];
  A3_BARRIER();

  // statistics of a block into the accumulators (the C operands of its score products): this lane's 16 query rows are 8 g4 + 4 hi + 0..3
#define A3Q_STATS(SLOT)                                                                                    \
  do {                                                                                                     \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                     \
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(&s_stat[SLOT][0][8 * g4 + 4 * hi]);                 \
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(&s_stat[SLOT][1][8 * g4 + 4 * hi]);                 \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) { s[g4 * 4 + k] = l4[k]; dp[g4 * 4 + k] = d4[k]; }     \
    }                                                                                                      \
  } while (0)
#define A3Q_ROWFRAGS(SLOT)                                                                                 \
  do {                                                                                                     \
    _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) {                                                     \
      qr[ds] = att_frag_row(smem[0][SLOT], 0, ds, l31, hi);                                                \
      dr[ds] = att_frag_row(smem[1][SLOT], 0, ds, l31, hi);                                                \
    }                                                                                                      \
  } while (0)
  // S = Q K^T + (-lse), dP = dO V^T + (-delta)
#define A3Q_SCORES()                                                                                       \
  do {                                                                                                     \
    _Pragma("unroll") for (int ds = 0; ds < 4; ++ds) {                                                     \
      s = MFMA32(qr[ds], kf[ds], s);                                                                       \
      dp = MFMA32(dr[ds], vf[ds], dp);                                                                     \
    }                                                                                                      \
  } while (0)

  A3Q_STATS(0);
  A3Q_ROWFRAGS(0);
  A3Q_SCORES();                                         // block 0, both groups at once (the only un-phased segment)
  A3_PIN16(s); A3_PIN16(dp);
  if (grp == 1) A3_BARRIER();
  int cur = 0;                                          // slot of block i
  for (int i = 0; i < nt; ++i) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    const int stg = nxt == 2 ? 0 : nxt + 1;             // slot of block i + 2
    // ---- vector segment: P and dS of block i, packed; the operand fragments of the matrix segment that follows; this group's staged block from
    //      registers to LDS, the next one requested ----
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dt[c2][db] = att_frag_tr(smem[1][cur], 16 * c2, db, lane);
        qt[c2][db] = att_frag_tr(smem[0][cur], 16 * c2, db, lane);
      }
    A3Q_ROWFRAGS(nxt);                                  // (last block: an old slot — the scores made from it are never used)
    {
      float pv[16], dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(s[r]);
        dsv[r] = pv[r] * dp[r];                        // (the factor of dS is applied once to the finished dK)
      }
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) { pa[c2] = pack8_bf16(&pv[c2 * 8]); dsa[c2] = pack8_bf16(&dsv[c2 * 8]); }
    }
    A3_PIN4(pa[0]); A3_PIN4(pa[1]); A3_PIN4(dsa[0]); A3_PIN4(dsa[1]);
    A3Q_STATS(nxt);                                     // s, dp are free again: the C operands of block i+1's products
    if (i + 2 < nt) {
      a3q_sstore(rs, sbuf[stg], tg);
      if (stat_thr) s_stat[stg][tg >> 5][tg & 31] = rstat * stat_mul;
    }
    if (i + 3 < nt) {
      rs = a3q_gload(Sp, Ss, (i + 3) * 32, tg);
      if (stat_thr) rstat = statp[(i + 3) * 32 + (tg & 31)];
    }
    A3_PIN16(s); A3_PIN16(dp);
    A3_BARRIER();
    // ---- matrix segment, registers only: dV += dO(i)^T P, dK += Q(i)^T dS, then the scores of block i+1 ----
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dv[db] = MFMA32(dt[c2][db], pa[c2], dv[db]);
        dk[db] = MFMA32(qt[c2][db], dsa[c2], dk[db]);
      }
    A3Q_SCORES();
    A3_PIN16(dv[0]); A3_PIN16(dv[1]); A3_PIN16(dk[0]); A3_PIN16(dk[1]); A3_PIN16(s); A3_PIN16(dp);
    A3_BARRIER();
    cur = nxt;
  }
  if (grp == 0) A3_BARRIER();
#undef A3Q_STATS
#undef A3Q_ROWFRAGS
#undef A3Q_SCORES

  // D^T[d][key]: lane (key = key0 + l31, hi) holds d = db*32 + 8*(r>>2) + 4*hi + (r&3)
  uint16_t* dkp = dqkv + ((int64_t)b * N + key0 + l31) * RS + H * ATT_D + h * ATT_D;
  uint16_t* dvp = dkp + H * ATT_D;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      const u32x2 wk = {pack_bf16x2(dk[db][g4 * 4 + 0] * kscale, dk[db][g4 * 4 + 1] * kscale), pack_bf16x2(dk[db][g4 * 4 + 2] * kscale, dk[db][g4 * 4 + 3] * kscale)};
      const u32x2 wv = {pack_bf16x2(dv[db][g4 * 4 + 0], dv[db][g4 * 4 + 1]), pack_bf16x2(dv[db][g4 * 4 + 2], dv[db][g4 * 4 + 3])};
      *reinterpret_cast<u32x2*>(dkp + d0) = wk;
      *reinterpret_cast<u32x2*>(dvp + d0) = wv;
    }
}

void attn_bwd_dkv3_launch(const uint16_t* qkv, const uint16_t* d_o, const float* lse, const float* delta, int B, int N, int H, float kscale, uint16_t* dqkv,
                          hipStream_t s) {
  const int64_t nblk = N / 256, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));
  attn_bwd_dkv3_kernel<<<grid, 512, 0, s>>>(qkv, d_o, lse, delta, B, N, H, kscale, dqkv);
}
