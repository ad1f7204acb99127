// x3.hip — "split-bf16" (x3) operands: the parity-grade encoder forward on the bf16 matrix cores (round 4).
//
// The reference's forward is fp32 end to end (enhancing/modules/stage1/layers.py:118-132,145-150, vitvqgan.py:61-66); a bf16-operand MFMA product
// carries 2^-9 per operand and flips ~2 % of the 8192-way argmin decisions downstream.  Here a value v is carried as the PAIR
//     hi = bf16(v),  lo = bf16(v - hi)            (v - hi - lo <= 2^-17 |v|)
// and a product a.b is formed as  a_hi b_hi + a_lo b_hi + a_hi b_lo  in the fp32 MFMA accumulator (the dropped a_lo b_lo term is 2^-18): three bf16
// passes = 833 TF/s of peak instead of the 157 TF/s of the exact-f32 MFMA, at ~1e-5 relative error end to end (measured: tests/test_x3_gpu.py).
//
// GEMMs need NO new kernel: the three passes are ONE enh_gemm_bf16 call on K-concatenated operands
//     A' [M][3K] = [ a_hi | a_lo | a_hi ]      B' [N][3K] = [ b_hi | b_hi | b_lo ]          sum_k' A' B' = the three-term product,
// so this file only holds what produces those rows — the split of an f32 matrix (optionally through bias + tanh: FeedForward's activation,
// layers.py:99-100), the split of the packed q | k | v projection into two planes — and the attention forward on split operands
// (S = Q K^T and O = P V each as three MFMA passes, softmax statistics in fp32 as in attention.hip).  LayerNorm writes its x3 row itself
// (layernorm.hip, enh_layernorm_forward_x3).
#include "attention_common.h"
typedef BF16 OT;   // the x3 path is bf16 by construction (hi / lo bf16 planes): the shared helpers of attention_common.h are used at that operand type

// hi / lo of eight consecutive values -> two 16-byte packets
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t h = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    hi[i] = h;
    lo[i] = pack_bf16x2(v[2 * i] - __builtin_bit_cast(float, h << 16), v[2 * i + 1] - __builtin_bit_cast(float, h & 0xffff0000u));
  }
}

// y3[m] = [hi | lo | hi] (ORDER 0: activation operand) or [hi | hi | lo] (ORDER 1: weight operand) of f(x[m]), f = identity / (+ bias) / tanh(+ bias);
// optional contiguous copy of the hi plane (what the bf16 path would have stored: the backward's operand).  8 elements per thread.
template <int ORDER, bool TANH>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int K, const float* __restrict__ bias,
                                                     uint16_t* __restrict__ y3, int64_t ldy3, uint16_t* __restrict__ yh, int64_t ldyh) {
  const int kc = K >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * kc) return;
  const int64_t m = i / kc;
  const int c = (int)(i - m * kc);
  const float4 a = *reinterpret_cast<const float4*>(x + m * ldx + c * 8), b = *reinterpret_cast<const float4*>(x + m * ldx + c * 8 + 4);
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if (bias) {
    const float4 p = *reinterpret_cast<const float4*>(bias + c * 8), q = *reinterpret_cast<const float4*>(bias + c * 8 + 4);
    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w; v[4] += q.x; v[5] += q.y; v[6] += q.z; v[7] += q.w;
  }
  if (TANH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanh_x3(v[e]);      // (common.h; the fused GEMM epilogue uses the same function: identical bits)
  }
  u32x4 hi, lo;
  split8(v, hi, lo);
  uint16_t* r = y3 + m * ldy3 + c * 8;
  *reinterpret_cast<u32x4*>(r) = hi;
  *reinterpret_cast<u32x4*>(r + K) = ORDER == 0 ? lo : hi;
  *reinterpret_cast<u32x4*>(r + 2 * K) = ORDER == 0 ? hi : lo;
  if (yh) *reinterpret_cast<u32x4*>(yh + m * ldyh + c * 8) = hi;
}

__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, int64_t n8, uint16_t* __restrict__ yh, uint16_t* __restrict__ yl) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const float4 a = *reinterpret_cast<const float4*>(x + i * 8), b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32x4 hi, lo;
  split8(v, hi, lo);
  *reinterpret_cast<u32x4*>(yh + i * 8) = hi;
  *reinterpret_cast<u32x4*>(yl + i * 8) = lo;
}

extern "C" int enh_split3_bf16(const float* x, int64_t ldx, int64_t M, int64_t K, const float* bias, int act, int order, enh_bf16* y3, int64_t ldy3,
                               enh_bf16* y_hi, int64_t ldy_hi, void* stream) {
  ENH_REQUIRE(x && y3, ENH_E_BADARG, "enh_split3_bf16: null pointer");
  ENH_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && K < (1 << 28) && ldx % 4 == 0 && ldy3 % 8 == 0 && ldy3 >= 3 * K && (!y_hi || ldy_hi % 8 == 0), ENH_E_SHAPE,
              "enh_split3_bf16: need K %% 8 == 0, ldx %% 4 == 0, ldy3 %% 8 == 0 and >= 3 K (M=%lld K=%lld ldx=%lld ldy3=%lld)", (long long)M, (long long)K,
              (long long)ldx, (long long)ldy3);
  ENH_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(y3) & 15u) == 0 && (reinterpret_cast<uintptr_t>(y_hi) & 15u) == 0 &&
              (reinterpret_cast<uintptr_t>(bias) & 15u) == 0, ENH_E_SHAPE, "enh_split3_bf16: 16-byte aligned bases");
  ENH_REQUIRE((act == ENH_ACT_NONE || act == ENH_ACT_TANH) && (order == 0 || order == 1), ENH_E_BADARG, "enh_split3_bf16: act in {0, 1}, order in {0, 1}");
  const int64_t n = M * (K / 8);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (order == 0 && act == ENH_ACT_TANH) split3_kernel<0, true><<<grid, 256, 0, s>>>(x, ldx, M, (int)K, bias, y3, ldy3, y_hi, ldy_hi);
  else if (order == 0) split3_kernel<0, false><<<grid, 256, 0, s>>>(x, ldx, M, (int)K, bias, y3, ldy3, y_hi, ldy_hi);
  else if (act == ENH_ACT_TANH) split3_kernel<1, true><<<grid, 256, 0, s>>>(x, ldx, M, (int)K, bias, y3, ldy3, y_hi, ldy_hi);
  else split3_kernel<1, false><<<grid, 256, 0, s>>>(x, ldx, M, (int)K, bias, y3, ldy3, y_hi, ldy_hi);
  return enh_check_launch("enh_split3_bf16");
}

extern "C" int enh_split2_bf16(const float* x, int64_t n, enh_bf16* hi, enh_bf16* lo, void* stream) {
  ENH_REQUIRE(x && hi && lo, ENH_E_BADARG, "enh_split2_bf16: null pointer");
  ENH_REQUIRE(n > 0 && n % 8 == 0, ENH_E_SHAPE, "enh_split2_bf16: n %% 8 == 0 (n=%lld)", (long long)n);
  ENH_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 15u) == 0, ENH_E_SHAPE,
              "enh_split2_bf16: 16-byte aligned bases");
  split2_kernel<<<dim3((unsigned)((n / 8 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, n / 8, hi, lo);
  return enh_check_launch("enh_split2_bf16");
}

// =================================================================================================
// attention forward on split operands.  Same skeleton as attn_fwd_exact (attention_common.h): 128 queries per workgroup, S^T = K Q^T so that a lane owns
// one query column, K / V streamed in 64-key tiles through a two-stage LDS ring — here FOUR tiles per stage (K_hi, K_lo, V_hi, V_lo; 64 KiB of LDS, two
// workgroups per CU).  Per key tile: 24 MFMAs for S (small terms first), exact running maximum, numerators split in registers, 24 MFMAs for O.
// The output row is written as the x3 operand [hi | lo | hi] of to_out (row stride 3 H 64) and, optionally, as the plain bf16 tensor the backward reads.
// =================================================================================================
#define X3_STAGE_BYTES (4 * ATT_TILE_BYTES)
__global__ __launch_bounds__(256, 2) void attn_fwd_x3_kernel(const uint16_t* __restrict__ qh, const uint16_t* __restrict__ ql, int B, int N, int H,
                                                             float scale_log2, uint16_t* __restrict__ out3, uint16_t* __restrict__ out16,
                                                             float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 stages][K_hi | K_lo | V_hi | V_lo]
  int blk, head;
  if (!att_block_coords((N + 127) / 128, B * H, blk, head)) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = head / H, h = head - b * H;
  const int q0 = blk * 128 + wave * 32;
  const int64_t RS = (int64_t)3 * H * ATT_D;
  const int64_t base = (int64_t)b * N * RS + h * ATT_D;
  const uint16_t* Kh = qh + base + H * ATT_D;
  const uint16_t* Kl = ql + base + H * ATT_D;
  const uint16_t* Vh = Kh + H * ATT_D;
  const uint16_t* Vl = Kl + H * ATT_D;

  const bool active = q0 < N;
  const int qrow = active ? q0 + l31 : l31;
  s16x8 qfh[4], qfl[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qfh[ds] = *reinterpret_cast<const s16x8*>(qh + base + (int64_t)qrow * RS + ds * 16 + hi * 8);
    qfl[ds] = *reinterpret_cast<const s16x8*>(ql + base + (int64_t)qrow * RS + ds * 16 + hi * 8);
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -__builtin_inff(), l_part = 0.f;

  const int nt = N / 64;
  u32x4 rkh[2], rkl[2], rvh[2], rvl[2];
  att_gload(rkh, Kh, RS, 0, t); att_gload(rkl, Kl, RS, 0, t);
  att_gload(rvh, Vh, RS, 0, t); att_gload(rvl, Vl, RS, 0, t);
  att_sstore(rkh, smem, t); att_sstore(rkl, smem + ATT_TILE_BYTES, t);
  att_sstore(rvh, smem + 2 * ATT_TILE_BYTES, t); att_sstore(rvl, smem + 3 * ATT_TILE_BYTES, t);
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) { att_pin(qfh[ds]); att_pin(qfl[ds]); }
  ATT_LOOP_ENTRY();
  __syncthreads();
  for (int kt = 0; kt < nt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nt) {
      att_gload(rkh, Kh, RS, (kt + 1) * 64, t); att_gload(rkl, Kl, RS, (kt + 1) * 64, t);
      att_gload(rvh, Vh, RS, (kt + 1) * 64, t); att_gload(rvl, Vl, RS, (kt + 1) * 64, t);
    }
    const unsigned char* kh_ = smem + st * X3_STAGE_BYTES;
    const unsigned char* kl_ = kh_ + ATT_TILE_BYTES;
    const unsigned char* vh_ = kh_ + 2 * ATT_TILE_BYTES;
    const unsigned char* vl_ = kh_ + 3 * ATT_TILE_BYTES;
    // ---- S^T[key][q] = K Q^T : K_lo Q_hi + K_hi Q_lo + K_hi Q_hi ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const s16x8 fh = att_frag_row(kh_, kb * 32, ds, l31, hi), fl = att_frag_row(kl_, kb * 32, ds, l31, hi);
        s[kb] = MFMA32(fl, qfh[ds], s[kb]);
        s[kb] = MFMA32(fh, qfl[ds], s[kb]);
        s[kb] = MFMA32(fh, qfh[ds], s[kb]);
      }
    }
    // ---- online softmax for this lane's query column (fp32, exact running maximum) ----
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float p[2][16], pl[2][16];
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] * scale_log2 - m_new);
        psum += p[kb][r];
      }
    l_part = l_part * alpha + psum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    // ---- O^T[d][q] += V^T P^T : V_lo P_hi + V_hi P_lo + V_hi P_hi ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const s16x8 pbh = pack8<BF16>(&p[kb][c2 * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) pl[kb][c2 * 8 + e] = p[kb][c2 * 8 + e] - bf16_bits_to_f32((uint16_t)pbh[e]);
        const s16x8 pbl = pack8<BF16>(&pl[kb][c2 * 8]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const s16x8 fvh = att_frag_tr(vh_, kb * 32 + 16 * c2, db, lane), fvl = att_frag_tr(vl_, kb * 32 + 16 * c2, db, lane);
          o[db] = MFMA32(fvl, pbh, o[db]);
          o[db] = MFMA32(fvh, pbl, o[db]);
          o[db] = MFMA32(fvh, pbh, o[db]);
        }
      }
    if (kt + 1 < nt) {
      unsigned char* nx = smem + (st ^ 1) * X3_STAGE_BYTES;
      att_sstore(rkh, nx, t); att_sstore(rkl, nx + ATT_TILE_BYTES, t);
      att_sstore(rvh, nx + 2 * ATT_TILE_BYTES, t); att_sstore(rvl, nx + 3 * ATT_TILE_BYTES, t);
    }
    __syncthreads();
  }
  const float l = l_part + __shfl_xor(l_part, 32, 64);
  const float inv = 1.0f / l;
  if (!active) return;
  const int64_t OS = (int64_t)H * ATT_D;
  uint16_t* op3 = out3 + ((int64_t)b * N + q0 + l31) * (3 * OS) + h * ATT_D;
  uint16_t* op = out16 ? out16 + ((int64_t)b * N + q0 + l31) * OS + h * ATT_D : nullptr;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d0 = db * 32 + 8 * g4 + 4 * hi;
      const float v0 = o[db][g4 * 4 + 0] * inv, v1 = o[db][g4 * 4 + 1] * inv, v2 = o[db][g4 * 4 + 2] * inv, v3 = o[db][g4 * 4 + 3] * inv;
      const u32x2 wh = {pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
      const u32x2 wl = {pack_bf16x2(v0 - __builtin_bit_cast(float, wh[0] << 16), v1 - __builtin_bit_cast(float, wh[0] & 0xffff0000u)),
                        pack_bf16x2(v2 - __builtin_bit_cast(float, wh[1] << 16), v3 - __builtin_bit_cast(float, wh[1] & 0xffff0000u))};
      *reinterpret_cast<u32x2*>(op3 + d0) = wh;
      *reinterpret_cast<u32x2*>(op3 + OS + d0) = wl;
      *reinterpret_cast<u32x2*>(op3 + 2 * OS + d0) = wh;
      if (op) *reinterpret_cast<u32x2*>(op + d0) = wh;
    }
  if (hi == 0) lse[((int64_t)b * H + h) * N + q0 + l31] = (m_run + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
}

extern "C" int enh_attention_forward_x3(const enh_bf16* qkv_hi, const enh_bf16* qkv_lo, int B, int N, int H, float scale, enh_bf16* out3, enh_bf16* out_bf16,
                                        float* lse, void* stream) {
  ENH_REQUIRE(qkv_hi && qkv_lo && out3 && lse, ENH_E_BADARG, "enh_attention_forward_x3: null pointer");
  ENH_REQUIRE(B > 0 && H > 0 && N > 0 && N % 64 == 0, ENH_E_SHAPE, "enh_attention_forward_x3: need N %% 64 == 0 (B=%d N=%d H=%d)", B, N, H);
  ENH_REQUIRE(scale > 0.f, ENH_E_BADARG, "enh_attention_forward_x3: scale must be positive");
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X3_STAGE_BYTES);
    return true;
  }();
  (void)attr;
  const int64_t nblk = (N + 127) / 128, heads = (int64_t)B * H;
  const dim3 grid((unsigned)(((heads + 7) / 8) * 8 * nblk));
  attn_fwd_x3_kernel<<<grid, 256, 2 * X3_STAGE_BYTES, (hipStream_t)stream>>>(qkv_hi, qkv_lo, B, N, H, scale * 1.4426950408889634f, out3, out_bf16, lse);
  return enh_check_launch("enh_attention_forward_x3");
}
