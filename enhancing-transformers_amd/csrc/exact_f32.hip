// exact_f32.hip — fp32 "exact mode" kernels: the same contractions as gemm.hip / attention.hip with fp32 operands, fp32 accumulation in
// a fixed ascending-k order, no reduced-precision storage anywhere.  Purpose: end-to-end parity runs against the fp32 CPU oracle
// (SURVEY.md §8d metric 3: relative Frobenius error <= 1e-3 for h, xrec and per-layer outputs; index parity end to end).  These kernels
// favour obviousness over speed (plain v_fma on LDS tiles, one thread per attention row); the bf16 MFMA kernels are the product path.
#include "common.h"

#define E_BM 64
#define E_BN 64
#define E_BK 16

struct GemmF32Args {
  const float* A; int64_t lda; const float* B; int64_t ldb;
  int64_t M, N, K;
  const float* bias; int act; const float* aux; int64_t ldaux;
  const float* res; int64_t ldres; int64_t res_rows; int accumulate;
  float* C; int64_t ldc;
};

// C[m][n] = epilogue( sum_k A(m,k) * B(n,k) ); trans_* as in enh_gemm_bf16.  64x64 tile, 256 threads, 4x4 outputs per thread.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args g) {
  __shared__ float sA[E_BK][E_BM + 1];
  __shared__ float sB[E_BK][E_BN + 1];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * E_BM, n0 = (int64_t)blockIdx.x * E_BN;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int64_t k0 = 0; k0 < g.K; k0 += E_BK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = t + 256 * e;  // 1024 elements per operand tile
      {
        const int kk = TA ? idx / E_BM : idx % E_BK, mm = TA ? idx % E_BM : idx / E_BK;
        const int64_t m = m0 + mm, k = k0 + kk;
        sA[kk][mm] = (m < g.M && k < g.K) ? (TA ? g.A[k * g.lda + m] : g.A[m * g.lda + k]) : 0.f;
      }
      {
        const int kk = TB ? idx / E_BN : idx % E_BK, nn = TB ? idx % E_BN : idx / E_BK;
        const int64_t n = n0 + nn, k = k0 + kk;
        sB[kk][nn] = (n < g.N && k < g.K) ? (TB ? g.B[k * g.ldb + n] : g.B[n * g.ldb + k]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < E_BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      float v = acc[i][j];
      if (g.bias) v += g.bias[n];
      if (g.act == ENH_ACT_TANH) v = tanhf(v);
      else if (g.act == ENH_ACT_DTANH) { const float h = g.aux[m * g.ldaux + n]; v *= 1.f - h * h; }
      if (g.res) v += g.res[(m % g.res_rows) * g.ldres + n];
      if (g.accumulate) v += g.C[m * g.ldc + n];
      g.C[m * g.ldc + n] = v;
    }
  }
}

// ---- attention, fp32, one thread per query (or key) row; qkv packed [B, N, 3*H*64] like the bf16 path ----
#define A_D 64
__global__ __launch_bounds__(64) void attn_f32_fwd_kernel(const float* __restrict__ qkv, int N, int H, float scale, float* __restrict__ out,
                                                          float* __restrict__ lse) {
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
  const int64_t RS = (int64_t)3 * H * A_D;
  const float* Q = qkv + (int64_t)b * N * RS + h * A_D;
  const float* K = Q + H * A_D;
  const float* V = K + H * A_D;
  const bool live = i < N;
  const int ii = live ? i : 0;
  float q[A_D], o[A_D];
#pragma unroll
  for (int d = 0; d < A_D; ++d) { q[d] = Q[(int64_t)ii * RS + d]; o[d] = 0.f; }
  float m = -__builtin_inff(), l = 0.f;
  for (int j = 0; j < N; ++j) {
    const float* kj = K + (int64_t)j * RS;
    const float* vj = V + (int64_t)j * RS;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < A_D; ++d) s = fmaf(q[d], kj[d], s);
    s *= scale;
    const float mn = fmaxf(m, s);
    const float alpha = expf(m - mn), p = expf(s - mn);
    l = l * alpha + p;
#pragma unroll
    for (int d = 0; d < A_D; ++d) o[d] = fmaf(p, vj[d], o[d] * alpha);
    m = mn;
  }
  if (!live) return;
  const float inv = 1.f / l;
  float* op = out + ((int64_t)b * N + i) * (H * A_D) + h * A_D;
#pragma unroll
  for (int d = 0; d < A_D; ++d) op[d] = o[d] * inv;
  lse[((int64_t)b * H + h) * N + i] = m + logf(l);
}

__global__ void attn_f32_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, int64_t BN, int N, int H, float* __restrict__ delta) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b*N + q) * H + h
  if (idx >= BN * H) return;
  const int h = (int)(idx % H);
  const int64_t bq = idx / H, bb = bq / N, q = bq % N;
  float acc = 0.f;
  for (int d = 0; d < A_D; ++d) acc = fmaf(o[idx * A_D + d], d_o[idx * A_D + d], acc);
  delta[(bb * H + h) * N + q] = acc;
}

// dQ_i = sum_j dS_ij K_j ,  dS_ij = P_ij (dO_i . V_j - delta_i) * scale
__global__ __launch_bounds__(64) void attn_f32_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, int N, int H, float scale, float* __restrict__ dqkv) {
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
  const int64_t RS = (int64_t)3 * H * A_D, OS = (int64_t)H * A_D;
  const float* Q = qkv + (int64_t)b * N * RS + h * A_D;
  const float* K = Q + H * A_D;
  const float* V = K + H * A_D;
  const bool live = i < N;
  const int ii = live ? i : 0;
  float q[A_D], g[A_D], dq[A_D];
#pragma unroll
  for (int d = 0; d < A_D; ++d) { q[d] = Q[(int64_t)ii * RS + d]; g[d] = d_o[((int64_t)b * N + ii) * OS + h * A_D + d]; dq[d] = 0.f; }
  const float li = lse[((int64_t)b * H + h) * N + ii], di = delta[((int64_t)b * H + h) * N + ii];
  for (int j = 0; j < N; ++j) {
    const float* kj = K + (int64_t)j * RS;
    const float* vj = V + (int64_t)j * RS;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < A_D; ++d) { s = fmaf(q[d], kj[d], s); dp = fmaf(g[d], vj[d], dp); }
    const float p = expf(s * scale - li);
    const float ds = p * (dp - di) * scale;
#pragma unroll
    for (int d = 0; d < A_D; ++d) dq[d] = fmaf(ds, kj[d], dq[d]);
  }
  if (!live) return;
  float* op = dqkv + ((int64_t)b * N + i) * RS + h * A_D;
#pragma unroll
  for (int d = 0; d < A_D; ++d) op[d] = dq[d];
}

// dV_j = sum_i P_ij dO_i ; dK_j = sum_i dS_ij Q_i
__global__ __launch_bounds__(64) void attn_f32_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, int N, int H, float scale, float* __restrict__ dqkv) {
  const int b = blockIdx.z, h = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  const int64_t RS = (int64_t)3 * H * A_D, OS = (int64_t)H * A_D;
  const float* Q = qkv + (int64_t)b * N * RS + h * A_D;
  const float* K = Q + H * A_D;
  const float* V = K + H * A_D;
  const bool live = j < N;
  const int jj = live ? j : 0;
  float k[A_D], v[A_D], dk[A_D], dv[A_D];
#pragma unroll
  for (int d = 0; d < A_D; ++d) { k[d] = K[(int64_t)jj * RS + d]; v[d] = V[(int64_t)jj * RS + d]; dk[d] = 0.f; dv[d] = 0.f; }
  for (int i = 0; i < N; ++i) {
    const float* qi = Q + (int64_t)i * RS;
    const float* gi = d_o + ((int64_t)b * N + i) * OS + h * A_D;
    const float li = lse[((int64_t)b * H + h) * N + i], di = delta[((int64_t)b * H + h) * N + i];
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < A_D; ++d) { s = fmaf(qi[d], k[d], s); dp = fmaf(gi[d], v[d], dp); }
    const float p = expf(s * scale - li);
    const float ds = p * (dp - di) * scale;
#pragma unroll
    for (int d = 0; d < A_D; ++d) { dv[d] = fmaf(p, gi[d], dv[d]); dk[d] = fmaf(ds, qi[d], dk[d]); }
  }
  if (!live) return;
  float* kp = dqkv + ((int64_t)b * N + j) * RS + H * A_D + h * A_D;
  float* vp = kp + H * A_D;
#pragma unroll
  for (int d = 0; d < A_D; ++d) { kp[d] = dk[d]; vp[d] = dv[d]; }
}

// out[n] (+)= sum_m x[m][n] for f32 x: one thread per column chunk, deterministic order inside a block, atomics across row chunks
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, int64_t M, int64_t N, int64_t ldx, int64_t rows_per_block,
                                                         float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t m1 = m0 + rows_per_block;
  if (m1 > M) m1 = M;
  float acc = 0.f;
  for (int64_t m = m0; m < m1; ++m) acc += x[m * ldx + n];
  atomicAdd(&out[n], acc);
}

// patch gather / scatter in f32 (same permutation as elementwise.hip)
__global__ void patch_perm_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W, int p, int to_patches,
                                      int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int gx_n = W / p, gy_n = H / p;
  int64_t r = i;
  const int pw = (int)(r % p); r /= p;
  const int ph = (int)(r % p); r /= p;
  const int c = (int)(r % C); r /= C;
  const int gx = (int)(r % gx_n);
  const int gy = (int)((r / gx_n) % gy_n);
  const int b = (int)(r / ((int64_t)gx_n * gy_n));
  const size_t off = (((size_t)b * C + c) * H + (size_t)gy * p + ph) * W + (size_t)gx * p + pw;
  if (to_patches) dst[i] = src[off]; else dst[off] = src[i];
}

extern "C" int enh_gemm_f32(const float* A, int64_t lda, int trans_a, const float* B, int64_t ldb, int trans_b, int64_t M, int64_t N, int64_t K,
                            const float* bias, int act, const float* aux, int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows,
                            int accumulate, float* C, int64_t ldc, void* stream) {
  ENH_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, ENH_E_BADARG, "enh_gemm_f32: bad argument");
  ENH_REQUIRE(act == ENH_ACT_NONE || act == ENH_ACT_TANH || (act == ENH_ACT_DTANH && aux), ENH_E_BADARG, "enh_gemm_f32: bad act/aux");
  ENH_REQUIRE(!res || res_rows > 0, ENH_E_BADARG, "enh_gemm_f32: res needs res_rows");
  GemmF32Args g{A, lda, B, ldb, M, N, K, bias, act, aux, ldaux, res, ldres, res_rows, accumulate, C, ldc};
  const dim3 grid((unsigned)((N + E_BN - 1) / E_BN), (unsigned)((M + E_BM - 1) / E_BM));
  hipStream_t s = (hipStream_t)stream;
  if (!trans_a && !trans_b) gemm_f32_kernel<false, false><<<grid, 256, 0, s>>>(g);
  else if (!trans_a && trans_b) gemm_f32_kernel<false, true><<<grid, 256, 0, s>>>(g);
  else if (trans_a && !trans_b) gemm_f32_kernel<true, false><<<grid, 256, 0, s>>>(g);
  else gemm_f32_kernel<true, true><<<grid, 256, 0, s>>>(g);
  return enh_check_launch("enh_gemm_f32");
}

extern "C" int enh_attention_forward_f32(const float* qkv, int B, int N, int H, float scale, float* out, float* lse, void* stream) {
  ENH_REQUIRE(qkv && out && lse && B > 0 && N > 0 && H > 0 && scale > 0.f, ENH_E_BADARG, "enh_attention_forward_f32: bad argument");
  attn_f32_fwd_kernel<<<dim3((N + 63) / 64, H, B), 64, 0, (hipStream_t)stream>>>(qkv, N, H, scale, out, lse);
  return enh_check_launch("enh_attention_forward_f32");
}

extern "C" int enh_attention_backward_f32(const float* qkv, const float* out, const float* dout, const float* lse, int B, int N, int H, float scale,
                                          float* dqkv, float* delta_ws, void* stream) {
  ENH_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws && B > 0 && N > 0 && H > 0 && scale > 0.f, ENH_E_BADARG, "enh_attention_backward_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t BN = (int64_t)B * N;
  attn_f32_delta_kernel<<<(int)((BN * H + 255) / 256), 256, 0, s>>>(out, dout, BN, N, H, delta_ws);
  const dim3 grid((N + 63) / 64, H, B);
  attn_f32_dq_kernel<<<grid, 64, 0, s>>>(qkv, dout, lse, delta_ws, N, H, scale, dqkv);
  attn_f32_dkv_kernel<<<grid, 64, 0, s>>>(qkv, dout, lse, delta_ws, N, H, scale, dqkv);
  return enh_check_launch("enh_attention_backward_f32");
}

extern "C" int enh_colsum_f32(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* stream) {
  ENH_REQUIRE(x && out && M > 0 && N > 0, ENH_E_BADARG, "enh_colsum_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) { const int rc = enh_zero_f32_launch(out, N, s); if (rc) return rc; }
  int64_t chunks = (M + 1023) / 1024;
  if (chunks > 128) chunks = 128;
  const int64_t rpb = (M + chunks - 1) / chunks;
  colsum_f32_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)chunks), 256, 0, s>>>(x, M, N, ldx, rpb, out);
  return enh_check_launch("enh_colsum_f32");
}

extern "C" int enh_patch_perm_f32(const float* src, float* dst, int B, int C, int H, int W, int p, int to_patches, void* stream) {
  ENH_REQUIRE(src && dst && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, ENH_E_BADARG, "enh_patch_perm_f32: bad argument");
  const int64_t total = (int64_t)B * C * H * W;
  patch_perm_f32_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, B, C, H, W, p, to_patches, total);
  return enh_check_launch("enh_patch_perm_f32");
}

// pixel loss + patch-layout gradient in f32 (exact-mode twin of unpatchify_loss_kernel in elementwise.hip)
__global__ __launch_bounds__(256) void unpatchify_loss_f32_kernel(const float* __restrict__ pix, const float* __restrict__ target, int B, int C, int H,
                                                                  int W, int p, float w_l1, float w_l2, float inv_numel, float* __restrict__ xrec,
                                                                  double* __restrict__ sums, float* __restrict__ dpix, int64_t total) {
  __shared__ float s_l1[4], s_l2[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float l1 = 0.f, l2 = 0.f;
  if (i < total) {
    const int gx_n = W / p, gy_n = H / p;
    int64_t r = i;
    const int pw = (int)(r % p); r /= p;
    const int ph = (int)(r % p); r /= p;
    const int c = (int)(r % C); r /= C;
    const int gx = (int)(r % gx_n);
    const int gy = (int)((r / gx_n) % gy_n);
    const int b = (int)(r / ((int64_t)gx_n * gy_n));
    const size_t off = (((size_t)b * C + c) * H + (size_t)gy * p + ph) * W + (size_t)gx * p + pw;
    const float v = pix[i];
    xrec[off] = v;
    const float d = v - target[off];
    l1 = fabsf(d);
    l2 = d * d;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    dpix[i] = (w_l1 * sg + w_l2 * 2.f * d) * inv_numel;
  }
  l1 = wave_sum(l1);
  l2 = wave_sum(l2);
  if ((threadIdx.x & 63) == 0) { s_l1[threadIdx.x >> 6] = l1; s_l2[threadIdx.x >> 6] = l2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (double)((s_l1[0] + s_l1[1]) + (s_l1[2] + s_l1[3])));
    atomicAdd(&sums[1], (double)((s_l2[0] + s_l2[1]) + (s_l2[2] + s_l2[3])));
  }
}

extern "C" int enh_unpatchify_loss_f32(const float* pix, const float* target, int B, int C, int H, int W, int p, float w_l1, float w_l2, float* xrec,
                                       double* sums, float* dpix, void* stream) {
  ENH_REQUIRE(pix && target && xrec && sums && dpix, ENH_E_BADARG, "enh_unpatchify_loss_f32: null pointer");
  ENH_REQUIRE(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, ENH_E_SHAPE, "enh_unpatchify_loss_f32: H, W must be divisible by p");
  const int64_t total = (int64_t)B * C * H * W;
  unpatchify_loss_f32_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(pix, target, B, C, H, W, p, w_l1, w_l2, 1.0f / (float)total, xrec,
                                                                                       sums, dpix, total);
  return enh_check_launch("enh_unpatchify_loss_f32");
}
