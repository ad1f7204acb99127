// conv_ops.hip — lowering of the StyleGAN2 discriminator's equalised-lr convolutions (reference enhancing/losses/layers.py:163-185:
// EqualConv2d = conv2d(input, weight * scale, stride 1 | 2, padding k/2 | 0), k = 1 | 3) onto the MFMA GEMM of gemm.hip (bf16 / fp16 columns) or onto the
// exact-f32 GEMM of exact_f32.hip (f32 columns: the parity instrument of the discriminator, StyleDiscriminator(lowering="im2col") under
// conv2d_gradfix.operand_dtype("fp32")).
//
//   forward   y[Cout, B*Ho*Wo]   = W[Cout, Kp] . cols[B*Ho*Wo, Kp]^T          cols = enh_im2col(x)
//   wgrad     dW[Cout, Kp]       = dy[Cout, B*Ho*Wo] . cols[B*Ho*Wo, Kp]
//   dgrad     dcols[B*Ho*Wo, Kp] = dy^T . W ;  dx = enh_col2im(dcols)
//
// cols row = (b, ho, wo), column = c*k*k + kh*k + kw — the order of weight.view(Cout, -1) — zero-padded to Kp = ld (a multiple of 8,
// the GEMM's alignment unit).  The image operand is addressed through explicit batch / channel strides, so the same kernels read the
// module input ([B,C,H,W]) and the discriminator's internal channel-major activations ([C,B,H,W], which is what the GEMM above
// writes).  Both kernels are HBM-bound data movement: global accesses are contiguous along W (image side) and along the column index
// (cols side); the (c,kh,kw) <-> (h,w) transposition happens in LDS.
#include "common.h"

#define CV_TW 32   // output columns (wo) per workgroup
#define CV_CC 32   // channels per workgroup (im2col)
#define CV_MAXK 3

// ---- im2col ---------------------------------------------------------------------------------------------------------------------
// grid: (ceil(Wo / CV_TW), Ho, B * ceil(C / CV_CC)); 256 threads
// DT: element type of the columns — ENH_DT_BF16 / ENH_DT_F16 (16-bit MFMA operands, round-to-nearest-even) or ENH_DT_F32 (exact)
template <int DT>
__device__ __forceinline__ uint16_t cv_pack1(float v) { return DT == ENH_DT_F16 ? pack1<F16>(v) : pack1<BF16>(v); }
template <int DT>
__device__ __forceinline__ float cv_unpack1(const void* base, int64_t i) {
  if (DT == ENH_DT_F32) return static_cast<const float*>(base)[i];
  const uint16_t h = static_cast<const uint16_t*>(base)[i];
  return DT == ENH_DT_F16 ? unpack1<F16>(h) : unpack1<BF16>(h);
}
template <int DT>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int64_t sb, int64_t sc, int C, int H, int W,
                                                     int k, int stride, int pad, int Ho, int Wo, int nchunk,
                                                     void* __restrict__ cols_any, int64_t ld) {
  __shared__ float patch[CV_CC * CV_MAXK * ((CV_TW - 1) * 2 + CV_MAXK + 1)];
  const int wo0 = blockIdx.x * CV_TW, ho = blockIdx.y;
  const int b = blockIdx.z / nchunk, c0 = (blockIdx.z % nchunk) * CV_CC;
  const int kk = k * k;
  const int Wp = (CV_TW - 1) * stride + k;  // input columns feeding CV_TW outputs
  const int t = threadIdx.x;
  // (1) stage the [CC][k][Wp] input patch; zero outside the image / beyond C
  for (int e = t; e < CV_CC * k * Wp; e += 256) {
    const int c = e / (k * Wp), r = e - c * (k * Wp);
    const int kh = r / Wp, w = r - kh * Wp;
    const int hi = ho * stride - pad + kh, wi = wo0 * stride - pad + w;
    float v = 0.f;
    if (c0 + c < C && hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[(int64_t)b * sb + (int64_t)(c0 + c) * sc + (int64_t)hi * W + wi];
    patch[e] = v;
  }
  __syncthreads();
  // (2) write CV_TW rows x (CC*kk) columns, two columns (4 bytes) per lane, contiguous along the row
  const int ncol = CV_CC * kk, half = ncol >> 1;  // ncol is even
  for (int e = t; e < CV_TW * half; e += 256) {
    const int r = e / half, j0 = (e - r * half) * 2;
    const int wo = wo0 + r;
    const int64_t col = (int64_t)c0 * kk + j0;
    if (wo >= Wo || col >= ld) continue;  // ld is even, so a pair is entirely inside or outside
    float pv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u;
      const int c = j / kk, q = j - c * kk;
      const int kh = q / k, kw = q - kh * k;
      pv[u] = patch[(c * k + kh) * Wp + r * stride + kw];
    }
    const int64_t row = ((int64_t)b * Ho + ho) * Wo + wo;
    if (DT == ENH_DT_F32) *reinterpret_cast<float2*>(static_cast<float*>(cols_any) + row * ld + col) = make_float2(pv[0], pv[1]);
    else *reinterpret_cast<uint32_t*>(static_cast<uint16_t*>(cols_any) + row * ld + col) = (uint32_t)cv_pack1<DT>(pv[0]) | ((uint32_t)cv_pack1<DT>(pv[1]) << 16);
  }
}

// ---- col2im ---------------------------------------------------------------------------------------------------------------------
// dx[b,c,h,w] = sum over (kh,kw) with (h + pad - kh) = ho*stride, (w + pad - kw) = wo*stride, 0 <= ho < Ho, 0 <= wo < Wo of
//               dcols[(b,ho,wo), c*kk + kh*k + kw]            (gather form: no atomics, every dx element written exactly once)
// grid: (ceil(W / CV_TW), H, B * ceil(C / 64)); 256 threads: lane -> channel (the cols side is contiguous in c*kk), wave -> 8 of the
// 32 w positions; the [64 c][32 w] result tile is transposed through LDS so the image side is written contiguously along w.
template <int DT>
__global__ __launch_bounds__(256) void col2im_kernel(const void* __restrict__ dcols, int64_t ld, int C, int H, int W, int k,
                                                         int stride, int pad, int Ho, int Wo, int nchunk, float* __restrict__ dx,
                                                         int64_t sb, int64_t sc) {
  __shared__ float tile[64][CV_TW + 1];
  const int w0 = blockIdx.x * CV_TW, h = blockIdx.y;
  const int b = blockIdx.z / nchunk, c0 = (blockIdx.z % nchunk) * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = k * k;
  const int c = c0 + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int wl = wave * 8 + i, w = w0 + wl;
    float acc = 0.f;
    if (c < C && w < W) {
      for (int kh = 0; kh < k; ++kh) {
        const int hs = h + pad - kh;
        if (hs < 0 || hs % stride) continue;
        const int ho = hs / stride;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < k; ++kw) {
          const int ws = w + pad - kw;
          if (ws < 0 || ws % stride) continue;
          const int wo = ws / stride;
          if (wo >= Wo) continue;
          const int64_t row = ((int64_t)b * Ho + ho) * Wo + wo;
          acc += cv_unpack1<DT>(dcols, row * ld + (int64_t)c * kk + kh * k + kw);
        }
      }
    }
    tile[lane][wl] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * CV_TW; e += 256) {
    const int cl = e / CV_TW, wl = e - cl * CV_TW;
    if (c0 + cl < C && w0 + wl < W) dx[(int64_t)b * sb + (int64_t)(c0 + cl) * sc + (int64_t)h * W + w0 + wl] = tile[cl][wl];
  }
}

static bool conv_geom_ok(int B, int C, int H, int W, int k, int stride, int pad, int Ho, int Wo, int64_t ld) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2) || pad < 0 || pad > k) return false;
  if (Ho != (H + 2 * pad - k) / stride + 1 || Wo != (W + 2 * pad - k) / stride + 1 || Ho <= 0 || Wo <= 0) return false;
  return ld % 8 == 0 && ld >= (int64_t)C * k * k && ld < (int64_t)C * k * k + 8;
}

#define CV_DISPATCH(dtype, CALL)                                   \
  do {                                                             \
    if ((dtype) == ENH_DT_F32) { constexpr int DT = ENH_DT_F32; CALL; }      \
    else if ((dtype) == ENH_DT_F16) { constexpr int DT = ENH_DT_F16; CALL; } \
    else { constexpr int DT = ENH_DT_BF16; CALL; }                  \
  } while (0)

extern "C" int enh_im2col(const float* x, int64_t stride_b, int64_t stride_c, int B, int C, int H, int W, int k, int stride,
                          int pad, int Ho, int Wo, void* cols, int64_t ld, int dtype, void* stream) {
  ENH_REQUIRE(dtype == ENH_DT_BF16 || dtype == ENH_DT_F16 || dtype == ENH_DT_F32, ENH_E_BADARG, "enh_im2col: dtype must be ENH_DT_BF16, ENH_DT_F16 or ENH_DT_F32");
  ENH_REQUIRE(x && cols, ENH_E_BADARG, "enh_im2col: null pointer");
  ENH_REQUIRE(conv_geom_ok(B, C, H, W, k, stride, pad, Ho, Wo, ld), ENH_E_SHAPE,
              "enh_im2col: need k in {1,3}, stride in {1,2}, Ho/Wo = (H + 2 pad - k) / stride + 1, ld = C*k*k rounded up to 8 "
              "(B=%d C=%d H=%d W=%d k=%d stride=%d pad=%d Ho=%d Wo=%d ld=%lld)", B, C, H, W, k, stride, pad, Ho, Wo, (long long)ld);
  const int nchunk = (C + CV_CC - 1) / CV_CC;
  ENH_REQUIRE((int64_t)B * nchunk <= 65535 && Ho <= 65535, ENH_E_SHAPE, "enh_im2col: grid too large");
  const dim3 grid((unsigned)((Wo + CV_TW - 1) / CV_TW), (unsigned)Ho, (unsigned)(B * nchunk));
  CV_DISPATCH(dtype, (im2col_kernel<DT><<<grid, 256, 0, (hipStream_t)stream>>>(x, stride_b, stride_c, C, H, W, k, stride, pad, Ho, Wo, nchunk, cols, ld)));
  return enh_check_launch("enh_im2col");
}

extern "C" int enh_col2im(const void* dcols, int64_t ld, int B, int C, int H, int W, int k, int stride, int pad, int Ho,
                          int Wo, float* dx, int64_t stride_b, int64_t stride_c, int dtype, void* stream) {
  ENH_REQUIRE(dtype == ENH_DT_BF16 || dtype == ENH_DT_F16 || dtype == ENH_DT_F32, ENH_E_BADARG, "enh_col2im: dtype must be ENH_DT_BF16, ENH_DT_F16 or ENH_DT_F32");
  ENH_REQUIRE(dcols && dx, ENH_E_BADARG, "enh_col2im: null pointer");
  ENH_REQUIRE(conv_geom_ok(B, C, H, W, k, stride, pad, Ho, Wo, ld), ENH_E_SHAPE,
              "enh_col2im: need k in {1,3}, stride in {1,2}, Ho/Wo = (H + 2 pad - k) / stride + 1, ld = C*k*k rounded up to 8 "
              "(B=%d C=%d H=%d W=%d k=%d stride=%d pad=%d Ho=%d Wo=%d ld=%lld)", B, C, H, W, k, stride, pad, Ho, Wo, (long long)ld);
  const int nchunk = (C + 63) / 64;
  ENH_REQUIRE((int64_t)B * nchunk <= 65535 && H <= 65535, ENH_E_SHAPE, "enh_col2im: grid too large");
  const dim3 grid((unsigned)((W + CV_TW - 1) / CV_TW), (unsigned)H, (unsigned)(B * nchunk));
  CV_DISPATCH(dtype, (col2im_kernel<DT><<<grid, 256, 0, (hipStream_t)stream>>>(dcols, ld, C, H, W, k, stride, pad, Ho, Wo, nchunk, dx, stride_b, stride_c)));
  return enh_check_launch("enh_col2im");
}
