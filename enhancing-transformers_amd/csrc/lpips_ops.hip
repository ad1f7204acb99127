// lpips_ops.hip — the non-GEMM pieces of the LPIPS perceptual term (lpips 0.1.4, net = "vgg"; reference call sites
// enhancing/losses/vqperceptual.py:29,43,74,115): the first VGG convolution fused with the input scaling layer, 2x2 max-pooling, and the LPIPS
// head (channel unit-normalisation, squared difference, 1x1 "lin" layer, spatial mean), each with its backward.  Activations are channels-last
// bf16 [B, H, W, C] — what the implicit-GEMM convolution of gemm.hip reads and writes.  All kernels are HBM-bound streaming kernels.
#include "common.h"

// ---- first convolution: img [B,3,H,W] f32 -> x = ((a*img + b) - shift) / scale (lpips ScalingLayer; a, b = 2, -1 for images in [0,1] = lpips'
//      normalize=True, 1, 0 for images already in [-1,1]) -> conv 3x3 (3 -> 64, zero padding of
//      the SCALED tensor) + bias + ReLU -> out [B,H,W,64] bf16.  Direct form on the vector ALUs (K = 27): one thread per (pixel, 4 output channels).
template <typename OT>
__global__ __launch_bounds__(256) void vgg_conv1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w /* [64][3][3][3] (co, ci, kh, kw) */,
                                                            const float* __restrict__ bias, const float* __restrict__ shift, const float* __restrict__ scale,
                                                            float a_in, float b_in, int B, int H, int W, uint16_t* __restrict__ out) {
  __shared__ float s_w[64 * 27];
  __shared__ float s_b[64];
  for (int e = threadIdx.x; e < 64 * 27; e += 256) s_w[e] = w[e];
  if (threadIdx.x < 64) s_b[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (pixel, group of 4 channels): 16 groups per pixel
  const int64_t P = (int64_t)B * H * W;
  if (idx >= P * 16) return;
  const int g = (int)(idx & 15);
  const int64_t p = idx >> 4;
  const int b = (int)(p / ((int64_t)H * W));
  const int hw = (int)(p - (int64_t)b * H * W), h = hw / W, x = hw - h * W;
  float in[27];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
    const float sh = shift[ci], inv = 1.f / scale[ci];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hh = h + kh - 1, ww = x + kw - 1;
        float v = 0.f;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = ((a_in * img[(((int64_t)b * 3 + ci) * H + hh) * W + ww] + b_in) - sh) * inv;
        in[ci * 9 + kh * 3 + kw] = v;
      }
  }
  float o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int co = g * 4 + q;
    float a = s_b[co];
#pragma unroll
    for (int e = 0; e < 27; ++e) a = fmaf(s_w[co * 27 + e], in[e], a);
    o[q] = fmaxf(a, 0.f);
  }
  const u32x2 pk = {pack2<OT>(o[0], o[1]), pack2<OT>(o[2], o[3])};
  *reinterpret_cast<u32x2*>(out + p * 64 + g * 4) = pk;
}

// backward of the above with respect to the image: gpre [B,H,W,64] bf16 = gradient at the convolution output BEFORE the ReLU (already masked)
//   dimg[b,ci,h,w] = (a / scale[ci]) * sum over (kh,kw,co) of gpre[b, h-kh+1, w-kw+1, co] * w[co][ci][kh][kw]
// one thread per pixel (all 3 channels)
template <typename OT>
__global__ __launch_bounds__(256) void vgg_conv1_bwd_kernel(const uint16_t* __restrict__ gpre, const float* __restrict__ w, const float* __restrict__ scale,
                                                            float a_in, int B, int H, int W, float* __restrict__ dimg) {
  __shared__ float s_w[64 * 27];
  for (int e = threadIdx.x; e < 64 * 27; e += 256) s_w[e] = w[e];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t P = (int64_t)B * H * W;
  if (p >= P) return;
  const int b = (int)(p / ((int64_t)H * W));
  const int hw = (int)(p - (int64_t)b * H * W), h = hw / W, x = hw - h * W;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int kh = 0; kh < 3; ++kh) {
    const int hh = h - kh + 1;
    if (hh < 0 || hh >= H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = x - kw + 1;
      if (ww < 0 || ww >= W) continue;
      const uint16_t* gp = gpre + (((int64_t)b * H + hh) * W + ww) * 64;
#pragma unroll 8
      for (int c8 = 0; c8 < 8; ++c8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(gp + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gv = unpack1<OT>((uint16_t)((v[e >> 1] >> (16 * (e & 1))) & 0xffffu));
          const int co = c8 * 8 + e;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) acc[ci] = fmaf(gv, s_w[co * 27 + ci * 9 + kh * 3 + kw], acc[ci]);
        }
      }
    }
  }
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) dimg[(((int64_t)b * 3 + ci) * H + h) * W + x] = acc[ci] * (a_in / scale[ci]);
}

// ---- 2x2 max-pool, stride 2, channels-last bf16.  One thread per (output pixel, 8 channels).
template <typename OT>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const uint16_t* __restrict__ x, int B, int H, int W, int C, uint16_t* __restrict__ y) {
  const int Ho = H >> 1, Wo = W >> 1, c8n = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * c8n) return;
  const int c8 = (int)(idx % c8n);
  const int64_t po = idx / c8n;
  const int wo = (int)(po % Wo), ho = (int)((po / Wo) % Ho), b = (int)(po / ((int64_t)Wo * Ho));
  const uint16_t* p00 = x + ((((int64_t)b * H + 2 * ho) * W + 2 * wo) * C) + c8 * 8;
  const u32x4 a = *reinterpret_cast<const u32x4*>(p00), bq = *reinterpret_cast<const u32x4*>(p00 + C);
  const u32x4 c = *reinterpret_cast<const u32x4*>(p00 + (int64_t)W * C), d = *reinterpret_cast<const u32x4*>(p00 + (int64_t)W * C + C);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t r = 0;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int sh = 16 * hlf;
      const float va = unpack1<OT>((uint16_t)(a[e] >> sh)), vb = unpack1<OT>((uint16_t)(bq[e] >> sh));
      const float vc = unpack1<OT>((uint16_t)(c[e] >> sh)), vd = unpack1<OT>((uint16_t)(d[e] >> sh));
      const float m = fmaxf(fmaxf(va, vb), fmaxf(vc, vd));
      r |= (uint32_t)pack1<OT>(m) << sh;
    }
    o[e] = r;
  }
  *reinterpret_cast<u32x4*>(y + po * C + c8 * 8) = o;
}

// backward: gpre_x[b,h,w,c] = (x > 0) * ( (x is the FIRST maximum of its 2x2 window, scan order (0,0),(0,1),(1,0),(1,1) — torch's max_pool2d tie
// rule) ? gy[b,h/2,w/2,c] : 0  +  add[b,h,w,c] )      add = gradient from the LPIPS head at this (pre-pool, post-ReLU) activation, optional.
// One thread per (output pixel, 8 channels), writing the four input pixels.
template <typename OT>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gy, const uint16_t* __restrict__ add,
                                                           int B, int H, int W, int C, uint16_t* __restrict__ gx) {
  const int Ho = H >> 1, Wo = W >> 1, c8n = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * c8n) return;
  const int c8 = (int)(idx % c8n);
  const int64_t po = idx / c8n;
  const int wo = (int)(po % Wo), ho = (int)((po / Wo) % Ho), b = (int)(po / ((int64_t)Wo * Ho));
  const int64_t off[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
  const int64_t base = ((((int64_t)b * H + 2 * ho) * W + 2 * wo) * C) + c8 * 8;
  u32x4 xv[4], av[4], ov[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xv[q] = *reinterpret_cast<const u32x4*>(x + base + off[q]);
    av[q] = add ? *reinterpret_cast<const u32x4*>(add + base + off[q]) : (u32x4){0u, 0u, 0u, 0u};
  }
  const u32x4 g = *reinterpret_cast<const u32x4*>(gy + po * C + c8 * 8);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int sh = 16 * hlf;
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = unpack1<OT>((uint16_t)(xv[q][e] >> sh));
      int arg = 0;
      float m = v[0];
#pragma unroll
      for (int q = 1; q < 4; ++q) if (v[q] > m) { m = v[q]; arg = q; }
      const float gv = unpack1<OT>((uint16_t)(g[e] >> sh));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t = (q == arg ? gv : 0.f) + unpack1<OT>((uint16_t)(av[q][e] >> sh));
        r[q] |= (uint32_t)pack1<OT>(v[q] > 0.f ? t : 0.f) << sh;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ov[q][e] = r[q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(gx + base + off[q]) = ov[q];
}

// ---- LPIPS head of one VGG slice (lpips/__init__.py normalize_tensor + lpips.py LPIPS.forward): features f [2B, h, w, C] bf16, images 0..B-1 = the
// reference inputs, B..2B-1 = the reconstructions.  Per pixel p of image b:  n0 = f0 / (||f0|| + 1e-10), n1 = f1 / (||f1|| + 1e-10),
//   val[b, p] = sum_c lin[c] * (n0_c - n1_c)^2 ;   out[b] += mean_p val[b, p]   (done by lpips_head_reduce_kernel: fixed order, no atomics).
// One wave per pixel; C in {64, 128, 256, 512}: C / 64 channels per lane.
template <typename OT, int CPL>
__global__ __launch_bounds__(256) void lpips_head_fwd_kernel(const uint16_t* __restrict__ f, const float* __restrict__ lin, int B, int64_t HW,
                                                             float* __restrict__ val /* [B*HW] */) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= (int64_t)B * HW) return;
  constexpr int C = CPL * 64;
  const uint16_t* f0 = f + p * C + lane * CPL;
  const uint16_t* f1 = f + (p + (int64_t)B * HW) * C + lane * CPL;
  float a[CPL], b[CPL], s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { a[i] = unpack1<OT>(f0[i]); b[i] = unpack1<OT>(f1[i]); s0 = fmaf(a[i], a[i], s0); s1 = fmaf(b[i], b[i], s1); }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const float i0 = 1.f / (sqrtf(s0) + 1e-10f), i1 = 1.f / (sqrtf(s1) + 1e-10f);
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { const float u = a[i] * i0 - b[i] * i1; v = fmaf(lin[lane * CPL + i] * u, u, v); }
  v = wave_sum(v);
  if (lane == 0) val[p] = v;
}

// out[b] (+)= (1 / HW) * sum_p val[b, p]   — one workgroup per image, fixed summation order
__global__ __launch_bounds__(256) void lpips_head_reduce_kernel(const float* __restrict__ val, int64_t HW, float* __restrict__ out, int accumulate) {
  __shared__ float s_part[4];
  const float* v = val + (int64_t)blockIdx.x * HW;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < HW; i += 256) acc += v[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)HW;
    out[blockIdx.x] = accumulate ? out[blockIdx.x] + r : r;
  }
}

// backward with respect to the reconstruction features f1 only (the inputs carry no gradient, the lin / VGG weights are frozen):
//   u = n0 - n1 ; g_c = -2 lin_c u_c * gout[b] / HW ;  df1_k = g_k * i1 - (sum_c g_c f1_c) * f1_k * i1^2 / ||f1||     (0 where ||f1|| = 0)
template <typename OT, int CPL>
__global__ __launch_bounds__(256) void lpips_head_bwd_kernel(const uint16_t* __restrict__ f, const float* __restrict__ lin, const float* __restrict__ gout,
                                                             int B, int64_t HW, uint16_t* __restrict__ df1 /* [B*HW, C] */) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= (int64_t)B * HW) return;
  constexpr int C = CPL * 64;
  const uint16_t* f0 = f + p * C + lane * CPL;
  const uint16_t* f1 = f + (p + (int64_t)B * HW) * C + lane * CPL;
  float a[CPL], b[CPL], s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { a[i] = unpack1<OT>(f0[i]); b[i] = unpack1<OT>(f1[i]); s0 = fmaf(a[i], a[i], s0); s1 = fmaf(b[i], b[i], s1); }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  const float n1 = sqrtf(s1);
  const float i0 = 1.f / (sqrtf(s0) + 1e-10f), i1 = 1.f / (n1 + 1e-10f);
  const float go = gout[p / HW] / (float)HW;
  float g[CPL], dot = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) { const float u = a[i] * i0 - b[i] * i1; g[i] = -2.f * lin[lane * CPL + i] * u * go; dot = fmaf(g[i], b[i], dot); }
  dot = wave_sum(dot);
  const float k2 = n1 > 0.f ? dot * i1 * i1 / n1 : 0.f;
  uint16_t* o = df1 + p * C + lane * CPL;
#pragma unroll
  for (int i = 0; i < CPL; ++i) o[i] = pack1<OT>(g[i] * i1 - k2 * b[i]);
}

extern "C" int enh_vgg_conv1(const float* img, const float* w, const float* bias, const float* shift, const float* scale, int normalize, int B, int H, int W,
                             enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_vgg_conv1");
  ENH_REQUIRE(img && w && bias && shift && scale && out && B > 0 && H > 0 && W > 0, ENH_E_BADARG, "enh_vgg_conv1: bad argument");
  const int64_t n = (int64_t)B * H * W * 16;
  ENH_DT_DISPATCH(dtype, (vgg_conv1_fwd_kernel<OT><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(img, w, bias, shift, scale, normalize ? 2.f : 1.f, normalize ? -1.f : 0.f, B, H, W, out)));
  return enh_check_launch("enh_vgg_conv1");
}

extern "C" int enh_vgg_conv1_backward(const enh_h16* gpre, const float* w, const float* scale, int normalize, int B, int H, int W, float* dimg, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_vgg_conv1_backward");
  ENH_REQUIRE(gpre && w && scale && dimg && B > 0 && H > 0 && W > 0, ENH_E_BADARG, "enh_vgg_conv1_backward: bad argument");
  const int64_t n = (int64_t)B * H * W;
  ENH_DT_DISPATCH(dtype, (vgg_conv1_bwd_kernel<OT><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(gpre, w, scale, normalize ? 2.f : 1.f, B, H, W, dimg)));
  return enh_check_launch("enh_vgg_conv1_backward");
}

extern "C" int enh_maxpool2_nhwc_h16(const enh_h16* x, int B, int H, int W, int C, enh_h16* y, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_maxpool2_nhwc_h16");
  ENH_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, ENH_E_BADARG, "enh_maxpool2_nhwc_h16: bad argument");
  ENH_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, ENH_E_SHAPE, "enh_maxpool2_nhwc_h16: H, W even and C %% 8 == 0");
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  ENH_DT_DISPATCH(dtype, (maxpool2_fwd_kernel<OT><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, B, H, W, C, y)));
  return enh_check_launch("enh_maxpool2_nhwc_h16");
}

extern "C" int enh_maxpool2_nhwc_h16_backward(const enh_h16* x, const enh_h16* gy, const enh_h16* add, int B, int H, int W, int C, enh_h16* gx, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_maxpool2_nhwc_h16_backward");
  ENH_REQUIRE(x && gy && gx && B > 0 && H > 0 && W > 0 && C > 0, ENH_E_BADARG, "enh_maxpool2_nhwc_h16_backward: bad argument");
  ENH_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0, ENH_E_SHAPE, "enh_maxpool2_nhwc_h16_backward: H, W even and C %% 8 == 0");
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  ENH_DT_DISPATCH(dtype, (maxpool2_bwd_kernel<OT><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, gy, add, B, H, W, C, gx)));
  return enh_check_launch("enh_maxpool2_nhwc_h16_backward");
}

extern "C" int enh_lpips_head(const enh_h16* feat, const float* lin, int B, int64_t HW, int C, float* val_ws, float* out, int accumulate, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_lpips_head");
  ENH_REQUIRE(feat && lin && val_ws && out && B > 0 && HW > 0, ENH_E_BADARG, "enh_lpips_head: bad argument");
  ENH_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, ENH_E_SHAPE, "enh_lpips_head: C must be 64, 128, 256 or 512 (the VGG16 slices)");
  const int64_t P = (int64_t)B * HW;
  const unsigned grid = (unsigned)((P + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) ENH_DT_DISPATCH(dtype, (lpips_head_fwd_kernel<OT, 1><<<grid, 256, 0, s>>>(feat, lin, B, HW, val_ws)));
  else if (C == 128) ENH_DT_DISPATCH(dtype, (lpips_head_fwd_kernel<OT, 2><<<grid, 256, 0, s>>>(feat, lin, B, HW, val_ws)));
  else if (C == 256) ENH_DT_DISPATCH(dtype, (lpips_head_fwd_kernel<OT, 4><<<grid, 256, 0, s>>>(feat, lin, B, HW, val_ws)));
  else ENH_DT_DISPATCH(dtype, (lpips_head_fwd_kernel<OT, 8><<<grid, 256, 0, s>>>(feat, lin, B, HW, val_ws)));
  lpips_head_reduce_kernel<<<(unsigned)B, 256, 0, s>>>(val_ws, HW, out, accumulate);
  return enh_check_launch("enh_lpips_head");
}

extern "C" int enh_lpips_head_backward(const enh_h16* feat, const float* lin, const float* gout, int B, int64_t HW, int C, enh_h16* dfeat1, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_lpips_head_backward");
  ENH_REQUIRE(feat && lin && gout && dfeat1 && B > 0 && HW > 0, ENH_E_BADARG, "enh_lpips_head_backward: bad argument");
  ENH_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, ENH_E_SHAPE, "enh_lpips_head_backward: C must be 64, 128, 256 or 512");
  const int64_t P = (int64_t)B * HW;
  const unsigned grid = (unsigned)((P + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) ENH_DT_DISPATCH(dtype, (lpips_head_bwd_kernel<OT, 1><<<grid, 256, 0, s>>>(feat, lin, gout, B, HW, dfeat1)));
  else if (C == 128) ENH_DT_DISPATCH(dtype, (lpips_head_bwd_kernel<OT, 2><<<grid, 256, 0, s>>>(feat, lin, gout, B, HW, dfeat1)));
  else if (C == 256) ENH_DT_DISPATCH(dtype, (lpips_head_bwd_kernel<OT, 4><<<grid, 256, 0, s>>>(feat, lin, gout, B, HW, dfeat1)));
  else ENH_DT_DISPATCH(dtype, (lpips_head_bwd_kernel<OT, 8><<<grid, 256, 0, s>>>(feat, lin, gout, B, HW, dfeat1)));
  return enh_check_launch("enh_lpips_head_backward");
}
