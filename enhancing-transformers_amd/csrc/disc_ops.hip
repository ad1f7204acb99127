// disc_ops.hip — gfx950 versions of the reference's ONLY native code: the two StyleGAN2-discriminator ops
//   fused_bias_act  (reference enhancing/losses/op/fused_bias_act_kernel.cu:18-65, bound at fused_bias_act.cpp:17-31)
//   upfirdn2d       (reference enhancing/losses/op/upfirdn2d_kernel.cu:49-207, bound at upfirdn2d.cpp:17-30)
// Both are HBM-bound streaming kernels: 16-byte accesses along the contiguous (W) axis, one pass over the data.  They serve the layer classes' own
// NCHW f32 API; the discriminator's forward runs on the channels-last bf16 kernels further down (blur, gate, layout change, minibatch standard
// deviation) and on the implicit-GEMM convolutions of conv_igemm.hip.
#include "common.h"
#include <stdlib.h>

// y = act(x + b[(i / step_b) % size_b]) * scale      (act = leaky-relu, reference "act*10+grad" cases 30 / 31)
//   grad == 0: act(v) = v > 0 ? v : alpha * v
//   grad == 1: y = (x + b) * (ref > 0 ? 1 : alpha) * scale   — first and second derivative paths, gated by the saved OUTPUT ref
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ ref,
                                      float* __restrict__ y, int64_t n, int64_t step_b, int size_b, int grad, float alpha,
                                      float scale) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
  const bool full = i0 + 3 < n;
  if (full) {
    const float4 t = *reinterpret_cast<const float4*>(x + i0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    if (grad) { const float4 q = *reinterpret_cast<const float4*>(ref + i0); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
  } else {
    for (int k = 0; k < 4; ++k) { v[k] = i0 + k < n ? x[i0 + k] : 0.f; if (grad) r[k] = i0 + k < n ? ref[i0 + k] : 0.f; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = v[k];
    if (b) t += b[((i0 + k) / step_b) % size_b];
    if (grad == 0) t = (t > 0.f ? t : alpha * t) * scale;
    else t = t * (r[k] > 0.f ? 1.f : alpha) * scale;
    v[k] = t;
  }
  if (full) *reinterpret_cast<float4*>(y + i0) = make_float4(v[0], v[1], v[2], v[3]);
  else for (int k = 0; k < 4 && i0 + k < n; ++k) y[i0 + k] = v[k];
}


// upfirdn2d on [major, in_h, in_w] planes (minor = 1, as the reference reshapes NCHW at upfirdn2d.py:100):
//   out[m][oy][ox] = sum_{ky,kx} kernel[kh-1-ky][kw-1-kx] * P[oy*down_y + ky][ox*down_x + kx],
//   P = zero-upsampled (up_x, up_y) input shifted by (pad_x0, pad_y0), zero outside   (upfirdn2d.py:168-209)
__global__ void upfirdn2d_kernel(const float* __restrict__ in, const float* __restrict__ kernel, float* __restrict__ out,
                                 int64_t major, int in_h, int in_w, int out_h, int out_w, int kh, int kw, int up_x, int up_y,
                                 int down_x, int down_y, int pad_x0, int pad_y0) {
  __shared__ float s_k[64];
  if (threadIdx.x < kh * kw) s_k[threadIdx.x] = kernel[(kh - 1 - threadIdx.x / kw) * kw + (kw - 1 - threadIdx.x % kw)];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = major * out_h * out_w;
  if (idx >= total) return;
  const int ox = (int)(idx % out_w);
  const int oy = (int)((idx / out_w) % out_h);
  const int64_t m = idx / ((int64_t)out_w * out_h);
  const float* plane = in + m * (int64_t)in_h * in_w;
  float acc = 0.f;
  for (int ky = 0; ky < kh; ++ky) {
    const int py = oy * down_y + ky - pad_y0;   // coordinate in the upsampled image
    if (py < 0 || py % up_y) continue;
    const int iy = py / up_y;
    if (iy >= in_h) continue;
    for (int kx = 0; kx < kw; ++kx) {
      const int px = ox * down_x + kx - pad_x0;
      if (px < 0 || px % up_x) continue;
      const int ix = px / up_x;
      if (ix >= in_w) continue;
      acc = fmaf(s_k[ky * kw + kx], plane[(int64_t)iy * in_w + ix], acc);
    }
  }
  out[idx] = acc;
}

// ---- register-blocked blur and split channel sums (validated on MI355X in round 2: same parity tests, adversarial step 68.8 -> 89.0 images/s) ------
// profiles/r01_adv_step_kernel_stats.csv: the one-output-per-thread blur was 24 % and the per-channel reduction 6 % of the adversarial training
// step — the first did 16 scalar loads per output, the second ran C (128..512) workgroups over a tensor of hundreds of MB.
//
// upfirdn2d with up = down = 1 (every use in the discriminator: Blur forward and both of its derivatives): a thread produces a
// 2 (y) x 4 (x) output block from a (kh+1) x (kw+3) input window held in registers -> (kh+1)(kw+3)/8 loads per output instead of kh*kw.
template <int KH, int KW>
__global__ __launch_bounds__(256) void upfirdn2d_unit_kernel(const float* __restrict__ in, const float* __restrict__ kernel,
                                                             float* __restrict__ out, int64_t major, int in_h, int in_w, int out_h,
                                                             int out_w, int pad_x0, int pad_y0) {
  __shared__ float s_k[KH * KW];
  if (threadIdx.x < KH * KW) s_k[threadIdx.x] = kernel[(KH - 1 - threadIdx.x / KW) * KW + (KW - 1 - threadIdx.x % KW)];
  __syncthreads();
  const int bx = (out_w + 3) >> 2, by = (out_h + 1) >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= major * by * bx) return;
  const int ox = (int)(idx % bx) * 4;
  const int oy = (int)((idx / bx) % by) * 2;
  const int64_t m = idx / ((int64_t)bx * by);
  const float* plane = in + m * (int64_t)in_h * in_w;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int r = 0; r < KH + 1; ++r) {
    const int iy = oy + r - pad_y0;
    float row[KW + 3];
    const bool yok = iy >= 0 && iy < in_h;
#pragma unroll
    for (int c = 0; c < KW + 3; ++c) {
      const int ix = ox + c - pad_x0;
      row[c] = (yok && ix >= 0 && ix < in_w) ? plane[(int64_t)iy * in_w + ix] : 0.f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int ky = r - dy;  // output row oy + dy uses input row (oy + dy) + ky - pad
      if (ky < 0 || ky >= KH) continue;
#pragma unroll
      for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) acc[dy][dx] = fmaf(s_k[ky * KW + kx], row[dx + kx], acc[dy][dx]);
    }
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    if (oy + dy >= out_h) continue;
    float* o = out + (m * out_h + oy + dy) * (int64_t)out_w + ox;
    if (ox + 3 < out_w && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(acc[dy][0], acc[dy][1], acc[dy][2], acc[dy][3]);
    } else {
#pragma unroll
      for (int dx = 0; dx < 4; ++dx)
        if (ox + dx < out_w) o[dx] = acc[dy][dx];
    }
  }
}

// channel sums with the (batch, inner) space split over blockIdx.y: 16-byte loads when the rows are 16-byte aligned
__global__ __launch_bounds__(256) void channel_sum_split_kernel(const float* __restrict__ x, int C, int64_t inner, int nchunk, int64_t chunk,
                                                                float* __restrict__ out) {
  __shared__ float s_part[4];
  const int c = blockIdx.x;
  const int b = blockIdx.y / nchunk, ch = blockIdx.y % nchunk;
  const int64_t i0 = (int64_t)ch * chunk;
  int64_t i1 = i0 + chunk;
  if (i1 > inner) i1 = inner;
  const float* p = x + ((int64_t)b * C + c) * inner;
  float acc = 0.f;
  if ((inner & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {  // chunk is a multiple of 4 (launcher)
    for (int64_t i = i0 + (int64_t)threadIdx.x * 4; i < i1; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(p + i);
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) acc += p[i];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[c], (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

extern "C" int enh_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, int64_t n, int64_t step_b, int size_b,
                                  int act, int grad, float alpha, float scale, void* stream) {
  ENH_REQUIRE(x && y && n > 0, ENH_E_BADARG, "enh_fused_bias_act: bad argument");
  ENH_REQUIRE(act == 3 && (grad == 0 || grad == 1), ENH_E_SHAPE, "enh_fused_bias_act: only leaky-relu (act = 3), grad 0 or 1, is used by the reference");
  ENH_REQUIRE(grad == 0 || ref, ENH_E_BADARG, "enh_fused_bias_act: grad = 1 needs the saved output `ref`");
  ENH_REQUIRE(!bias || (step_b > 0 && size_b > 0), ENH_E_BADARG, "enh_fused_bias_act: bias needs step_b / size_b");
  const int64_t n4 = (n + 3) / 4;
  fused_bias_act_kernel<<<(int)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, bias, ref, y, n, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1,
                                                                               grad, alpha, scale);
  return enh_check_launch("enh_fused_bias_act");
}

extern "C" int enh_channel_sum_f32(const float* x, int B, int C, int64_t inner, float* out, int accumulate, void* stream) {
  ENH_REQUIRE(x && out && B > 0 && C > 0 && inner > 0, ENH_E_BADARG, "enh_channel_sum_f32: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = enh_zero_f32_launch(out, C, s);
    if (rc) return rc;
  }
  int64_t chunk = (inner + 63) / 64;               // up to 64 slices of the inner axis per (batch, channel) row ...
  if (chunk < 4096) chunk = 4096;                   // ... but never less than 16 KiB of work per workgroup
  chunk = (chunk + 3) / 4 * 4;
  const int nchunk = (int)((inner + chunk - 1) / chunk);
  ENH_REQUIRE((int64_t)B * nchunk <= 65535, ENH_E_SHAPE, "enh_channel_sum_f32: grid too large");
  channel_sum_split_kernel<<<dim3(C, B * nchunk), 256, 0, s>>>(x, C, inner, nchunk, chunk, out);
  return enh_check_launch("enh_channel_sum_f32");
}

extern "C" int enh_upfirdn2d(const float* in, const float* kernel, float* out, int64_t major, int in_h, int in_w, int kh, int kw, int up_x,
                             int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  ENH_REQUIRE(in && kernel && out && major > 0 && in_h > 0 && in_w > 0, ENH_E_BADARG, "enh_upfirdn2d: bad argument");
  ENH_REQUIRE(kh > 0 && kw > 0 && kh * kw <= 64 && up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, ENH_E_SHAPE, "enh_upfirdn2d: kernel up to 64 taps, positive up / down factors");
  ENH_REQUIRE(pad_x0 >= 0 && pad_x1 >= 0 && pad_y0 >= 0 && pad_y1 >= 0, ENH_E_SHAPE, "enh_upfirdn2d: negative pads (cropping) are not used by the reference's Blur and are not supported");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  ENH_REQUIRE(out_h > 0 && out_w > 0, ENH_E_SHAPE, "enh_upfirdn2d: empty output");
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh == 4 && kw == 4) {
    const int64_t blocks = major * ((out_h + 1) / 2) * ((out_w + 3) / 4);
    upfirdn2d_unit_kernel<4, 4><<<(unsigned)((blocks + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, kernel, out, major, in_h, in_w, out_h, out_w,
                                                                                                  pad_x0, pad_y0);
    return enh_check_launch("enh_upfirdn2d");
  }
  const int64_t total = major * out_h * out_w;
  upfirdn2d_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, kernel, out, major, in_h, in_w, out_h, out_w, kh, kw, up_x, up_y,
                                                                             down_x, down_y, pad_x0, pad_y0);
  return enh_check_launch("enh_upfirdn2d");
}

// =================================================================================================
// Channels-last 16-bit element-wise kernels of the implicit-GEMM discriminator path (activations [B,H,W,C] bf16 | fp16: templates over the operand type tag OT,
// common.h; C % 8 == 0).
// Each is linear in its data argument and closed under differentiation (the derivative of every one is another launch of
// the same kernel with other arguments), which is what lets the R1 penalty differentiate through the backward pass.
// =================================================================================================
template <typename OT>
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = unpack_lo<OT>(v[k]); f[2 * k + 1] = unpack_hi<OT>(v[k]); }
}
template <typename OT>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return (u32x4){pack2<OT>(f[0], f[1]), pack2<OT>(f[2], f[3]), pack2<OT>(f[4], f[5]), pack2<OT>(f[6], f[7])};
}

// FIR filter with unit up / down factors (Blur, reference enhancing/losses/layers.py:140-160 -> upfirdn2d(input, kernel, pad)):
//   out[b,oy,ox,c] = sum_{i,j} w(i,j) * x[b, oy + i - pad_y0, ox + j - pad_x0, c],  w(i,j) = kernel[kh-1-i][kw-1-j]  (flip = 0: upfirdn2d's convention)
//                                                                                   or   kernel[i][j]            (flip = 1: its adjoint)
// one thread = 8 channels x 4 consecutive output columns of one row: (kw + 3) * kh 16-byte loads for 4 outputs
template <typename OT>
__global__ __launch_bounds__(256) void blur_nhwc_kernel(const uint16_t* __restrict__ x, const float* __restrict__ kernel, uint16_t* __restrict__ out,
                                                        int B, int H, int W, int C, int Ho, int Wo, int kh, int kw, int pad_y0, int pad_x0, int flip) {
  __shared__ float s_k[64];
  if ((int)threadIdx.x < kh * kw) {
    const int i = threadIdx.x / kw, j = threadIdx.x % kw;
    s_k[threadIdx.x] = flip ? kernel[i * kw + j] : kernel[(kh - 1 - i) * kw + (kw - 1 - j)];
  }
  __syncthreads();
  const int c8n = C >> 3, wq = (Wo + 3) >> 2;
  const int64_t total = (int64_t)B * Ho * wq * c8n;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % c8n);
  int64_t q = idx / c8n;
  const int xq = (int)(q % wq); q /= wq;
  const int oy = (int)(q % Ho);
  const int64_t b = q / Ho;
  const int ox0 = xq * 4;
  float acc[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[o][k] = 0.f;
  for (int i = 0; i < kh; ++i) {
    const int iy = oy + i - pad_y0;
    if (iy < 0 || iy >= H) continue;
    const uint16_t* row = x + ((b * H + iy) * (int64_t)W) * C + c8 * 8;
    for (int jj = 0; jj < kw + 3; ++jj) {
      const int ix = ox0 + jj - pad_x0;
      if (ix < 0 || ix >= W) continue;
      float f[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(row + (int64_t)ix * C), f);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int j = jj - o;
        if (j >= 0 && j < kw) {
          const float wv = s_k[i * kw + j];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[o][k] += wv * f[k];
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
    if (ox0 + o < Wo) *reinterpret_cast<u32x4*>(out + ((b * Ho + oy) * (int64_t)Wo + ox0 + o) * C + c8 * 8) = pack8<OT>(acc[o]);
}

// The 4 x 4 case (every Blur of the discriminator) as a column march: one thread = 8 channels x 2 output columns x a strip of RS output rows.  Each input
// row of the strip is loaded ONCE (5 16-byte loads, requested one row ahead) and unpacked once; it contributes to the up to four output rows that are in
// flight (a ring of accumulators whose slot is a compile-time constant: the march is unrolled in groups of four rows).  16 loads per output (4 per output
// with the one-row kernel's 4-column reuse) become 2.5 * (RS + 3) / RS.  A given output receives its 16 products in the same order as in
// blur_nhwc_kernel (rows ascending, columns ascending), so the two kernels agree bit for bit.
// Lab (B = 16, 256^2 x 128 / 128^2 x 256 / 64^2 x 512, profiles/r04_conv_layers.txt): one-row kernel 245 / 123 / 65 us; this form 117 / 60 / 35 us
// (4.6 TB/s of input + output); 4 columns per thread 153 / 86 / 49 (282 registers: one wave per SIMD; 285 when forced to two waves, it spills);
// loads issued unconditionally from clamped addresses and masked instead of predicated: 130 / 64 / 38; 1 column per thread at three waves: 321.
template <typename OT, int RS>
__global__ __launch_bounds__(256, 2) void blur4x4_nhwc_kernel(const uint16_t* __restrict__ x, const float* __restrict__ kernel, uint16_t* __restrict__ out,
                                                              int B, int H, int W, int C, int Ho, int Wo, int pad_y0, int pad_x0, int flip) {
  static_assert((RS + 3) % 4 == 0, "the march runs in groups of four input rows");
  constexpr int NC = 2;
  float w[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w[i][j] = flip ? kernel[i * 4 + j] : kernel[(3 - i) * 4 + (3 - j)];
  const int c8n = C >> 3, wq = (Wo + NC - 1) / NC, ns = (Ho + RS - 1) / RS;
  const int64_t total = (int64_t)B * ns * wq * c8n;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % c8n);
  int64_t q = idx / c8n;
  const int xq = (int)(q % wq); q /= wq;
  const int st = (int)(q % ns);
  const int64_t b = q / ns;
  const int ox0 = xq * NC, oy0 = st * RS;
  const uint16_t* xb = x + b * H * (int64_t)W * C + c8 * 8;
  uint16_t* ob = out + b * Ho * (int64_t)Wo * C + c8 * 8;
  int coff[NC + 3];        // element offset of input column jj inside a row, -1 outside the image
#pragma unroll
  for (int jj = 0; jj < NC + 3; ++jj) {
    const int ix = ox0 + jj - pad_x0;
    coff[jj] = (ix >= 0 && ix < W) ? ix * C : -1;
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_row = [&](int t, u32x4 (&r)[NC + 3]) {
    const int iy = oy0 + t - pad_y0;
    const bool rok = iy >= 0 && iy < H && t < RS + 3;
    const uint16_t* row = xb + (int64_t)(rok ? iy : 0) * W * C;
#pragma unroll
    for (int jj = 0; jj < NC + 3; ++jj) r[jj] = (rok && coff[jj] >= 0) ? *reinterpret_cast<const u32x4*>(row + coff[jj]) : zero4;
  };
  float acc[4][NC][8];   // [ring slot = output row & 3][output column][channel]
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
    for (int o = 0; o < NC; ++o)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[s_][o][k] = 0.f;
  u32x4 cur[NC + 3], nxt[NC + 3];
  load_row(0, cur);
  for (int t4 = 0; t4 < RS + 3; t4 += 4) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int t = t4 + tt;                       // input row oy0 - pad_y0 + t feeds output rows oy0 + t - i, i = 0..3 (ring slot (tt - i) & 3)
      load_row(t + 1, nxt);
#pragma unroll
      for (int jj = 0; jj < NC + 3; ++jj) {
        float f[8];
        unpack8<OT>(cur[jj], f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int o = 0; o < NC; ++o) {
            const int j = jj - o;
            if (j >= 0 && j < 4) {
#pragma unroll
              for (int k = 0; k < 8; ++k) acc[(tt - i) & 3][o][k] += w[i][j] * f[k];
            }
          }
      }
      // output row oy0 + t - 3 is complete (its i = 3 term was this row): store it and hand its slot to output row oy0 + t + 1
      const int orow = t - 3;
      if (orow >= 0 && orow < RS && oy0 + orow < Ho) {
        uint16_t* dst = ob + (int64_t)(oy0 + orow) * Wo * C + (int64_t)ox0 * C;
#pragma unroll
        for (int o = 0; o < NC; ++o)
          if (ox0 + o < Wo) *reinterpret_cast<u32x4*>(dst + (int64_t)o * C) = pack8<OT>(acc[(tt - 3) & 3][o]);
      }
#pragma unroll
      for (int o = 0; o < NC; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[(tt - 3) & 3][o][k] = 0.f;
#pragma unroll
      for (int jj = 0; jj < NC + 3; ++jj) cur[jj] = nxt[jj];
    }
  }
}

static int g_blur_variant = 0;   // 0 = per shape, 1 = the one-row kernel everywhere (A/B, tests)
extern "C" int enh_blur_set_kernel(int variant) {
  ENH_REQUIRE(variant == 0 || variant == 1, ENH_E_BADARG, "enh_blur_set_kernel: variant must be 0 (auto) or 1 (one-row kernel)");
  g_blur_variant = variant;
  return ENH_OK;
}

extern "C" int enh_blur_nhwc_h16(const enh_h16* x, const float* kernel, int B, int H, int W, int C, int kh, int kw, int pad_y0, int pad_y1, int pad_x0,
                                  int pad_x1, int flip, enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_blur_nhwc_h16");
  ENH_REQUIRE(x && kernel && out && B > 0 && H > 0 && W > 0, ENH_E_BADARG, "enh_blur_nhwc_h16: bad argument");
  ENH_REQUIRE(C > 0 && C % 8 == 0 && kh > 0 && kw > 0 && kh * kw <= 64, ENH_E_SHAPE, "enh_blur_nhwc_h16: C must be a multiple of 8 and the kernel at most 64 taps");
  const int Ho = H + pad_y0 + pad_y1 - kh + 1, Wo = W + pad_x0 + pad_x1 - kw + 1;
  ENH_REQUIRE(Ho > 0 && Wo > 0, ENH_E_SHAPE, "enh_blur_nhwc_h16: empty output");
  if (kh == 4 && kw == 4 && g_blur_variant == 0 && (int64_t)W * C < (1ll << 31)) {
    // strips of 13 rows when that still gives every CU several workgroups, else 5
    const int64_t cols = (int64_t)B * ((Wo + 1) / 2) * (C / 8);
    const int64_t t13 = cols * ((Ho + 12) / 13), t5 = cols * ((Ho + 4) / 5);
    if (t13 >= 256ll * 256 * 4)
      ENH_DT_DISPATCH(dtype, (blur4x4_nhwc_kernel<OT, 13><<<dim3((unsigned)((t13 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, kernel, out, B, H, W, C, Ho, Wo, pad_y0, pad_x0, flip)));
    else
      ENH_DT_DISPATCH(dtype, (blur4x4_nhwc_kernel<OT, 5><<<dim3((unsigned)((t5 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, kernel, out, B, H, W, C, Ho, Wo, pad_y0, pad_x0, flip)));
    return enh_check_launch("enh_blur_nhwc_h16");
  }
  const int64_t total = (int64_t)B * Ho * ((Wo + 3) / 4) * (C / 8);
  ENH_DT_DISPATCH(dtype, (blur_nhwc_kernel<OT><<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, kernel, out, B, H, W, C, Ho, Wo, kh, kw, pad_y0, pad_x0, flip)));
  return enh_check_launch("enh_blur_nhwc_h16");
}

// y = g * (ref > 0 ? 1 : slope) * scale   (the derivative of FusedLeakyReLU through its saved OUTPUT, fused_act.py:21-45; ref == NULL: y = g * scale)
template <typename OT>
__global__ __launch_bounds__(256) void lrelu_gate_h16_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ ref, uint16_t* __restrict__ y, int64_t n8,
                                                              float slope, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float f[8], r[8];
  unpack8<OT>(*reinterpret_cast<const u32x4*>(g + i * 8), f);
  if (ref) {
    unpack8<OT>(*reinterpret_cast<const u32x4*>(ref + i * 8), r);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] *= (r[k] > 0.f ? 1.f : slope) * scale;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] *= scale;
  }
  *reinterpret_cast<u32x4*>(y + i * 8) = pack8<OT>(f);
}

extern "C" int enh_lrelu_gate_h16(const enh_h16* g, const enh_h16* ref, int64_t n, float slope, float scale, enh_h16* y, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_lrelu_gate_h16");
  ENH_REQUIRE(g && y && n > 0, ENH_E_BADARG, "enh_lrelu_gate_h16: bad argument");
  ENH_REQUIRE(n % 8 == 0, ENH_E_SHAPE, "enh_lrelu_gate_h16: n must be a multiple of 8");
  ENH_DT_DISPATCH(dtype, (lrelu_gate_h16_kernel<OT><<<dim3((unsigned)((n / 8 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(g, ref, y, n / 8, slope, scale)));
  return enh_check_launch("enh_lrelu_gate_h16");
}

// img [B,C,H,W] f32 (C <= 8) -> [B,H,W,8] bf16 with channels C..7 zero, and its adjoint ([B,H,W,8] bf16 -> [B,C,H,W] f32, the padding channels dropped)
template <typename OT>
__global__ __launch_bounds__(256) void img_to_nhwc8_kernel(const float* __restrict__ img, uint16_t* __restrict__ out, int C, int64_t HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index over B*H*W
  if (i >= total) return;
  const int64_t b = i / HW, p = i - b * HW;
  float f[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) f[c] = c < C ? img[(b * C + c) * HW + p] : 0.f;
  *reinterpret_cast<u32x4*>(out + i * 8) = pack8<OT>(f);
}
template <typename OT>
__global__ __launch_bounds__(256) void nhwc8_to_img_kernel(const uint16_t* __restrict__ src, float* __restrict__ img, int C, int64_t HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / HW, p = i - b * HW;
  float f[8];
  unpack8<OT>(*reinterpret_cast<const u32x4*>(src + i * 8), f);
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < C) img[(b * C + c) * HW + p] = f[c];
}

extern "C" int enh_img_to_nhwc8(const float* img, int B, int C, int H, int W, enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_img_to_nhwc8");
  ENH_REQUIRE(img && out && B > 0 && H > 0 && W > 0 && C > 0 && C <= 8, ENH_E_BADARG, "enh_img_to_nhwc8: bad argument (1 <= C <= 8)");
  const int64_t total = (int64_t)B * H * W;
  ENH_DT_DISPATCH(dtype, (img_to_nhwc8_kernel<OT><<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(img, out, C, (int64_t)H * W, total)));
  return enh_check_launch("enh_img_to_nhwc8");
}
extern "C" int enh_nhwc8_to_img(const enh_h16* src, int B, int C, int H, int W, float* img, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_nhwc8_to_img");
  ENH_REQUIRE(img && src && B > 0 && H > 0 && W > 0 && C > 0 && C <= 8, ENH_E_BADARG, "enh_nhwc8_to_img: bad argument (1 <= C <= 8)");
  const int64_t total = (int64_t)B * H * W;
  ENH_DT_DISPATCH(dtype, (nhwc8_to_img_kernel<OT><<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(src, img, C, (int64_t)H * W, total)));
  return enh_check_launch("enh_nhwc8_to_img");
}

// Minibatch standard deviation (reference enhancing/losses/layers.py:358-367, stddev_feat = 1) on channels-last bf16, fused with the concatenation and
// the channel padding of the final convolution's input: sample b belongs to slot b % n (n = B / group); for every position p = (h,w,c) the standard
// deviation over the `group` samples of the slot, sd_p = sqrt(var_p + 1e-8) (biased variance), is averaged over p into ONE scalar per slot:
//   out[b,h,w,0..C-1] = x[b,h,w,:] ; out[b,h,w,C] = mean_p sd_p of b's slot ; out[b,h,w,C+1..Cp-1] = 0
// One workgroup per slot (the tensor is B x 4 x 4 x 512: a few hundred KiB); fixed-order reductions, no atomics.
__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

template <typename OT>
__global__ __launch_bounds__(256) void stddev_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int group, int n, int HW, int C, int Cp) {
  __shared__ float s_red[4];
  const int slot = blockIdx.x;
  const int64_t P = (int64_t)HW * C;
  float part = 0.f;
  for (int64_t p8 = threadIdx.x; p8 < P / 8; p8 += 256) {
    float m[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = 0.f; q[k] = 0.f; }
    for (int gi = 0; gi < group; ++gi) {
      float f[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(x + ((int64_t)gi * n + slot) * P + p8 * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] += f[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] /= (float)group;
    for (int gi = 0; gi < group; ++gi) {
      float f[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(x + ((int64_t)gi * n + slot) * P + p8 * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] += (f[k] - m[k]) * (f[k] - m[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part += sqrtf(q[k] / (float)group + 1e-8f);
  }
  const float sd = block_sum_256(part, s_red) / (float)P;
  // write the group's samples: copy, the statistic, zero padding
  const int cq = Cp / 8;                       // 16-byte chunks per output pixel
  for (int gi = 0; gi < group; ++gi) {
    const int64_t b = (int64_t)gi * n + slot;
    for (int64_t i = threadIdx.x; i < (int64_t)HW * cq; i += 256) {
      const int64_t pix = i / cq;
      const int ch = (int)(i - pix * cq) * 8;
      u32x4 v;
      if (ch + 8 <= C) v = *reinterpret_cast<const u32x4*>(x + (b * HW + pix) * C + ch);
      else {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = ch + k < C ? unpack1<OT>(x[(b * HW + pix) * C + ch + k]) : (ch + k == C ? sd : 0.f);
        v = pack8<OT>(f);
      }
      *reinterpret_cast<u32x4*>(out + (b * HW + pix) * Cp + ch) = v;
    }
  }
}

// dx[b,p] = g[b,p] + (sum over the slot's samples and pixels of g[.., C]) / P * (x[b,p] - mean_p) / (group * sd_p)
template <typename OT>
__global__ __launch_bounds__(256) void stddev_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ g, uint16_t* __restrict__ dx, int group, int n,
                                                         int HW, int C, int Cp) {
  __shared__ float s_red[4];
  const int slot = blockIdx.x;
  const int64_t P = (int64_t)HW * C;
  float part = 0.f;
  for (int i = threadIdx.x; i < group * HW; i += 256) {
    const int64_t b = (int64_t)(i / HW) * n + slot;
    part += unpack1<OT>(g[(b * HW + i % HW) * Cp + C]);
  }
  const float gsd = block_sum_256(part, s_red) / (float)P;
  const int c8 = C / 8;
  for (int64_t p8 = threadIdx.x; p8 < P / 8; p8 += 256) {
    const int64_t pix = p8 / c8;
    const int ch = (int)(p8 - pix * c8) * 8;
    float m[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = 0.f; q[k] = 0.f; }
    for (int gi = 0; gi < group; ++gi) {
      float f[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(x + ((int64_t)gi * n + slot) * P + p8 * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] += f[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] /= (float)group;
    for (int gi = 0; gi < group; ++gi) {
      float f[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(x + ((int64_t)gi * n + slot) * P + p8 * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] += (f[k] - m[k]) * (f[k] - m[k]);
    }
    float coef[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) coef[k] = gsd / ((float)group * sqrtf(q[k] / (float)group + 1e-8f));
    for (int gi = 0; gi < group; ++gi) {
      const int64_t b = (int64_t)gi * n + slot;
      float f[8], gv[8];
      unpack8<OT>(*reinterpret_cast<const u32x4*>(x + b * P + p8 * 8), f);
      unpack8<OT>(*reinterpret_cast<const u32x4*>(g + (b * HW + pix) * Cp + ch), gv);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = gv[k] + coef[k] * (f[k] - m[k]);
      *reinterpret_cast<u32x4*>(dx + b * P + p8 * 8) = pack8<OT>(f);
    }
  }
}

static int stddev_check(const void* a, const void* b, int B, int HW, int C, int Cp, int group, const char* who) {
  ENH_REQUIRE(a && b && B > 0 && HW > 0 && group > 0, ENH_E_BADARG, "%s: bad argument", who);
  ENH_REQUIRE(B % group == 0 && C % 8 == 0 && Cp % 8 == 0 && Cp > C, ENH_E_SHAPE, "%s: B %% group == 0, C %% 8 == 0 and Cp > C (a multiple of 8) required", who);
  return ENH_OK;
}
extern "C" int enh_minibatch_stddev_nhwc(const enh_h16* x, int B, int HW, int C, int Cp, int group, enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_minibatch_stddev_nhwc");
  const int rc = stddev_check(x, out, B, HW, C, Cp, group, "enh_minibatch_stddev_nhwc");
  if (rc != ENH_OK) return rc;
  ENH_DT_DISPATCH(dtype, (stddev_fwd_kernel<OT><<<dim3((unsigned)(B / group)), 256, 0, (hipStream_t)stream>>>(x, out, group, B / group, HW, C, Cp)));
  return enh_check_launch("enh_minibatch_stddev_nhwc");
}
extern "C" int enh_minibatch_stddev_nhwc_backward(const enh_h16* x, const enh_h16* g, int B, int HW, int C, int Cp, int group, enh_h16* dx, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_minibatch_stddev_nhwc_backward");
  const int rc = stddev_check(x, dx, B, HW, C, Cp, group, "enh_minibatch_stddev_nhwc_backward");
  if (rc != ENH_OK) return rc;
  ENH_REQUIRE(g, ENH_E_BADARG, "enh_minibatch_stddev_nhwc_backward: g is NULL");
  ENH_DT_DISPATCH(dtype, (stddev_bwd_kernel<OT><<<dim3((unsigned)(B / group)), 256, 0, (hipStream_t)stream>>>(x, g, dx, group, B / group, HW, C, Cp)));
  return enh_check_launch("enh_minibatch_stddev_nhwc_backward");
}
