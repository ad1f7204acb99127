// elementwise.hip — HBM-bound data movement, reductions and the optimizer step for gfx950.
//   patchify / unpatchify(+pixel loss): the gather / scatter halves of Conv2d(k=s=p) and
//   ConvTranspose2d(k=s=p) (reference enhancing/modules/stage1/layers.py:168-171,178,202-205,212) with the
//   pixel losses of enhancing/losses/vqperceptual.py:113-114 fused into the scatter pass;
//   colsum: bias gradients; adamw: torch.optim.AdamW as configured at vitvqgan.py:160.
// All kernels move 8-16 bytes per lane with the contiguous side chosen for the larger tensor.  Kernels that read or write 16-bit operands are templates
// over the operand type tag OT = BF16 | F16 (common.h), chosen by the `dtype` argument of their C entry.
#include "common.h"

// img [B,C,H,W] f32 -> patches [M=B*gy*gx, C*p*p] (16-bit operand), element order (c, ph, pw)
template <typename OT>
__global__ void patchify_kernel(const float* __restrict__ img, int B, int C, int H, int W, int p,
                                uint16_t* __restrict__ out, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int p4 = p >> 2, gx_n = W / p, gy_n = H / p;
  int64_t r = i;
  const int q = (int)(r % p4); r /= p4;
  const int ph = (int)(r % p); r /= p;
  const int c = (int)(r % C); r /= C;       // r = m
  const int gx = (int)(r % gx_n);
  const int gy = (int)((r / gx_n) % gy_n);
  const int b = (int)(r / ((int64_t)gx_n * gy_n));
  const float4 v = *reinterpret_cast<const float4*>(img + (((size_t)b * C + c) * H + (size_t)gy * p + ph) * W + (size_t)gx * p + q * 4);
  *reinterpret_cast<uint2*>(out + (size_t)i * 4) = make_uint2(pack2<OT>(v.x, v.y), pack2<OT>(v.z, v.w));
}

// pix [M, C*p*p] f32 -> xrec [B,C,H,W] f32 ; optional pixel-loss sums (double atomics: order-independent
// to 1e-16, hence deterministic after rounding to f32) and the patch-layout 16-bit loss gradient.
template <typename OT>
__global__ __launch_bounds__(256) void unpatchify_loss_kernel(
    const float* __restrict__ pix, const float* __restrict__ target, int B, int C, int H, int W, int p, float w_l1,
    float w_l2, float inv_numel, float* __restrict__ xrec, double* __restrict__ sums, uint16_t* __restrict__ dpix,
    int64_t total4, const float* __restrict__ gscale) {
  __shared__ float s_l1[4], s_l2[4];
  if (gscale) inv_numel *= *gscale;      // the loss scale of the fp16 backward (a device scalar: GradScaler's scale, updated on the device) multiplies the GRADIENT only
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float l1 = 0.f, l2 = 0.f;
  if (i < total4) {
    const int p4 = p >> 2, gx_n = W / p, gy_n = H / p;
    int64_t r = i;
    const int q = (int)(r % p4); r /= p4;
    const int ph = (int)(r % p); r /= p;
    const int c = (int)(r % C); r /= C;
    const int gx = (int)(r % gx_n);
    const int gy = (int)((r / gx_n) % gy_n);
    const int b = (int)(r / ((int64_t)gx_n * gy_n));
    const size_t off = (((size_t)b * C + c) * H + (size_t)gy * p + ph) * W + (size_t)gx * p + q * 4;
    const float4 v = *reinterpret_cast<const float4*>(pix + (size_t)i * 4);
    if (xrec) *reinterpret_cast<float4*>(xrec + off) = v;
    if (target) {
      const float4 x = *reinterpret_cast<const float4*>(target + off);
      const float d[4] = {v.x - x.x, v.y - x.y, v.z - x.z, v.w - x.w};
      float g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        l1 += fabsf(d[k]);
        l2 += d[k] * d[k];
        const float sg = d[k] > 0.f ? 1.f : (d[k] < 0.f ? -1.f : 0.f);
        g[k] = (w_l1 * sg + w_l2 * 2.f * d[k]) * inv_numel;
      }
      if (dpix) *reinterpret_cast<uint2*>(dpix + (size_t)i * 4) = make_uint2(pack2<OT>(g[0], g[1]), pack2<OT>(g[2], g[3]));
    }
  }
  if (sums) {
    l1 = wave_sum(l1);
    l2 = wave_sum(l2);
    if ((threadIdx.x & 63) == 0) { s_l1[threadIdx.x >> 6] = l1; s_l2[threadIdx.x >> 6] = l2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&sums[0], (double)((s_l1[0] + s_l1[1]) + (s_l1[2] + s_l1[3])));
      atomicAdd(&sums[1], (double)((s_l2[0] + s_l2[1]) + (s_l2[2] + s_l2[3])));
    }
  }
}

// out[n] += sum_m x[m,n] ; x 16-bit (OT).  Wave = 128 columns (4 B per lane), 4 waves split the rows of a chunk.
// part_stride == 0: every row chunk adds into out[n] with f32 atomics; part_stride == N: chunk y stores its partial at out[y*N + n] (deterministic
// form: colsum_reduce_kernel then adds the chunks in a fixed order)
template <typename OT>
__global__ __launch_bounds__(256) void colsum_h16_kernel(const uint16_t* __restrict__ x, int64_t M, int64_t N,
                                                          int64_t ldx, int64_t rows_per_block, float* __restrict__ out, int64_t part_stride) {
  __shared__ float s_part[3][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * 128 + lane * 2;
  const int64_t m_begin = (int64_t)blockIdx.y * rows_per_block;
  int64_t m_end = m_begin + rows_per_block;
  if (m_end > M) m_end = M;
  float a0 = 0.f, a1 = 0.f;
  if (n0 < N) {  // N is even (N % 8 == 0 enforced by the launcher)
    for (int64_t m = m_begin + wave; m < m_end; m += 4) {
      const uint32_t u = *reinterpret_cast<const uint32_t*>(x + (size_t)m * ldx + n0);
      a0 += unpack_lo<OT>(u);
      a1 += unpack_hi<OT>(u);
    }
  }
  if (wave > 0) { s_part[wave - 1][lane * 2] = a0; s_part[wave - 1][lane * 2 + 1] = a1; }
  __syncthreads();
  if (wave == 0 && n0 < N) {
    for (int ww = 0; ww < 3; ++ww) { a0 += s_part[ww][lane * 2]; a1 += s_part[ww][lane * 2 + 1]; }
    { if (part_stride) out[(int64_t)blockIdx.y * part_stride + n0] = a0; else atomicAdd(&out[n0], a0); }
    { if (part_stride) out[(int64_t)blockIdx.y * part_stride + n0 + 1] = a1; else atomicAdd(&out[n0 + 1], a1); }
  }
}

// Wide variant (N % 8 == 0, 16-byte aligned rows): a lane owns 8 columns (one 16-byte load per row), a wave covers 512 contiguous
// columns = 1 KiB of a row, and four rows are in flight per wave.
template <typename OT>
__global__ __launch_bounds__(256) void colsum_h16_wide_kernel(const uint16_t* __restrict__ x, int64_t M, int64_t N, int64_t ldx,
                                                               int64_t rows_per_block, float* __restrict__ out, int64_t part_stride) {
  __shared__ float s_part[3][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * 512 + lane * 8;
  const int64_t m_begin = (int64_t)blockIdx.y * rows_per_block;
  int64_t m_end = m_begin + rows_per_block;
  if (m_end > M) m_end = M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n0 < N) {
    const uint16_t* col = x + n0;
    int64_t m = m_begin + wave;
    for (; m + 12 < m_end; m += 16) {
      u32x4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const u32x4*>(col + (size_t)(m + 4 * r) * ldx);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[2 * k] += unpack_lo<OT>(v[r][k]);
          acc[2 * k + 1] += unpack_hi<OT>(v[r][k]);
        }
    }
    for (; m < m_end; m += 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(col + (size_t)m * ldx);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] += unpack_lo<OT>(v[k]);
        acc[2 * k + 1] += unpack_hi<OT>(v[k]);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_part[wave - 1][lane * 8 + k] = acc[k];
  }
  __syncthreads();
  if (wave == 0 && n0 < N) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float a = acc[k];
      for (int ww = 0; ww < 3; ++ww) a += s_part[ww][lane * 8 + k];
      { if (part_stride) out[(int64_t)blockIdx.y * part_stride + n0 + k] = a; else atomicAdd(&out[n0 + k], a); }
    }
  }
}

template <typename OT>
__global__ void cast_f32_h16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<uint2*>(y + i) = make_uint2(pack2<OT>(v.x, v.y), pack2<OT>(v.z, v.w));
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = pack1<OT>(x[k]);
  }
}

// AdamW, decoupled weight decay, bias correction (torch.optim.AdamW semantics; vitvqgan.py:160).  skip: optional device flag — non-zero = this step's
// gradients held an inf / nan (enh_nonfinite_flag): nothing is written, the step is dropped the way torch.cuda.amp.GradScaler.step drops it under the
// reference's --use_amp (main.py:25,52).
template <typename OT>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, uint16_t* __restrict__ p16, int64_t n, float lr, float beta1,
                             float beta2, float eps, float wd, float gscale, float inv_bc1, float inv_sqrt_bc2, const float* __restrict__ skip,
                             const float* __restrict__ loss_scale) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (skip && *skip != 0.f) return;
  if (loss_scale) gscale /= *loss_scale;      // GradScaler's unscale, folded into the step (the scale is a power of two: exact)
  float pv[4], gv[4], mv[4], vv[4];
  const bool full = i + 3 < n;
  if (full) {
    const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
    const float4 c = *reinterpret_cast<const float4*>(m + i), d = *reinterpret_cast<const float4*>(v + i);
    pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
    mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
  } else {
    for (int k = 0; k < 4; ++k) {
      const bool ok = i + k < n;
      pv[k] = ok ? p[i + k] : 0.f; gv[k] = ok ? g[i + k] : 0.f; mv[k] = ok ? m[i + k] : 0.f; vv[k] = ok ? v[i + k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gg = gv[k] * gscale;
    pv[k] = pv[k] * (1.0f - lr * wd);
    mv[k] = mv[k] * beta1 + gg * (1.0f - beta1);
    vv[k] = vv[k] * beta2 + gg * gg * (1.0f - beta2);
    const float denom = sqrtf(vv[k]) * inv_sqrt_bc2 + eps;
    pv[k] = pv[k] - (lr * inv_bc1) * (mv[k] / denom);
  }
  if (full) {
    *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *reinterpret_cast<float4*>(m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (p16) *reinterpret_cast<uint2*>(p16 + i) = make_uint2(pack2<OT>(pv[0], pv[1]), pack2<OT>(pv[2], pv[3]));
  } else {
    for (int k = 0; k < 4 && i + k < n; ++k) {
      p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k];
      if (p16) p16[i + k] = pack1<OT>(pv[k]);
    }
  }
}

extern "C" int enh_patchify(const float* img, int B, int C, int H, int W, int p, enh_h16* patches, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_patchify");
  ENH_REQUIRE(img && patches, ENH_E_BADARG, "enh_patchify: null pointer");
  ENH_REQUIRE(B > 0 && C > 0 && p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0, ENH_E_SHAPE, "enh_patchify: need p %% 4 == 0 and H,W divisible by p (B=%d C=%d H=%d W=%d p=%d)", B, C, H, W, p);
  const int64_t total4 = (int64_t)B * C * H * W / 4;
  ENH_DT_DISPATCH(dtype, (patchify_kernel<OT><<<(int)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(img, B, C, H, W, p, patches, total4)));
  return enh_check_launch("enh_patchify");
}

extern "C" int enh_unpatchify_loss(const float* pix, const float* target, int B, int C, int H, int W, int p, float w_l1,
                                   float w_l2, float* xrec, double* sums, enh_h16* dpix_bf16, const float* grad_scale_dev, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_unpatchify_loss");
  ENH_REQUIRE(pix && (xrec || target), ENH_E_BADARG, "enh_unpatchify_loss: null pointer");
  ENH_REQUIRE(B > 0 && C > 0 && p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0, ENH_E_SHAPE, "enh_unpatchify_loss: need p %% 4 == 0 and H,W divisible by p");
  const int64_t numel = (int64_t)B * C * H * W, total4 = numel / 4;
  ENH_DT_DISPATCH(dtype, (unpatchify_loss_kernel<OT><<<(int)((total4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      pix, target, B, C, H, W, p, w_l1, w_l2, 1.0f / (float)numel, xrec, target ? sums : nullptr, dpix_bf16, total4, grad_scale_dev)));
  return enh_check_launch("enh_unpatchify_loss");
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int chunks, int64_t N, float* __restrict__ out, int accumulate) {
  __shared__ float s_red[256];
  const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const float t = fixed_order_rowsum16(part, chunks, N, n, n < N, s_red);
  if ((threadIdx.x >> 4) == 0 && n < N) out[n] = accumulate ? out[n] + t : t;
}

// fixed-order sum of `chunks` partial rows [chunks][N] into out[N] — also the second pass of the bias gradient that gemm.hip's tanh' epilogue produces
int enh_colsum_reduce_launch(const float* part, int chunks, int64_t N, float* out, int accumulate, hipStream_t s) {
  colsum_reduce_kernel<<<dim3((unsigned)((N + 15) / 16)), 256, 0, s>>>(part, chunks, N, out, accumulate);
  return enh_check_launch("colsum_reduce");
}

static int64_t colsum_chunks(int64_t M) {
  const int64_t chunks = (M + 511) / 512;
  return chunks > 256 ? 256 : chunks;
}

extern "C" size_t enh_colsum_h16_workspace_bytes(int64_t M, int64_t N) { return M > 0 && N > 0 ? (size_t)colsum_chunks(M) * N * sizeof(float) : 0; }

static int colsum_h16_impl(const enh_h16* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, float* part, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_colsum_h16");
  ENH_REQUIRE(x && out, ENH_E_BADARG, "enh_colsum_h16: null pointer");
  ENH_REQUIRE(M > 0 && N > 0 && N % 2 == 0 && ldx % 2 == 0, ENH_E_SHAPE, "enh_colsum_h16: N and ldx must be even");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && !part) {
    const int rc = enh_zero_f32_launch(out, N, s);
    if (rc) return rc;
  }
  const int64_t chunks = colsum_chunks(M);
  const int64_t rows_per_block = (M + chunks - 1) / chunks;
  float* dst = part ? part : out;
  const int64_t stride = part ? N : 0;
  if (N % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0) {
    dim3 grid((unsigned)((N + 511) / 512), (unsigned)chunks);
    ENH_DT_DISPATCH(dtype, (colsum_h16_wide_kernel<OT><<<grid, 256, 0, s>>>(x, M, N, ldx, rows_per_block, dst, stride)));
  } else {
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)chunks);
    ENH_DT_DISPATCH(dtype, (colsum_h16_kernel<OT><<<grid, 256, 0, s>>>(x, M, N, ldx, rows_per_block, dst, stride)));
  }
  if (part) colsum_reduce_kernel<<<dim3((unsigned)((N + 15) / 16)), 256, 0, s>>>(part, (int)chunks, N, out, accumulate);
  return enh_check_launch("enh_colsum_h16");
}

extern "C" int enh_colsum_h16(const enh_h16* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, int dtype, void* stream) {
  return colsum_h16_impl(x, M, N, ldx, out, accumulate, nullptr, dtype, stream);
}

// deterministic form: per-chunk partial rows in `ws` (enh_colsum_h16_workspace_bytes), added in a fixed order
extern "C" int enh_colsum_h16_ws(const enh_h16* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream) {
  ENH_REQUIRE(ws && ws_bytes >= enh_colsum_h16_workspace_bytes(M, N), ENH_E_WORKSPACE, "enh_colsum_h16_ws: workspace too small (%zu bytes)", ws_bytes);
  return colsum_h16_impl(x, M, N, ldx, out, accumulate, (float*)ws, dtype, stream);
}

// y[i] = bf16(x[i] * (i < n_scaled ? alpha : 1)): the forward operand of a packed q | k | v projection whose q rows carry the softmax scale (one rounding,
// from the fp32 master) — include/enh_hip.h enh_attention_forward, q_prescaled
// blockIdx.y = one of `count` equally spaced blocks (the same projection of successive transformer layers in the flat parameter store): one launch for a tower
template <typename OT>
__global__ void cast_f32_h16_head_scaled_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n, int64_t n_scaled, float alpha,
                                                 int64_t x_stride, int64_t y_stride) {
  x += (int64_t)blockIdx.y * x_stride; y += (int64_t)blockIdx.y * y_stride;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    if (i < n_scaled) { v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha; }     // n_scaled % 4 == 0 (checked by the launcher)
    *reinterpret_cast<uint2*>(y + i) = make_uint2(pack2<OT>(v.x, v.y), pack2<OT>(v.z, v.w));
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = pack1<OT>(x[k] * (k < n_scaled ? alpha : 1.0f));
  }
}

extern "C" int enh_cast_f32_h16_head_scaled(const float* x, enh_h16* y, int64_t n, int64_t n_scaled, float alpha, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_cast_f32_h16_head_scaled");
  ENH_REQUIRE(x && y && n > 0 && n_scaled >= 0 && n_scaled <= n && n_scaled % 4 == 0, ENH_E_BADARG, "enh_cast_f32_h16_head_scaled: bad argument");
  const int64_t n4 = (n + 3) / 4;
  ENH_DT_DISPATCH(dtype, (cast_f32_h16_head_scaled_kernel<OT><<<(int)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, n_scaled, alpha, 0, 0)));
  return enh_check_launch("enh_cast_f32_h16_head_scaled");
}

extern "C" int enh_cast_f32_h16_head_scaled_strided(const float* x, int64_t x_stride, enh_h16* y, int64_t y_stride, int64_t n, int64_t n_scaled, float alpha,
                                                     int count, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_cast_f32_h16_head_scaled_strided");
  ENH_REQUIRE(x && y && n > 0 && n_scaled >= 0 && n_scaled <= n && n_scaled % 4 == 0 && count > 0 && count <= 65535, ENH_E_BADARG,
              "enh_cast_f32_h16_head_scaled_strided: bad argument");
  ENH_REQUIRE(x_stride % 4 == 0 && y_stride % 4 == 0 && y_stride >= n, ENH_E_SHAPE, "enh_cast_f32_h16_head_scaled_strided: strides must be multiples of 4 elements and the outputs disjoint");
  const int64_t n4 = (n + 3) / 4;
  ENH_DT_DISPATCH(dtype, (cast_f32_h16_head_scaled_kernel<OT><<<dim3((unsigned)((n4 + 255) / 256), (unsigned)count), 256, 0, (hipStream_t)stream>>>(x, y, n, n_scaled, alpha, x_stride, y_stride)));
  return enh_check_launch("enh_cast_f32_h16_head_scaled_strided");
}

extern "C" int enh_cast_f32_h16(const float* x, enh_h16* y, int64_t n, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_cast_f32_h16");
  ENH_REQUIRE(x && y && n > 0, ENH_E_BADARG, "enh_cast_f32_h16: bad argument");
  const int64_t n4 = (n + 3) / 4;
  ENH_DT_DISPATCH(dtype, (cast_f32_h16_kernel<OT><<<(int)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n)));
  return enh_check_launch("enh_cast_f32_h16");
}

extern "C" int enh_adamw_step(float* p, const float* g, float* m, float* v, enh_h16* p_bf16, int64_t n, int step,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                              const float* skip_flag, const float* loss_scale_dev, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_adamw_step");
  ENH_REQUIRE(p && g && m && v && n > 0 && step >= 1, ENH_E_BADARG, "enh_adamw_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t n4 = (n + 3) / 4;
  ENH_DT_DISPATCH(dtype, (adamw_kernel<OT><<<(int)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay,
                                                                                            grad_scale, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), skip_flag, loss_scale_dev)));
  return enh_check_launch("enh_adamw_step");
}

// torch.cuda.amp.GradScaler.update() on the device (the reference's --use_amp, main.py:25,52): found_inf != 0 -> scale *= backoff, tracker = 0; otherwise
// ++tracker and, after growth_interval clean steps in a row, scale *= growth, tracker = 0.  scale stays inside [1, 2^24] (powers of two with the default factors
// 2 / 0.5: scaling and unscaling are exact).  One thread; no host round trip, HIP-graph safe.
__global__ void loss_scale_update_kernel(float* __restrict__ scale, const float* __restrict__ found_inf, int* __restrict__ tracker, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = *scale;
  int t = *tracker;
  if (*found_inf != 0.f) { s *= backoff; t = 0; }
  else if (interval > 0 && ++t >= interval) { s *= growth; t = 0; }
  *scale = fminf(fmaxf(s, 1.0f), 16777216.0f);
  *tracker = t;
}

extern "C" int enh_loss_scale_update(float* scale, const float* found_inf, int* growth_tracker, float growth_factor, float backoff_factor, int growth_interval,
                                     void* stream) {
  ENH_REQUIRE(scale && found_inf && growth_tracker && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 0, ENH_E_BADARG,
              "enh_loss_scale_update: bad argument (growth >= 1, 0 < backoff <= 1, interval >= 0)");
  loss_scale_update_kernel<<<1, 64, 0, (hipStream_t)stream>>>(scale, found_inf, growth_tracker, growth_factor, backoff_factor, growth_interval);
  return enh_check_launch("enh_loss_scale_update");
}

// flag[0] = 1 if any of x[0..n) is inf / nan, else unchanged (the caller zeroes it): the found-inf check of torch.cuda.amp.GradScaler.unscale_ over one flat
// gradient buffer (the reference's --use_amp, main.py:25,52).  One read of x at HBM rate; the flag is written with a plain store (every writer stores 1).
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  const uint32_t* __restrict__ u = reinterpret_cast<const uint32_t*>(x);
  const int64_t n4 = n & ~(int64_t)3;
  uint32_t any = 0u;      // bit 0 set once an element with an all-ones exponent (inf or nan) has been seen
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n4; i += stride) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(u + i);
    const u32x4 e = (v >> 23) & 0xffu;                               // biased exponents
    any |= (uint32_t)(e[0] == 0xffu) | (uint32_t)(e[1] == 0xffu) | (uint32_t)(e[2] == 0xffu) | (uint32_t)(e[3] == 0xffu);
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4)) any |= (uint32_t)(((u[n4 + threadIdx.x] >> 23) & 0xffu) == 0xffu);      // the (at most three) tail elements
  if (__builtin_amdgcn_ballot_w64(any != 0u) != 0 && (threadIdx.x & 63) == 0) *flag = 1.0f;
}

extern "C" int enh_nonfinite_flag(const float* x, int64_t n, float* flag, void* stream) {
  ENH_REQUIRE(x && flag && n > 0 && ((uintptr_t)x & 15) == 0, ENH_E_BADARG, "enh_nonfinite_flag: bad argument (x 16-byte aligned)");
  const int64_t n4 = (n + 3) / 4;
  const int64_t want = (n4 + 255) / 256;
  const int grid = (int)(want < 2048 ? want : 2048);
  nonfinite_flag_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, n, flag);
  return enh_check_launch("enh_nonfinite_flag");
}

// Device-side tail of the input pipeline (reference enhancing/dataloader/imagenet.py:26-54: Resize -> RandomCrop / CenterCrop -> RandomHorizontalFlip ->
// ToTensor): crop window, flip and ToTensor (uint8 HWC -> float CHW / 255) of a batch of decoded, resized images in one pass.  The images arrive as
// uint8 [B, Hs, Ws, 3] (each resized image in the top-left corner of its Hs x Ws slot), meta[b] = (y0, x0, flip).  Pure data movement + one division:
// bit-identical to the host path (numpy crop, [:, ::-1], astype(float32) / 255).
__global__ __launch_bounds__(256) void crop_flip_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ meta, float* __restrict__ out, int B, int Hs,
                                                           int Ws, int R) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b, y, x)
  const int64_t total = (int64_t)B * R * R;
  if (idx >= total) return;
  const int x = (int)(idx % R), y = (int)((idx / R) % R);
  const int64_t b = idx / ((int64_t)R * R);
  const int y0 = meta[b * 3 + 0], x0 = meta[b * 3 + 1], flip = meta[b * 3 + 2];
  const int sx = x0 + (flip ? R - 1 - x : x), sy = y0 + y;
  const uint8_t* p = src + ((b * Hs + sy) * (int64_t)Ws + sx) * 3;
  float* o = out + b * 3 * (int64_t)R * R + (int64_t)y * R + x;
  o[0] = (float)p[0] / 255.0f;
  o[(int64_t)R * R] = (float)p[1] / 255.0f;
  o[2 * (int64_t)R * R] = (float)p[2] / 255.0f;
}

extern "C" int enh_crop_flip_u8(const uint8_t* src, int B, int Hs, int Ws, const int32_t* meta, int R, float* out, void* stream) {
  ENH_REQUIRE(src && meta && out && B > 0 && R > 0 && Hs >= R && Ws >= R, ENH_E_BADARG, "enh_crop_flip_u8: bad argument (the staging slot must hold an R x R window)");
  const int64_t total = (int64_t)B * R * R;
  crop_flip_u8_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(src, meta, out, B, Hs, Ws, R);
  return enh_check_launch("enh_crop_flip_u8");
}
