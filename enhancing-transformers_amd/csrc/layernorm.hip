// layernorm.hip — LayerNorm forward / backward for gfx950 (HBM-bound: one wave per token row, 16-byte
// coalesced accesses, statistics and all accumulation in f32).
// Replaces nn.LayerNorm(dim) inside PreNorm and Transformer.norm (reference
// enhancing/modules/stage1/layers.py:85-92,143): eps 1e-5, biased variance, affine.
// Forward writes the 16-bit operand (bf16 or fp16: the kernels are templates over the operand type tag OT, common.h) the following MFMA GEMM consumes
// (and optionally an f32 copy); backward fuses the residual-stream gradient add and emits the 16-bit copy the wgrad/dgrad GEMMs consume.
#include "common.h"

// X3: additionally writes the split-bf16 operand row y3 [M][3*D] = [hi | lo | hi] with hi = bf16(y), lo = bf16(y - hi) (x3.hip: the A operand of a
// K-concatenated three-pass product); statistics and y are the same bits as in the plain form.
template <int NCH, bool X3, typename OT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int64_t M, int D, float eps,
                                                     uint16_t* __restrict__ y16, float* __restrict__ y32,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, uint16_t* __restrict__ y3) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < nch ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      q += (a * a + bb * bb) + (cc * cc + dd * dd);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 g = w4[c], be = b4[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + be.x;
      o.y = (v[i].y - mean) * rstd * g.y + be.y;
      o.z = (v[i].z - mean) * rstd * g.z + be.z;
      o.w = (v[i].w - mean) * rstd * g.w + be.w;
      if (y32) reinterpret_cast<float4*>(y32 + (size_t)row * D)[c] = o;
      const uint2 hi2 = make_uint2(pack2<OT>(o.x, o.y), pack2<OT>(o.z, o.w));
      if (y16) reinterpret_cast<uint2*>(y16 + (size_t)row * D)[c] = hi2;
      if (X3) {   // (bf16 only)
        const uint2 lo2 = make_uint2(pack_bf16x2(o.x - __builtin_bit_cast(float, hi2.x << 16), o.y - __builtin_bit_cast(float, hi2.x & 0xffff0000u)),
                                     pack_bf16x2(o.z - __builtin_bit_cast(float, hi2.y << 16), o.w - __builtin_bit_cast(float, hi2.y & 0xffff0000u)));
        uint2* r3 = reinterpret_cast<uint2*>(y3 + (size_t)row * 3 * D);
        r3[c] = hi2; r3[nch + c] = lo2; r3[2 * nch + c] = hi2;
      }
    }
  }
}

// DY16: dy is a 16-bit tensor (the dgrad GEMM's 16-bit output) instead of f32.  The dres loads are issued together with dy / x, not after the row
// reduction (D = 768: 128 VGPRs, still 4 waves / SIMD).  Measured at M = 131072, D = 768 with distinct buffers
// (tools/ln_bench.py): 328 us = 4.9 TB/s of algorithmic traffic (16 B / element).
template <int NCH, bool DY16, typename OT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_any, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     int64_t M, int D, float* __restrict__ dx32,
                                                     uint16_t* __restrict__ dx16, float* __restrict__ dw,
                                                     float* __restrict__ db, float* __restrict__ dxsum, float* __restrict__ part) {
  __shared__ float s_dw[3][NCH * 256];  // waves 1..3 park their column partials here
  __shared__ float s_db[3][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 gw[NCH], adw[NCH], adb[NCH], adx[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    gw[i] = c < nch ? w4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    adw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    adb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    adx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  // Software pipeline over the rows of this wave: the loads of the NEXT row are issued before the stores of the current one.  CDNA4 retires vector
  // memory operations in order (vmcnt counts stores too), so loads issued after a row's stores could only be waited for together with those stores'
  // acknowledgements — one exposed HBM round trip per row with only two waves per SIMD to cover it.
  struct RowIn { float4 xv[NCH]; float4 dv[NCH]; float4 rr[NCH]; float mu, rs; };
  auto load_row = [&](int64_t row, RowIn& in) {
    in.mu = mean[row]; in.rs = rstd[row];
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        in.xv[i] = xr[c];
        if (DY16) {
          const uint2 r = reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(dy_any) + (size_t)row * D)[c];
          in.dv[i] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), 0.f, 0.f);   // raw 16-bit pairs, unpacked where they are used
        } else {
          in.dv[i] = reinterpret_cast<const float4*>(static_cast<const float*>(dy_any) + (size_t)row * D)[c];
        }
        if (dres) in.rr[i] = reinterpret_cast<const float4*>(dres + (size_t)row * D)[c];
      }
    }
  };
  RowIn cur, nxt;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row < M) load_row(row, cur);
  for (; row < M; row += stride) {
    const bool more = row + stride < M;
    if (more) load_row(row + stride, nxt);
    const float mu = cur.mu, rs = cur.rs;
    float4 xh[NCH], g[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 xv = cur.xv[i];
        float4 dv;
        if (DY16) {
          const uint32_t r0 = __float_as_uint(cur.dv[i].x), r1 = __float_as_uint(cur.dv[i].y);
          dv = make_float4(unpack_lo<OT>(r0), unpack_hi<OT>(r0), unpack_lo<OT>(r1), unpack_hi<OT>(r1));
        } else {
          dv = cur.dv[i];
        }
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        g[i] = make_float4(dv.x * gw[i].x, dv.y * gw[i].y, dv.z * gw[i].z, dv.w * gw[i].w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        adw[i].x += dv.x * xh[i].x; adw[i].y += dv.y * xh[i].y; adw[i].z += dv.z * xh[i].z; adw[i].w += dv.w * xh[i].w;
        adb[i].x += dv.x; adb[i].y += dv.y; adb[i].z += dv.z; adb[i].w += dv.w;
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        g[i] = xh[i];
      }
    }
    const float c1 = wave_sum(s1) / (float)D;
    const float c2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float4 o;
        o.x = rs * (g[i].x - c1 - xh[i].x * c2);
        o.y = rs * (g[i].y - c1 - xh[i].y * c2);
        o.z = rs * (g[i].z - c1 - xh[i].z * c2);
        o.w = rs * (g[i].w - c1 - xh[i].w * c2);
        if (dres) {
          const float4 r = cur.rr[i];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4*>(dx32 + (size_t)row * D)[c] = o;
        if (dx16) reinterpret_cast<uint2*>(dx16 + (size_t)row * D)[c] = make_uint2(pack2<OT>(o.x, o.y), pack2<OT>(o.z, o.w));
        adx[i].x += o.x; adx[i].y += o.y; adx[i].z += o.z; adx[i].w += o.w;
      }
    }
    if (more) cur = nxt;
  }
  // column partials: 4 waves -> 1 through LDS, then one f32 atomic per column per workgroup
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float* pw = &s_dw[wave - 1][(lane + 64 * i) * 4];
      float* pb = &s_db[wave - 1][(lane + 64 * i) * 4];
      pw[0] = adw[i].x; pw[1] = adw[i].y; pw[2] = adw[i].z; pw[3] = adw[i].w;
      pb[0] = adb[i].x; pb[1] = adb[i].y; pb[2] = adb[i].z; pb[3] = adb[i].w;
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float aw[4] = {adw[i].x, adw[i].y, adw[i].z, adw[i].w};
        float ab[4] = {adb[i].x, adb[i].y, adb[i].z, adb[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          for (int ww = 0; ww < 3; ++ww) { aw[k] += s_dw[ww][c * 4 + k]; ab[k] += s_db[ww][c * 4 + k]; }
          if (part) {   // deterministic form: this workgroup's partial row; ln_bwd_reduce_kernel adds the rows in a fixed order
            part[((size_t)blockIdx.x * 3 + 0) * D + c * 4 + k] = aw[k];
            part[((size_t)blockIdx.x * 3 + 1) * D + c * 4 + k] = ab[k];
          } else {
            atomicAdd(&dw[c * 4 + k], aw[k]);
            atomicAdd(&db[c * 4 + k], ab[k]);
          }
        }
      }
    }
  }
  // optional: column sums of dx (= the bias gradient of the Linear that produced this LayerNorm's input stream)
  if (dxsum) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float* pw = &s_dw[wave - 1][(lane + 64 * i) * 4];
        pw[0] = adx[i].x; pw[1] = adx[i].y; pw[2] = adx[i].z; pw[3] = adx[i].w;
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
          float ax[4] = {adx[i].x, adx[i].y, adx[i].z, adx[i].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            for (int ww = 0; ww < 3; ++ww) ax[k] += s_dw[ww][c * 4 + k];
            if (part) part[((size_t)blockIdx.x * 3 + 2) * D + c * 4 + k] = ax[k];
            else atomicAdd(&dxsum[c * 4 + k], ax[k]);
          }
        }
      }
    }
  }
}

template <bool X3, typename OT>
static void ln_fwd_launch(int grid, hipStream_t s, const float* x, const float* w, const float* b, int64_t M, int D, float eps, enh_h16* y_bf16, float* y_f32,
                          float* mean, float* rstd, enh_h16* y3) {
  switch ((D + 255) / 256) {  // float4 chunks per lane: registers (hence occupancy) scale with it
    case 1: ln_fwd_kernel<1, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
    case 2: ln_fwd_kernel<2, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
    case 3: ln_fwd_kernel<3, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
    case 4: ln_fwd_kernel<4, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
    case 5: ln_fwd_kernel<5, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
    default: ln_fwd_kernel<8, X3, OT><<<grid, 256, 0, s>>>(x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3); break;
  }
}

extern "C" int enh_layernorm_forward(const float* x, const float* w, const float* b, int64_t M, int D, float eps,
                                     enh_h16* y_bf16, float* y_f32, float* mean, float* rstd, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_layernorm_forward");
  ENH_REQUIRE(x && w && b && (y_bf16 || y_f32), ENH_E_BADARG, "enh_layernorm_forward: null pointer");
  ENH_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, ENH_E_SHAPE, "enh_layernorm_forward: need D %% 4 == 0 and D <= 2048, got M=%lld D=%d", (long long)M, D);
  ENH_DT_DISPATCH(dtype, (ln_fwd_launch<false, OT>((int)((M + 3) / 4), (hipStream_t)stream, x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, nullptr)));
  return enh_check_launch("enh_layernorm_forward");
}

extern "C" int enh_layernorm_forward_x3(const float* x, const float* w, const float* b, int64_t M, int D, float eps, enh_h16* y3,
                                        enh_h16* y_bf16, float* y_f32, float* mean, float* rstd, void* stream) {
  ENH_REQUIRE(x && w && b && y3, ENH_E_BADARG, "enh_layernorm_forward_x3: null pointer");
  ENH_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, ENH_E_SHAPE, "enh_layernorm_forward_x3: need D %% 4 == 0 and D <= 2048, got M=%lld D=%d", (long long)M, D);
  ln_fwd_launch<true, BF16>((int)((M + 3) / 4), (hipStream_t)stream, x, w, b, M, D, eps, y_bf16, y_f32, mean, rstd, y3);
  return enh_check_launch("enh_layernorm_forward_x3");
}

template <bool DY16, typename OT>
static void ln_bwd_launch(int grid, hipStream_t s, const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                          const float* dres, int64_t M, int D, float* dx_f32, enh_h16* dx_bf16, float* dw, float* db, float* dx_colsum, float* part) {
  switch ((D + 255) / 256) {
    case 1: ln_bwd_kernel<1, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
    case 2: ln_bwd_kernel<2, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
    case 3: ln_bwd_kernel<3, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
    case 4: ln_bwd_kernel<4, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
    case 5: ln_bwd_kernel<5, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
    default: ln_bwd_kernel<8, DY16, OT><<<grid, 256, 0, s>>>(dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part); break;
  }
}

// second pass of the deterministic form: out[j] += sum over the workgroups of part[wg][which][j], in a fixed order (common.h fixed_order_rowsum16)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nwg, int D, float* __restrict__ dw, float* __restrict__ db,
                                                            float* __restrict__ dxsum) {
  __shared__ float s_red[256];
  const int which = blockIdx.y;
  float* out = which == 0 ? dw : (which == 1 ? db : dxsum);
  if (!out) return;   // (uniform per workgroup)
  const int j = blockIdx.x * 16 + (threadIdx.x & 15);
  const float t = fixed_order_rowsum16(part + (size_t)which * D, nwg, (int64_t)3 * D, j, j < D, s_red);
  if ((threadIdx.x >> 4) == 0 && j < D) out[j] += t;
}

static int ln_bwd_grid(int64_t M, bool for_workspace = false) {
  // Persistent grid, ONE workgroup per CU: with the row loop software-pipelined a workgroup keeps its CU's memory pipe busy by itself, and whole
  // multiples of the 256 CUs matter more than occupancy — measured 294 / 343 / 317 / 340 / 345 / 373 us at 256 / 384 / 512 / 768 / 1024 / 2048
  // workgroups (M = 131072, D = 768, bf16 dy; the atomic form of round 1 was best at 512).
  // The row partition is static (the per-workgroup partial sums and their fixed-order second pass are what makes the gradient bits reproducible), so
  // under a CU budget (collectives beside the backward pass) the grid shrinks with it: a workgroup without a CU would wait for another to retire.
  const int64_t want = (M + 3) / 4;
  const int cus = for_workspace ? enh_device_cus() : enh_cu_budget();
  return (int)(want < cus ? want : cus);
}

extern "C" size_t enh_layernorm_backward_workspace_bytes(int64_t M, int D) {
  return M > 0 && D > 0 ? (size_t)ln_bwd_grid(M, true) * 3 * D * sizeof(float) : 0;
}

static int ln_bwd_impl(const float* dy, const enh_h16* dy_bf16, const float* x, const float* w, const float* mean, const float* rstd, const float* dres,
                       int64_t M, int D, float* dx_f32, enh_h16* dx_bf16, float* dw, float* db, float* dx_colsum, float* part, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_layernorm_backward");
  ENH_REQUIRE((dy != nullptr) != (dy_bf16 != nullptr), ENH_E_BADARG, "enh_layernorm_backward: pass exactly one of dy (f32) / dy_bf16");
  ENH_REQUIRE(x && w && mean && rstd && dx_f32 && dw && db, ENH_E_BADARG, "enh_layernorm_backward: null pointer");
  ENH_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, ENH_E_SHAPE, "enh_layernorm_backward: need D %% 4 == 0 and D <= 2048, got M=%lld D=%d", (long long)M, D);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ln_bwd_grid(M);
  if (dy_bf16) ENH_DT_DISPATCH(dtype, (ln_bwd_launch<true, OT>(grid, s, dy_bf16, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part)));
  else ENH_DT_DISPATCH(dtype, (ln_bwd_launch<false, OT>(grid, s, dy, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, part)));
  if (part) ln_bwd_reduce_kernel<<<dim3((unsigned)((D + 15) / 16), 3), 256, 0, s>>>(part, grid, D, dw, db, dx_colsum);
  return enh_check_launch("enh_layernorm_backward");
}

extern "C" int enh_layernorm_backward(const float* dy, const enh_h16* dy_bf16, const float* x, const float* w, const float* mean,
                                      const float* rstd, const float* dres, int64_t M, int D, float* dx_f32,
                                      enh_h16* dx_bf16, float* dw, float* db, float* dx_colsum, int dtype, void* stream) {
  return ln_bwd_impl(dy, dy_bf16, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, nullptr, dtype, stream);
}

extern "C" int enh_layernorm_backward_ws(const float* dy, const enh_h16* dy_bf16, const float* x, const float* w, const float* mean,
                                         const float* rstd, const float* dres, int64_t M, int D, float* dx_f32,
                                         enh_h16* dx_bf16, float* dw, float* db, float* dx_colsum, void* ws, size_t ws_bytes, int dtype, void* stream) {
  ENH_REQUIRE(ws && ws_bytes >= enh_layernorm_backward_workspace_bytes(M, D), ENH_E_WORKSPACE, "enh_layernorm_backward_ws: workspace too small (%zu bytes)", ws_bytes);
  return ln_bwd_impl(dy, dy_bf16, x, w, mean, rstd, dres, M, D, dx_f32, dx_bf16, dw, db, dx_colsum, (float*)ws, dtype, stream);
}
