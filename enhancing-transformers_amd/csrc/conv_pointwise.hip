// conv_pointwise.hip — the 1 x 1 convolution between an 8-channel tensor and a wide one, for gfx950 (CDNA4), wave64.
//
// The discriminator's first layer (FromRGB: ConvLayer(3, 128, 1), reference enhancing/losses/layers.py:220-264,296-306) maps the image — padded to 8 channels
// in the channels-last layout — to 128 channels at full resolution: 2 * 8 * 128 FLOP per pixel against 272 bytes of traffic.  As an implicit GEMM its
// contraction is 8 deep (forward, weight gradient) or its output 8 wide (input gradient): the MFMA tiles run at 16 / 19 / 7 TF/s (profiles/r04_conv_layers.txt)
// and none of the three roles comes near the memory system.  Here they are what they are — streaming kernels on the vector ALU:
//   forward          out[p][n] = epilogue( sum_c x[p][c] * w[n][c] )              one thread = one pixel x 8 output channels, weights in registers
//   input gradient   dx[p][c]  = sum_n dy[p][n] * wt[c][n]                         N/8 lanes per pixel, 64 partial products each, a butterfly over the lanes
//   weight gradient  dw[n][c]  = sum_p dy[p][n] * x[p][c]                          64 accumulators per thread over a pixel slice, fixed-order reduction:
//                                                                                  lanes (shuffles) -> waves (LDS) -> workgroups (workspace slabs)
// All arithmetic in f32 on exact bf16 products, deterministic.  Called by enh_conv_nhwc_h16 / enh_conv_wgrad_nhwc_h16 (conv_igemm.hip) when the geometry is
// a dense 1 x 1, stride 1 one with C == 8 (forward / weight gradient) or N == 8 (input gradient).
#include "gemm_tiles.h"

template <typename OT>
__device__ __forceinline__ void pw_unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = unpack_lo<OT>(v[k]); f[2 * k + 1] = unpack_hi<OT>(v[k]); }
}

// ---- forward: x [M][8], w [N][8] (packed operand of the implicit-GEMM kernels), out [M][N]; mode 2 (plain) or 3 (bias + leaky-ReLU * p1) -------------
template <typename OT>
__global__ __launch_bounds__(256) void conv_pw_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                          uint16_t* __restrict__ out, int64_t M, int N, int mode, float p0, float p1, int pix_per_wg) {
  const int groups = N >> 3, t = threadIdx.x;
  const int ng = t % groups, pl = t / groups, ppb = 256 / groups;
  float wr[8][8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pw_unpack8<OT>(*reinterpret_cast<const u32x4*>(w + (int64_t)(ng * 8 + j) * 8), wr[j]);
    b[j] = (mode == 3 && bias) ? bias[ng * 8 + j] : 0.f;
  }
  const int64_t p0_ = (int64_t)blockIdx.x * pix_per_wg;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int it = 0; it < pix_per_wg; it += 4 * ppb) {          // four pixels per thread in flight (pix_per_wg is a multiple of 4 * ppb)
    u32x4 xin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0_ + it + u * ppb + pl;
      xin[u] = p < M ? *reinterpret_cast<const u32x4*>(x + p * 8) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0_ + it + u * ppb + pl;
      if (p >= M) continue;
      float f[8], v[8];
      pw_unpack8<OT>(xin[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) a += wr[j][c] * f[c];
        if (mode == 3) { a += b[j]; a = (a > 0.f ? a : a * p0) * p1; }
        v[j] = a;
      }
      const u32x4 o = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3]), pack2<OT>(v[4], v[5]), pack2<OT>(v[6], v[7])};
      *reinterpret_cast<u32x4*>(out + p * N + ng * 8) = o;
    }
  }
}

// ---- input gradient: src [M][K] (dy), wt [8][K] (transposed packed operand), out [M][8]; mode 2 ----------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void conv_pw_dgrad_kernel(const uint16_t* __restrict__ src, const uint16_t* __restrict__ wt, uint16_t* __restrict__ out,
                                                            int64_t M, int K, int pix_per_wg) {
  const int groups = K >> 3, t = threadIdx.x;          // lanes per pixel: a power of two <= 64, so a pixel lives inside one wave
  const int kg = t % groups, pl = t / groups, ppb = 256 / groups;
  float wr[8][8];                                       // wr[j][k] = wt[j][kg*8 + k]
#pragma unroll
  for (int j = 0; j < 8; ++j) pw_unpack8<OT>(*reinterpret_cast<const u32x4*>(wt + (int64_t)j * K + kg * 8), wr[j]);
  const int64_t p0_ = (int64_t)blockIdx.x * pix_per_wg;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int it = 0; it < pix_per_wg; it += 4 * ppb) {          // four pixels per thread in flight; no early exit: the reductions need every lane of the wave
    u32x4 sin[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0_ + it + u * ppb + pl;
      sin[u] = p < M ? *reinterpret_cast<const u32x4*>(src + p * K + kg * 8) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0_ + it + u * ppb + pl;
      float f[8], v[8];
      pw_unpack8<OT>(sin[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += wr[j][k] * f[k];
        v[j] = a;
      }
      if (groups == 16) {          // one pixel = one 16-lane DPP row: sum of the row's rotations by 8, 4, 2, 1 on the vector ALU (no LDS crossbar)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0x128, 0xf, 0xf, false));
          v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0x124, 0xf, 0xf, false));
          v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0x122, 0xf, 0xf, false));
          v[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[j]), 0x121, 0xf, 0xf, false));
        }
      } else {
        for (int d = 1; d < groups; d <<= 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += __shfl_xor(v[j], d, 64);
        }
      }
      if (p < M && kg == 0) {
        const u32x4 o = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3]), pack2<OT>(v[4], v[5]), pack2<OT>(v[6], v[7])};
        *reinterpret_cast<u32x4*>(out + p * 8) = o;
      }
    }
  }
}

// ---- weight gradient: x [M][8], dy [M][N] -> partial slab [N][8] f32 per workgroup --------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void conv_pw_wgrad_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, float* __restrict__ ws,
                                                            int64_t M, int N, int pix_per_wg) {
  __shared__ float s_part[4][1024];                     // per wave: [N/8 groups <= 16][8 n][8 c]
  const int groups = N >> 3, t = threadIdx.x;           // groups in {1, 2, 4, 8, 16}: a wave holds 64 / groups pixel lanes of every group
  const int ng = t % groups, pl = t / groups, ppb = 256 / groups;
  float acc[8][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[j][c] = 0.f;
  const int64_t p0_ = (int64_t)blockIdx.x * pix_per_wg;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int it = 0; it < pix_per_wg; it += 4 * ppb) {          // four pixels per thread in flight; a pixel beyond M contributes zeros
    u32x4 xin[4], din[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = p0_ + it + u * ppb + pl;
      xin[u] = p < M ? *reinterpret_cast<const u32x4*>(x + p * 8) : zero4;
      din[u] = p < M ? *reinterpret_cast<const u32x4*>(dy + p * N + ng * 8) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8], d[8];
      pw_unpack8<OT>(xin[u], f);
      pw_unpack8<OT>(din[u], d);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[j][c] += d[j] * f[c];
    }
  }
  // pixel lanes of one wave (lane bits above log2(groups)), then the four waves, each in a fixed order
  for (int dlt = groups; dlt < 64; dlt <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[j][c] += __shfl_xor(acc[j][c], dlt, 64);
  }
  const int lane = t & 63, wave = t >> 6;
  if (lane < groups) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) s_part[wave][(lane * 8 + j) * 8 + c] = acc[j][c];
  }
  __syncthreads();
  for (int i = t; i < N * 8; i += 256) ws[(int64_t)blockIdx.x * N * 8 + i] = ((s_part[0][i] + s_part[1][i]) + s_part[2][i]) + s_part[3][i];
}

// out[i] = sum over slabs of ws[slab][i], i < MN: one workgroup per 64 outputs, 16 waves each adding every 16th slab in ascending order, then the
// 16 partial sums in ascending order — a fixed tree (deterministic), parallel over the slabs (the generic split-K second pass walks them one by one)
__global__ __launch_bounds__(1024) void conv_pw_reduce_kernel(const float* __restrict__ ws, int slabs, int MN, float* __restrict__ out) {
  __shared__ float s_sum[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float a = 0.f;
  if (i < MN)
    for (int s_ = wave; s_ < slabs; s_ += 16) a += ws[(int64_t)s_ * MN + i];
  s_sum[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && i < MN) {
    float t = s_sum[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += s_sum[w][lane];
    out[i] = t;
  }
}

static bool pw_groups_ok(int n) { const int g = n / 8; return n % 8 == 0 && g >= 1 && g <= 16 && (g & (g - 1)) == 0; }

// dense 1 x 1, stride 1, no padding: GEMM row m = pixel m of source and output alike
static bool pw_geom(const enh_conv_geom& g) {
  return g.nty == 1 && g.ntx == 1 && g.gs == 1 && g.oy0 == 0 && g.ox0 == 0 && g.os == 1 && g.oph == 0 && g.opw == 0 && g.Hm == g.Hs && g.Wm == g.Ws &&
         g.HO == g.Hm && g.WO == g.Wm;
}

static int pw_pix_per_wg(int64_t M, int ppb, int wgs = 2048) {
  // ~8 workgroups per CU (weight gradient: 2, its slabs are added up afterwards), whole passes of four pixels per thread
  const int unit = 4 * ppb;
  int64_t per = (M + wgs - 1) / wgs;
  per = ((per + unit - 1) / unit) * unit;
  return (int)(per < unit ? unit : per);
}

// returns 1 if the launch was taken here, 0 if the geometry is not a pointwise one (the caller runs the implicit-GEMM kernels)
int conv_pointwise_forward(const uint16_t* src, const uint16_t* wt, const enh_conv_geom& g, int mode, const float* bias, float p0, float p1, uint16_t* out,
                           int dtype, hipStream_t stream) {
  if (!pw_geom(g) || (mode != 2 && mode != 3)) return 0;
  const int64_t M = (int64_t)g.B * g.Hm * g.Wm;
  if (g.C == 8 && pw_groups_ok(g.N)) {
    const int ppb = 256 / (g.N / 8), per = pw_pix_per_wg(M, ppb);
    ENH_DT_DISPATCH(dtype, (conv_pw_fwd_kernel<OT><<<dim3((unsigned)((M + per - 1) / per)), 256, 0, stream>>>(src, wt, bias, out, M, g.N, mode, p0, p1, per)));
    return 1;
  }
  if (g.N == 8 && mode == 2 && g.C >= 8 && g.C <= 512 && (g.C & (g.C - 1)) == 0) {
    const int ppb = 256 / (g.C / 8), per = pw_pix_per_wg(M, ppb);
    ENH_DT_DISPATCH(dtype, (conv_pw_dgrad_kernel<OT><<<dim3((unsigned)((M + per - 1) / per)), 256, 0, stream>>>(src, wt, out, M, g.C, per)));
    return 1;
  }
  return 0;
}

// weight gradient: number of partial slabs (0 = not a pointwise geometry)
int conv_pointwise_wgrad_slabs(const enh_conv_geom& g) {
  if (!(g.nty == 1 && g.ntx == 1 && g.gs == 1 && g.oy0 == 0 && g.ox0 == 0 && g.Hm == g.Hs && g.Wm == g.Ws) || g.C != 8 || !pw_groups_ok(g.N)) return 0;
  const int64_t M = (int64_t)g.B * g.Hm * g.Wm;
  const int ppb = 256 / (g.N / 8), per = pw_pix_per_wg(M, ppb, 512);
  return (int)((M + per - 1) / per);
}

// ws: slabs x [N][8] f32 (conv_pointwise_wgrad_slabs) ; dw [N][8]
void conv_pointwise_wgrad(const uint16_t* src, const uint16_t* dy, const enh_conv_geom& g, float* ws, float* dw, int dtype, hipStream_t stream) {
  const int64_t M = (int64_t)g.B * g.Hm * g.Wm;
  const int ppb = 256 / (g.N / 8), per = pw_pix_per_wg(M, ppb, 512);
  const int slabs = (int)((M + per - 1) / per), MN = g.N * 8;
  ENH_DT_DISPATCH(dtype, (conv_pw_wgrad_kernel<OT><<<dim3((unsigned)slabs), 256, 0, stream>>>(src, dy, slabs == 1 ? dw : ws, M, g.N, per)));
  if (slabs > 1) conv_pw_reduce_kernel<<<dim3((unsigned)((MN + 63) / 64)), 1024, 0, stream>>>(ws, slabs, MN, dw);
}
