// vq.hip — fused nearest-neighbour vector / residual quantizer for gfx950.
//
// Replaces VectorQuantizer.quantize + BaseQuantizer.forward (reference
// enhancing/modules/stage1/quantizers.py:38-92): l2-normalise -> pairwise distance against all K codes ->
// argmin -> gather -> commit/codebook loss -> (residual loop) -> straight-through, WITHOUT materialising the
// [M,K] distance matrix.  The M x K x 32 contraction runs on the exact-f32 matrix pipe
// (v_mfma_f32_32x32x2_f32): indices must match the reference's fp32 argmin, and bf16 operands do not
// (99.29 %), so this kernel is bound by the 157.3 TFLOP/s f32 MFMA roof, not by HBM.
//
// Work decomposition: one wave = 32 tokens (MFMA B operand, held in registers for the whole K sweep);
// a 256-thread workgroup (4 waves, 128 tokens) streams the pre-normalised codebook through LDS in
// 128-code tiles (MFMA A operand), so each D[32 codes x 32 tokens] tile leaves every lane with 16 candidate
// codes of ONE token: the running argmin is lane-local, one cross-half exchange at the end.
//
// The arithmetic contract (summation orders) is stated in include/enh_hip.h and restated bit-exactly by
// oracle/vq_oracle.c.  Compiled with -ffp-contract=off so that every fused multiply-add is explicit.
#include "common.h"

#define VQ_D 32
#define VQ_TILE 128
#define VQ_PITCH 36  // floats per staged code row (144 B: 16-B aligned, breaks the 128-B bank period)

__device__ inline float chain16_sq(const float* x) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s = fmaf(x[j], x[j], s);
  return s;
}

// ---------------------------------------------------------------------------------------------
// codebook preparation: en = n(E) (or E), ee = S(en), enrm = max(||E||, 1e-12)   [quantizers.py:76,79]
// ---------------------------------------------------------------------------------------------
__global__ void vq_prep_kernel(const float* __restrict__ E, float* __restrict__ en, float* __restrict__ ee,
                               float* __restrict__ enrm, int K, int Kpad, int use_norm) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Kpad) return;
  if (k >= K) {
    for (int j = 0; j < VQ_D; ++j) en[(size_t)k * VQ_D + j] = 0.f;
    ee[k] = __builtin_inff();
    enrm[k] = 1.f;
    return;
  }
  float x[VQ_D];
  const float4* src = reinterpret_cast<const float4*>(E + (size_t)k * VQ_D);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float4 v = src[c];
    x[c * 4 + 0] = v.x; x[c * 4 + 1] = v.y; x[c * 4 + 2] = v.z; x[c * 4 + 3] = v.w;
  }
  float den = 1.f;
  if (use_norm) {
    float s = chain16_sq(x) + chain16_sq(x + 16);
    den = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
    for (int j = 0; j < VQ_D; ++j) x[j] = x[j] / den;
  }
  enrm[k] = den;
  ee[k] = chain16_sq(x) + chain16_sq(x + 16);
  float4* dst = reinterpret_cast<float4*>(en + (size_t)k * VQ_D);
#pragma unroll
  for (int c = 0; c < 8; ++c) dst[c] = make_float4(x[c * 4], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
}

// codebook tile staging: 128 codes x 32 floats (16 KB) per tile, 4 x 16 B per thread, fully coalesced
__device__ __forceinline__ void vq_tile_gload(f32x4 (&pre)[4], float& pre_ee, const float* __restrict__ en,
                                              const float* __restrict__ ee, int kt, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = t + 256 * i;
    pre[i] = *reinterpret_cast<const f32x4*>(en + ((size_t)kt * VQ_TILE + (id >> 3)) * VQ_D + (id & 7) * 4);
  }
  if (t < VQ_TILE) pre_ee = ee[(size_t)kt * VQ_TILE + t];
}
__device__ __forceinline__ void vq_tile_sstore(const f32x4 (&pre)[4], float pre_ee, float* tile, float* tile_ee, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = t + 256 * i;
    *reinterpret_cast<f32x4*>(&tile[(id >> 3) * VQ_PITCH + (id & 7) * 4]) = pre[i];
  }
  if (t < VQ_TILE) tile_ee[t] = pre_ee;
}

// ---------------------------------------------------------------------------------------------
// main forward kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_nn_kernel(
    const float* __restrict__ z, const float* __restrict__ en, const float* __restrict__ ee, int64_t M, int Kpad,
    int depth, int use_norm, float* __restrict__ zq_out, uint16_t* __restrict__ zq_bf16, int f16,
    int64_t* __restrict__ idx_out, float* __restrict__ loss_partials) {
  __shared__ __attribute__((aligned(16))) float s_tile[2][VQ_TILE * VQ_PITCH];
  __shared__ __attribute__((aligned(16))) float s_ee[2][VQ_TILE];
  __shared__ float s_red[4];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const int64_t tok = ((int64_t)blockIdx.x * 4 + wave) * 32 + col;
  const bool live = tok < M;
  const int ntiles = Kpad / VQ_TILE;

  float z0[16], r[16], zn[16], zq_acc[16];
  {
    const float4* src = reinterpret_cast<const float4*>(z + (size_t)(live ? tok : 0) * VQ_D + hi * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 v = live ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      z0[c * 4 + 0] = v.x; z0[c * 4 + 1] = v.y; z0[c * 4 + 2] = v.z; z0[c * 4 + 3] = v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) { r[j] = z0[j]; zq_acc[j] = 0.f; }

  for (int dpt = 0; dpt < depth; ++dpt) {
    // ---- zn = n(r), zz = S(zn) ----
    if (use_norm) {
      float sp = chain16_sq(r);
      float s = sp + __shfl_xor(sp, 32, 64);
      float den = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
      for (int j = 0; j < 16; ++j) zn[j] = r[j] / den;
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) zn[j] = r[j];
    }
    float zzp = chain16_sq(zn);
    const float zz = zzp + __shfl_xor(zzp, 32, 64);

    float best_d = __builtin_inff();
    int best_i = 0;

    // ---- sweep the codebook: LDS double buffer, register prefetch ----
    f32x4 pre[4];
    float pre_ee = 0.f;
    __syncthreads();  // previous depth's readers are done with both buffers
    vq_tile_gload(pre, pre_ee, en, ee, 0, t);
    vq_tile_sstore(pre, pre_ee, s_tile[0], s_ee[0], t);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < ntiles) vq_tile_gload(pre, pre_ee, en, ee, kt + 1, t);
#pragma unroll 1
      for (int sub = 0; sub < 4; ++sub) {
        float a[16];
        const float4* ap = reinterpret_cast<const float4*>(&s_tile[buf][(sub * 32 + col) * VQ_PITCH + hi * 16]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float4 v = ap[c];
          a[c * 4 + 0] = v.x; a[c * 4 + 1] = v.y; a[c * 4 + 2] = v.z; a[c * 4 + 3] = v.w;
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], zn[kk], acc, 0, 0, 0);
        // D row (code) of acc[q]: (q&3) + 8*(q>>2) + 4*hi ; column = this lane's token
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int row0 = sub * 32 + 8 * g4 + 4 * hi;
          const float4 e4 = *reinterpret_cast<const float4*>(&s_ee[buf][row0]);
          const float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = (zz + ev[q]) - 2.0f * acc[g4 * 4 + q];
            const int code = kt * VQ_TILE + row0 + q;
            if (d < best_d) { best_d = d; best_i = code; }
          }
        }
      }
      if (kt + 1 < ntiles) vq_tile_sstore(pre, pre_ee, s_tile[buf ^ 1], s_ee[buf ^ 1], t);
      __syncthreads();
    }
    // ---- merge the two half-waves (same token): lowest distance, lowest index on ties ----
    {
      const float od = __shfl_xor(best_d, 32, 64);
      const int oi = __shfl_xor(best_i, 32, 64);
      if (od < best_d || (od == best_d && oi < best_i)) { best_d = od; best_i = oi; }
    }
    if (live && hi == 0) idx_out[(size_t)tok * depth + dpt] = (int64_t)best_i;

    // ---- gather en[idx], loss partial, residual update ----
    float lp = 0.f;
    {
      const float4* ep = reinterpret_cast<const float4*>(en + (size_t)best_i * VQ_D + hi * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v = ep[c];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = c * 4 + q;
          const float df = e[q] - zn[j];
          lp = fmaf(df, df, lp);
          zq_acc[j] = zq_acc[j] + e[q];
          r[j] = r[j] - e[q];
        }
      }
    }
    if (!live) lp = 0.f;
    lp = wave_sum(lp);
    if (lane == 0) s_red[wave] = lp;
    __syncthreads();
    if (t == 0) loss_partials[(size_t)blockIdx.x * depth + dpt] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }

  if (live) {
    float o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = z0[j] + (zq_acc[j] - z0[j]);  // straight-through value, quantizers.py:61
    float4* dst = reinterpret_cast<float4*>(zq_out + (size_t)tok * VQ_D + hi * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(o[c * 4], o[c * 4 + 1], o[c * 4 + 2], o[c * 4 + 3]);
    if (zq_bf16) {
      uint4* d16 = reinterpret_cast<uint4*>(zq_bf16 + (size_t)tok * VQ_D + hi * 16);
      d16[0] = f16 ? make_uint4(pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3]), pack2<F16>(o[4], o[5]), pack2<F16>(o[6], o[7]))
                   : make_uint4(pack2<BF16>(o[0], o[1]), pack2<BF16>(o[2], o[3]), pack2<BF16>(o[4], o[5]), pack2<BF16>(o[6], o[7]));
      d16[1] = f16 ? make_uint4(pack2<F16>(o[8], o[9]), pack2<F16>(o[10], o[11]), pack2<F16>(o[12], o[13]), pack2<F16>(o[14], o[15]))
                   : make_uint4(pack2<BF16>(o[8], o[9]), pack2<BF16>(o[10], o[11]), pack2<BF16>(o[12], o[13]), pack2<BF16>(o[14], o[15]));
    }
  }
}

// loss = mean_i( beta*m_i + m_i ), m_i = S_i / (M*32)      [quantizers.py:56,89-90]
__global__ void vq_loss_finalize_kernel(const float* __restrict__ partials, int nblocks, int depth, int64_t M,
                                        float beta, float* __restrict__ loss_out) {
  __shared__ double s_acc[256];
  float total = 0.f;
  for (int dpt = 0; dpt < depth; ++dpt) {
    double a = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) a += (double)partials[(size_t)b * depth + dpt];
    s_acc[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) s_acc[threadIdx.x] += s_acc[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float m = (float)(s_acc[0] / ((double)M * VQ_D));
      total += beta * m + m;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = total / (float)depth;
}

// ---------------------------------------------------------------------------------------------
// codebook gradient, deterministic (round 4).  dE[k] = sum over the (token, depth) entries that picked code k of their contribution vector — a
// scatter-add whose natural forms (per-token atomics; the LDS hash + one global atomic per distinct code of rounds 1-3) add in an order that changes
// from run to run.  Here the backward kernel only WRITES each entry's vector and code (entry e = depth * M + token), and the sum is an "owner scans"
// reduction with a fixed order: a workgroup owns 32 consecutive codes and one of 16 token slices; it walks its slice's codes in ascending entry order
// (256 per step, hits compacted in order by ballot); hit q of a step goes to half-wave q mod 8, which adds its vector (one column per lane) into ITS copy
// of the 32 x 32 accumulator in LDS; the eight copies and then the 16 slice partials of a code are added in a fixed order.  No sort, no float atomics:
// the same bits on every run, which is what a diff of two data-parallel runs needs (quantizers.py:85-90 is the reference's autograd
// scatter).  Usage collapse is the stress case (an untrained model uses a handful of codes): all hits of a slice then belong to one workgroup, whose eight
// half-waves share them (128-way parallel over the launch) with four loads in flight each.
// ---------------------------------------------------------------------------------------------
#define VQ_CPW 32   // codes per workgroup
#define VQ_RS 16    // token slices
__global__ __launch_bounds__(256) void vq_de_partial_kernel(const int* __restrict__ codes, const float* __restrict__ contrib, int64_t n, int K,
                                                            float* __restrict__ part) {
  __shared__ int s_list[256];
  __shared__ int s_wc[4];
  __shared__ float s_acc[8][VQ_CPW][VQ_D];       // [half-wave][code][column]: each half-wave adds into its own copy (32 KiB)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, grp = t >> 5, gl = t & 31;
  const int cb = blockIdx.x * VQ_CPW, slice = blockIdx.y;
  const int64_t per = ((n + (int64_t)VQ_RS * 256 - 1) / ((int64_t)VQ_RS * 256)) * 256;
  const int64_t e0 = (int64_t)slice * per, e1 = e0 + per < n ? e0 + per : n;
#pragma unroll
  for (int k = 0; k < 8 * VQ_CPW * VQ_D / 256; ++k) (&s_acc[0][0][0])[t + 256 * k] = 0.f;
  for (int64_t base = e0; base < e1; base += 256) {
    const int64_t e = base + t;
    const unsigned lc = e < e1 ? (unsigned)(codes[e] - cb) : 0xffffffffu;
    const bool hit = lc < (unsigned)VQ_CPW;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wc[wave] = __popcll(m);
    __syncthreads();                                   // (also orders the zero-fill / the previous chunk's list reads before this chunk's list writes)
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_wc[w]; if (w < wave) off += c; total += c; }
    if (hit) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = t | ((int)lc << 8);
    __syncthreads();
    // hit q of the chunk (ascending entry order) belongs to half-wave q mod 8; four hits per trip so that their vector loads are in flight together
    // (a code that attracts every token — an untrained model has a handful of codes in use — makes this loop the whole kernel: one load per trip
    // cost 4.7 ms per launch at 131 072 tokens on 6 codes)
    for (int q0 = grp; q0 < total; q0 += 32) {
      float v[4];
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + 8 * u;
        c[u] = -1;
        v[u] = 0.f;
        if (q < total) {
          const int item = s_list[q];
          c[u] = item >> 8;
          v[u] = contrib[(size_t)(base + (item & 255)) * VQ_D + gl];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c[u] >= 0) s_acc[grp][c[u]][gl] += v[u];
    }
  }
  __syncthreads();
  // the eight half-wave copies of a code are added in a fixed order
#pragma unroll
  for (int k = 0; k < VQ_CPW * VQ_D / 256; ++k) {
    const int idx = t + 256 * k, c = idx >> 5, col = idx & 31;
    float a = s_acc[0][c][col];
#pragma unroll
    for (int g = 1; g < 8; ++g) a += s_acc[g][c][col];
    if (cb + c < K) part[((size_t)slice * K + cb + c) * VQ_D + col] = a;
  }
}
__global__ __launch_bounds__(256) void vq_de_finalize_kernel(const float* __restrict__ part, int K, float* __restrict__ dE) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * VQ_D) return;
  float s = part[i];
#pragma unroll
  for (int r = 1; r < VQ_RS; ++r) s += part[(size_t)r * K * VQ_D + i];
  dE[i] += s;
}

// ---------------------------------------------------------------------------------------------
// backward: replay r_i, then reverse recursion for the cross-depth term
// ---------------------------------------------------------------------------------------------
template <int MAXD>
__global__ __launch_bounds__(256) void vq_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ E, const float* __restrict__ en,
    const float* __restrict__ enrm, const int64_t* __restrict__ idx, const float* __restrict__ g_out, float g_loss,
    const float* __restrict__ g_loss_dev, int64_t M, int depth, int use_norm, int use_residual, float beta,
    float* __restrict__ dz, uint16_t* __restrict__ dz_bf16, int f16, float* __restrict__ contrib, int* __restrict__ codes) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, hi = lane >> 5;
  const int64_t tok = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + col;
  const bool live = tok < M;  // whole-wave uniform shuffles below still execute for dead lanes
  const int64_t tk = live ? tok : 0;
  float gL = g_loss * (g_loss_dev ? g_loss_dev[0] : 1.f) / (float)depth;
  const float c = 2.0f / ((float)M * (float)VQ_D);

  float r[16], zn[MAXD][16], e[MAXD][16], den[MAXD];
  int code[MAXD];
  {
    const float4* src = reinterpret_cast<const float4*>(z + (size_t)tk * VQ_D + hi * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { float4 v = src[q]; r[q * 4] = v.x; r[q * 4 + 1] = v.y; r[q * 4 + 2] = v.z; r[q * 4 + 3] = v.w; }
  }
#pragma unroll
  for (int i = 0; i < MAXD; ++i) {
    if (i < depth) {
      if (use_norm) {
        float sp = chain16_sq(r);
        float s = sp + __shfl_xor(sp, 32, 64);
        den[i] = fmaxf(sqrtf(s), 1e-12f);
      } else den[i] = 1.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) zn[i][j] = use_norm ? r[j] / den[i] : r[j];
      code[i] = (int)idx[(size_t)tk * depth + i];
      const float4* ep = reinterpret_cast<const float4*>(en + (size_t)code[i] * VQ_D + hi * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) { float4 v = ep[q]; e[i][q * 4] = v.x; e[i][q * 4 + 1] = v.y; e[i][q * 4 + 2] = v.z; e[i][q * 4 + 3] = v.w; }
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = r[j] - e[i][j];
    }
  }
  float G[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) G[j] = 0.f;
  float own0[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) own0[j] = 0.f;
#pragma unroll
  for (int i = MAXD - 1; i >= 0; --i) {
    if (i < depth) {
      // codebook side: u = J^T(E[idx_i])[ gL*c*(en - zn) - G ]
      float v[16], dotp = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { v[j] = gL * c * (e[i][j] - zn[i][j]) - G[j]; dotp = fmaf(e[i][j], v[j], dotp); }
      if (use_norm) {
        const float dot = dotp + __shfl_xor(dotp, 32, 64);
        const float nr = enrm[code[i]];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (v[j] - e[i][j] * dot) / nr;
      }
      if (live) {     // this entry's share of the codebook gradient: summed per code, in a fixed order, by vq_de_partial_kernel
        const size_t e = (size_t)i * (size_t)M + (size_t)tok;
        float4* cp = reinterpret_cast<float4*>(contrib + e * VQ_D + hi * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) cp[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
        if (hi == 0) codes[e] = code[i];
      }
      // encoder side: own_i = J^T(r_i)[ gL*beta*c*(zn - en) ]
      float w[16], dp2 = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) { w[j] = gL * beta * c * (zn[i][j] - e[i][j]); dp2 = fmaf(zn[i][j], w[j], dp2); }
      if (use_norm) {
        const float dot2 = dp2 + __shfl_xor(dp2, 32, 64);
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = (w[j] - zn[i][j] * dot2) / den[i];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) { G[j] = G[j] + w[j]; if (i == 0) own0[j] = w[j]; }
    }
  }
  if (live) {
    float o[16];
    const float4* gp = reinterpret_cast<const float4*>(g_out + (size_t)tok * VQ_D + hi * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { float4 v = gp[q]; o[q * 4] = v.x; o[q * 4 + 1] = v.y; o[q * 4 + 2] = v.z; o[q * 4 + 3] = v.w; }
    if (!use_residual) {
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = o[j] + own0[j];
    }
    float4* dst = reinterpret_cast<float4*>(dz + (size_t)tok * VQ_D + hi * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[q * 4], o[q * 4 + 1], o[q * 4 + 2], o[q * 4 + 3]);
    if (dz_bf16) {
      uint4* d16 = reinterpret_cast<uint4*>(dz_bf16 + (size_t)tok * VQ_D + hi * 16);
      d16[0] = f16 ? make_uint4(pack2<F16>(o[0], o[1]), pack2<F16>(o[2], o[3]), pack2<F16>(o[4], o[5]), pack2<F16>(o[6], o[7]))
                   : make_uint4(pack2<BF16>(o[0], o[1]), pack2<BF16>(o[2], o[3]), pack2<BF16>(o[4], o[5]), pack2<BF16>(o[6], o[7]));
      d16[1] = f16 ? make_uint4(pack2<F16>(o[8], o[9]), pack2<F16>(o[10], o[11]), pack2<F16>(o[12], o[13]), pack2<F16>(o[14], o[15]))
                   : make_uint4(pack2<BF16>(o[8], o[9]), pack2<BF16>(o[10], o[11]), pack2<BF16>(o[12], o[13]), pack2<BF16>(o[14], o[15]));
    }
  }
}

// decode_codes front half: out = sum_i n(E[idx_i])   (vitvqgan.py:82-87)
__global__ void vq_lookup_kernel(const float* __restrict__ E, const int64_t* __restrict__ idx, int64_t M, int depth,
                                 int use_norm, float* __restrict__ out, uint16_t* __restrict__ out_bf16, int f16) {
  const int64_t tok = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= M) return;
  float acc[VQ_D];
  for (int j = 0; j < VQ_D; ++j) acc[j] = 0.f;
  for (int i = 0; i < depth; ++i) {
    const float* row = E + (size_t)idx[(size_t)tok * depth + i] * VQ_D;
    float x[VQ_D];
    for (int j = 0; j < VQ_D; ++j) x[j] = row[j];
    if (use_norm) {
      float s = chain16_sq(x) + chain16_sq(x + 16);
      float den = fmaxf(sqrtf(s), 1e-12f);
      for (int j = 0; j < VQ_D; ++j) x[j] = x[j] / den;
    }
    for (int j = 0; j < VQ_D; ++j) acc[j] = acc[j] + x[j];
  }
  for (int j = 0; j < VQ_D; ++j) {
    if (out) out[(size_t)tok * VQ_D + j] = acc[j];
    if (out_bf16) out_bf16[(size_t)tok * VQ_D + j] = f16 ? pack1<F16>(acc[j]) : pack1<BF16>(acc[j]);
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline int vq_kpad(int K) { return (K + VQ_TILE - 1) / VQ_TILE * VQ_TILE; }
static inline int64_t vq_nblocks(int64_t M) { return (M + 127) / 128; }

struct VqWs { float *en, *ee, *enrm, *partials, *contrib, *depart; int* codes; };
static VqWs vq_carve(void* ws, int K, int64_t M, int depth) {
  const size_t kp = (size_t)vq_kpad(K);
  VqWs w;
  w.en = reinterpret_cast<float*>(ws);
  w.ee = w.en + kp * VQ_D;
  w.enrm = w.ee + kp;
  w.partials = w.enrm + kp;
  const size_t n = (size_t)M * (size_t)(depth < 1 ? 1 : depth);
  // (backward only) per-entry contribution vectors, the 8 slice partials of the codebook gradient, per-entry codes — each 16-byte aligned
  w.contrib = w.partials + ((size_t)vq_nblocks(M) * (size_t)(depth < 1 ? 1 : depth) + 3) / 4 * 4;
  w.depart = w.contrib + n * VQ_D;
  w.codes = reinterpret_cast<int*>(w.depart + (size_t)VQ_RS * (size_t)K * VQ_D);
  return w;
}

extern "C" size_t enh_vq_workspace_bytes(int64_t M, int n_embed, int depth) {
  const size_t kp = (size_t)vq_kpad(n_embed);
  const size_t d = (size_t)(depth < 1 ? 1 : depth), n = (size_t)M * d;
  return (kp * VQ_D + 2 * kp + ((size_t)vq_nblocks(M) * d + 3) / 4 * 4 + n * VQ_D + (size_t)VQ_RS * (size_t)n_embed * VQ_D + n) * sizeof(float) + 256;
}

extern "C" int enh_vq_forward(const float* z, const float* codebook, int64_t M, int n_embed, int embed_dim,
                              float beta, int depth, int use_norm, float* zq_out, enh_h16* zq_bf16,
                              int64_t* idx_out, float* loss_out, void* workspace, size_t workspace_bytes,
                              int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_vq_forward");
  ENH_REQUIRE(z && codebook && zq_out && idx_out && loss_out && workspace, ENH_E_BADARG, "enh_vq_forward: null pointer");
  ENH_REQUIRE(M > 0 && n_embed > 0 && depth >= 1, ENH_E_BADARG, "enh_vq_forward: M=%lld n_embed=%d depth=%d", (long long)M, n_embed, depth);
  ENH_REQUIRE(embed_dim == VQ_D, ENH_E_SHAPE, "enh_vq_forward: embed_dim must be 32, got %d", embed_dim);
  ENH_REQUIRE(workspace_bytes >= enh_vq_workspace_bytes(M, n_embed, depth), ENH_E_WORKSPACE, "enh_vq_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int kp = vq_kpad(n_embed);
  VqWs w = vq_carve(workspace, n_embed, M, depth);
  vq_prep_kernel<<<(kp + 255) / 256, 256, 0, s>>>(codebook, w.en, w.ee, w.enrm, n_embed, kp, use_norm);
  const int nb = (int)vq_nblocks(M);
  vq_nn_kernel<<<nb, 256, 0, s>>>(z, w.en, w.ee, M, kp, depth, use_norm, zq_out, zq_bf16, dtype == ENH_DT_F16, idx_out, w.partials);
  vq_loss_finalize_kernel<<<1, 256, 0, s>>>(w.partials, nb, depth, M, beta, loss_out);
  return enh_check_launch("enh_vq_forward");
}

extern "C" int enh_vq_backward(const float* z, const float* codebook, const int64_t* idx, const float* g_out,
                               float g_loss, const float* g_loss_dev, int64_t M, int n_embed, int embed_dim,
                               float beta, int depth, int use_residual, int use_norm, float* dz,
                               enh_h16* dz_bf16, float* d_codebook, void* workspace, size_t workspace_bytes,
                               int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_vq_backward");
  ENH_REQUIRE(z && codebook && idx && g_out && dz && d_codebook && workspace, ENH_E_BADARG, "enh_vq_backward: null pointer");
  const int D = depth;
  ENH_REQUIRE(M > 0 && n_embed > 0 && D >= 1 && D <= 8, ENH_E_BADARG, "enh_vq_backward: M=%lld n_embed=%d depth=%d (|depth| <= 8)", (long long)M, n_embed, depth);
  ENH_REQUIRE(embed_dim == VQ_D, ENH_E_SHAPE, "enh_vq_backward: embed_dim must be 32, got %d", embed_dim);
  ENH_REQUIRE(workspace_bytes >= enh_vq_workspace_bytes(M, n_embed, D), ENH_E_WORKSPACE, "enh_vq_backward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int kp = vq_kpad(n_embed);
  VqWs w = vq_carve(workspace, n_embed, M, D);
  vq_prep_kernel<<<(kp + 255) / 256, 256, 0, s>>>(codebook, w.en, w.ee, w.enrm, n_embed, kp, use_norm);
  const int nb = (int)vq_nblocks(M);
  if (D == 1)
    vq_bwd_kernel<1><<<nb, 256, 0, s>>>(z, codebook, w.en, w.enrm, idx, g_out, g_loss, g_loss_dev, M, D, use_norm, use_residual, beta, dz, dz_bf16, dtype == ENH_DT_F16, w.contrib, w.codes);
  else if (D <= 4)
    vq_bwd_kernel<4><<<nb, 256, 0, s>>>(z, codebook, w.en, w.enrm, idx, g_out, g_loss, g_loss_dev, M, D, use_norm, use_residual, beta, dz, dz_bf16, dtype == ENH_DT_F16, w.contrib, w.codes);
  else
    vq_bwd_kernel<8><<<nb, 256, 0, s>>>(z, codebook, w.en, w.enrm, idx, g_out, g_loss, g_loss_dev, M, D, use_norm, use_residual, beta, dz, dz_bf16, dtype == ENH_DT_F16, w.contrib, w.codes);
  vq_de_partial_kernel<<<dim3((unsigned)((n_embed + VQ_CPW - 1) / VQ_CPW), VQ_RS), 256, 0, s>>>(w.codes, w.contrib, M * (int64_t)D, n_embed, w.depart);
  vq_de_finalize_kernel<<<(unsigned)(((int64_t)n_embed * VQ_D + 255) / 256), 256, 0, s>>>(w.depart, n_embed, d_codebook);
  return enh_check_launch("enh_vq_backward");
}

extern "C" int enh_vq_lookup(const float* codebook, const int64_t* idx, int64_t M, int n_embed, int embed_dim,
                             int depth, int use_norm, float* out, enh_h16* out_bf16, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_vq_lookup");
  ENH_REQUIRE(codebook && idx && (out || out_bf16), ENH_E_BADARG, "enh_vq_lookup: null pointer");
  ENH_REQUIRE(embed_dim == VQ_D && depth >= 1 && M > 0, ENH_E_SHAPE, "enh_vq_lookup: embed_dim must be 32, depth >= 1");
  vq_lookup_kernel<<<(int)((M + 127) / 128), 128, 0, (hipStream_t)stream>>>(codebook, idx, M, depth, use_norm, out, out_bf16, dtype == ENH_DT_F16);
  return enh_check_launch("enh_vq_lookup");
}
