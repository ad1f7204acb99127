// conv_igemm.hip — implicit-GEMM convolutions on channels-last bf16 activations for gfx950 (CDNA4), wave64.
//
// Serves (a) the StyleGAN2 discriminator's equalised-lr convolutions — 3x3 / 1x1, stride 1 / 2, forward, input gradient and weight
// gradient, each of them also as a term of the R1 penalty's second-order pass (reference enhancing/losses/layers.py:163-185 EqualConv2d,
// enhancing/losses/op/conv2d_gradfix.py:22-42,81-195) and (b) the 3x3 convolutions of the LPIPS VGG16 trunk (lpips 0.1.4
// pretrained_networks.vgg16; reference call sites enhancing/losses/vqperceptual.py:29,43,74,115).
//
// No `cols` tensor exists.  One geometry description (enh_conv_geom) covers every role:
//   GEMM row m = (b, y, x) of a logical grid Hm x Wm ; contraction index k = (tap (jy, jx), channel c) ; the A operand element is
//       src[b, y*gs + oy0 + jy*sty, x*gs + ox0 + jx*stx, c]      (zero outside the source image)
//   and the result row is written at pixel (y*os + oph, x*os + opw) of the output tensor.
//     forward, stride s, padding p, k x k :  gs = s, oy0 = -p, sty = +1, nty = k, dense output
//     input gradient, stride 1            :  src = dy, gs = 1, oy0 = +p, sty = -1 (the taps run backwards), weights transposed
//     input gradient, stride 2            :  one launch per output parity class (ph, pw): only the taps kh = (ph+p)&1, +2, .. reach
//                                            rows of that parity, their sources are dy[(y' + (ph+p-kh0)/2 - jy)], and the rows are
//                                            written with os = 2, oph = ph (a class without taps writes zeros)
// The operand is GATHERED in the load stage — 8 consecutive channels of one tap per 16-byte load — into the same XOR-swizzled LDS
// images the dense kernels use (gemm_tiles.h); the weights are a small [N][taps*C] matrix packed by enh_conv_pack_weight.
// The weight gradient is the transposed problem: dW[co][(tap, ci)] = sum over pixels of dy[pix][co] * src[gathered pix, tap][ci] — both
// operands contraction-major, staged as they lie and read with the LDS transpose read; the pixel axis is split over the grid and the
// partial slabs are added in a fixed order (deterministic, no atomics).
// Register-staged double buffer (the gather needs per-lane predication, which global_load_lds cannot do), 128 x 128 x 64 tile,
// 4 waves of 64 x 64 (v_mfma_f32_16x16x32_bf16), 2 workgroups per CU.  C and N multiples of 8; any grid size.
#include "gemm_tiles.h"

// conv_pointwise.hip: the dense 1 x 1 geometries with an 8-channel side (the discriminator's first layer) as streaming kernels
int conv_pointwise_forward(const uint16_t* src, const uint16_t* wt, const enh_conv_geom& g, int mode, const float* bias, float p0, float p1, uint16_t* out,
                           int dtype, hipStream_t stream);
int conv_pointwise_wgrad_slabs(const enh_conv_geom& g);
void conv_pointwise_wgrad(const uint16_t* src, const uint16_t* dy, const enh_conv_geom& g, float* ws, float* dw, int dtype, hipStream_t stream);

struct ConvArgs {
  const uint16_t* X; const uint16_t* Wt;
  enh_conv_geom g;
  int64_t M, K;
  const float* bias; int mode; const uint16_t* aux; const uint16_t* add; float p0, p1;
  uint16_t* out;
  int nbm, nbn;
  float* ws; int splits, st_per_split;   // split over the K stages (small grids): partial slabs [split][M][N] f32, finished by conv_splitk_finish_kernel
};

// Workgroup b runs on XCD b % 8 (private 4 MiB L2 each): hand every XCD a CONTIGUOUS run of the linear tile order, so that the vertical halo of
// a 3x3 tap grid (image rows y-1, y+1 = tiles +-W/128 away) and the nine taps' re-reads of one pixel range hit the same L2 instead of being
// fetched from the Infinity Cache once per XCD.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// output element offset (in pixels) of GEMM row m
__device__ __forceinline__ int64_t conv_out_pixel(const enh_conv_geom& g, int64_t m) {
  // 32-bit divisions (the launcher requires fewer than 2^31 GEMM rows): a 64-bit one costs ~5x the instructions, and a strided epilogue does four per lane
  const unsigned hw = (unsigned)g.Hm * (unsigned)g.Wm, mm = (unsigned)m;
  const unsigned b = mm / hw, rem = mm - b * hw;
  const unsigned y = rem / (unsigned)g.Wm, x = rem - y * (unsigned)g.Wm;
  return ((int64_t)b * g.HO + (int64_t)y * g.os + g.oph) * g.WO + (int64_t)x * g.os + g.opw;
}

// the five epilogue modes on four consecutive output columns (bias b4, saved activation ax, addend ad as they were loaded)
template <typename OT>
__device__ __forceinline__ void conv_epi_value(const ConvArgs& args, float (&v)[4], const float4& b4, const uint2& ax, const uint2& ad) {
  const uint32_t d0 = ad.x, d1 = ad.y;
  const float e0 = unpack1<OT>((uint16_t)(d0 & 0xffffu)), e1 = unpack1<OT>((uint16_t)(d0 >> 16));
  const float e2 = unpack1<OT>((uint16_t)(d1 & 0xffffu)), e3 = unpack1<OT>((uint16_t)(d1 >> 16));
  if (args.mode == 0) {
    v[0] = fmaxf(v[0] + b4.x, 0.f); v[1] = fmaxf(v[1] + b4.y, 0.f); v[2] = fmaxf(v[2] + b4.z, 0.f); v[3] = fmaxf(v[3] + b4.w, 0.f);
  } else if (args.mode == 1) {
    const uint32_t a0 = ax.x, a1 = ax.y;
    v[0] = (a0 & 0x7fffu) && !(a0 & 0x8000u) ? v[0] + e0 : 0.f;
    v[1] = ((a0 >> 16) & 0x7fffu) && !(a0 >> 31) ? v[1] + e1 : 0.f;
    v[2] = (a1 & 0x7fffu) && !(a1 & 0x8000u) ? v[2] + e2 : 0.f;
    v[3] = ((a1 >> 16) & 0x7fffu) && !(a1 >> 31) ? v[3] + e3 : 0.f;
  } else if (args.mode == 3) {
    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (v[r] > 0.f ? v[r] : v[r] * args.p0) * args.p1;
  } else if (args.mode == 4) {
    v[0] += args.p0 * e0; v[1] += args.p0 * e1; v[2] += args.p0 * e2; v[3] += args.p0 * e3;
  }
}

template <typename OT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& args, f32x4 (&acc)[4][4], int64_t m0, int64_t n0, int wm, int wn, int lg, int l16,
                                              unsigned char* stage) {
  const enh_conv_geom& g = args.g;
  const bool dense = g.os == 1 && g.HO == g.Hm && g.WO == g.Wm && g.oph == 0 && g.opw == 0;
  // wave-uniform: the wave's 64 columns all exist and the rows are consecutive output pixels -> its 64 x 64 block leaves through LDS (see below; the
  // strided rows of a stride-2 input gradient gain nothing from whole-row stores and pay for eight pixel decompositions per lane: 377 -> 412 us)
  const bool staged = dense && n0 + wn * 64 + 64 <= g.N;
  // epilogue: lane (lg, l16) holds out[m = m0 + wm*64 + i*16 + l16][n = n0 + wn*64 + j*16 + lg*4 + 0..3].  Everything the epilogue READS (bias, aux, add)
  // is requested for all 16 element groups before the first store: CDNA4's vmcnt retires loads and stores in order, so a load issued after a store
  // can only be waited for together with that store's acknowledgement (gemm_tiles.h epi_bias).
  int64_t orow[4];
  float4 b4[4];
  uint2 ax[4][4], ad[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
    b4[j] = ((args.mode == 0 || args.mode == 3) && args.bias && n < g.N) ? *reinterpret_cast<const float4*>(args.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + l16;
    orow[i] = m < args.M ? (dense ? m : conv_out_pixel(g, m)) * g.N : -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      ax[i][j] = make_uint2(0u, 0u); ad[i][j] = make_uint2(0u, 0u);
      if (orow[i] >= 0 && n < g.N) {
        if (args.mode == 1) ax[i][j] = *reinterpret_cast<const uint2*>(args.aux + orow[i] + n);
        if ((args.mode == 1 || args.mode == 4) && args.add) ad[i][j] = *reinterpret_cast<const uint2*>(args.add + orow[i] + n);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (orow[i] < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= g.N) continue;   // N % 8 == 0: the 4 columns are in or out together
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const uint32_t d0 = ad[i][j].x, d1 = ad[i][j].y;
      const float e0 = unpack1<OT>((uint16_t)(d0 & 0xffffu)), e1 = unpack1<OT>((uint16_t)(d0 >> 16));
      const float e2 = unpack1<OT>((uint16_t)(d1 & 0xffffu)), e3 = unpack1<OT>((uint16_t)(d1 >> 16));
      if (args.mode == 0) {
        v[0] = fmaxf(v[0] + b4[j].x, 0.f); v[1] = fmaxf(v[1] + b4[j].y, 0.f); v[2] = fmaxf(v[2] + b4[j].z, 0.f); v[3] = fmaxf(v[3] + b4[j].w, 0.f);
      } else if (args.mode == 1) {
        const uint32_t a0 = ax[i][j].x, a1 = ax[i][j].y;
        v[0] = (a0 & 0x7fffu) && !(a0 & 0x8000u) ? v[0] + e0 : 0.f;
        v[1] = ((a0 >> 16) & 0x7fffu) && !(a0 >> 31) ? v[1] + e1 : 0.f;
        v[2] = (a1 & 0x7fffu) && !(a1 & 0x8000u) ? v[2] + e2 : 0.f;
        v[3] = ((a1 >> 16) & 0x7fffu) && !(a1 >> 31) ? v[3] + e3 : 0.f;
      } else if (args.mode == 3) {
        v[0] += b4[j].x; v[1] += b4[j].y; v[2] += b4[j].z; v[3] += b4[j].w;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] > 0.f ? v[r] : v[r] * args.p0) * args.p1;
      } else if (args.mode == 4) {
        v[0] += args.p0 * e0; v[1] += args.p0 * e1; v[2] += args.p0 * e2; v[3] += args.p0 * e3;
      }
      const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
      if (staged) {
        const int row = i * 16 + l16;
        *reinterpret_cast<u32x2*>(stage + row * 128 + (((j * 2 + (lg >> 1)) ^ (row & 7)) << 4) + (lg & 1) * 8) = o_;
      } else {
        *reinterpret_cast<u32x2*>(args.out + orow[i] + n) = o_;
      }
    }
  }
  if (staged) {
    // The accumulator layout offers 8 bytes per lane, 16 rows x 32 B per store instruction; through a wave-private XOR-swizzled 8-KiB LDS tile the block
    // leaves as 16 bytes per lane, whole 128-byte row segments (gemm.hip gemm_epilogue32_loops has the measurements: 3.1-3.9 -> ~5 TB/s of stores).
    // Rows that do not exist (orow < 0 above) were skipped by their writer lanes and are skipped by their reader lanes.
    const int lane = lg * 16 + l16;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = p * 8 + (lane >> 3), c = lane & 7;
      const int64_t m = m0 + wm * 64 + row;
      const u32x4 w = *reinterpret_cast<const u32x4*>(stage + row * 128 + ((c ^ (row & 7)) << 4));
      if (m < args.M) *reinterpret_cast<u32x4*>(args.out + m * g.N + n0 + wn * 64 + c * 8) = w;
    }
  }
}

//   mode 0: out = relu(acc + bias[n])                                   (VGG16 conv + ReLU)
//   mode 1: out = (acc + add[o,n]) * (aux[o,n] > 0)                     (VGG16 input gradient through a ReLU; add optional)
//   mode 2: out = acc
//   mode 3: out = lrelu(acc + bias[n], slope p0) * p1                   (EqualConv2d + FusedLeakyReLU; bias optional)
//   mode 4: out = acc + p0 * add[o,n]                                   (the residual merge of a StyleBlock folded into its skip convolution)
template <typename OT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile | B tile]
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;
  const int bid_ = xcd_contiguous((int)blockIdx.x, args.nbm * args.nbn);
  const int tile_m = bid_ % args.nbm, tile_n = bid_ / args.nbm;
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int nk = (int)((args.K + G_BK - 1) / G_BK);

  // this thread gathers chunk c (8 channels) of rows r0 + 32*i: the pixel of a row does not change along K
  const int c = t & 7, r0 = t >> 3;
  int py[4], px[4];
  int64_t pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = m0 + r0 + 32 * i;
    if (row < args.M) {
      const unsigned hw = (unsigned)g.Hm * (unsigned)g.Wm, rr = (unsigned)row;      // < 2^31 rows (launcher): 32-bit divisions
      const unsigned b = rr / hw, rem = rr - b * hw;
      const int y = (int)(rem / (unsigned)g.Wm), x = (int)(rem - (unsigned)y * (unsigned)g.Wm);
      py[i] = y * g.gs; px[i] = x * g.gs;
      pb[i] = (int64_t)b * g.Hs * g.Ws;
    } else { py[i] = -(1 << 28); px[i] = -(1 << 28); pb[i] = 0; }   // every tap of an out-of-range row falls outside the image -> zeros
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // (a second register set for the gathered operand — loads two K steps ahead — was measured SLOWER: 285-383 vs 405 TF/s; the allocator then
  //  fills all 256 registers and spills)
  u32x4 ra[4], rb[4];
  auto gather = [&](int64_t k0) {
    const int64_t kk = k0 + c * 8;
    const int tap = (int)(kk / g.C), ch = (int)(kk - (int64_t)tap * g.C);
    const int jy = tap / g.ntx, jx = tap - jy * g.ntx;
    const int dy = g.oy0 + jy * g.sty, dx = g.ox0 + jx * g.stx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = py[i] + dy, sx = px[i] + dx;
      const bool ok = kk < args.K && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      ra[i] = ok ? *reinterpret_cast<const u32x4*>(args.X + (pb[i] + (int64_t)sy * g.Ws + sx) * g.C + ch) : zero4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t co = n0 + r0 + 32 * i;
      rb[i] = (co < g.N && kk < args.K) ? *reinterpret_cast<const u32x4*>(args.Wt + co * args.K + kk) : zero4;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    gather(0);
    tile_sstore<false>(ra, smem, t);
    tile_sstore<false>(rb, smem + G_TILE_BYTES, t);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) gather((int64_t)(kt + 1) * G_BK);
    const unsigned char* sa = smem + stage * (2 * G_TILE_BYTES);
    const unsigned char* sb = sa + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = tile_frag<false>(sa, wm * 64 + i * 16, ks, lg, l16);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = tile_frag<false>(sb, wn * 64 + j * 16, ks, lg, l16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<OT>(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      unsigned char* na = smem + (stage ^ 1) * (2 * G_TILE_BYTES);
      tile_sstore<false>(ra, na, t);
      tile_sstore<false>(rb, na + G_TILE_BYTES, t);
    }
    __syncthreads();
  }
  conv_epilogue<OT>(args, acc, m0, n0, wm, wn, lg, l16, smem + wave * 8192);   // the stages are free: no LDS read follows the loop's last barrier
}

// =================================================================================================
// LDS-DMA form of the same convolution for C % 64 == 0 (every 3x3 / 1x1 layer of the discriminator past the first, the VGG16 trunk): a 64-deep K step
// then lies inside ONE tap, so the gather is "row pointer + a tap offset that is uniform for the workgroup", and global_load_lds (16 B per lane straight
// into the swizzled LDS image, no staging registers, no ds_write pass) can fetch it — the zero padding by pointing the lanes whose tap falls outside
// the image at a zero page instead of predicating them.  K loop = gemm_bf16_pipe2_kernel's (gemm.hip): one mid-iteration barrier, the loads of stage
// kt+2 spread one per two MFMAs, fragments of the next half-step read under the MFMAs of the current one.
// =================================================================================================
__device__ __attribute__((aligned(16))) uint32_t g_conv_zero_page[4] = {0u, 0u, 0u, 0u};

template <typename OT>
__global__ __launch_bounds__(256, 2) void conv_igemm_glds_kernel(const ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile | B tile]
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;
  const int ntile = args.nbm * args.nbn;
  const int lin_ = xcd_contiguous((int)blockIdx.x, ntile * args.splits);
  const int split = lin_ / ntile, bid_ = lin_ - split * ntile;
  const int tile_m = bid_ % args.nbm, tile_n = bid_ / args.nbm;
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int ks0 = split * args.st_per_split;                 // this workgroup's K stages: [ks0, ks0 + nk)
  const int nk_all = (int)(args.K / G_BK);
  const int nk = nk_all - ks0 < args.st_per_split ? nk_all - ks0 : args.st_per_split;

  // A operand: slab i of this wave = rows (wave*4 + i)*8 + (lane>>3), physical chunk lane&7 holding logical chunk c = pc ^ ((r>>1)&7) (gemm_tiles.h row image)
  int py[4], px[4], coff[4];
  int64_t pb[4];
  const uint16_t* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((r >> 1) & 7);
    coff[i] = c * 8;
    const int64_t row = m0 + r;
    if (row < args.M) {
      const unsigned hw = (unsigned)g.Hm * (unsigned)g.Wm, rr = (unsigned)row;      // < 2^31 rows (launcher): 32-bit divisions
      const unsigned b = rr / hw, rem = rr - b * hw;
      const int y = (int)(rem / (unsigned)g.Wm), x = (int)(rem - (unsigned)y * (unsigned)g.Wm);
      py[i] = y * g.gs; px[i] = x * g.gs; pb[i] = (int64_t)b * g.Hs * g.Ws;
    } else { py[i] = -(1 << 28); px[i] = -(1 << 28); pb[i] = 0; }
    int64_t co = n0 + r;                       // B operand (weights [N][K]): rows beyond N are clamped, their products land in columns that are never stored
    if (co > g.N - 1) co = g.N - 1;
    bsrc[i] = args.Wt + co * args.K + (int64_t)ks0 * G_BK + c * 8;
  }
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_conv_zero_page);
  // source pointers of K step `ks` of this workgroup (uniform tap) for the four A slabs
  auto a_ptrs = [&](int ks, const uint16_t* (&ap)[4]) {
    const int k0 = (ks0 + ks) * G_BK;
    const int tap = k0 / g.C, ch0 = k0 - tap * g.C;
    const int jy = tap / g.ntx, jx = tap - jy * g.ntx;
    const int dy = g.oy0 + jy * g.sty, dx = g.ox0 + jx * g.stx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = py[i] + dy, sx = px[i] + dx;
      const bool ok = sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      ap[i] = ok ? args.X + (pb[i] + (int64_t)sy * g.Ws + sx) * g.C + ch0 + coff[i] : zero;
    }
  };
#define CG_LOAD(PTR, BUF, WHICH, I)                                                                                                          \
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)(PTR), (LDS_AS void*)(smem + (BUF) * (2 * G_TILE_BYTES) + (WHICH) * G_TILE_BYTES + (wave * 4 + (I)) * 1024), 16, 0, ENH_GLDS_AUX)
#define CG_READ(FA, FB, BUF, KS)                                                                                         \
  do {                                                                                                                   \
    const unsigned char* sa_ = smem + (BUF) * (2 * G_TILE_BYTES);                                                        \
    const unsigned char* sb_ = sa_ + G_TILE_BYTES;                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) FA[i_] = tile_frag<false>(sa_, wm * 64 + i_ * 16, KS, lg, l16);     \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) FB[j_] = tile_frag<false>(sb_, wn * 64 + j_ * 16, KS, lg, l16);     \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  const uint16_t* ap[4];

  if (nk > 0) {
    a_ptrs(0, ap);
#pragma unroll
    for (int i = 0; i < 4; ++i) { CG_LOAD(ap[i], 0, 0, i); CG_LOAD(bsrc[i], 0, 1, i); bsrc[i] += G_BK; }
    if (nk > 1) {
      a_ptrs(1, ap);
#pragma unroll
      for (int i = 0; i < 4; ++i) { CG_LOAD(ap[i], 1, 0, i); CG_LOAD(bsrc[i], 1, 1, i); bsrc[i] += G_BK; }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 landed (stage 1's 8 loads may be outstanding)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    CG_READ(fa0, fb0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  }
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    CG_READ(fa1, fb1, buf, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = mfma16<OT>(fb0[j], fa0[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): F1 in registers, my share of stage kt+1 landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) CG_READ(fa0, fb0, buf ^ 1, 0);
    const bool more = kt + 2 < nk;
    if (more) a_ptrs(kt + 2, ap);
    __builtin_amdgcn_sched_barrier(0);
    // second half: the 8 loads of stage kt+2 (into the buffer stage kt just vacated) one per two MFMAs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        acc[i][jj * 2] = mfma16<OT>(fb1[jj * 2], fa1[i], acc[i][jj * 2]);
        acc[i][jj * 2 + 1] = mfma16<OT>(fb1[jj * 2 + 1], fa1[i], acc[i][jj * 2 + 1]);
        if (more) {
          if (jj == 0) CG_LOAD(ap[i], buf, 0, i);
          else { CG_LOAD(bsrc[i], buf, 1, i); bsrc[i] += G_BK; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): next F0 has arrived under the MFMAs above
    buf ^= 1;
  }
#undef CG_LOAD
#undef CG_READ
  if (args.ws) {   // split over K: the f32 partial tile goes to this split's slab as it lies in the accumulators (16 bytes per lane)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + wm * 64 + i * 16 + l16;
      if (m >= args.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
        if (n < g.N) *reinterpret_cast<f32x4*>(args.ws + ((int64_t)split * args.M + m) * g.N + n) = acc[i][j];
      }
    }
    return;
  }
  conv_epilogue<OT>(args, acc, m0, n0, wm, wn, lg, l16, smem + wave * 8192);   // the stages are free: no LDS read follows the loop's last barrier
}

// second pass of a split convolution: out[o, n..n+3] = epilogue( sum over the slabs in ascending order ) — one thread per four columns
template <typename OT>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs args) {
  const enh_conv_geom& g = args.g;
  const int n4 = g.N >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= args.M * n4) return;
  const int64_t m = idx / n4;
  const int n = (int)(idx - m * n4) * 4;
  f32x4 a = *reinterpret_cast<const f32x4*>(args.ws + m * g.N + n);
  for (int s_ = 1; s_ < args.splits; ++s_) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(args.ws + ((int64_t)s_ * args.M + m) * g.N + n);
    a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
  }
  const bool dense = g.os == 1 && g.HO == g.Hm && g.WO == g.Wm && g.oph == 0 && g.opw == 0;
  const int64_t orow = (dense ? m : conv_out_pixel(g, m)) * g.N;
  float v[4] = {a[0], a[1], a[2], a[3]};
  const float4 b4 = ((args.mode == 0 || args.mode == 3) && args.bias) ? *reinterpret_cast<const float4*>(args.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint2 ax = args.mode == 1 ? *reinterpret_cast<const uint2*>(args.aux + orow + n) : make_uint2(0u, 0u);
  const uint2 ad = ((args.mode == 1 || args.mode == 4) && args.add) ? *reinterpret_cast<const uint2*>(args.add + orow + n) : make_uint2(0u, 0u);
  conv_epi_value<OT>(args, v, b4, ax, ad);
  const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
  *reinterpret_cast<u32x2*>(args.out + orow + n) = o_;
}

// =================================================================================================
// "w256" form of the convolution (round 4): the dense w256 main loop (gemm.hip gemm_bf16_w256_kernel — 256 x 256 tile, FOUR waves of 128 x 128, one wave per
// SIMD, v_mfma_f32_32x32x16_bf16, two 64-KiB LDS slots, one barrier per 64-deep K stage, one fragment read per MFMA and one global_load_lds per two
// MFMAs) fed by the gather of conv_igemm_glds_kernel: C % 64 == 0, so a K stage lies inside ONE tap and a staged row is "pixel offset + a tap offset that is
// uniform for the wave"; lanes whose tap falls outside the image read the zero page.  Waves 0, 1 stage the two 128-pixel halves of the A tile, waves 2, 3 the
// weights; BOTH run the same instruction stream — a staging lane keeps, per 1-KiB piece u, an element offset and a packed (y, x), and the wave keeps
// (dy, dx, tap offset) in scalar registers (weights: y = x = dy = dx = 0, the offset advances by 64) — so the pointer of a piece costs ~10 vector
// instructions in the MFMA slot where it is issued and no divergence.
// LDS image per slot: the dense kernel's [A0 | A1 | B0 | B1].  N % 256 == 0; other multiples of 128 run conv_igemm_w512_kernel below.
// Output: dense rows leave through a wave-private LDS tile as whole row segments; strided rows (stride-2 input gradient) as 8-byte pieces.
// =================================================================================================
#define CW_SLOT (4 * G_TILE_BYTES)
#define CW_LDS_BYTES (2 * CW_SLOT + 2048)

// accumulator layout of the swapped 32x32 MFMA: acc[i][j][r] = C[mw + i*32 + (lane&31)][nw + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)]
template <typename OT, int NJ>
__device__ __forceinline__ void conv_epilogue32(const ConvArgs& args, f32x16 (&acc)[4][NJ], int64_t mw, int64_t nw, int lane, float* wave_bias, unsigned char* stage) {
  const enh_conv_geom& g = args.g;
  const int l31 = lane & 31, hi = lane >> 5;
  const bool dense = g.os == 1 && g.HO == g.Hm && g.WO == g.Wm && g.oph == 0 && g.opw == 0;
  const bool has_bias = (args.mode == 0 || args.mode == 3) && args.bias;
  const float4 bv = (lane < NJ * 8 && has_bias) ? *reinterpret_cast<const float4*>(args.bias + nw + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool want_aux = args.mode == 1, want_add = (args.mode == 1 || args.mode == 4) && args.add;
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's last fragment reads are done
  __builtin_amdgcn_s_barrier();         // ... and everybody else's: `stage` (and, in the 512-row kernel, the bias strip) overlay the K slots
  if (lane < NJ * 8) *reinterpret_cast<float4*>(wave_bias + lane * 4) = bv;
  constexpr int CH = 4 * NJ;            // 16-byte chunks per staged row (32*NJ columns of bf16)
  constexpr int RPP = 64 / CH;          // rows per read-back pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = mw + i * 32 + l31;
    const int64_t orow = m < args.M ? (dense ? m : conv_out_pixel(g, m)) * g.N : -1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      uint2 ax[4], ad[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int64_t n = nw + j * 32 + 8 * g4 + 4 * hi;
        ax[g4] = make_uint2(0u, 0u); ad[g4] = make_uint2(0u, 0u);
        if (orow >= 0) {
          if (want_aux) ax[g4] = *reinterpret_cast<const uint2*>(args.aux + orow + n);
          if (want_add) ad[g4] = *reinterpret_cast<const uint2*>(args.add + orow + n);
        }
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v[4] = {acc_read(acc[i][j][g4 * 4 + 0]), acc_read(acc[i][j][g4 * 4 + 1]), acc_read(acc[i][j][g4 * 4 + 2]), acc_read(acc[i][j][g4 * 4 + 3])};
        const float4 b4 = *reinterpret_cast<const float4*>(wave_bias + j * 32 + 8 * g4 + 4 * hi);
        conv_epi_value<OT>(args, v, b4, ax[g4], ad[g4]);
        const u32x2 o_ = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
        if (dense) *reinterpret_cast<u32x2*>(stage + l31 * (CH * 16) + (((j * 4 + g4) ^ (l31 & (CH - 1))) << 4) + hi * 8) = o_;
        else if (orow >= 0) *reinterpret_cast<u32x2*>(args.out + orow + nw + j * 32 + 8 * g4 + 4 * hi) = o_;
      }
    }
    if (dense) {
#pragma unroll
      for (int p = 0; p < 32 / RPP; ++p) {
        const int row = p * RPP + lane / CH, c = lane % CH;
        const u32x4 w = *reinterpret_cast<const u32x4*>(stage + row * (CH * 16) + ((c ^ (row & (CH - 1))) << 4));
        const int64_t mm = mw + i * 32 + row;
        if (mm < args.M) *reinterpret_cast<u32x4*>(args.out + mm * g.N + nw + c * 8) = w;
      }
    }
  }
}

template <typename OT, int NJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_igemm_w256_kernel(const ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 slots][A0 | A1 | B0 | B1] + bias strips
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid_ = xcd_contiguous((int)blockIdx.x, args.nbm * args.nbn);
  const int tile_m = bid_ % args.nbm, tile_n = bid_ / args.nbm;
  const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * (64 * NJ);
  const int nst = (int)(args.K / G_BK);   // >= 2 (launcher)

  // ---- staging state -----------------------------------------------------------------------------------------------------------
  static_assert(NJ == 4, "256 x 256 tiles");
  const bool stage_a = wave < 2;                                   // wave-uniform
  constexpr int slab0 = 0;
  unsigned char* const my_sub = smem + wave * G_TILE_BYTES;        // the 16-KiB sub-tile of a slot this wave fills
  const uint16_t* const gbase = stage_a ? args.X : args.Wt;
  const uint16_t* const zero = reinterpret_cast<const uint16_t*>(g_conv_zero_page);
  int rowoff[16], pyx[16];
  {
    // pixel (b, y, x) of this lane's row of piece 0 by one 32-bit division; the rows of pieces 1..15 are 8 pixels further each
    const unsigned hw = (unsigned)g.Hm * (unsigned)g.Wm;
    const unsigned row0 = (unsigned)m0 + wave * 128 + slab0 * 8 + (lane >> 3);
    unsigned pb = row0 / hw, prem = row0 - pb * hw;
    unsigned py_ = prem / (unsigned)g.Wm, px_ = prem - py_ * (unsigned)g.Wm;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = (slab0 + u) * 8 + (lane >> 3), pc = lane & 7;    // row of the sub-tile, physical chunk ; logical chunk of the row image:
      const int c = pc ^ ((r >> 1) & 7);
      if (stage_a) {
        const bool exists = (int64_t)m0 + wave * 128 + r < args.M;
        rowoff[u] = exists ? (int)(((pb * g.Hs + py_ * g.gs) * g.Ws + px_ * g.gs) * g.C) + c * 8 : 0;
        pyx[u] = exists ? (int)(((py_ * g.gs) << 16) | (px_ * g.gs)) : 0x40004000;   // every tap of a row that does not exist falls outside the image
        px_ += 8;
        while (px_ >= (unsigned)g.Wm) { px_ -= (unsigned)g.Wm; ++py_; }
        while (py_ >= (unsigned)g.Hm) { py_ -= (unsigned)g.Hm; ++pb; }
      } else {
        rowoff[u] = (int)((n0 + (wave - 2) * 128 + r) * args.K) + c * 8;
        pyx[u] = 0;
      }
    }
  }
  // scalar tap state of the stage the NEXT request belongs to
  int s_ch = 0, s_jx = 0, s_jy = 0;
  int s_dy = stage_a ? g.oy0 : 0, s_dx = stage_a ? g.ox0 : 0;
  int s_off = stage_a ? (s_dy * g.Ws + s_dx) * g.C : 0;
  const unsigned lim_y = stage_a ? (unsigned)g.Hs : 1u, lim_x = stage_a ? (unsigned)g.Ws : 1u;
#define CW_ADVANCE()                                                                                                              \
  do {                                                                                                                            \
    if (stage_a) {                                                                                                                \
      s_ch += G_BK;                                                                                                               \
      if (s_ch == g.C) { s_ch = 0; if (++s_jx == g.ntx) { s_jx = 0; ++s_jy; } s_dy = g.oy0 + s_jy * g.sty; s_dx = g.ox0 + s_jx * g.stx; } \
      s_off = (s_dy * g.Ws + s_dx) * g.C + s_ch;                                                                                  \
    } else s_off += G_BK;                                                                                                         \
  } while (0)
#define CW_ISSUE_ONE(SLOT, U)                                                                                                     \
  do {                                                                                                                            \
    {                                                                                                                             \
      const unsigned sy_ = (unsigned)((pyx[U] >> 16) + s_dy), sx_ = (unsigned)((pyx[U] & 0xffff) + s_dx);                         \
      const uint16_t* p_ = (sy_ < lim_y && sx_ < lim_x) ? gbase + (rowoff[U] + s_off) : zero;                                     \
      __builtin_amdgcn_global_load_lds((const GLB_AS void*)p_, (LDS_AS void*)(my_sub + (SLOT) * CW_SLOT + (U) * 1024), 16, 0, ENH_GLDS_AUX);  \
    }                                                                                                                             \
  } while (0)
#define CW_READ_ONE(FA, FB, SLOT, S, U)                                                                                           \
  do {                                                                                                                            \
    if ((U) < 4) FA[(U) & 3] = frag32<false>(smem + (SLOT) * CW_SLOT + wm * G_TILE_BYTES, ((U) & 3) * 32, S, lane);               \
    else FB[(U) & 3] = frag32<false>(smem + (SLOT) * CW_SLOT + (2 + wn) * G_TILE_BYTES, ((U) & 3) * 32, S, lane);                 \
  } while (0)
#define CW_MM(Q, FA, FB)                                                                                                          \
  acc[(Q) / NJ][(Q) % NJ] = mfma32<OT>(FB[(Q) % NJ], FA[(Q) / NJ], acc[(Q) / NJ][(Q) % NJ])
#define CW_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one k16 step: 16 MFMAs on (FA, FB); under the first 8 one fragment read each (k-step RS of slot RSLOT into RA / RB); 8 staging requests (pieces
  // G0 .. G0+7 into slot GSLOT) under the odd MFMAs
#define CW_KSTEP(FA, FB, RA, RB, RSLOT, RS, DO_READ, GSLOT, G0, DO_ISSUE)                                                         \
  do {                                                                                                                            \
    CW_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4 * NJ; ++q_) {                                                                       \
      CW_MM(q_, FA, FB);                                                                                                          \
      if ((DO_READ) && q_ < 4 + NJ) { CW_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                       \
      if ((DO_ISSUE) && (q_ & 1)) { CW_ISSUE_ONE(GSLOT, (G0) + (q_ >> 1)); }                                                      \
      CW_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];

  // prologue: stage 0 -> slot 0 completely; pieces 0-7 of stage 1 -> slot 1 (pieces 8-15 follow under the first k-step)
#pragma unroll
  for (int u = 0; u < 16; ++u) CW_ISSUE_ONE(0, u);
  CW_ADVANCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) CW_ISSUE_ONE(1, u);
  __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): stage 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 4 + NJ; ++u) CW_READ_ONE(fa0, fb0, 0, 0, u);
  CW_FENCE();

  // invariant at the top of iteration j: the tap state is at stage j+1, whose pieces 0-7 are already requested into slot (j+1)&1
  int j = 0;
  for (; j + 2 < nst; ++j) {
    const int slot = j & 1;
    CW_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage j+1
    CW_ADVANCE();
    CW_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    CW_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): stage j+1 landed (nothing newer outstanding) ; lgkmcnt(0): this slot is read out
    __builtin_amdgcn_s_barrier();
    CW_FENCE();
    CW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, slot, 0, true);      // + pieces 0-7 of stage j+2 into the slot just vacated
  }
  {  // tail: stages nst-2 and nst-1
    const int slot = j & 1;
    CW_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage nst-1
    CW_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    CW_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    CW_FENCE();
    CW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, 0, 0, false);
    CW_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 1, true, 0, 0, false);
    CW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 2, true, 0, 0, false);
    CW_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 3, true, 0, 0, false);
    CW_KSTEP(fa1, fb1, fa0, fb0, 0, 0, false, 0, 0, false);
  }
#undef CW_ADVANCE
#undef CW_ISSUE_ONE
#undef CW_READ_ONE
#undef CW_MM
#undef CW_KSTEP
#undef CW_FENCE
  conv_epilogue32<OT, NJ>(args, acc, m0 + wm * 128, n0 + wn * (32 * NJ), lane, reinterpret_cast<float*>(smem + 2 * CW_SLOT) + wave * 128, smem + wave * 8192);
}

// =================================================================================================
// "w512": the same main loop for N = 128 layers (the discriminator's 128 -> 128 convolution at 256^2 and its input gradient — the largest single layer).
// 512 x 128 tile, the four waves stacked along M (each 128 x 128: the accumulators, fragment reads and MFMA count per wave are those of the 256 x 256
// kernel), so each wave owns the 128-pixel A sub-tile it multiplies and the ONE 16-KiB weight sub-tile is shared by all four.  LDS image per slot
// [A0 | A1 | A2 | A3 | B0] = 80 KiB, two slots = the CU's whole 160 KiB; every wave stages its own pixels (16 pieces per K stage) and a quarter of the
// weights (4 pieces): 10 requests under the 16 MFMAs of each of the two k16 steps that carry requests.  The epilogue's staging tiles and bias strips
// overlay the slots.  (A 256 x 128 tile with 128 x 64 waves was built first and lost to the 128 x 128 kernel: 614 -> 513 TF/s — half the workgroups
// in flight and 1.5x the LDS traffic per MFMA; profiles/r04_conv_layers.txt.)
// =================================================================================================
#define CX_SLOT (5 * G_TILE_BYTES)
#define CX_LDS_BYTES (2 * CX_SLOT)

template <typename OT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_igemm_w512_kernel(const ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 slots][A0 | A1 | A2 | A3 | B0]
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int bid_ = xcd_contiguous((int)blockIdx.x, args.nbm * args.nbn);
  const int tile_m = bid_ % args.nbm, tile_n = bid_ / args.nbm;
  const int64_t m0 = (int64_t)tile_m * 512, n0 = (int64_t)tile_n * 128;
  const int nst = (int)(args.K / G_BK);   // >= 2 (launcher)

  // ---- staging state: 16 pieces of this wave's pixels (offset + packed (y, x) each), 4 pieces of the weights -------------------------
  const uint16_t* const zero = reinterpret_cast<const uint16_t*>(g_conv_zero_page);
  int rowoff[16], pyx[16], boff[4];
  {
    const unsigned hw = (unsigned)g.Hm * (unsigned)g.Wm;
    const unsigned row0 = (unsigned)m0 + wave * 128 + (lane >> 3);
    unsigned pb = row0 / hw, prem = row0 - pb * hw;
    unsigned py_ = prem / (unsigned)g.Wm, px_ = prem - py_ * (unsigned)g.Wm;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = u * 8 + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((r >> 1) & 7);
      const bool exists = (int64_t)m0 + wave * 128 + r < args.M;
      rowoff[u] = exists ? (int)(((pb * g.Hs + py_ * g.gs) * g.Ws + px_ * g.gs) * g.C) + c * 8 : 0;
      pyx[u] = exists ? (int)(((py_ * g.gs) << 16) | (px_ * g.gs)) : 0x40004000;   // every tap of a row that does not exist falls outside the image
      px_ += 8;
      while (px_ >= (unsigned)g.Wm) { px_ -= (unsigned)g.Wm; ++py_; }
      while (py_ >= (unsigned)g.Hm) { py_ -= (unsigned)g.Hm; ++pb; }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = (wave * 4 + v) * 8 + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((r >> 1) & 7);
      boff[v] = (int)((n0 + r) * args.K) + c * 8;
    }
  }
  unsigned char* const my_a = smem + wave * G_TILE_BYTES;
  unsigned char* const my_b = smem + 4 * G_TILE_BYTES + wave * 4096;
  int s_ch = 0, s_jx = 0, s_jy = 0, s_k = 0;
  int s_dy = g.oy0, s_dx = g.ox0;
  int s_off = (s_dy * g.Ws + s_dx) * g.C;
#define CX_ADVANCE()                                                                                                              \
  do {                                                                                                                            \
    s_k += G_BK; s_ch += G_BK;                                                                                                    \
    if (s_ch == g.C) { s_ch = 0; if (++s_jx == g.ntx) { s_jx = 0; ++s_jy; } s_dy = g.oy0 + s_jy * g.sty; s_dx = g.ox0 + s_jx * g.stx; } \
    s_off = (s_dy * g.Ws + s_dx) * g.C + s_ch;                                                                                    \
  } while (0)
#define CX_ISSUE_A(SLOT, U)                                                                                                       \
  do {                                                                                                                            \
    const unsigned sy_ = (unsigned)((pyx[U] >> 16) + s_dy), sx_ = (unsigned)((pyx[U] & 0xffff) + s_dx);                           \
    const uint16_t* p_ = (sy_ < (unsigned)g.Hs && sx_ < (unsigned)g.Ws) ? args.X + (rowoff[U] + s_off) : zero;                    \
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)p_, (LDS_AS void*)(my_a + (SLOT) * CX_SLOT + (U) * 1024), 16, 0, ENH_GLDS_AUX);     \
  } while (0)
#define CX_ISSUE_B(SLOT, V)                                                                                                       \
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)(args.Wt + (boff[V] + s_k)), (LDS_AS void*)(my_b + (SLOT) * CX_SLOT + (V) * 1024), 16, 0, ENH_GLDS_AUX)
#define CX_READ_ONE(FA, FB, SLOT, S, U)                                                                                           \
  do {                                                                                                                            \
    if ((U) < 4) FA[(U) & 3] = frag32<false>(smem + (SLOT) * CX_SLOT + wave * G_TILE_BYTES, ((U) & 3) * 32, S, lane);             \
    else FB[(U) & 3] = frag32<false>(smem + (SLOT) * CX_SLOT + 4 * G_TILE_BYTES, ((U) & 3) * 32, S, lane);                        \
  } while (0)
#define CX_MM(Q, FA, FB)                                                                                                          \
  acc[(Q) >> 2][(Q) & 3] = mfma32<OT>(FB[(Q) & 3], FA[(Q) >> 2], acc[(Q) >> 2][(Q) & 3])
#define CX_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one k16 step: 16 MFMAs; under the first 8 one fragment read each; HALF = 0 / 1: requests of the first / second half of a stage (A pieces 8*HALF..+7 under
  // the odd MFMAs, weight pieces 2*HALF, +1 under MFMAs 4 and 10)
#define CX_KSTEP(FA, FB, RA, RB, RSLOT, RS, DO_READ, GSLOT, HALF, DO_ISSUE)                                                       \
  do {                                                                                                                            \
    CX_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      CX_MM(q_, FA, FB);                                                                                                          \
      if ((DO_READ) && q_ < 8) { CX_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                            \
      if ((DO_ISSUE) && (q_ & 1)) { CX_ISSUE_A(GSLOT, (HALF) * 8 + (q_ >> 1)); }                                                  \
      if ((DO_ISSUE) && (q_ == 4 || q_ == 10)) { CX_ISSUE_B(GSLOT, (HALF) * 2 + (q_ == 10)); }                                    \
      CX_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];

  // prologue: stage 0 -> slot 0 completely; the first half of stage 1 -> slot 1
#pragma unroll
  for (int u = 0; u < 16; ++u) CX_ISSUE_A(0, u);
#pragma unroll
  for (int v = 0; v < 4; ++v) CX_ISSUE_B(0, v);
  CX_ADVANCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) CX_ISSUE_A(1, u);
  CX_ISSUE_B(1, 0); CX_ISSUE_B(1, 1);
  __builtin_amdgcn_s_waitcnt(0x0F7A);   // vmcnt(10): stage 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 8; ++u) CX_READ_ONE(fa0, fb0, 0, 0, u);
  CX_FENCE();

  // invariant at the top of iteration j: the tap state is at stage j+1, whose first half is already requested into slot (j+1)&1
  int j = 0;
  for (; j + 2 < nst; ++j) {
    const int slot = j & 1;
    CX_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 1, true);      // + second half of stage j+1
    CX_ADVANCE();
    CX_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    CX_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): stage j+1 landed (nothing newer outstanding) ; lgkmcnt(0): this slot is read out
    __builtin_amdgcn_s_barrier();
    CX_FENCE();
    CX_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, slot, 0, true);      // + first half of stage j+2 into the slot just vacated
  }
  {  // tail: stages nst-2 and nst-1
    const int slot = j & 1;
    CX_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 1, true);      // + second half of stage nst-1
    CX_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    CX_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    CX_FENCE();
    CX_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, 0, 0, false);
    CX_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 1, true, 0, 0, false);
    CX_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 2, true, 0, 0, false);
    CX_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 3, true, 0, 0, false);
    CX_KSTEP(fa1, fb1, fa0, fb0, 0, 0, false, 0, 0, false);
  }
#undef CX_ADVANCE
#undef CX_ISSUE_A
#undef CX_ISSUE_B
#undef CX_READ_ONE
#undef CX_MM
#undef CX_KSTEP
#undef CX_FENCE
  conv_epilogue32<OT, 4>(args, acc, m0 + wave * 128, n0, lane, reinterpret_cast<float*>(smem + 32768) + wave * 128, smem + wave * 8192);
}

static int conv_geom_check(const enh_conv_geom* g, const char* who) {
  ENH_REQUIRE(g, ENH_E_BADARG, "%s: geometry is NULL", who);
  ENH_REQUIRE(g->B > 0 && g->Hs > 0 && g->Ws > 0 && g->Hm > 0 && g->Wm > 0 && g->HO > 0 && g->WO > 0, ENH_E_BADARG, "%s: non-positive size", who);
  ENH_REQUIRE(g->C > 0 && g->N > 0 && g->C % 8 == 0 && g->N % 8 == 0, ENH_E_SHAPE, "%s: C and N must be multiples of 8 (C=%d N=%d)", who, g->C, g->N);
  ENH_REQUIRE(g->nty >= 0 && g->ntx >= 0 && g->nty <= 16 && g->ntx <= 16 && g->gs >= 1 && g->os >= 1, ENH_E_SHAPE, "%s: bad tap grid / stride", who);
  ENH_REQUIRE((int64_t)(g->Hm - 1) * g->os + g->oph < g->HO && (int64_t)(g->Wm - 1) * g->os + g->opw < g->WO && g->oph >= 0 && g->opw >= 0, ENH_E_SHAPE,
              "%s: the logical grid %dx%d (stride %d, offset %d,%d) does not fit the output %dx%d", who, g->Hm, g->Wm, g->os, g->oph, g->opw, g->HO, g->WO);
  return ENH_OK;
}

// 0 = per-shape choice, 1 = register-staged kernel everywhere, 2 = no 256-row kernel (the round-2 choice), 3 = the 256-row kernel wherever the shape allows
// it, however few tiles (A/B measurements and tests; explicit state like enh_gemm_set_kernel)
static int g_conv_variant = 0;
extern "C" int enh_conv_set_kernel(int variant) {
  ENH_REQUIRE(variant >= 0 && variant <= 3, ENH_E_BADARG, "enh_conv_set_kernel: variant must be 0 (auto), 1 (register-staged), 2 (128-row kernels), 3 (256-row kernel wherever possible)");
  g_conv_variant = variant;
  return ENH_OK;
}

// the 256 / 512-row kernels: whole K stages inside one tap, whole N tiles, 32-bit element offsets, (y, x) in 14 bits each.
// 0 = not applicable, 4 = 256 x 256 tiles (N % 256 == 0), 5 = 512 x 128 tiles (other multiples of 128)
static int conv_w256_nj(const ConvArgs& a) {
  const enh_conv_geom& g = a.g;
  if (g_conv_variant == 1 || g_conv_variant == 2) return 0;
  if (g.C % G_BK != 0 || a.K < 4 * G_BK || g.N % 128 != 0) return 0;
  if (a.M >= (1ll << 31) - 1024 || (int64_t)g.B * g.Hs * g.Ws * g.C >= (1ll << 31) || (int64_t)g.N * a.K >= (1ll << 31) || g.Hs > 16000 || g.Ws > 16000) return 0;
  if ((int64_t)(g.Hm - 1) * g.gs > 16000 || (int64_t)(g.Wm - 1) * g.gs > 16000) return 0;
  const int nj = g.N % 256 == 0 ? 4 : 5;
  if (g_conv_variant == 3) return nj;
  // per-shape choice (B = 16 layer table, profiles/r04_conv_layers.txt): with dense output rows the large tiles win 1.2-1.4x (760 -> 930, 764 -> 1052 TF/s);
  // the parity classes of a stride-2 input gradient (1-4 taps: two to eight K stages per tile) do not amortise the deeper prologue (408 -> 326)
  const bool dense = g.os == 1 && g.HO == g.Hm && g.WO == g.Wm && g.oph == 0 && g.opw == 0;
  if (!dense) return 0;
  const int64_t tiles = nj == 4 ? ((a.M + 255) / 256) * (g.N / 256) : ((a.M + 511) / 512) * (g.N / 128);
  return tiles >= enh_device_cus() ? nj : 0;   // below one tile per CU the 128-row kernels (four times the workgroups, two per CU) fill the chip better
}

template <typename OT>
static void conv_lds_attr_once() {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_glds_kernel<OT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G_TILE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<OT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G_TILE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_w256_kernel<OT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_w512_kernel<OT>), hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS_BYTES);
    return true;
  }();
  (void)attr_set;
}

// Small grids (the <= 16^2 layers at 16 images: 128 tiles of a 72-stage K loop on 256 CUs) are split over the K stages so that every CU has two
// workgroups; {1, all} = not split.  Only the LDS-DMA 128 x 128 kernel has the split form.
struct ConvSplit { int splits, st_per_split; };
static ConvSplit conv_split_plan(const enh_conv_geom& g, int64_t M, int64_t K) {
  const int nst = (int)(K / G_BK);
  const ConvSplit none = {1, nst};
  if (g_conv_variant != 0 || g.C % G_BK != 0 || nst < 16 || g.N % 4 != 0) return none;   // (the A/B families run their own kernels unsplit)
  const int64_t tiles = ((M + G_BM - 1) / G_BM) * ((g.N + G_BN - 1) / G_BN);
  const int cus = enh_device_cus();
  if (tiles * 2 > cus) return none;                        // at least half a round of workgroups already
  int splits = (int)((2 * cus) / tiles);
  if (splits > 8) splits = 8;
  if (splits > nst / 4) splits = nst / 4;                   // at least four stages per slice
  if (splits < 2) return none;
  const int per = (nst + splits - 1) / splits;
  splits = (nst + per - 1) / per;
  if (nst - (splits - 1) * per < 2) return none;            // the pipelined K loop wants two stages
  return {splits, per};
}

extern "C" size_t enh_conv_workspace_bytes(const enh_conv_geom* g) {
  if (!g || g->N <= 0 || g->C <= 0) return 0;
  const int64_t M = (int64_t)g->B * g->Hm * g->Wm, K = (int64_t)g->nty * g->ntx * g->C;
  const int keep = g_conv_variant;
  g_conv_variant = 0;                                       // sized for the per-shape choice whatever family is selected now
  const ConvSplit sp = conv_split_plan(*g, M, K);
  g_conv_variant = keep;
  return sp.splits > 1 ? (size_t)sp.splits * M * g->N * sizeof(float) : 0;
}

static int conv_nhwc_impl(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                          const enh_h16* add, float p0, float p1, enh_h16* out, void* ws, size_t ws_bytes, int dtype, void* stream);

extern "C" int enh_conv_nhwc_h16(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                                  const enh_h16* add, float p0, float p1, enh_h16* out, int dtype, void* stream) {
  return conv_nhwc_impl(src, wt, g, mode, bias, aux, add, p0, p1, out, nullptr, 0, dtype, stream);
}

// the same with a caller-provided workspace (enh_conv_workspace_bytes): small grids are then split over the contraction
extern "C" int enh_conv_nhwc_h16_ws(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                                     const enh_h16* add, float p0, float p1, enh_h16* out, void* ws, size_t ws_bytes, int dtype, void* stream) {
  return conv_nhwc_impl(src, wt, g, mode, bias, aux, add, p0, p1, out, ws, ws_bytes, dtype, stream);
}

static int conv_nhwc_impl(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                          const enh_h16* add, float p0, float p1, enh_h16* out, void* ws, size_t ws_bytes, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_conv_nhwc_h16");
  ENH_REQUIRE(src && wt && out, ENH_E_BADARG, "enh_conv_nhwc_h16: bad argument");
  const int rc = conv_geom_check(g, "enh_conv_nhwc_h16");
  if (rc != ENH_OK) return rc;
  ENH_REQUIRE(mode >= 0 && mode <= 4, ENH_E_BADARG, "enh_conv_nhwc_h16: mode must be 0..4");
  ENH_REQUIRE((mode != 0 || bias) && (mode != 1 || aux) && (mode != 4 || add), ENH_E_BADARG, "enh_conv_nhwc_h16: mode 0 needs bias, mode 1 aux, mode 4 add");
  if (g_conv_variant == 0 && conv_pointwise_forward(src, wt, *g, mode, bias, p0, p1, out, dtype, (hipStream_t)stream)) return enh_check_launch("enh_conv_nhwc_h16");
  ConvArgs a;
  a.X = src; a.Wt = wt; a.g = *g;
  a.M = (int64_t)g->B * g->Hm * g->Wm; a.K = (int64_t)g->nty * g->ntx * g->C;
  a.bias = bias; a.mode = mode; a.aux = aux; a.add = add; a.p0 = p0; a.p1 = p1; a.out = out;
  a.nbm = (int)((a.M + G_BM - 1) / G_BM); a.nbn = (g->N + G_BN - 1) / G_BN;
  a.ws = nullptr; a.splits = 1; a.st_per_split = (int)(a.K / G_BK);
  ENH_REQUIRE((int64_t)a.nbm * a.nbn < (1ll << 30) && a.M < (1ll << 31), ENH_E_SHAPE, "enh_conv_nhwc_h16: grid too large (2^31 GEMM rows or more)");
  ENH_DT_DISPATCH(dtype, (conv_lds_attr_once<OT>()));
  const ConvSplit sp = ws ? conv_split_plan(*g, a.M, a.K) : ConvSplit{1, 0};
  if (sp.splits > 1 && ws_bytes >= (size_t)sp.splits * a.M * g->N * sizeof(float)) {
    a.ws = (float*)ws; a.splits = sp.splits; a.st_per_split = sp.st_per_split;
    ENH_DT_DISPATCH(dtype, (conv_igemm_glds_kernel<OT><<<dim3((unsigned)(a.nbm * a.nbn * a.splits)), 256, 4 * G_TILE_BYTES, (hipStream_t)stream>>>(a)));
    ENH_DT_DISPATCH(dtype, (conv_splitk_finish_kernel<OT><<<dim3((unsigned)((a.M * (g->N / 4) + 255) / 256)), 256, 0, (hipStream_t)stream>>>(a)));
    return enh_check_launch("enh_conv_nhwc_h16");
  }
  const int nj = conv_w256_nj(a);
  if (nj) {
    if (nj == 4) {
      a.nbm = (int)((a.M + 255) / 256); a.nbn = g->N / 256;
      ENH_DT_DISPATCH(dtype, (conv_igemm_w256_kernel<OT, 4><<<dim3((unsigned)(a.nbm * a.nbn)), 256, CW_LDS_BYTES, (hipStream_t)stream>>>(a)));
    } else {
      a.nbm = (int)((a.M + 511) / 512); a.nbn = g->N / 128;
      ENH_DT_DISPATCH(dtype, (conv_igemm_w512_kernel<OT><<<dim3((unsigned)(a.nbm * a.nbn)), 256, CX_LDS_BYTES, (hipStream_t)stream>>>(a)));
    }
  } else if (g->C % G_BK == 0 && a.K >= 2 * G_BK && g_conv_variant != 1)
    ENH_DT_DISPATCH(dtype, (conv_igemm_glds_kernel<OT><<<dim3((unsigned)(a.nbm * a.nbn)), 256, 4 * G_TILE_BYTES, (hipStream_t)stream>>>(a)));
  else
    ENH_DT_DISPATCH(dtype, (conv_igemm_kernel<OT><<<dim3((unsigned)(a.nbm * a.nbn)), 256, 4 * G_TILE_BYTES, (hipStream_t)stream>>>(a)));
  return enh_check_launch("enh_conv_nhwc_h16");
}

// the LPIPS entry point: 3x3, stride 1, padding 1 on the general kernel
extern "C" int enh_conv3x3_nhwc_h16(const enh_h16* x, const enh_h16* wt, int B, int H, int W, int Cin, int Cout, const float* bias, int mode,
                                     const enh_h16* aux, const enh_h16* add, enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE(mode >= 0 && mode <= 2, ENH_E_BADARG, "enh_conv3x3_nhwc_h16: mode must be 0, 1 or 2");
  enh_conv_geom g;
  g.B = B; g.Hs = H; g.Ws = W; g.C = Cin; g.Hm = H; g.Wm = W; g.gs = 1; g.oy0 = -1; g.ox0 = -1; g.nty = 3; g.ntx = 3; g.sty = 1; g.stx = 1;
  g.N = Cout; g.HO = H; g.WO = W; g.os = 1; g.oph = 0; g.opw = 0;
  return enh_conv_nhwc_h16(x, wt, &g, mode, bias, aux, add, 0.f, 1.f, out, dtype, stream);
}

// =================================================================================================
// weight gradient: dW[co][(tap, ci)] = sum over pixels (b, y, x) of dy[b, y, x, co] * src[b, y*gs + oy0 + jy*sty, x*gs + ox0 + jx*stx, ci]
// =================================================================================================
struct ConvWgradArgs {
  const uint16_t* X; const uint16_t* DY;
  enh_conv_geom g;      // g.N = channels of dy ; (Hm, Wm) = dy's spatial size ; output addressing fields unused
  GemmArgs e;           // epilogue description: M = g.N, N = taps*C, ws / c_f32 / ldc / accumulate / splits / k_per_split
};

template <typename OT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_igemm_kernel(const ConvWgradArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile (dy, kmaj) | B tile (gathered src, kmaj)]
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;
  const int nwg = args.e.nbm * args.e.nbn;
  const int lin = xcd_contiguous((int)blockIdx.x, nwg * args.e.splits);   // all tiles of one pixel slice on one XCD: they re-read the same pixels
  const int split = lin / nwg, bid = lin - split * nwg;
  const int tile_m = bid % args.e.nbm, tile_n = bid / args.e.nbm;
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int64_t k_begin = (int64_t)split * args.e.k_per_split;
  int64_t k_end = k_begin + args.e.k_per_split;
  if (k_end > args.e.K) k_end = args.e.K;
  const int nk = k_end > k_begin ? (int)((k_end - k_begin + G_BK - 1) / G_BK) : 0;

  // B operand: this thread gathers columns n0 + c*8 .. +7 (one tap, 8 channels) of pixel rows r0 + 16*i of every K step
  const int c = t & 15, r0 = t >> 4;
  const int64_t ncol = n0 + c * 8;
  const bool n_ok = ncol < args.e.N;
  const int tap = n_ok ? (int)(ncol / g.C) : 0, ch = n_ok ? (int)(ncol - (int64_t)tap * g.C) : 0;
  const int jy = tap / (g.ntx > 0 ? g.ntx : 1), jx = tap - jy * g.ntx;
  const int dy_ = g.oy0 + jy * g.sty, dx_ = g.ox0 + jx * g.stx;
  // pixel state of the 4 rows; all advance by 64 pixels per K step
  const int64_t hw = (int64_t)g.Hm * g.Wm;
  const int step_b = (int)(G_BK / hw), step_rem = (int)(G_BK - step_b * hw);
  const int step_y = step_rem / g.Wm, step_x = step_rem - step_y * g.Wm;
  int pb[4], py[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t pix = k_begin + r0 + 16 * i;
    const int64_t b = pix / hw, rem = pix - b * hw;
    pb[i] = (int)b; py[i] = (int)(rem / g.Wm); px[i] = (int)(rem - (int64_t)py[i] * g.Wm);
  }
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 ra[4], rb[4];   // one register set (a second one, as conv_igemm_kernel has for its gathered operand, spills here: 12 pixel-state registers)
  auto gather = [&](int64_t k0) {
    tile_gload<true>(ra, args.DY, g.N, m0, g.N, k0, k_end, t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t pix = k0 + r0 + 16 * i;
      const int sy = py[i] * g.gs + dy_, sx = px[i] * g.gs + dx_;
      const bool ok = n_ok && pix < k_end && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      rb[i] = ok ? *reinterpret_cast<const u32x4*>(args.X + (((int64_t)pb[i] * g.Hs + sy) * g.Ws + sx) * g.C + ch) : zero4;
      // advance this row to the next K step
      px[i] += step_x;
      const int cx = px[i] >= g.Wm;
      px[i] -= cx ? g.Wm : 0;
      py[i] += step_y + cx;
      const int cy = py[i] >= g.Hm;
      py[i] -= cy ? g.Hm : 0;
      pb[i] += step_b + cy;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    gather(k_begin);
    tile_sstore<true>(ra, smem, t);
    tile_sstore<true>(rb, smem + G_TILE_BYTES, t);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) gather(k_begin + (int64_t)(kt + 1) * G_BK);
    const unsigned char* sa = smem + stage * (2 * G_TILE_BYTES);
    const unsigned char* sb = sa + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = tile_frag<true>(sa, wm * 64 + i * 16, ks, lg, l16);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = tile_frag<true>(sb, wn * 64 + j * 16, ks, lg, l16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16<OT>(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      unsigned char* na = smem + (stage ^ 1) * (2 * G_TILE_BYTES);
      tile_sstore<true>(ra, na, t);
      tile_sstore<true>(rb, na + G_TILE_BYTES, t);
    }
    __syncthreads();
  }
  gemm_epilogue<OT>(args.e, acc, m0, n0, wm, wn, lg, l16, split);
}

// =================================================================================================
// "w256" form of the weight gradient (round 4): the dense split-K weight-gradient kernel (gemm_bf16_w256_kernel<true, true, EPI_WS>: both operands
// contraction-major in the "kmaj2" LDS image, transpose reads, 256 x 256 tile, one wave per SIMD) with the gathered operand fetched by global_load_lds.
// The contraction runs over pixels; a 64-pixel K stage of a grid whose width is a multiple of 64 lies inside ONE image row, and a 128-column sub-tile of
// (tap, channel) with C % 128 == 0 inside ONE tap — so the wave that stages a B sub-tile keeps (image, row, first column, validity of the source row, base
// pointer) in SCALAR registers, and a lane keeps two 32-bit offsets: per 1-KiB piece the pointer is "scalar base + lane offset", the horizontal padding
// a single unsigned compare, lanes outside the image read the zero page.  Waves 0, 1 stage dy (a plain matrix) through the same instruction stream.
// Split over the pixel axis; partial slabs to the workspace, added in a fixed order by splitk_reduce_kernel (deterministic).
// =================================================================================================
template <typename OT, bool BOUNDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wgrad_w256_kernel(const ConvWgradArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 slots][A0 | A1 | B0 | B1]
  const enh_conv_geom& g = args.g;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = args.e.nbm * args.e.nbn;
  const int lin = xcd_contiguous((int)blockIdx.x, nwg * args.e.splits);   // all tiles of one pixel slice on one XCD: they re-read the same pixels
  const int split = lin / nwg, bid = lin - split * nwg;
  const int tile_m = bid % args.e.nbm, tile_n = bid / args.e.nbm;
  const int64_t m0 = (int64_t)tile_m * 256, n0 = (int64_t)tile_n * 256;
  const int64_t k_begin = (int64_t)split * args.e.k_per_split;
  int64_t k_end = k_begin + args.e.k_per_split;
  if (k_end > args.e.K) k_end = args.e.K;
  const int nst = (int)((k_end - k_begin) / G_BK);   // >= 2 (launcher)

  // ---- staging state: pointer of piece u = s_base + (u odd ? voff_o : voff_e) + (u >> 1) * step2, valid iff s_ok && vx0 + u * vxstep + s_x < lim ----
  const bool stage_a = wave < 2;                                   // wave-uniform
  const int l4 = lane >> 4, pp = lane & 15;
  const int col_e = (((pp >> 1) ^ (l4 << 1)) << 4) + (pp & 1) * 8;        // kmaj2 image, k-row 4u + l4: chunk q ^ (2 * (k & 3) | (k >> 2) & 1)
  const int col_o = (((pp >> 1) ^ ((l4 << 1) | 1)) << 4) + (pp & 1) * 8;
  const uint16_t* const zero = reinterpret_cast<const uint16_t*>(g_conv_zero_page);
  int voff_e, voff_o, step2;
  unsigned vx0, vxstep, lim, s_x;
  bool s_ok;
  const uint16_t* s_base;
  int s_b = 0, s_y = 0, s_x0 = 0, dy_ = 0, dx_ = 0, ch0 = 0;
  bool sub_ok = true;
  if (stage_a) {
    const int ld = g.N;
    voff_e = l4 * ld + col_e; voff_o = (4 + l4) * ld + col_o; step2 = 8 * ld;
    vx0 = 0u; vxstep = 0u; lim = 1u; s_x = 0u; s_ok = true;
    s_base = args.DY + k_begin * ld + m0 + wave * 128;
  } else {
    const int64_t nb = n0 + (wave - 2) * 128;
    sub_ok = nb < args.e.N;
    const int tap = sub_ok ? (int)(nb / g.C) : 0;
    ch0 = sub_ok ? (int)(nb - (int64_t)tap * g.C) : 0;
    const int jy = tap / g.ntx, jx = tap - jy * g.ntx;
    dy_ = g.oy0 + jy * g.sty; dx_ = g.ox0 + jx * g.stx;
    const int pix = g.gs * g.C;
    voff_e = l4 * pix + col_e; voff_o = (4 + l4) * pix + col_o; step2 = 8 * pix;
    vx0 = (unsigned)(l4 * g.gs); vxstep = (unsigned)(4 * g.gs); lim = (unsigned)g.Ws;
    const int64_t hw = (int64_t)g.Hm * g.Wm;
    s_b = (int)(k_begin / hw);
    const int rem = (int)(k_begin - (int64_t)s_b * hw);
    s_y = rem / g.Wm; s_x0 = rem - s_y * g.Wm;
    const int sy = s_y * g.gs + dy_;
    s_ok = sub_ok && (unsigned)sy < (unsigned)g.Hs;
    s_x = (unsigned)(s_x0 * g.gs + dx_);
    s_base = args.X + (((int64_t)s_b * g.Hs + sy) * g.Ws + (int64_t)s_x0 * g.gs + dx_) * g.C + ch0;
  }
  unsigned char* const my_sub = smem + wave * G_TILE_BYTES;
#define WW_ADVANCE()                                                                                                              \
  do {                                                                                                                            \
    if (stage_a) s_base += (int64_t)G_BK * g.N;                                                                                   \
    else {                                                                                                                        \
      s_x0 += G_BK;                                                                                                               \
      if (s_x0 == g.Wm) { s_x0 = 0; if (++s_y == g.Hm) { s_y = 0; ++s_b; } }                                                      \
      const int sy_ = s_y * g.gs + dy_;                                                                                           \
      s_ok = sub_ok && (unsigned)sy_ < (unsigned)g.Hs;                                                                            \
      s_x = (unsigned)(s_x0 * g.gs + dx_);                                                                                        \
      s_base = args.X + (((int64_t)s_b * g.Hs + sy_) * g.Ws + (int64_t)s_x0 * g.gs + dx_) * g.C + ch0;                            \
    }                                                                                                                             \
  } while (0)
#define WW_ISSUE_ONE(SLOT, U)                                                                                                     \
  do {                                                                                                                            \
    const unsigned vx_ = vx0 + (unsigned)(U) * vxstep + s_x;                                                                      \
    const uint16_t* p_ = (s_ok && vx_ < lim) ? s_base + ((((U) & 1) ? voff_o : voff_e) + ((U) >> 1) * step2) : zero;              \
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)p_, (LDS_AS void*)(my_sub + (SLOT) * CW_SLOT + (U) * 1024), 16, 0, ENH_GLDS_AUX);    \
  } while (0)
#define WW_READ_ONE(FA, FB, SLOT, S, U)                                                                                           \
  do {                                                                                                                            \
    if ((U) < 4) FA[(U) & 3] = frag32<true>(smem + (SLOT) * CW_SLOT + wm * G_TILE_BYTES, ((U) & 3) * 32, S, lane);                \
    else FB[(U) & 3] = frag32<true>(smem + (SLOT) * CW_SLOT + (2 + wn) * G_TILE_BYTES, ((U) & 3) * 32, S, lane);                  \
  } while (0)
#define WW_MM(Q, FA, FB)                                                                                                          \
  acc[(Q) >> 2][(Q) & 3] = mfma32<OT>(FB[(Q) & 3], FA[(Q) >> 2], acc[(Q) >> 2][(Q) & 3])
#define WW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WW_KSTEP(FA, FB, RA, RB, RSLOT, RS, DO_READ, GSLOT, G0, DO_ISSUE)                                                         \
  do {                                                                                                                            \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* the asm transpose reads are invisible to the compiler's wait-count pass */               \
    WW_FENCE();                                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                                           \
      WW_MM(q_, FA, FB);                                                                                                          \
      if ((DO_READ) && q_ < 8) { WW_READ_ONE(RA, RB, RSLOT, RS, q_); }                                                            \
      if ((DO_ISSUE) && (q_ & 1)) { WW_ISSUE_ONE(GSLOT, (G0) + (q_ >> 1)); }                                                      \
      WW_FENCE();                                                                                                                 \
    }                                                                                                                             \
  } while (0)

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  s16x8 fa0[4], fb0[4], fa1[4], fb1[4];

#pragma unroll
  for (int u = 0; u < 16; ++u) WW_ISSUE_ONE(0, u);
  WW_ADVANCE();
#pragma unroll
  for (int u = 0; u < 8; ++u) WW_ISSUE_ONE(1, u);
  __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): stage 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int u = 0; u < 8; ++u) WW_READ_ONE(fa0, fb0, 0, 0, u);
  WW_FENCE();

  // invariant at the top of iteration j: the staging state is at stage j+1, whose pieces 0-7 are already requested into slot (j+1)&1
  int j = 0;
  for (; j + 2 < nst; ++j) {
    const int slot = j & 1;
    WW_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage j+1
    WW_ADVANCE();
    WW_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    WW_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);    // vmcnt(0): stage j+1 landed ; lgkmcnt(0): this slot is read out
    __builtin_amdgcn_s_barrier();
    WW_FENCE();
    WW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, slot, 0, true);      // + pieces 0-7 of stage j+2 into the slot just vacated
  }
  {  // tail: stages nst-2 and nst-1
    const int slot = j & 1;
    WW_KSTEP(fa0, fb0, fa1, fb1, slot, 1, true, slot ^ 1, 8, true);      // + pieces 8-15 of stage nst-1
    WW_KSTEP(fa1, fb1, fa0, fb0, slot, 2, true, 0, 0, false);
    WW_KSTEP(fa0, fb0, fa1, fb1, slot, 3, true, 0, 0, false);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    WW_FENCE();
    WW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 0, true, 0, 0, false);
    WW_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 1, true, 0, 0, false);
    WW_KSTEP(fa1, fb1, fa0, fb0, slot ^ 1, 2, true, 0, 0, false);
    WW_KSTEP(fa0, fb0, fa1, fb1, slot ^ 1, 3, true, 0, 0, false);
    WW_KSTEP(fa1, fb1, fa0, fb0, 0, 0, false, 0, 0, false);
  }
#undef WW_ADVANCE
#undef WW_ISSUE_ONE
#undef WW_READ_ONE
#undef WW_MM
#undef WW_KSTEP
#undef WW_FENCE
  gemm_epilogue32_loops<EPI_WS, 4, BOUNDS, OT>(args.e, acc, m0 + wm * 128, n0 + wn * 128, lane, split, reinterpret_cast<float*>(smem + 2 * CW_SLOT) + wave * 128,
                                           smem + wave * 8192, smem + wave * 16384);
}

struct WgradPlan { int splits; int64_t k_per_split; };
static WgradPlan conv_wgrad_plan(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
  const int64_t ksteps = (K + G_BK - 1) / G_BK;
  int64_t splits = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU in total
  if (splits > ksteps / 4) splits = ksteps / 4;         // at least 4 K steps per slice
  if (splits < 1) splits = 1;
  const int64_t per = ((ksteps + splits - 1) / splits) * G_BK;
  splits = (K + per - 1) / per;
  return {(int)splits, per};
}

// the 256-row kernel's plan: one workgroup per CU, the pixel axis cut so that tiles x slices just fills the CU budget; {0, 0} = not applicable
static WgradPlan conv_wgrad_w256_plan(const enh_conv_geom& g, int64_t M, int64_t N, int64_t K, int cus) {
  const WgradPlan none = {0, 0};
  if (g_conv_variant == 1 || g_conv_variant == 2) return none;
  if (M % 256 != 0 || g.C % 128 != 0 || g.Wm % G_BK != 0 || K % G_BK != 0 || g.ntx <= 0 || g.nty <= 0) return none;
  if ((int64_t)g.B * g.Hs * g.Ws * g.C >= (1ll << 31) || K >= (1ll << 31) || (int64_t)g.gs * g.C * 64 >= (1 << 30) || (int64_t)g.N * 64 >= (1 << 30)) return none;
  const int64_t tiles = (M / 256) * ((N + 255) / 256), stages = K / G_BK;
  int64_t splits = cus / tiles;
  if (splits > stages / 8) splits = stages / 8;         // at least 8 K stages per slice
  if (splits < 2) return none;                          // (the epilogue instantiated for this kernel is the workspace one)
  const int64_t per = ((stages + splits - 1) / splits) * G_BK;
  splits = (K + per - 1) / per;
  if (splits < 2 || K - (splits - 1) * per < 2 * G_BK) return none;
  return {(int)splits, per};
}

static void conv_wgrad_dims(const enh_conv_geom* g, int64_t& M, int64_t& N, int64_t& K) {
  M = g->N; N = (int64_t)g->nty * g->ntx * g->C; K = (int64_t)g->B * g->Hm * g->Wm;
}

// sized for whichever kernel family and CU budget is selected LATER as well (enh_conv_set_kernel / enh_set_cu_budget may change between the query and the call)
extern "C" size_t enh_conv_wgrad_workspace_bytes(const enh_conv_geom* g) {
  if (!g || g->N <= 0 || g->C <= 0) return 0;
  int64_t M, N, K;
  conv_wgrad_dims(g, M, N, K);
  const WgradPlan p = conv_wgrad_plan(M, N, K);
  const int keep = g_conv_variant;
  g_conv_variant = 0;
  const WgradPlan q = conv_wgrad_w256_plan(*g, M, N, K, enh_device_cus());
  g_conv_variant = keep;
  int splits = p.splits > q.splits ? p.splits : q.splits;
  const int pw = conv_pointwise_wgrad_slabs(*g);
  if (pw > splits) splits = pw;
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

template <typename OT>
static void conv_wgrad_attr_once() {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_igemm_kernel<OT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G_TILE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_w256_kernel<OT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_w256_kernel<OT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS_BYTES);
    return true;
  }();
  (void)attr_set;
}

extern "C" int enh_conv_wgrad_nhwc_h16(const enh_h16* src, const enh_h16* dy, const enh_conv_geom* g, float* dw, void* ws, size_t ws_bytes, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_conv_wgrad_nhwc_h16");
  ENH_REQUIRE(src && dy && dw, ENH_E_BADARG, "enh_conv_wgrad_nhwc_h16: bad argument");
  ENH_REQUIRE(g, ENH_E_BADARG, "enh_conv_wgrad_nhwc_h16: geometry is NULL");
  enh_conv_geom gg = *g; gg.HO = g->Hm; gg.WO = g->Wm; gg.os = 1; gg.oph = 0; gg.opw = 0;   // the output-addressing fields are not used by this role
  const int rc = conv_geom_check(&gg, "enh_conv_wgrad_nhwc_h16");
  if (rc != ENH_OK) return rc;
  ENH_REQUIRE(g->nty > 0 && g->ntx > 0, ENH_E_SHAPE, "enh_conv_wgrad_nhwc_h16: empty tap grid");
  ENH_REQUIRE((int64_t)g->B * g->Hm * g->Wm < (1ll << 31) && (int64_t)g->B * g->Hs * g->Ws < (1ll << 31), ENH_E_SHAPE, "enh_conv_wgrad_nhwc_h16: more than 2^31 pixels");
  const int pw_slabs = g_conv_variant == 0 ? conv_pointwise_wgrad_slabs(gg) : 0;
  if (pw_slabs > 0) {
    const int64_t MN = (int64_t)g->N * 8;
    ENH_REQUIRE(pw_slabs == 1 || (ws && ws_bytes >= (size_t)pw_slabs * MN * sizeof(float)), ENH_E_WORKSPACE, "enh_conv_wgrad_nhwc_h16: workspace too small (%zu < %zu bytes)",
                ws_bytes, (size_t)pw_slabs * MN * sizeof(float));
    conv_pointwise_wgrad(src, dy, gg, (float*)ws, dw, dtype, (hipStream_t)stream);
    return enh_check_launch("enh_conv_wgrad_nhwc_h16");
  }
  ConvWgradArgs a;
  a.X = src; a.DY = dy; a.g = gg;
  GemmArgs& e = a.e;
  e.A = nullptr; e.lda = 0; e.B = nullptr; e.ldb = 0;
  conv_wgrad_dims(g, e.M, e.N, e.K);
  const WgradPlan big = conv_wgrad_w256_plan(gg, e.M, e.N, e.K, enh_cu_budget());
  const WgradPlan p = big.splits ? big : conv_wgrad_plan(e.M, e.N, e.K);
  e.k_per_split = p.k_per_split; e.splits = p.splits;
  e.bias = nullptr; e.act = ENH_ACT_NONE; e.aux = nullptr; e.ldaux = 0; e.res = nullptr; e.ldres = 0; e.res_rows = 0;
  e.c_bf16 = nullptr; e.c_f32 = dw; e.ldc = e.N; e.ws = nullptr; e.accumulate = 0;
  const int64_t MN = e.M * e.N;
  if (p.splits > 1) {
    ENH_REQUIRE(ws && ws_bytes >= (size_t)p.splits * MN * sizeof(float), ENH_E_WORKSPACE, "enh_conv_wgrad_nhwc_h16: workspace too small (%zu < %zu bytes)",
                ws_bytes, (size_t)p.splits * MN * sizeof(float));
    e.ws = (float*)ws; e.accumulate = 3;
  }
  ENH_DT_DISPATCH(dtype, (conv_wgrad_attr_once<OT>()));
  hipStream_t s = (hipStream_t)stream;
  if (big.splits) {
    e.nbm = (int)(e.M / 256); e.nbn = (int)((e.N + 255) / 256);
    const dim3 grid((unsigned)(e.nbm * e.nbn * p.splits));
    if (e.N % 256 == 0) ENH_DT_DISPATCH(dtype, (conv_wgrad_w256_kernel<OT, false><<<grid, 256, CW_LDS_BYTES, s>>>(a)));
    else ENH_DT_DISPATCH(dtype, (conv_wgrad_w256_kernel<OT, true><<<grid, 256, CW_LDS_BYTES, s>>>(a)));
  } else {
    e.nbm = (int)((e.M + G_BM - 1) / G_BM); e.nbn = (int)((e.N + G_BN - 1) / G_BN);
    ENH_DT_DISPATCH(dtype, (conv_wgrad_igemm_kernel<OT><<<dim3((unsigned)(e.nbm * e.nbn * p.splits)), 256, 4 * G_TILE_BYTES, s>>>(a)));
  }
  if (p.splits > 1) splitk_reduce_kernel<<<dim3((unsigned)((MN / 4 + 255) / 256)), 256, 0, s>>>(e.ws, p.splits, MN, e.N, dw, e.N, 0);
  return enh_check_launch("enh_conv_wgrad_nhwc_h16");
}

// =================================================================================================
// weight packing: parameter layout [Cout][Cin][k][k] f32  <->  the [rows][taps * cols] matrices the kernels above read / write
// =================================================================================================
// transposed = 0: out[co][(jy*ntx + jx)*Cp + ci] = scale * w[co][ci][kh0 + jy*kstep][kw0 + jx*kstep]   rows co < Rp (Rp >= Cout), ci < Cp (Cp >= Cin)
// transposed = 1: out[ci][(jy*ntx + jx)*Cp + co] = scale * w[co][ci][kh0 + jy*kstep][kw0 + jx*kstep]   rows ci < Rp (Rp >= Cin),  co < Cp (Cp >= Cout)
// rows / columns beyond the real channel counts are zero
template <typename OT>
__global__ void conv_pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int k, float scale, int transposed, int kh0, int kw0, int kstep,
                                        int nty, int ntx, int Rp, int Cp, uint16_t* __restrict__ out) {
  const int64_t total = (int64_t)Rp * nty * ntx * Cp;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int col = (int)(i % Cp);
  const int64_t q = i / Cp;
  const int tap = (int)(q % (nty * ntx)), row = (int)(q / (nty * ntx));
  const int jy = tap / ntx, jx = tap - jy * ntx;
  const int co = transposed ? col : row, ci = transposed ? row : col;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = scale * w[(((int64_t)co * Cin + ci) * k + kh0 + jy * kstep) * k + kw0 + jx * kstep];
  out[i] = pack1<OT>(v);
}

extern "C" int enh_conv_pack_weight(const float* w, int Cout, int Cin, int k, float scale, int transposed, int kh0, int kw0, int kstep, int nty, int ntx,
                                    int rows_padded, int cols_padded, enh_h16* out, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_conv_pack_weight");
  ENH_REQUIRE(w && out && Cout > 0 && Cin > 0 && k > 0 && kstep > 0, ENH_E_BADARG, "enh_conv_pack_weight: bad argument");
  ENH_REQUIRE(nty >= 0 && ntx >= 0 && kh0 >= 0 && kw0 >= 0 && (nty == 0 || kh0 + (nty - 1) * kstep < k) && (ntx == 0 || kw0 + (ntx - 1) * kstep < k), ENH_E_SHAPE,
              "enh_conv_pack_weight: tap selection outside the %dx%d kernel", k, k);
  ENH_REQUIRE(rows_padded >= (transposed ? Cin : Cout) && cols_padded >= (transposed ? Cout : Cin), ENH_E_SHAPE, "enh_conv_pack_weight: padded sizes too small");
  const int64_t total = (int64_t)rows_padded * nty * ntx * cols_padded;
  if (total == 0) return ENH_OK;
  ENH_DT_DISPATCH(dtype, (conv_pack_weight_kernel<OT><<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(w, Cout, Cin, k, scale, transposed, kh0, kw0, kstep, nty, ntx,
                                                                                                  rows_padded, cols_padded, out)));
  return enh_check_launch("enh_conv_pack_weight");
}

// dw[co][ci][kh][kw] = scale * dwp[co][(kh*k + kw)*Cp + ci]
__global__ void conv_unpack_wgrad_kernel(const float* __restrict__ dwp, int Cout, int Cin, int Cp, int k, float scale, float* __restrict__ dw) {
  const int64_t total = (int64_t)Cout * Cin * k * k;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int kw = (int)(i % k);
  int64_t q = i / k;
  const int kh = (int)(q % k); q /= k;
  const int ci = (int)(q % Cin), co = (int)(q / Cin);
  dw[i] = scale * dwp[((int64_t)co * k * k + kh * k + kw) * Cp + ci];
}

extern "C" int enh_conv_unpack_wgrad(const float* dwp, int Cout, int Cin, int cin_padded, int k, float scale, float* dw, void* stream) {
  ENH_REQUIRE(dwp && dw && Cout > 0 && Cin > 0 && k > 0 && cin_padded >= Cin, ENH_E_BADARG, "enh_conv_unpack_wgrad: bad argument");
  const int64_t total = (int64_t)Cout * Cin * k * k;
  conv_unpack_wgrad_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(dwp, Cout, Cin, cin_padded, k, scale, dw);
  return enh_check_launch("enh_conv_unpack_wgrad");
}
