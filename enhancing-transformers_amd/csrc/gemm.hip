// gemm.hip — bf16 MFMA GEMM with fused epilogues for gfx950 (CDNA4), wave64.
//
// One kernel family serves every dense contraction of the ViT towers (reference
// enhancing/modules/stage1/layers.py:99-101,118,120,169,204 and vitvqgan.py:38-39) in all three roles:
//   forward  y  = x  W^T        : A [M][K] row-major,            B = W  [N][K]            (trans_a=0, trans_b=0)
//   dgrad    dx = dy W          : A = dy [M][N_out] row-major,   B = W  stored [K=N_out][N=K_in] (trans_b=1)
//   wgrad    dW = dy^T x        : A = dy stored [K=tokens][M=N_out] (trans_a=1), B = x stored [K=tokens][N] (trans_b=1)
// so no transposed copies of activations or weights are ever written to HBM: operands whose contraction
// index is the slow storage index are staged as-is and read from LDS with the hardware transpose read
// ds_read_b64_tr_b16 (semantics verified on MI355X, profiles/hw_probe_r01.txt).
//
// Tiling: 128x128x64 workgroup tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16 (f32 accumulate).  Global -> register -> LDS staging, double-buffered, one
// barrier per K-step (the next tile's global loads are in flight during the MFMAs).  LDS images are
// XOR-swizzled so that ds_write_b128, ds_read_b128 and the transpose reads are bank-conflict free under the
// gfx950 bank model (MI355X_MICROARCH.md §LDS; checked by tools/lds_bank_check.py).
// The MFMA is issued with swapped operands (D = B_frag x A_frag) so each lane ends up with 4 CONSECUTIVE
// output columns of one row: the epilogue reads bias / residual / aux and writes C with 16-byte (f32) or
// 8-byte (bf16) accesses.  Workgroup ids are remapped so each XCD (private L2) walks a contiguous run of
// tiles, n-fastest, sharing the A row-panel and the weight matrix in that L2.
#include "common.h"

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_TILE_BYTES 16384  // one operand tile (either layout)

// ---- LDS layouts -------------------------------------------------------------------------------
// "row" layout  (operand stored [rows][K]):   128 rows x 128 B ; 16-B chunk c (0..7) of row r lives at
//     r*128 + ((c ^ ((r>>1)&7)) << 4)
// "kmaj" layout (operand stored [K][cols]):    64 k-rows x 256 B ; 32-B chunk q (0..7) of k-row k lives at
//     k*256 + ((q ^ ((k&3) | (((k>>3)&1)<<2))) << 5)      (8-byte pieces inside a chunk stay in order)
__device__ __forceinline__ int lds_row_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int lds_kmaj_off(int k, int q) { return k * 256 + ((q ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 5); }

struct GemmArgs {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  int64_t M, N, K;
  int64_t k_per_split;  // multiple of G_BK
  const float* bias; int act; const uint16_t* aux; int64_t ldaux;
  const float* res; int64_t ldres; int64_t res_rows;
  int accumulate;       // 1: += C_old ; 2: split-K partial -> f32 atomicAdd into c_f32
  float* c_f32; uint16_t* c_bf16; int64_t ldc;
  int nbm, nbn;
};

// global -> registers: one 128 x 64 (row layout) or 64 x 128 (kmaj layout) bf16 operand tile, 4 x 16 B per thread
template <bool TR>
__device__ __forceinline__ void tile_gload(u32x4 (&r)[4], const uint16_t* __restrict__ P, int64_t ld, int64_t x0,
                                           int64_t X, int64_t k0, int64_t k_end, int t) {
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  if (!TR) {
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = x0 + r0 + 32 * i, kk = k0 + c * 8;
      r[i] = (row < X && kk < k_end) ? *reinterpret_cast<const u32x4*>(P + row * ld + kk) : zero4;
    }
  } else {
    const int c = t & 15, r0 = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t kk = k0 + r0 + 16 * i, col = x0 + c * 8;
      r[i] = (kk < k_end && col < X) ? *reinterpret_cast<const u32x4*>(P + kk * ld + col) : zero4;
    }
  }
}
// registers -> LDS (swizzled image)
template <bool TR>
__device__ __forceinline__ void tile_sstore(const u32x4 (&r)[4], unsigned char* tile, int t) {
  if (!TR) {
    const int c = t & 7, r0 = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tile + lds_row_off(r0 + 32 * i, c)) = r[i];
  } else {
    const int c = t & 15, r0 = t >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tile + lds_kmaj_off(r0 + 16 * i, c >> 1) + ((c & 1) << 4)) = r[i];
  }
}
// MFMA 16x16x32 operand fragment: lane (lg, l16) gets tile index base + l16, k = ks*32 + lg*8 + 0..7
template <bool TR>
__device__ __forceinline__ s16x8 tile_frag(const unsigned char* tile, int base, int ks, int lg, int l16) {
  if (!TR) {
    return *reinterpret_cast<const s16x8*>(tile + lds_row_off(base + l16, ks * 4 + lg));
  } else {
    // loader role of this lane inside its 16-lane group: k-row (l16>>2), 4 columns starting at (l16&3)*4
    const int kr = ks * 32 + lg * 8 + (l16 >> 2);
    const int q = base >> 4;  // 32-byte chunk = 16 columns
    const s16x4 lo = lds_tr_read_b64(tile + lds_kmaj_off(kr, q) + (l16 & 3) * 8);
    const s16x4 hi = lds_tr_read_b64(tile + lds_kmaj_off(kr + 4, q) + (l16 & 3) * 8);
    s16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 stages][A tile | B tile]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, lg = lane >> 4;

  // ---- XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles ----
  const int nwg = args.nbm * args.nbn;  // tiles per K-split
  const int split = blockIdx.x / nwg;
  int bid = blockIdx.x - split * nwg;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  }
  const int tile_m = bid / args.nbn, tile_n = bid - tile_m * args.nbn;
  const int64_t m0 = (int64_t)tile_m * G_BM, n0 = (int64_t)tile_n * G_BN;
  const int64_t k_begin = (int64_t)split * args.k_per_split;
  int64_t k_end = k_begin + args.k_per_split;
  if (k_end > args.K) k_end = args.K;
  const int nk = (int)((k_end - k_begin + G_BK - 1) / G_BK);

  u32x4 ra[4], rb[4];

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    tile_gload<TA>(ra, args.A, args.lda, m0, args.M, k_begin, k_end, t);
    tile_gload<TB>(rb, args.B, args.ldb, n0, args.N, k_begin, k_end, t);
    tile_sstore<TA>(ra, smem, t);
    tile_sstore<TB>(rb, smem + G_TILE_BYTES, t);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) {
      const int64_t k0 = k_begin + (int64_t)(kt + 1) * G_BK;
      tile_gload<TA>(ra, args.A, args.lda, m0, args.M, k0, k_end, t);
      tile_gload<TB>(rb, args.B, args.ldb, n0, args.N, k0, k_end, t);
    }
    const unsigned char* sa = smem + stage * (2 * G_TILE_BYTES);
    const unsigned char* sb = sa + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = tile_frag<TA>(sa, wm * 64 + i * 16, ks, lg, l16);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = tile_frag<TB>(sb, wn * 64 + j * 16, ks, lg, l16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      unsigned char* na = smem + (stage ^ 1) * (2 * G_TILE_BYTES);
      tile_sstore<TA>(ra, na, t);
      tile_sstore<TB>(rb, na + G_TILE_BYTES, t);
    }
    __syncthreads();
  }

  // ---- epilogue: lane (lg, l16) holds C[m = .. + l16][n = .. + lg*4 + 0..3] ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + l16;
    if (m >= args.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= args.N) continue;  // N % 4 == 0: the 4 columns are in or out together
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      float* cp = args.c_f32 ? args.c_f32 + m * args.ldc + n : nullptr;
      if (args.accumulate == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
        continue;
      }
      if (args.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(args.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (args.act == ENH_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      } else if (args.act == ENH_ACT_DTANH) {
        const uint2 a2 = *reinterpret_cast<const uint2*>(args.aux + m * args.ldaux + n);
        const float h0 = bf16_bits_to_f32((uint16_t)(a2.x & 0xffffu)), h1 = bf16_bits_to_f32((uint16_t)(a2.x >> 16));
        const float h2 = bf16_bits_to_f32((uint16_t)(a2.y & 0xffffu)), h3 = bf16_bits_to_f32((uint16_t)(a2.y >> 16));
        v[0] *= 1.f - h0 * h0; v[1] *= 1.f - h1 * h1; v[2] *= 1.f - h2 * h2; v[3] *= 1.f - h3 * h3;
      }
      if (args.res) {
        const float4 r4 = *reinterpret_cast<const float4*>(args.res + (m % args.res_rows) * args.ldres + n);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
      if (args.accumulate == 1 && cp) {
        const float4 o4 = *reinterpret_cast<const float4*>(cp);
        v[0] += o4.x; v[1] += o4.y; v[2] += o4.z; v[3] += o4.w;
      }
      if (cp) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      if (args.c_bf16) *reinterpret_cast<uint2*>(args.c_bf16 + m * args.ldc + n) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int enh_gemm_bf16(const enh_bf16* A, int64_t lda, int trans_a, const enh_bf16* B, int64_t ldb, int trans_b,
                             int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_bf16* aux,
                             int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                             float* c_f32, enh_bf16* c_bf16, int64_t ldc, void* stream) {
  ENH_REQUIRE(A && B && (c_f32 || c_bf16), ENH_E_BADARG, "enh_gemm_bf16: null pointer");
  ENH_REQUIRE(M > 0 && N > 0 && K > 0, ENH_E_BADARG, "enh_gemm_bf16: M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  ENH_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B), ENH_E_SHAPE,
              "enh_gemm_bf16: K, lda, ldb must be multiples of 8 and A, B 16-byte aligned (K=%lld lda=%lld ldb=%lld)", (long long)K, (long long)lda, (long long)ldb);
  ENH_REQUIRE(N % 4 == 0 && ldc % 4 == 0, ENH_E_SHAPE, "enh_gemm_bf16: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", (long long)N, (long long)ldc);
  ENH_REQUIRE(!trans_a || M % 8 == 0, ENH_E_SHAPE, "enh_gemm_bf16: trans_a needs M %% 8 == 0");
  ENH_REQUIRE(!trans_b || N % 8 == 0, ENH_E_SHAPE, "enh_gemm_bf16: trans_b needs N %% 8 == 0");
  ENH_REQUIRE(act == ENH_ACT_NONE || act == ENH_ACT_TANH || (act == ENH_ACT_DTANH && aux && ldaux % 4 == 0), ENH_E_BADARG, "enh_gemm_bf16: bad act/aux");
  ENH_REQUIRE(!res || (res_rows > 0 && ldres % 4 == 0), ENH_E_BADARG, "enh_gemm_bf16: res needs res_rows > 0 and ldres %% 4 == 0");
  ENH_REQUIRE(accumulate == 0 || (accumulate == 1 && c_f32), ENH_E_BADARG, "enh_gemm_bf16: accumulate needs an f32 output");
  ENH_REQUIRE((!c_f32 || aligned16(c_f32)) && (!c_bf16 || (reinterpret_cast<uintptr_t>(c_bf16) & 7u) == 0), ENH_E_SHAPE, "enh_gemm_bf16: output alignment");

  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.act = act; g.aux = aux; g.ldaux = ldaux; g.res = res; g.ldres = ldres; g.res_rows = res_rows;
  g.accumulate = accumulate; g.c_f32 = c_f32; g.c_bf16 = c_bf16; g.ldc = ldc;
  g.nbm = (int)((M + G_BM - 1) / G_BM);
  g.nbn = (int)((N + G_BN - 1) / G_BN);
  const int64_t tiles = (int64_t)g.nbm * g.nbn;
  ENH_REQUIRE(tiles < (1ll << 30), ENH_E_SHAPE, "enh_gemm_bf16: grid too large");
  // split-K (f32 atomics into a pre-initialised C) when a weight-gradient-shaped problem cannot fill 256 CUs
  int splits = 1;
  if (accumulate == 1 && c_f32 && !c_bf16 && !bias && act == ENH_ACT_NONE && !res && tiles < 384 && K >= 2048) {
    const int64_t ksteps = (K + G_BK - 1) / G_BK;
    int64_t want = (768 + tiles - 1) / tiles;
    if (want > ksteps / 8) want = ksteps / 8;
    if (want > 64) want = 64;
    if (want >= 2) splits = (int)want;
  }
  const int64_t ksteps = (K + G_BK - 1) / G_BK;
  g.k_per_split = ((ksteps + splits - 1) / splits) * G_BK;
  splits = (int)((K + g.k_per_split - 1) / g.k_per_split);
  if (splits > 1) g.accumulate = 2;
  const dim3 grid((unsigned)(tiles * splits));
  const size_t lds = 4 * G_TILE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  static const bool attr_set = [] {
    const int bytes = 4 * G_TILE_BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return true;
  }();
  (void)attr_set;
  if (!trans_a && !trans_b) gemm_bf16_kernel<false, false><<<grid, 256, lds, s>>>(g);
  else if (!trans_a && trans_b) gemm_bf16_kernel<false, true><<<grid, 256, lds, s>>>(g);
  else if (trans_a && !trans_b) gemm_bf16_kernel<true, false><<<grid, 256, lds, s>>>(g);
  else gemm_bf16_kernel<true, true><<<grid, 256, lds, s>>>(g);
  return enh_check_launch("enh_gemm_bf16");
}
