// gemm.hip — planner and C ABI of the 16-bit-operand MFMA GEMM (bf16 or fp16 operands per call, f32 accumulation) with fused epilogues, gfx950 (CDNA4).
//
// One kernel family serves every dense contraction of the ViT towers (reference
// enhancing/modules/stage1/layers.py:99-101,118,120,169,204 and vitvqgan.py:38-39) in all three roles:
//   forward  y  = x  W^T        : A [M][K] row-major,            B = W  [N][K]            (trans_a=0, trans_b=0)
//   dgrad    dx = dy W          : A = dy [M][N_out] row-major,   B = W  stored [K=N_out][N=K_in] (trans_b=1)
//   wgrad    dW = dy^T x        : A = dy stored [K=tokens][M=N_out] (trans_a=1), B = x stored [K=tokens][N] (trans_b=1)
// The kernels live in gemm_kernels.h (templates over the operand type) and are instantiated by gemm_bf16.hip / gemm_f16.hip; this file chooses the
// family, the split-K plan and the tile schedule per shape — identically for both operand types — and owns the library state behind the enh_gemm_set_* /
// enh_debug_gemm_* calls.  Kernel families (enh_gemm_h16_variant() reports the per-shape choice; enh_gemm_set_kernel() overrides it):
//   gemm_w256_kernel   256x256x64 tile, 4 waves (one per SIMD); persistent forms gemm_w256p_kernel / gemm_w256r_kernel for forward / input-gradient roles
//   gemm_pipe2_kernel  128x128x64 tile, 4 waves, 2 workgroups per CU: shapes that are not whole 256-tiles or cannot fill the chip
//   gemm_kernel        register-staged 128x128x64 fallback for K not a multiple of 64 (zero-fills partial tiles)
#include <stdlib.h>
#include "common.h"
#include "gemm_launch.h"


static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// kernel family: 0 = register-staged fallback (any K % 8), 3 = pipe2 (128x128), 7 = w256 (256x256, 4 waves, one tile per workgroup everywhere),
// 8 = w256 with the persistent form w256p wherever that serves, 9 = as 8 with the A-in-registers form w256r where THAT serves (= the per-shape default)
static int g_kernel_override = -1;   // set by enh_gemm_set_kernel(): explicit state behind an explicit call, no environment lookups in the library

extern "C" int enh_gemm_set_kernel(int family) {
  ENH_REQUIRE(family == -1 || family == 0 || family == 3 || family == 7 || family == 8 || family == 9, ENH_E_BADARG,
              "enh_gemm_set_kernel: family must be -1 (auto), 0, 3, 7, 8 or 9");
  g_kernel_override = family;
  return ENH_OK;
}

// CU budget (common.cpp, enh_set_cu_budget): persistent grids take cu_budget() workgroups; one-round split-K plans count on whole XCD rows of it —
// workgroups go to the XCDs round-robin and so do a collective's, so with b CUs free the fullest XCD has floor(b / 8) of them (measured: a 252-workgroup
// plan against 4 held CUs overflowed XCDs 0-3 and took 2 rounds, profiles/r04_comm_contention.txt)
static int cu_budget() { return enh_cu_budget(); }
static int cu_budget_rows() { const int b = enh_cu_budget(); return b >= 8 ? (b / 8) * 8 : b; }

// tile schedule of the persistent kernels: 0 = static partition (workgroup b walks b, b + grid, ...), 1 = tiles claimed from per-XCD queues (see
// gemm_w256p_kernel).  The counters are library-owned device words (not an allocation): 64 launch slots x 8 queues, used round-robin — a slot is
// reused 64 persistent launches later, by which time the launch that used it has long retired (they run in stream order on one stream; two streams
// would have to keep 64 persistent GEMMs in flight to collide).
static int g_w256_lab = 0;   // enh_debug_gemm_lab: measurement-only forms of the split-K weight-gradient loop (wrong results)
extern "C" int enh_debug_gemm_lab(int variant) {
  ENH_REQUIRE(variant >= 0 && variant <= 9, ENH_E_BADARG, "enh_debug_gemm_lab: 0 (off) .. 9");
  g_w256_lab = variant;
  return ENH_OK;
}
static int g_dyn_schedule = 1;
// tile order of the non-split forms (gemm_tile_coords_of).  Default (grp_rows = 0): per shape — 8-row groups, rows fastest (an 8 x 4 patch of tiles in flight
// per XCD), except long-K problems (K >= 2048: fc2 forward, the qkv / fc1 input gradients), which walk row-major over the tiles (every A panel — 1-1.5 MB
// there — is fetched by one XCD once): -1 ... -4 % per launch on those, nothing elsewhere (profiles/r05_gemm_landing_lab.txt; results are bit-identical).
static int g_grp_rows = 0, g_col_fast = 0;
extern "C" int enh_debug_gemm_order(int grp_rows, int col_fast) {
  ENH_REQUIRE(grp_rows >= 0 && grp_rows <= 4096 && (col_fast == 0 || col_fast == 1), ENH_E_BADARG, "enh_debug_gemm_order: grp_rows 0 (auto) .. 4096, col_fast 0 | 1");
  g_grp_rows = grp_rows; g_col_fast = col_fast;
  return ENH_OK;
}
__device__ unsigned int g_tile_ctr[64][8];
extern "C" int enh_gemm_set_scheduler(int dynamic) {
  ENH_REQUIRE(dynamic == 0 || dynamic == 1, ENH_E_BADARG, "enh_gemm_set_scheduler: 0 (static) or 1 (dynamic)");
  g_dyn_schedule = dynamic;
  return ENH_OK;
}
static unsigned int* next_tile_counters() {
  // the counter words are a __device__ symbol: one instance per device.  Address and launch-slot sequence are kept per device (keyed by the current
  // device: a process that drives several GPUs claims tiles from the counters of the device it launches on)
  static unsigned int* base[ENH_MAX_DEVICES] = {nullptr};
  static unsigned seq[ENH_MAX_DEVICES] = {0};
  const int dev = enh_current_device();
  if (!base[dev]) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_ctr));
    base[dev] = (unsigned int*)p;
  }
  return base[dev] ? base[dev] + (size_t)((seq[dev]++) & 63u) * 8 : nullptr;
}

struct GemmPlan { int family, splits; int64_t k_per_split; };
static int g_force_splits = 0;   // enh_debug_gemm_splits: K slices of the split-K plans (0 = the planner's choice)
extern "C" int enh_debug_gemm_splits(int splits) {
  ENH_REQUIRE(splits >= 0 && splits <= 256, ENH_E_BADARG, "enh_debug_gemm_splits: 0 (auto) .. 256");
  g_force_splits = splits;
  return ENH_OK;
}

// splittable: 0 = no K slices; 1 = the weight-gradient kind (accumulate-only epilogue: may take the 256 x 256 kernel with K slices; f32 atomics when the
// caller gives no workspace); 2 = the forward kind (round 6: f32 output with optional bias / residual / accumulate, applied by the second pass) — the
// 128 x 128 family only, only with a workspace: what fills the chip at 2 - 4 images per GPU (M = 2048: the large towers' N = 1280 GEMMs are 160 tiles
// on 512 slots; K = 3840 / 5120).  At 8 images and above those calls have >= 256 tiles and are not split.
static GemmPlan gemm_plan(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int splittable) {
  const bool k64 = K % G_BK == 0 && (!trans_a || M >= 8) && (!trans_b || N >= 8);
  const int64_t ksteps = (K + G_BK - 1) / G_BK;
  GemmPlan pl = {0, 1, ksteps * G_BK};
  if (!k64) return pl;
  // split-K (weight-gradient-shaped problems that cannot fill 256 CUs): as many K slices as fit ONE round of resident workgroups, each at least
  // 8 stages long (more slices only add partial-sum traffic, a ragged second round costs more: profiles/r01_gemm_ablation.txt)
  auto split_for = [&](int64_t tiles, int64_t slots, int64_t cap) -> int {
    if (!splittable || tiles >= slots / 2 || K < 2048) return 1;
    if (g_force_splits > 0 && g_force_splits <= ksteps / 2) return g_force_splits;   // enh_debug_gemm_splits (lab)
    int64_t want = slots / tiles;
    if (want > ksteps / 8) want = ksteps / 8;
    if (want > cap) want = cap;
    if (want < 2) return 1;
    const int64_t per = (ksteps + want - 1) / want;
    return (int)((ksteps + per - 1) / per);
  };
  const bool w256_ok = M % 256 == 0 && N % 256 == 0 && ksteps >= 2;
  int family = g_kernel_override >= 0 ? g_kernel_override : -1;
  if (family == 8 || family == 9) family = 7;   // the persistent forms share w256's plan; the launcher upgrades the calls it covers
  if (family == 7 && !w256_ok) family = -1;
  if (family < 0) {
    // w256 whenever its tiles (times K slices) occupy at least 3/4 of the CUs; else the 128x128 pipe2 kernel (four times as many workgroups)
    family = 3;
    if (w256_ok) {
      const int64_t tiles = (M / 256) * (N / 256);
      const int sp = splittable == 1 ? split_for(tiles, cu_budget_rows(), 64) : 1;
      const int64_t per = (ksteps + sp - 1) / sp;
      if (tiles * sp >= (3 * cu_budget()) / 4 && ksteps - (sp - 1) * per >= 2) family = 7;
    }
  }
  pl.family = family;
  const int64_t bm = family == 7 ? 256 : 128;
  const int64_t tiles = ((M + bm - 1) / bm) * ((N + bm - 1) / bm);
  pl.splits = (family == 7 && splittable != 1) ? 1 : split_for(tiles, family == 7 ? cu_budget_rows() : 2 * cu_budget_rows(), 64);
  const int64_t per = (ksteps + pl.splits - 1) / pl.splits;
  pl.k_per_split = per * G_BK;
  pl.splits = (int)((ksteps + per - 1) / per);
  if (family == 7 && ksteps - (pl.splits - 1) * per < 2) { pl.splits = 1; pl.k_per_split = ksteps * G_BK; }   // every slice needs two stages
  return pl;
}

static int gemm_splittable(int accumulate, const float* c_f32, const enh_h16* c_bf16, const float* bias, int act, const float* res, const void* workspace) {
  if (act != ENH_ACT_NONE || (c_f32 && c_bf16)) return 0;
  if (c_bf16) return (workspace && !bias && !res && accumulate == 0) ? 2 : 0;      // plain 16-bit output (token gradients): the second pass packs
  if (accumulate == 1 && !bias && !res) return 1;
  return workspace ? 2 : 0;
}

extern "C" const char* enh_gemm_h16_variant(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K);
// the persistent form of w256 serves: A stored [M][K], no split-K, at least three K stages, one of the five epilogue modes it implements
static bool gemm_persistent(const GemmPlan& pl, int trans_a, int64_t K, int mode) {
  if (g_kernel_override == 7 || pl.family != 7 || trans_a || pl.splits != 1 || K / G_BK < 3) return false;
  return mode == EPI_BF16 || mode == EPI_BF16_BIAS_TANH || mode == EPI_BF16_DTANH || mode == EPI_F32_BIAS_RES || mode == EPI_F32;
}

// ... and, of those, the A-in-registers form: bf16 / bf16 + bias + tanh / f32 outputs, an even number (>= 6) of K stages
static bool gemm_regstaged(int64_t K, int mode) {
  const int64_t nst = K / G_BK;
  return g_kernel_override != 8 && (mode == EPI_BF16 || mode == EPI_BF16_BIAS_TANH || mode == EPI_F32) && nst % 2 == 0 && nst >= 6;
}

extern "C" const char* enh_gemm_h16_variant_mode(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int epi_mode) {
  const GemmPlan pl = gemm_plan(trans_a, trans_b, M, N, K, (epi_mode == EPI_WS || epi_mode == EPI_ATOMIC) ? 1 : 0);
  if (gemm_persistent(pl, trans_a, K, epi_mode)) return gemm_regstaged(K, epi_mode) ? "gemm_w256r_kernel" : "gemm_w256p_kernel";
  return enh_gemm_h16_variant(trans_a, trans_b, M, N, K);
}

extern "C" const char* enh_gemm_h16_variant(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K) {
  static const char* names[8] = {"gemm_kernel", "", "", "gemm_pipe2_kernel", "", "", "", "gemm_w256_kernel"};
  // weight-gradient-shaped calls (both operands contraction-major) are the ones issued with accumulate -> report their split-K plan
  return names[gemm_plan(trans_a, trans_b, M, N, K, (trans_a && trans_b) ? 1 : 0).family];
}

extern "C" size_t enh_gemm_h16_workspace_bytes(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K) {
  const GemmPlan p1 = gemm_plan(trans_a, trans_b, M, N, K, 1), p2 = gemm_plan(trans_a, trans_b, M, N, K, 2);   // (either kind of split: the caller sizes once)
  const int splits = p1.splits > p2.splits ? p1.splits : p2.splits;
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// colpart != null: the caller (enh_gemm_bf16_dtanh_colsum) has checked that the persistent tanh' kernel serves this call
static int gemm_h16_impl(const enh_h16* A, int64_t lda, int trans_a, const enh_h16* B, int64_t ldb, int trans_b,
                         int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_h16* aux,
                         int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                         float* c_f32, enh_h16* c_bf16, int64_t ldc, void* workspace, size_t workspace_bytes, int dtype, void* stream, float* colpart) {
  ENH_REQUIRE_DT(dtype, "enh_gemm_h16");
  ENH_REQUIRE(A && B && (c_f32 || c_bf16), ENH_E_BADARG, "enh_gemm_h16: null pointer");
  ENH_REQUIRE(M > 0 && N > 0 && K > 0, ENH_E_BADARG, "enh_gemm_h16: M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  ENH_REQUIRE(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B), ENH_E_SHAPE,
              "enh_gemm_h16: K, lda, ldb must be multiples of 8 and A, B 16-byte aligned (K=%lld lda=%lld ldb=%lld)", (long long)K, (long long)lda, (long long)ldb);
  ENH_REQUIRE(N % 4 == 0 && ldc % 4 == 0, ENH_E_SHAPE, "enh_gemm_h16: N and ldc must be multiples of 4 (N=%lld ldc=%lld)", (long long)N, (long long)ldc);
  ENH_REQUIRE(!trans_a || M % 8 == 0, ENH_E_SHAPE, "enh_gemm_h16: trans_a needs M %% 8 == 0");
  ENH_REQUIRE(!trans_b || N % 8 == 0, ENH_E_SHAPE, "enh_gemm_h16: trans_b needs N %% 8 == 0");
  ENH_REQUIRE(act == ENH_ACT_NONE || act == ENH_ACT_TANH || (act == ENH_ACT_DTANH && aux && ldaux % 4 == 0), ENH_E_BADARG, "enh_gemm_h16: bad act/aux");
  ENH_REQUIRE(!res || (res_rows > 0 && ldres % 4 == 0), ENH_E_BADARG, "enh_gemm_h16: res needs res_rows > 0 and ldres %% 4 == 0");
  ENH_REQUIRE(accumulate == 0 || (accumulate == 1 && c_f32), ENH_E_BADARG, "enh_gemm_h16: accumulate needs an f32 output");
  ENH_REQUIRE((!c_f32 || aligned16(c_f32)) && (!c_bf16 || (reinterpret_cast<uintptr_t>(c_bf16) & 7u) == 0), ENH_E_SHAPE, "enh_gemm_h16: output alignment");
  ENH_REQUIRE(!workspace || aligned16(workspace), ENH_E_SHAPE, "enh_gemm_h16: workspace must be 16-byte aligned");

  const int split_kind = gemm_splittable(accumulate, c_f32, c_bf16, bias, act, res, workspace);
  GemmPlan pl = gemm_plan(trans_a, trans_b, M, N, K, split_kind);
  if (split_kind == 2 && pl.splits > 1 && workspace_bytes < (size_t)pl.splits * (size_t)M * (size_t)N * sizeof(float))
    pl = gemm_plan(trans_a, trans_b, M, N, K, 0);      // (the forward kind is an optimisation: a workspace sized for something else means "do not split")
  const int family = pl.family;
  const int bm = family == 7 ? G4_BM : G_BM;
  const int bn = family == 7 ? G4_BN : G_BN;

  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.act = act; g.aux = aux; g.ldaux = ldaux; g.res = res; g.ldres = ldres; g.res_rows = res_rows;
  g.accumulate = accumulate; g.c_f32 = c_f32; g.c_bf16 = c_bf16; g.ldc = ldc; g.ws = nullptr;
  g.colpart = colpart;
  g.c2 = g.c3 = g.clo = nullptr; g.ldc2 = g.ldc3 = g.ldlo = 0;
  g.tile_ctr = nullptr;
  if (g_grp_rows > 0) { g.grp_rows = g_grp_rows; g.col_fast = g_col_fast; }
  else { g.col_fast = K >= 2048 ? 1 : 0; g.grp_rows = g.col_fast ? 4 : 8; }
  g.nbm = (int)((M + bm - 1) / bm);
  g.nbn = (int)((N + bn - 1) / bn);
  const int64_t tiles = (int64_t)g.nbm * g.nbn;
  ENH_REQUIRE(tiles < (1ll << 30), ENH_E_SHAPE, "enh_gemm_h16: grid too large");
  g.k_per_split = pl.k_per_split;
  g.splits = pl.splits;
  bool two_pass = false;
  if (pl.splits > 1) {
    // with a workspace: partial slabs + a fixed-order second pass (deterministic); without: f32 atomics into C (N % 4 == 0 and a dense slab need ldc == N
    // only for the workspace form)
    const size_t need = (size_t)pl.splits * (size_t)M * (size_t)N * sizeof(float);
    if (workspace) {   // every kernel family shares the epilogue, so the two-pass form is not tied to the 256 x 256 kernel
      ENH_REQUIRE(workspace_bytes >= need, ENH_E_WORKSPACE, "enh_gemm_h16: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
      g.accumulate = 3; g.ws = (float*)workspace; two_pass = true;
    } else {
      g.accumulate = 2;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  GemmLaunch L;
  L.family = family; L.trans_a = trans_a ? 1 : 0; L.trans_b = trans_b ? 1 : 0;
  L.mode = epi_mode(g); L.form = 0; L.dyn = 0; L.lab = 0;
  L.grid = (unsigned)(tiles * pl.splits);
  if (family == 7) {
    if (g_w256_lab && dtype == ENH_DT_BF16 && trans_a && trans_b && L.mode == EPI_WS) L.lab = g_w256_lab;   // measurement only (enh_debug_gemm_lab)
    // (the persistent epilogues use 16-byte accesses everywhere: operands that only meet the API's weaker alignment rules take the one-tile kernel)
    const bool p_aligned = aligned16(c_bf16) && aligned16(aux) && aligned16(res) && aligned16(bias) && (!c_bf16 || ldc % 8 == 0) && (!aux || ldaux % 8 == 0);
    if (!L.lab && gemm_persistent(pl, trans_a, K, L.mode) && (!res || res_rows == M) && p_aligned) {
      // persistent form: one workgroup per CU walks the tiles
      const int n_cu = cu_budget();
      const int64_t wgs = tiles < n_cu ? tiles : n_cu;
      L.dyn = g_dyn_schedule && wgs >= 8 ? 1 : 0;     // (every XCD queue needs a workgroup that serves it)
      if (L.dyn) {
        g.tile_ctr = next_tile_counters();
        ENH_REQUIRE(g.tile_ctr, ENH_E_BADARG, "enh_gemm_h16: tile counters unavailable");
      }
      L.form = gemm_regstaged(K, L.mode) ? 2 : 1;
      L.grid = (unsigned)wgs;
    }
  }
  if (dtype == ENH_DT_F16) gemm_launch<F16>(g, L, s); else gemm_launch<BF16>(g, L, s);
  if (two_pass) {
    const int64_t MN = M * N;
    const dim3 rg((unsigned)((MN / 4 + 255) / 256));
    if (c_bf16) {
      ENH_DT_DISPATCH(dtype, (splitk_reduce16_kernel<OT><<<rg, 256, 0, s>>>(g.ws, pl.splits, MN, N, c_bf16, ldc)));
    } else {
      splitk_reduce_kernel<<<rg, 256, 0, s>>>(g.ws, pl.splits, MN, N, c_f32, ldc, accumulate, bias, res, ldres, res_rows);
    }
  }
  return enh_check_launch("enh_gemm_h16");
}

extern "C" int enh_gemm_h16_ws(const enh_h16* A, int64_t lda, int trans_a, const enh_h16* B, int64_t ldb, int trans_b,
                               int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_h16* aux,
                               int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                               float* c_f32, enh_h16* c_h16, int64_t ldc, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  return gemm_h16_impl(A, lda, trans_a, B, ldb, trans_b, M, N, K, bias, act, aux, ldaux, res, ldres, res_rows, accumulate, c_f32, c_h16, ldc,
                       workspace, workspace_bytes, dtype, stream, nullptr);
}

// C = (A B) * (1 - aux^2) -> bf16, AND colsum[n] (+)= sum_m C[m][n] over the stored (rounded) values: the input gradient through a tanh together with
// the bias gradient of the Linear in front of it (the reference's autograd of FeedForward: layers.py:99-101 Linear -> nn.Tanh -> Linear).  Where the
// persistent tanh' kernel serves the call, its epilogue leaves one partial row per 128 rows in `ws` and a fixed-order second pass adds them (the
// separate column-sum kernel would re-read all of C: 805 MB at the base config); otherwise: the plain GEMM, then enh_colsum_bf16_ws.
static bool dtanh_colsum_fused(int trans_b, int64_t M, int64_t N, int64_t K) {
  const GemmPlan pl = gemm_plan(0, trans_b, M, N, K, 0);
  return gemm_persistent(pl, 0, K, EPI_BF16_DTANH) && M % 256 == 0 && N % 256 == 0;
}
extern "C" size_t enh_gemm_h16_dtanh_colsum_workspace_bytes(int trans_b, int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0) return 0;
  return dtanh_colsum_fused(trans_b, M, N, K) ? (size_t)(M / 128) * (size_t)N * sizeof(float) : enh_colsum_h16_workspace_bytes(M, N);
}
extern "C" int enh_gemm_h16_dtanh_colsum(const enh_h16* A, int64_t lda, const enh_h16* B, int64_t ldb, int trans_b, int64_t M, int64_t N, int64_t K,
                                         const enh_h16* aux, int64_t ldaux, enh_h16* c_bf16, int64_t ldc, float* colsum, int accumulate_colsum,
                                         void* ws, size_t ws_bytes, int dtype, void* stream) {
  ENH_REQUIRE_DT(dtype, "enh_gemm_h16_dtanh_colsum");
  ENH_REQUIRE(colsum && ws && c_bf16 && aux, ENH_E_BADARG, "enh_gemm_h16_dtanh_colsum: null pointer");
  ENH_REQUIRE(ws_bytes >= enh_gemm_h16_dtanh_colsum_workspace_bytes(trans_b, M, N, K) && aligned16(ws), ENH_E_WORKSPACE,
              "enh_gemm_h16_dtanh_colsum: workspace of %zu bytes needed, %zu given", enh_gemm_h16_dtanh_colsum_workspace_bytes(trans_b, M, N, K), ws_bytes);
  if (!dtanh_colsum_fused(trans_b, M, N, K)) {
    const int rc = enh_gemm_h16(A, lda, 0, B, ldb, trans_b, M, N, K, nullptr, ENH_ACT_DTANH, aux, ldaux, nullptr, 0, 0, 0, nullptr, c_bf16, ldc, dtype, stream);
    return rc ? rc : enh_colsum_h16_ws(c_bf16, M, N, ldc, colsum, accumulate_colsum, ws, ws_bytes, dtype, stream);
  }
  const int rc = gemm_h16_impl(A, lda, 0, B, ldb, trans_b, M, N, K, nullptr, ENH_ACT_DTANH, aux, ldaux, nullptr, 0, 0, 0, nullptr, c_bf16, ldc, nullptr, 0,
                               dtype, stream, (float*)ws);
  return rc ? rc : enh_colsum_reduce_launch((const float*)ws, (int)(M / 128), N, colsum, accumulate_colsum, (hipStream_t)stream);
}

extern "C" int enh_gemm_h16(const enh_h16* A, int64_t lda, int trans_a, const enh_h16* B, int64_t ldb, int trans_b,
                            int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_h16* aux,
                            int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                            float* c_f32, enh_h16* c_h16, int64_t ldc, int dtype, void* stream) {
  return enh_gemm_h16_ws(A, lda, trans_a, B, ldb, trans_b, M, N, K, bias, act, aux, ldaux, res, ldres, res_rows, accumulate, c_f32, c_h16, ldc,
                         nullptr, 0, dtype, stream);
}

// ---- x3 producers fused (round 5) ------------------------------------------------------------------------------------------------------------
// v = A B^T (+ bias, tanh) is never written: the persistent kernel's epilogue leaves hi = bf16(v) and lo = bf16(v - hi) directly — what the x3 forward did
// with an f32 [M][N] round trip and csrc/x3.hip's split2 / split3 kernels (2.4 GB per qkv projection, 5.6 GB per fc1 at the base config, B = 128).
// hi goes to `hi` and, where non-null, `hi2` / `hi3` (the x3 A-operand row [hi | lo | hi] needs it twice, the backward's arena once more); lo to `lo`.
// Served where the persistent 256 x 256 kernel serves (enh_gemm_bf16_split_fused); the caller otherwise keeps the two-call form.
static bool gemm_split_plan_ok(int64_t M, int64_t N, int64_t K) {
  if (g_kernel_override == 0 || g_kernel_override == 3 || g_kernel_override == 7) return false;
  const GemmPlan pl = gemm_plan(0, 0, M, N, K, 0);
  return pl.family == 7 && pl.splits == 1 && K / G_BK >= 3 && M % 256 == 0 && N % 256 == 0;
}
extern "C" int enh_gemm_bf16_split_fused(int64_t M, int64_t N, int64_t K) { return gemm_split_plan_ok(M, N, K) ? 1 : 0; }

extern "C" int enh_gemm_bf16_split(const enh_h16* A, int64_t lda, const enh_h16* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const float* bias, int act,
                                   enh_h16* hi, int64_t ldhi, enh_h16* lo, int64_t ldlo, enh_h16* hi2, int64_t ldhi2, enh_h16* hi3, int64_t ldhi3,
                                   void* stream) {
  ENH_REQUIRE(A && B && hi && lo, ENH_E_BADARG, "enh_gemm_bf16_split: null pointer");
  ENH_REQUIRE(M > 0 && N > 0 && K > 0, ENH_E_BADARG, "enh_gemm_bf16_split: M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  ENH_REQUIRE(gemm_split_plan_ok(M, N, K), ENH_E_SHAPE, "enh_gemm_bf16_split: M=%lld N=%lld K=%lld is not served by the persistent 256 x 256 kernel (ask enh_gemm_bf16_split_fused)",
              (long long)M, (long long)N, (long long)K);
  ENH_REQUIRE((act == ENH_ACT_NONE && !bias) || (act == ENH_ACT_TANH && bias), ENH_E_BADARG, "enh_gemm_bf16_split: plain (no bias) or bias + tanh");
  ENH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldhi % 8 == 0 && ldlo % 8 == 0 && (!hi2 || ldhi2 % 8 == 0) && (!hi3 || ldhi3 % 8 == 0) && aligned16(A) && aligned16(B) &&
              aligned16(hi) && aligned16(lo) && aligned16(hi2) && aligned16(hi3) && aligned16(bias), ENH_E_SHAPE, "enh_gemm_bf16_split: 16-byte alignment / ld %% 8");
  GemmArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.act = act; g.aux = nullptr; g.ldaux = 0; g.res = nullptr; g.ldres = 0; g.res_rows = 0;
  g.accumulate = 0; g.c_f32 = nullptr; g.c_bf16 = hi; g.ldc = ldhi; g.ws = nullptr; g.colpart = nullptr;
  g.c2 = hi2; g.ldc2 = ldhi2; g.c3 = hi3; g.ldc3 = ldhi3; g.clo = lo; g.ldlo = ldlo;
  g.tile_ctr = nullptr;
  if (g_grp_rows > 0) { g.grp_rows = g_grp_rows; g.col_fast = g_col_fast; }
  else { g.col_fast = K >= 2048 ? 1 : 0; g.grp_rows = g.col_fast ? 4 : 8; }
  g.nbm = (int)(M / 256); g.nbn = (int)(N / 256);
  g.k_per_split = K; g.splits = 1;
  const int64_t tiles = (int64_t)g.nbm * g.nbn;
  const int64_t nst_ = K / G_BK;
  const int regst = (g_kernel_override != 8 && nst_ % 2 == 0 && nst_ >= 6) ? 1 : 0;
  const int n_cu = cu_budget();
  const int64_t wgs = tiles < n_cu ? tiles : n_cu;
  const int dyn = g_dyn_schedule && wgs >= 8 ? 1 : 0;
  if (dyn) {
    g.tile_ctr = next_tile_counters();
    ENH_REQUIRE(g.tile_ctr, ENH_E_BADARG, "enh_gemm_bf16_split: tile counters unavailable");
  }
  gemm_split_launch_bf16(g, regst, dyn, act == ENH_ACT_TANH ? 1 : 0, (unsigned)wgs, (hipStream_t)stream);
  return enh_check_launch("enh_gemm_bf16_split");
}
