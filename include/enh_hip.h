/*
 * enh_hip.h — C ABI of libenh_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the stage-1
 * ViT-VQGAN / RQ-VAE training path of thuanz123/enhancing-transformers.
 *
 * Boundary contract:
 *   - extern "C", plain pointers and sizes only; no torch / C++ types cross this boundary.
 *   - Every pointer is a DEVICE pointer owned by the caller (the Python host side allocates them as
 *     torch tensors and passes data_ptr()).  The library never allocates, frees or retains memory and
 *     never synchronises: all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - Return value: 0 (ENH_OK) or a negative ENH_E_* code; a HIP launch error is returned as
 *     -(1000 + hipError_t).  enh_last_error() gives a thread-local human-readable message.  This mirrors
 *     the reference's native-op precedent, where TORCH_CHECK failures surface as a Python RuntimeError
 *     (enhancing/losses/op/fused_bias_act.cpp:9-15); the Python wrapper raises RuntimeError on rc != 0.
 *   - All functions are stateless and re-entrant (they are called from autograd worker threads too), EXCEPT the explicit library
 *     state behind the enh_*_set_* / enh_set_cu_budget / enh_debug_* setters (process-global kernel-family switches, meant for A/B measurements).
 *     Device-bound state — the CU count, the CU budget and the persistent GEMMs' tile-claim counters — is keyed by the calling thread's CURRENT
 *     device (hipGetDevice) since round 6: a process may drive several GPUs through this ABI (the library still never sets a device).
 *
 * The reference reaches this path through stock PyTorch ops, not an FFI; each entry
 * below cites the reference lines whose arithmetic it replaces.  Paths are relative to the reference
 * repository root.
 */
#ifndef ENH_HIP_H
#define ENH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ENH_OK 0
#define ENH_E_BADARG (-1)      /* null pointer / non-positive size */
#define ENH_E_SHAPE (-2)       /* unsupported shape or alignment */
#define ENH_E_WORKSPACE (-3)   /* workspace too small */
#define ENH_E_HIP_BASE (-1000) /* rc = ENH_E_HIP_BASE - hipError_t */

/* 16-bit operand types.  Every MFMA operand of the product path is a 16-bit float in ONE of two formats, selected per call by the `dtype`
 * argument (always the parameter in front of `stream`):
 *   ENH_DT_BF16  bfloat16 (8 significand bits, f32's exponent range)         — v_mfma_f32_*_bf16, v_cvt_pk_bf16_f32
 *   ENH_DT_F16   IEEE binary16 (11 significand bits, max 65504, min normal 6.1e-5) — v_mfma_f32_*_f16, v_cvt_pk_f16_f32
 * Both roundings are round-to-nearest-even, accumulation is f32 in both, and the two run at the same MFMA rate.  fp16 is the reference's own
 * mixed-precision dtype (main.py:25,52: --use_amp -> Lightning precision=16 = fp16 autocast + GradScaler): its operand rounding is 8x smaller than
 * bf16's, which is what brings the activations within 1e-3 of the fp32 reference in ONE MFMA pass; gradients need a loss scale (enh_nonfinite_flag,
 * enh_adamw_step skip_flag).  Every 16-bit tensor of one call has the call's dtype; entries whose names still say "bf16" and take no dtype (the x3
 * split-operand family) are bf16-only.  The channels-last convolution family (discriminator, LPIPS trunk) takes the dtype too (round 6). */
#define ENH_DT_BF16 0
#define ENH_DT_F16 1
#define ENH_DT_F32 2   /* enh_im2col / enh_col2im only: f32 columns for the exact-f32 GEMM (the discriminator's parity instrument) */
typedef uint16_t enh_h16;  /* raw 16-bit float bits: bfloat16 or binary16, per the call's dtype */
typedef enh_h16 enh_bf16;  /* raw bfloat16 bits (bf16-only entries) */

const char* enh_last_error(void);
#define ENH_ABI_VERSION 17  /* bumped whenever a signature below changes; the bindings check it at load */
int enh_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Vector / residual quantizer  — enhancing/modules/stage1/quantizers.py:38-92
 * ------------------------------------------------------------------------------------------------
 * Arithmetic contract (bit-exactly restated in oracle/vq_oracle.c):
 *   n(x) = x / max(sqrt(S(x)), 1e-12),  S(x) = chain(x[0..15]) + chain(x[16..31]),
 *   chain = ascending fmaf chain from 0;  zz = S(zn), ee_k = S(en_k);
 *   dot_k = fmaf chain over the order k = 0,16,1,17,...,15,31 (f32 MFMA 32x32x2, exact f32);
 *   d_k = (zz + ee_k) - 2*dot_k ; idx = argmin_k d_k, lowest k on ties  (quantizers.py:78-83).
 * embed_dim must be 32 (every shipped config: configs/imagenet_vitvq_*.yaml:17-19).
 */

/* bytes of scratch needed by enh_vq_forward / enh_vq_backward for a codebook of n_embed codes and M tokens */
size_t enh_vq_workspace_bytes(int64_t M, int n_embed, int depth);

/* Forward of BaseQuantizer.forward + VectorQuantizer.quantize (quantizers.py:38-63,74-92).
 *   z        [M,32] f32   quantizer input (pre_quant output)
 *   codebook [K,32] f32   quantizer.embedding.weight
 *   depth    1 = plain VQ ; >1 = residual quantizer with one shared codebook (use_residual, num_quantizers)
 *   zq_out   [M,32] f32   straight-through VALUE  z + (sum_i en_i - z)          (quantizers.py:61)
 *   zq_h16   [M,32] 16-bit optional (may be NULL): same, rounded to `dtype` (operand of post_quant GEMM)
 *   idx_out  [M,depth] i64 code indices (stacked on the last axis as quantizers.py:55)
 *   loss_out [1] f32      mean_i( beta*mean((en_i-zn_i)^2) + mean((en_i-zn_i)^2) ) (quantizers.py:56,89-90)
 */
int enh_vq_forward(const float* z, const float* codebook, int64_t M, int n_embed, int embed_dim,
                   float beta, int depth, int use_norm, float* zq_out, enh_h16* zq_h16,
                   int64_t* idx_out, float* loss_out, void* workspace, size_t workspace_bytes,
                   int dtype, void* stream);

/* Backward (derived from the reference autograd graph):
 *   g_out   [M,32] f32  grad wrt returned z_q;  g_loss_dev: optional device scalar multiplying g_loss
 *   use_residual: 1 when the forward ran the residual loop (z is detached at quantizers.py:43, so the
 *           encoder then receives g_out only, and the codebook grad gets the cross-depth term)
 *   dz      [M,32] f32  (VQ: g_out + beta-term through the normalise Jacobian; RQ: g_out only)
 *   dz_h16 optional 16-bit copy (`dtype`) ; d_codebook [K,32] f32 is ACCUMULATED into (caller zeroes it).
 */
int enh_vq_backward(const float* z, const float* codebook, const int64_t* idx, const float* g_out,
                    float g_loss, const float* g_loss_dev, int64_t M, int n_embed, int embed_dim,
                    float beta, int depth, int use_residual, int use_norm, float* dz, enh_h16* dz_h16,
                    float* d_codebook, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* decode_codes front half (vitvqgan.py:81-87): out[m] = sum_i n(codebook[idx[m,i]]) as f32 and as a 16-bit operand (`dtype`) */
int enh_vq_lookup(const float* codebook, const int64_t* idx, int64_t M, int n_embed, int embed_dim,
                  int depth, int use_norm, float* out, enh_h16* out_h16, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm — nn.LayerNorm(dim), eps 1e-5, biased variance (enhancing/modules/stage1/layers.py:85-92,143)
 * ------------------------------------------------------------------------------------------------ */
/* y = (x-mean)*rstd*w + b.  x [M,D] f32; y_h16 [M,D] (GEMM operand, `dtype`) and/or y_f32 (either may be NULL);
 * mean,rstd [M] f32 saved for backward (may be NULL for inference).  D % 4 == 0, D <= 2048. */
int enh_layernorm_forward(const float* x, const float* w, const float* b, int64_t M, int D, float eps,
                          enh_h16* y_h16, float* y_f32, float* mean, float* rstd, int dtype, void* stream);
/* dx = LN-backward(dy) [+ dres];  the upstream gradient [M,D] is passed EITHER as dy (f32) OR as dy_h16 (the 16-bit output of the dgrad
 * GEMM that produced it) — exactly one non-NULL;  dres optional f32 residual-stream gradient that is added;
 * dx_f32 [M,D] f32 and optional dx_h16 copy; dw,db [D] f32 are ACCUMULATED (atomics; caller zeroes);
 * dx_colsum [D] f32, optional: += column sums of dx — the bias gradient of the Linear whose output feeds this
 * residual stream (to_out / fc2), fused here so no separate reduction pass over dx is needed. */
int enh_layernorm_backward(const float* dy, const enh_h16* dy_h16, const float* x, const float* w, const float* mean,
                           const float* rstd, const float* dres, int64_t M, int D, float* dx_f32,
                           enh_h16* dx_h16, float* dw, float* db, float* dx_colsum, int dtype, void* stream);
/* Deterministic form: the per-workgroup column partials of dw / db / dx_colsum go to `ws` (enh_layernorm_backward_workspace_bytes) and a second pass
 * adds them in a fixed order — bit-reproducible from run to run; enh_layernorm_backward uses f32 atomics instead. */
size_t enh_layernorm_backward_workspace_bytes(int64_t M, int D);
int enh_layernorm_backward_ws(const float* dy, const enh_h16* dy_h16, const float* x, const float* w, const float* mean,
                              const float* rstd, const float* dres, int64_t M, int D, float* dx_f32, enh_h16* dx_h16, float* dw,
                              float* db, float* dx_colsum, void* ws, size_t ws_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 16-bit-operand MFMA GEMM with fused epilogue — every nn.Linear / patch conv on the path
 * (layers.py:99-101,118,120,169,204; vitvqgan.py:38-39) and their dgrad / wgrad.
 * ------------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  A, B (and aux, c_h16) in `dtype`, f32 accumulation on v_mfma_f32_32x32x16_{bf16,f16} /
 * v_mfma_f32_16x16x32_{bf16,f16}.  Shapes, plans, tile order and split-K decisions do not depend on dtype.
 *   trans_a = 0: A stored [M][K] (lda elements between rows) ; 1: A stored [K][M]
 *   trans_b = 0: B stored [N][K] (nn.Linear.weight layout)   ; 1: B stored [K][N]
 *   epilogue, in this order:  v = acc ; v += bias[n] (f32, optional) ; v = tanh(v) if act == 1 ;
 *     v *= (1 - aux[m,n]^2) if act == 2 (aux = saved tanh output: tanh backward) ;
 *     v += res[(m % res_rows), n] (f32, optional; res_rows = M for a residual, n_tokens for a pos-embed) ;
 *     v += C_old if accumulate ;  store to c_f32 and/or c_h16 (ldc).
 *   Requirements: K % 8 == 0, lda/ldb % 8 == 0, 16-byte aligned bases; for trans_a M % 8 == 0; trans_b N % 8 == 0.
 */
#define ENH_ACT_NONE 0
#define ENH_ACT_TANH 1
#define ENH_ACT_DTANH 2
int enh_gemm_h16(const enh_h16* A, int64_t lda, int trans_a, const enh_h16* B, int64_t ldb, int trans_b,
                 int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_h16* aux,
                 int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                 float* c_f32, enh_h16* c_h16, int64_t ldc, int dtype, void* stream);

/* The same with a caller-owned split-K workspace.  Weight-gradient-shaped calls (accumulate = 1, f32 output only, no bias / act / res) whose
 * output tiles cannot fill the chip are split along K; with a workspace of enh_gemm_h16_workspace_bytes() the K slices write partial slabs
 * [splits][M][N] and a second pass adds them to C in a fixed order: bit-reproducible, no f32 atomics (needs ldc == N).  workspace = NULL
 * (what enh_gemm_h16 passes) falls back to f32 atomicAdd into C.  Replaces the autograd wgrad of every nn.Linear (layers.py:99-101,118,120).
 * Round 6 (the reference's shipped batch sizes, configs/imagenet_vitvq_large.yaml:31: 2 images per GPU): with a workspace, calls WITHOUT an activation whose
 * 128 x 128 tiles cover less than half of the resident workgroup slots and whose K >= 2048 are split as well — f32 output with any of bias / residual /
 * accumulate, or a plain 16-bit output; the second pass then carries the epilogue (bias, residual row m mod res_rows, previous C; or the one RNE pack).
 * Same bits for the same call and workspace size; a workspace smaller than enh_gemm_h16_workspace_bytes() simply means "do not split" for these calls.
 * Never chosen at training sizes (M >= 8192 token rows with N >= 768 has >= 256 tiles). */
int enh_gemm_h16_ws(const enh_h16* A, int64_t lda, int trans_a, const enh_h16* B, int64_t ldb, int trans_b,
                    int64_t M, int64_t N, int64_t K, const float* bias, int act, const enh_h16* aux,
                    int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows, int accumulate,
                    float* c_f32, enh_h16* c_h16, int64_t ldc, void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* C[M,N] = (A B) * (1 - aux^2) -> 16-bit AND colsum[n] (+)= sum_m C[m][n] over the stored (rounded) values: the input gradient through a tanh plus the
 * bias gradient of the Linear in front of it — the autograd of `nn.Linear -> nn.Tanh` in FeedForward (layers.py:99-101).  On whole 256 x 256 tiles
 * the tanh' kernel's epilogue leaves one partial row per 128 rows in `ws` and a fixed-order second pass adds them (bit-reproducible; the separate
 * column-sum kernel re-reads all of C: 805 MB per layer at the base config); any other shape runs enh_gemm_h16 then enh_colsum_h16_ws.
 * A is [M][K] (never transposed here); trans_b as in enh_gemm_h16. */
size_t enh_gemm_h16_dtanh_colsum_workspace_bytes(int trans_b, int64_t M, int64_t N, int64_t K);
int enh_gemm_h16_dtanh_colsum(const enh_h16* A, int64_t lda, const enh_h16* B, int64_t ldb, int trans_b, int64_t M, int64_t N, int64_t K,
                              const enh_h16* aux, int64_t ldaux, enh_h16* c_h16, int64_t ldc, float* colsum, int accumulate_colsum,
                              void* ws, size_t ws_bytes, int dtype, void* stream);
/* bytes of workspace the split-K plan of this shape needs, whichever kind of call (weight-gradient or forward kind) would split it (0 = never split) */
size_t enh_gemm_h16_workspace_bytes(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K);

/* name of the kernel family enh_gemm_h16 launches for this shape (measurement aid: lets callers label timings with
 * the symbol a profiler will report); the choice is per shape */
const char* enh_gemm_h16_variant(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K);
/* the same for a call with a given fused epilogue (epi_mode: 0 generic, 1 16-bit, 2 16-bit + bias + tanh, 3 16-bit * tanh', 4 f32 + bias + residual stream,
 * 5 f32, 6 / 7 split-K partials): forward / input-gradient calls of modes 1-5 on whole 256 x 256 tiles run a PERSISTENT form of the 256 x 256
 * kernel — "gemm_w256p_kernel": one workgroup per CU walks the tiles, the next tile's operands are requested before the stores of this one;
 * "gemm_w256r_kernel" (modes 1, 2, 5; an even number >= 6 of 64-deep K stages): the same with the A operand staged through registers two
 * stages ahead, which lets its requests stay in flight 1.75 stages instead of 0.75 */
const char* enh_gemm_h16_variant_mode(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, int epi_mode);
/* A/B measurement aid: force a kernel family for every later call that it can serve (-1 = per-shape choice [default], 0 = register-staged
 * fallback, 3 = pipe2 128x128, 7 = w256 256x256 / 4 waves with one tile per workgroup everywhere, 8 = w256 with the persistent form w256p where it
 * serves, 9 = as 8 plus w256r where that serves [what the default picks]).  Process-global, set explicitly by the caller (the Python
 * binding maps the ENH_GEMM_KERNEL environment variable onto it); the library itself reads no environment. */
int enh_gemm_set_kernel(int family);
/* Tile schedule of the persistent 256 x 256 kernels: 1 [default] = tiles are CLAIMED from one queue per XCD (atomic counters in library-owned device
 * words; the queue order is the static walk's, so operand slices still meet in one L2) — a workgroup that gets its CU late finds the queue empty and
 * exits, the launch slows by (CUs held by others) / CUs; 0 = static partition (workgroup b walks tiles b, b + grid, ...: every workgroup must run, a
 * collective's kernel holding k CUs at dispatch costs up to 2x on that launch).  Same result bits either way.  Process-global measurement switch. */
int enh_gemm_set_scheduler(int dynamic);
/* CU budget: the number of CUs GEMM launches may count on (persistent grid size, one-round split-K plans); 0 = all CUs of the device [default].
 * Data-parallel training sets it to (CUs - the collective's channels) while gradient buckets are in flight (engine/ddp.py; reference main.py:54-57 is
 * DDP over NCCL).  enh_get_cu_budget returns the effective value. */
int enh_set_cu_budget(int n_cus);
int enh_get_cu_budget(void);
/* Measurement aid: holds n_wg CUs (one workgroup each, the CU's whole LDS) for ms milliseconds (clamped to 2000) on `stream` — the stand-in for a
 * collective's kernel in the one-GPU contention experiment (tools/comm_contention.py). */
/* measurement only (results are WRONG while it is on): the split-K weight-gradient loop with 1 = plain 16-byte fragment reads instead of the
 * transposing ones (same LDS bytes), 2 = no fragment reads, 3 = neither fragment reads nor staging requests (MFMAs + barriers), 4 = every second
 * staging request, 5 = all fragment reads, no staging requests, 6 .. 9 = the shipped loop with cache-policy bits sc0 / nt / sc1 / sc0 + sc1 on its
 * staging requests (correct results); 0 = off (tools/wgrad_lab.py, profiles/r04_gemm_fill_lab.txt) */
int enh_debug_gemm_lab(int variant);
/* Measurement aid (results unchanged): tile order of the forward / input-gradient GEMMs — `grp_rows` row panels per group (8, rows fastest: the 32
 * workgroups of an XCD have an 8 x 4 patch of tiles in flight), col_fast = 1: columns fastest inside a group (a few rows x ALL column tiles in flight:
 * every A panel is fetched by one XCD once).  grp_rows = 0: the library's per-shape default (col_fast for K >= 2048).  tools/gemm_ld_lab.py,
 * profiles/r05_gemm_landing_lab.txt */
int enh_debug_gemm_order(int grp_rows, int col_fast);
/* Measurement aid: force the number of K slices of the split-K plans (weight gradients); 0 = the planner's choice.  With 8 x (tiles per slice)
 * workgroups every XCD holds whole slices (profiles/r05_gemm_landing_lab.txt §5). */
int enh_debug_gemm_splits(int splits);
int enh_debug_occupy_cus(int n_wg, float ms, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention — Attention.forward layers.py:122-132 without materialising the N x N matrix
 * ------------------------------------------------------------------------------------------------
 * qkv [B,N,3*H*64] 16-bit (`dtype`) packed exactly as to_qkv emits it (q | k | v thirds, head-major, layers.py:123-124);
 * out [B,N,H*64] 16-bit in the 'b n (h d)' layout to_out consumes (layers.py:130); lse [B,H,N] f32 =
 * row log-sum-exp of the scaled scores (saved for backward).  dim_head = 64, N % 64 == 0.
 * q_prescaled = 1: the q third already holds q * scale * log2(e) (the caller folded the softmax scale into the projection's q rows, once, in
 * fp32 before the 16-bit rounding).  The score products are then log2-domain logits and the kernels feed -max / -lse / -delta through the MFMA C
 * operand instead of spending vector instructions on them (the kernels are vector-issue bound, profiles/r03_attention_lab.txt).  Semantics are
 * unchanged: out / lse are those of softmax(q k^T scale) v for the UNSCALED q, dqkv's q third is the gradient with respect to the unscaled q.
 */
/* NaN note: the attention objects are built with -fno-honor-nans (the row maxima must fuse into v_max3_f32 without canonicalising moves): scores that
 * contain NaN (a diverged run) are NOT guaranteed to propagate as NaN through out / lse — check the loss / the gradients (enh_nonfinite_flag) instead. */
int enh_attention_forward(const enh_h16* qkv, int B, int N, int H, float scale, int q_prescaled, enh_h16* out, float* lse,
                          int dtype, void* stream);
/* kernel family per pass for A/B measurements (explicit library state, like enh_gemm_set_kernel), 0 = the library's choice:
 *   fwd: 1 four-wave kernel (round 2; serves both q conventions), 5 the same skeleton with the running reference as the MFMA C operand, a packed exact row
 *        sum and the K / V tiles by LDS-DMA (round 5; pre-scaled q, otherwise family 1) [default]
 *   dq : 1 round-2 arithmetic, 3 -delta (-lse too when q is pre-scaled) as MFMA C operands [default since round 5]
 *   dkv: 1 round-2 arithmetic, 2 -delta (-lse too when q is pre-scaled) as MFMA C operands [default]
 * (the software-pipelined round-3 kernels and the eight-wave antiphase kernels of round 4 were measured slower and are deleted.)
 * Same results up to rounding: every family passes the same parity and bit-reproducibility tests. */
int enh_attention_set_kernel(int fwd, int dq, int dkv);
/* dqkv [B,N,3*H*64] 16-bit ; delta_ws [B,H,N] f32 scratch.  With ENH_DT_F16 the caller keeps dout inside fp16's range (loss scale): p o (dP - delta) is
 * packed to fp16 before the dQ / dK products. */
int enh_attention_backward(const enh_h16* qkv, const enh_h16* out, const enh_h16* dout, const float* lse,
                           int B, int N, int H, float scale, int q_prescaled, enh_h16* dqkv, float* delta_ws, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * "x3" split-bf16 operands — the parity-grade ENCODER forward (round 4).  The reference's forward is fp32 end to end
 * (layers.py:118-132,145-150; vitvqgan.py:61-66; quantizers.py:74-92 consumes its output); bf16 operands flip ~2 % of the argmin decisions
 * downstream.  A value v is carried as hi = bf16(v), lo = bf16(v - hi) and a product as a_hi b_hi + a_lo b_hi + a_hi b_lo in the fp32 MFMA
 * accumulator: ~1e-5 relative end to end at a third of the bf16 MFMA rate (the exact-f32 MFMA has a sixteenth).
 * A GEMM is ONE enh_gemm_h16 (ENH_DT_BF16) call with K' = 3K on K-concatenated rows  A' = [a_hi | a_lo | a_hi],  B' = [b_hi | b_hi | b_lo].
 * ------------------------------------------------------------------------------------------------ */
/* y3[m] = the x3 row of f(x[m] (+ bias)), f = identity (act 0) or tanh (act 1: FeedForward's activation, layers.py:99-100);
 * order 0: [hi | lo | hi] (A operand), order 1: [hi | hi | lo] (B operand = weights).  x f32 [M,K] (ldx), y3 bf16 [M,3K] (ldy3),
 * y_hi optional bf16 [M,K] (ldy_hi): the hi plane alone (= what the bf16 path stores; the backward's operand).  K % 8 == 0. */
int enh_split3_bf16(const float* x, int64_t ldx, int64_t M, int64_t K, const float* bias, int act, int order, enh_bf16* y3, int64_t ldy3,
                    enh_bf16* y_hi, int64_t ldy_hi, void* stream);
/* hi[i] = bf16(x[i]), lo[i] = bf16(x[i] - hi[i]); n % 8 == 0 (the packed q | k | v projection for enh_attention_forward_x3) */
int enh_split2_bf16(const float* x, int64_t n, enh_bf16* hi, enh_bf16* lo, void* stream);
/* The x3 producers fused into the GEMM that computes their input (round 5): v = A B^T (A [M][K'], B [N][K'] row-major, K' = 3K x3 rows) or
 * v = tanh(A B^T + bias) — reference layers.py:118 (to_qkv) and :99-100 (Linear + Tanh) — leaves hi = bf16(v) in `hi` (and in `hi2`, `hi3` where non-null)
 * and lo = bf16(v - hi) in `lo` (each [M][N] with its own leading dimension), i.e. enh_gemm_h16 (f32 out) + enh_split2_bf16 / enh_split3_bf16 without the f32
 * round trip; plain mode bit-identical to that pair.  act: ENH_ACT_NONE (bias must be null) | ENH_ACT_TANH (bias required).  Served where the persistent
 * 256 x 256 kernel serves: enh_gemm_bf16_split_fused(M, N, K') == 1 (M, N multiples of 256, enough tiles to fill the chip); ENH_E_SHAPE otherwise. */
int enh_gemm_bf16_split_fused(int64_t M, int64_t N, int64_t K);
int enh_gemm_bf16_split(const enh_bf16* A, int64_t lda, const enh_bf16* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const float* bias, int act,
                        enh_bf16* hi, int64_t ldhi, enh_bf16* lo, int64_t ldlo, enh_bf16* hi2, int64_t ldhi2, enh_bf16* hi3, int64_t ldhi3, void* stream);
/* enh_layernorm_forward that writes the x3 row y3 [M,3D] (and, optionally, the plain bf16 / f32 outputs); same statistics bits */
int enh_layernorm_forward_x3(const float* x, const float* w, const float* b, int64_t M, int D, float eps, enh_bf16* y3, enh_bf16* y_bf16,
                             float* y_f32, float* mean, float* rstd, void* stream);
/* enh_attention_forward on split operands: qkv_hi / qkv_lo [B,N,3*H*64] (unscaled q); S = Q K^T and O = P V as three MFMA passes each,
 * softmax statistics in fp32.  out3 [B,N,3*H*64] = the x3 row [hi | lo | hi] of the 'b n (h d)' output (A operand of to_out);
 * out_bf16 optional [B,N,H*64] (= the hi plane, what enh_attention_backward reads); lse as enh_attention_forward. */
int enh_attention_forward_x3(const enh_bf16* qkv_hi, const enh_bf16* qkv_lo, int B, int N, int H, float scale, enh_bf16* out3,
                             enh_bf16* out_bf16, float* lse, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Patch (un)embedding data movement, pixel loss, reductions, optimizer
 * ------------------------------------------------------------------------------------------------ */
/* 'b c h w -> (b gy gx) (c ph pw)' gather of Conv2d(k=s=p) as a GEMM operand (layers.py:168-171,178);
 * also maps an image-layout gradient to the patch layout (same permutation). */
int enh_patchify(const float* img, int B, int C, int H, int W, int p, enh_h16* patches, int dtype, void* stream);
/* Inverse scatter of ConvTranspose2d(k=s=p) (layers.py:202-205,212) fused with the pixel losses
 * (vqperceptual.py:113-114): pix [M, C*p*p] f32 (to_pixel GEMM output incl. bias) -> xrec [B,C,H,W] f32;
 * if target != NULL: sums[0] += sum|xrec-x|, sums[1] += sum (xrec-x)^2 (f64 atomics; caller zeroes) and
 * dpix_h16 [M, C*p*p] (`dtype`) = (w_l1*sign(diff) + w_l2*2*diff) / numel  (grad of w_l1*L1 + w_l2*L2), optional; grad_scale_dev (optional device
 * float) multiplies dpix only — the loss scale of an ENH_DT_F16 backward, kept on the device (enh_loss_scale_update); the sums are unaffected. */
int enh_unpatchify_loss(const float* pix, const float* target, int B, int C, int H, int W, int p, float w_l1,
                        float w_l2, float* xrec, double* sums, enh_h16* dpix_h16, const float* grad_scale_dev, int dtype, void* stream);
/* out[n] (+)= sum_m x[m,n] (bias gradients); x 16-bit [M,N] (`dtype`) */
int enh_colsum_h16(const enh_h16* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, int dtype, void* stream);
/* deterministic form: per-row-chunk partials in `ws` (enh_colsum_h16_workspace_bytes), added in a fixed order; enh_colsum_h16 uses f32 atomics */
size_t enh_colsum_h16_workspace_bytes(int64_t M, int64_t N);
int enh_colsum_h16_ws(const enh_h16* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* ws, size_t ws_bytes, int dtype, void* stream);
/* f32 -> 16-bit cast, round-to-nearest-even (weight shadows) */
int enh_cast_f32_h16(const float* x, enh_h16* y, int64_t n, int dtype, void* stream);
/* y[i] = h16(x[i] * (i < n_scaled ? alpha : 1)), n_scaled % 4 == 0: forward operand of a packed q | k | v projection weight whose leading q rows carry
 * the softmax scale * log2(e) (one rounding from the fp32 master; see enh_attention_forward, q_prescaled) */
int enh_cast_f32_h16_head_scaled(const float* x, enh_h16* y, int64_t n, int64_t n_scaled, float alpha, int dtype, void* stream);
/* the same for `count` equally spaced blocks (block b reads x + b*x_stride, writes y + b*y_stride; strides in elements, multiples of 4): the to_qkv weights of
 * every layer of a tower in one launch */
int enh_cast_f32_h16_head_scaled_strided(const float* x, int64_t x_stride, enh_h16* y, int64_t y_stride, int64_t n, int64_t n_scaled, float alpha,
                                         int count, int dtype, void* stream);
/* torch.optim.AdamW(lr, betas=(0.9,0.99), weight_decay=1e-4) step over one flat buffer (vitvqgan.py:160),
 * also refreshes the 16-bit operand shadow p_h16 (`dtype`) used by the GEMMs.  grad_scale multiplies g first (DDP mean / accumulation / 1 / loss scale).
 * skip_flag (optional device float): non-zero = drop this step — nothing is written (p, m, v, p_h16 unchanged): the inf / nan skip of
 * torch.cuda.amp.GradScaler.step under the reference's --use_amp (main.py:25,52); the flag comes from enh_nonfinite_flag.
 * loss_scale_dev (optional device float): g is additionally divided by it (GradScaler's unscale folded into the step). */
int enh_adamw_step(float* p, const float* g, float* m, float* v, enh_h16* p_h16, int64_t n, int step, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float grad_scale, const float* skip_flag, const float* loss_scale_dev,
                   int dtype, void* stream);
/* GradScaler.update() on the device: found_inf != 0 -> *scale *= backoff_factor, *growth_tracker = 0; otherwise ++*growth_tracker and after
 * growth_interval clean steps in a row *scale *= growth_factor (growth_interval = 0: never grow = a static scale with backoff).  *scale is kept
 * inside [1, 2^24].  torch defaults: growth 2, backoff 0.5, interval 2000, initial scale 65536. */
int enh_loss_scale_update(float* scale, const float* found_inf, int* growth_tracker, float growth_factor, float backoff_factor, int growth_interval,
                          void* stream);
/* flag[0] = 1.0f if any of x[0..n) is inf or nan, otherwise untouched (the caller zeroes it once per step): GradScaler's found-inf check over one flat
 * gradient buffer, one pass at HBM rate.  x 16-byte aligned. */
int enh_nonfinite_flag(const float* x, int64_t n, float* flag, void* stream);

/* Device-side tail of the input pipeline (enhancing/dataloader/imagenet.py:26-54: ... RandomCrop / CenterCrop -> RandomHorizontalFlip -> ToTensor):
 * src uint8 [B,Hs,Ws,3] (decoded + resized images, each in the top-left corner of its slot), meta int32 [B,3] = (y0, x0, flip) -> out f32 [B,3,R,R] in [0,1].
 * The window (y0..y0+R, x0..x0+R) must lie inside the image; exact (uint8 / 255.0f). */
int enh_crop_flip_u8(const uint8_t* src, int B, int Hs, int Ws, const int32_t* meta, int R, float* out, void* stream);
/* Resize of the reference's dataset transforms (torchvision T.Resize = PIL.Image.resize(size, BILINEAR), enhancing/dataloader/imagenet.py:31,49) for a batch
 * of decoded 8-bit RGB images of ragged sizes, bit-exact with Pillow's antialiased separable resampler (horizontal pass, then vertical, 22-bit fixed-point
 * weights).  src [B][HS][WS][3] / dst [B][HD][WD][3]: image b occupies the top-left corner of its slot.  meta [B][10] int32 per image:
 * {h_in, w_in, h_out, w_out, hb_off, hk_off, hks, vb_off, vk_off, vks}: the horizontal pass reads bounds[hb_off + 2x] = (first input column, taps) and
 * weights[hk_off + x*hks + tap]; the vertical pass likewise (tables: enhancing/dataloader/resize.py, Pillow's precompute_coeffs / normalize_coeffs_8bpc).
 * workspace: enh_resize_u8_workspace_bytes(B, HS, WD) bytes (the horizontal pass's uint8 result). */
size_t enh_resize_u8_workspace_bytes(int B, int HS, int WD);
int enh_resize_u8(const uint8_t* src, int B, int HS, int WS, const int* meta, const int* bounds, const int* weights, uint8_t* dst, int HD, int WD,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * StyleGAN2-discriminator native ops ("next" row) — drop-ins for the reference's two pybind ops
 * ------------------------------------------------------------------------------------------------ */
/* fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale) — enhancing/losses/op/fused_bias_act.cpp:17-31.
 * y = lrelu(x + bias[(i / step_b) % size_b]) * scale (grad = 0) or (x + bias) * (ref > 0 ? 1 : alpha) * scale (grad = 1: first and
 * second derivative, gated by the saved output).  f32, contiguous; bias / ref may be NULL; only act = 3 (leaky-relu) is used. */
int enh_fused_bias_act(const float* x, const float* bias, const float* ref, float* y, int64_t n, int64_t step_b, int size_b,
                       int act, int grad, float alpha, float scale, void* stream);
/* out[c] (+)= sum_{b,i} x[b,c,i] : the bias gradient `grad_input.sum(dim)` of fused_act.py:37-41 */
int enh_channel_sum_f32(const float* x, int B, int C, int64_t inner, float* out, int accumulate, void* stream);
/* upfirdn2d_op.upfirdn2d(input [major,in_h,in_w,1], kernel [kh,kw], up, down, pads) — enhancing/losses/op/upfirdn2d.cpp:17-30;
 * out [major, out_h, out_w], out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y (same for w).  f32; pads >= 0. */
int enh_upfirdn2d(const float* in, const float* kernel, float* out, int64_t major, int in_h, int in_w, int kh, int kw, int up_x,
                  int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* Equalised-lr convolution lowering (EqualConv2d, enhancing/losses/layers.py:163-185: conv2d(x, weight*scale, stride, padding),
 * k in {1,3}, stride in {1,2}) onto enh_gemm_h16 / enh_gemm_f32:  cols[(b,ho,wo), c*k*k + kh*k + kw] = x[b,c,ho*stride-pad+kh,wo*stride-pad+kw]
 * (zero outside the image) in `dtype` (ENH_DT_BF16 / ENH_DT_F16: 16-bit MFMA operands; ENH_DT_F32: the exact-f32 instrument), row stride ld = C*k*k
 * rounded up to a multiple of 8 with the pad columns zeroed.  The image is addressed as x[b*stride_b + c*stride_c + h*W + w], so [B,C,H,W] and
 * channel-major [C,B,H,W] activations are both accepted.  enh_col2im is the adjoint (the convolution's input gradient given dcols = dy^T . W, in
 * `dtype`): dx (f32) is overwritten, no atomics. */
int enh_im2col(const float* x, int64_t stride_b, int64_t stride_c, int B, int C, int H, int W, int k, int stride, int pad,
               int Ho, int Wo, void* cols, int64_t ld, int dtype, void* stream);
int enh_col2im(const void* dcols, int64_t ld, int B, int C, int H, int W, int k, int stride, int pad, int Ho, int Wo,
               float* dx, int64_t stride_b, int64_t stride_c, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolutions on channels-last bf16 activations [B,H,W,C] (no im2col tensor): EqualConv2d forward, input gradient and
 * weight gradient of the StyleGAN2 discriminator (enhancing/losses/layers.py:163-185; the three roles conv2d_gradfix differentiates
 * through, enhancing/losses/op/conv2d_gradfix.py:81-195) and the 3x3 convolutions of the LPIPS VGG16 trunk.
 * ------------------------------------------------------------------------------------------------ */
/* One description for every role.  GEMM row m = (b, y, x) of a logical grid Hm x Wm; contraction index = (tap (jy, jx), channel c);
 * operand element = src[b, y*gs + oy0 + jy*sty, x*gs + ox0 + jx*stx, c] (zero outside the source); the result row goes to pixel
 * (y*os + oph, x*os + opw) of the output tensor [B,HO,WO,N].
 *   forward  (stride s, padding p, k x k): src = x,  gs = s, oy0 = ox0 = -p, sty = stx = +1, nty = ntx = k, (Hm,Wm) = (HO,WO) = output size, os = 1
 *   dgrad, s = 1:  src = dy, gs = 1, oy0 = ox0 = +p, sty = stx = -1, nty = ntx = k, (Hm,Wm) = (HO,WO) = input size, weights packed transposed
 *   dgrad, s = 2:  one launch per output parity class (ph, pw): taps kh = kh0 + 2 jy with kh0 = (ph + p) & 1, oy0 = (ph + p - kh0) / 2, sty = -1,
 *                  Hm = (H - ph + 1) / 2, os = 2, oph = ph (same in x); a class without taps (nty*ntx = 0) writes zeros
 *   wgrad:         src = x, (Hm,Wm) = dy's spatial size, gs / oy0 / sty / nty as in forward, N = channels of dy (output fields unused) */
typedef struct enh_conv_geom {
  int B;
  int Hs, Ws, C;               /* source tensor [B,Hs,Ws,C], C % 8 == 0 */
  int Hm, Wm;                  /* logical grid enumerated as GEMM rows */
  int gs, oy0, ox0, nty, ntx, sty, stx;
  int N;                       /* output channels, N % 8 == 0 */
  int HO, WO, os, oph, opw;    /* output tensor and the placement of the logical grid in it */
} enh_conv_geom;
/* out[o, n] = epilogue( sum_{tap,c} src[...] * wt[n][tap*C + c] ), wt [N][nty*ntx*C] bf16 (enh_conv_pack_weight), o = output pixel of the row
 *   mode 0: relu(acc + bias[n])                     mode 1: (acc + add[o,n]) * (aux[o,n] > 0)   (add optional)      mode 2: acc
 *   mode 3: lrelu(acc + bias[n], slope p0) * p1     (EqualConv2d + FusedLeakyReLU, layers.py:220-264 / fused_act.py:48-76; bias optional)
 *   mode 4: acc + p0 * add[o,n]                     (StyleBlock's (out + skip) / sqrt(2), layers.py:262, folded into the skip convolution) */
int enh_conv_nhwc_h16(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                      const enh_h16* add, float p0, float p1, enh_h16* out, int dtype, void* stream);
/* The same with a caller-provided workspace of enh_conv_workspace_bytes(g) bytes (0 = none needed): grids of less than half a round of workgroups
 * (the <= 16^2 layers at 16 images) are then split over the contraction — f32 partial slabs, added in ascending order by a second kernel that
 * applies the epilogue (deterministic).  ws == NULL or too small: not split. */
size_t enh_conv_workspace_bytes(const enh_conv_geom* g);
int enh_conv_nhwc_h16_ws(const enh_h16* src, const enh_h16* wt, const enh_conv_geom* g, int mode, const float* bias, const enh_h16* aux,
                         const enh_h16* add, float p0, float p1, enh_h16* out, void* ws, size_t ws_bytes, int dtype, void* stream);
/* kernel choice of enh_conv_nhwc_h16 / enh_conv_wgrad_nhwc_h16 for A/B measurements (explicit library state, like enh_gemm_set_kernel):
 * 0 = per shape (256-row tiles when C % 64 == 0, N % 128 == 0, at least four K stages and one tile per CU; else the 128 x 128 LDS-DMA kernel when
 * C % 64 == 0; else register-staged), 1 = register-staged everywhere, 2 = never the 256-row kernels (the round-2 choice), 3 = the 256-row kernels
 * wherever the shape allows them, however few tiles */
int enh_conv_set_kernel(int variant);
/* dw[n][tap*C + c] = sum over pixels (b,y,x) of dy[b,y,x,n] * src[b, y*gs + oy0 + jy*sty, x*gs + ox0 + jx*stx, c]   (f32, overwritten).
 * The pixel axis is split over the grid; partial slabs go to `ws` (enh_conv_wgrad_workspace_bytes) and are added in a fixed order. */
size_t enh_conv_wgrad_workspace_bytes(const enh_conv_geom* g);
int enh_conv_wgrad_nhwc_h16(const enh_h16* src, const enh_h16* dy, const enh_conv_geom* g, float* dw, void* ws, size_t ws_bytes, int dtype, void* stream);
/* parameter layout [Cout][Cin][k][k] f32 (x scale) -> packed bf16 operand, taps (kh0 + jy*kstep, kw0 + jx*kstep):
 *   transposed = 0: out[co][(jy*ntx+jx)*cols_padded + ci]  (forward)      transposed = 1: out[ci][(jy*ntx+jx)*cols_padded + co]  (input gradient)
 * rows / columns beyond the real channel counts are zero.  enh_conv_unpack_wgrad: dw[co][ci][kh][kw] = scale * dwp[co][(kh*k+kw)*cin_padded + ci] */
int enh_conv_pack_weight(const float* w, int Cout, int Cin, int k, float scale, int transposed, int kh0, int kw0, int kstep, int nty, int ntx,
                         int rows_padded, int cols_padded, enh_h16* out, int dtype, void* stream);
int enh_conv_unpack_wgrad(const float* dwp, int Cout, int Cin, int cin_padded, int k, float scale, float* dw, void* stream);
/* Blur (layers.py:140-160 -> upfirdn2d with unit up / down): out[b,oy,ox,c] = sum_{i,j} w(i,j) x[b, oy+i-pad_y0, ox+j-pad_x0, c] with
 * w(i,j) = kernel[kh-1-i][kw-1-j] (flip = 0, upfirdn2d's convention) or kernel[i][j] (flip = 1, the adjoint); out is [B, H+pad_y0+pad_y1-kh+1, W+..., C];
 * pads may be negative (crop) */
/* kernel choice of enh_blur_nhwc_h16 (explicit library state): 0 = per shape (4 x 4 filters: the row-marching kernel, bit-identical results),
 * 1 = the one-row kernel everywhere */
int enh_blur_set_kernel(int variant);
int enh_blur_nhwc_h16(const enh_h16* x, const float* kernel, int B, int H, int W, int C, int kh, int kw, int pad_y0, int pad_y1, int pad_x0,
                      int pad_x1, int flip, enh_h16* out, int dtype, void* stream);
/* y = g * (ref > 0 ? 1 : slope) * scale — the first / second derivative of FusedLeakyReLU through its saved output (fused_act.py:21-45);
 * ref == NULL: y = g * scale.  n % 8 == 0 */
int enh_lrelu_gate_h16(const enh_h16* g, const enh_h16* ref, int64_t n, float slope, float scale, enh_h16* y, int dtype, void* stream);
/* Minibatch standard deviation (layers.py:358-367, stddev_feat = 1) fused with the concatenation and the channel padding of the final convolution's input:
 * x [B,HW,C] bf16 -> out [B,HW,Cp] bf16 with out[..,0..C-1] = x, out[..,C] = mean over (h,w,c) of sqrt(var over the group + 1e-8) of the sample's slot
 * (sample b is in slot b % (B/group); biased variance), out[..,C+1..] = 0.  Backward: dx [B,HW,C] from g [B,HW,Cp].  B % group == 0, C % 8 == 0, Cp % 8 == 0, Cp > C. */
int enh_minibatch_stddev_nhwc(const enh_h16* x, int B, int HW, int C, int Cp, int group, enh_h16* out, int dtype, void* stream);
int enh_minibatch_stddev_nhwc_backward(const enh_h16* x, const enh_h16* g, int B, int HW, int C, int Cp, int group, enh_h16* dx, int dtype, void* stream);
/* img [B,C,H,W] f32, C <= 8  ->  [B,H,W,8] bf16 (channels C..7 zero), and the adjoint (padding channels dropped) */
int enh_img_to_nhwc8(const float* img, int B, int C, int H, int W, enh_h16* out, int dtype, void* stream);
int enh_nhwc8_to_img(const enh_h16* src, int B, int C, int H, int W, float* img, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 "exact mode": the same contractions with fp32 operands and fixed ascending-k fp32 accumulation (no bf16 anywhere), for end-to-end
 * parity runs against the fp32 CPU oracle (SURVEY.md §8d metric 3).  Same argument meaning as the bf16 entries above; all tensors f32.
 * ------------------------------------------------------------------------------------------------ */
int enh_gemm_f32(const float* A, int64_t lda, int trans_a, const float* B, int64_t ldb, int trans_b, int64_t M, int64_t N, int64_t K,
                 const float* bias, int act, const float* aux, int64_t ldaux, const float* res, int64_t ldres, int64_t res_rows,
                 int accumulate, float* C, int64_t ldc, void* stream);
int enh_attention_forward_f32(const float* qkv, int B, int N, int H, float scale, float* out, float* lse, void* stream);
int enh_attention_backward_f32(const float* qkv, const float* out, const float* dout, const float* lse, int B, int N, int H, float scale,
                               float* dqkv, float* delta_ws, void* stream);
int enh_colsum_f32(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* stream);
/* to_patches = 1: img [B,C,H,W] -> patches [M, C*p*p] ; 0: the inverse scatter */
int enh_patch_perm_f32(const float* src, float* dst, int B, int C, int H, int W, int p, int to_patches, void* stream);
int enh_unpatchify_loss_f32(const float* pix, const float* target, int B, int C, int H, int W, int p, float w_l1, float w_l2, float* xrec,
                            double* sums, float* dpix, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LPIPS perceptual term (lpips 0.1.4, net = "vgg": pinned third-party dependency of the reference, requirements.txt:2; call sites
 * enhancing/losses/vqperceptual.py:29,43,74,115).  Activations are channels-last bf16 [B,H,W,C].
 * ------------------------------------------------------------------------------------------------ */
/* 3x3 convolution, stride 1, zero padding 1, as an implicit GEMM (no im2col tensor).  x [B,H,W,Cin], wt [Cout][9*Cin] tap-major
 * (wt[co][(kh*3+kw)*Cin + ci] = weight[co][ci][kh][kw]), out [B,H,W,Cout]; Cin, Cout multiples of 8.
 *   mode 0: out = relu(acc + bias[co])                            (torchvision vgg16.features conv + ReLU)
 *   mode 1: out = (acc + add[m,co]) * (aux[m,co] > 0)             (input gradient: wt = flipped / transposed weights, aux = the saved post-ReLU
 *                                                                  activation this gradient is for, add = optional extra gradient at that activation)
 *   mode 2: out = acc                                             (input gradient in front of a max-pool) */
int enh_conv3x3_nhwc_h16(const enh_h16* x, const enh_h16* wt, int B, int H, int W, int Cin, int Cout, const float* bias, int mode,
                         const enh_h16* aux, const enh_h16* add, enh_h16* out, int dtype, void* stream);
/* ScalingLayer + first convolution: img [B,3,H,W] f32 -> relu(conv3x3(((a img + b) - shift) / scale, w [64,3,3,3]) + bias) as [B,H,W,64] bf16, with
 * (a, b) = (2, -1) if normalize (images in [0,1]: lpips' normalize=True, = the inputs*2-1 of vqperceptual.py:43) else (1, 0); lpips ScalingLayer: shift
 * (-.030,-.088,-.188), scale (.458,.448,.450) ; and its gradient w.r.t. img given the gradient at the convolution output before the ReLU */
int enh_vgg_conv1(const float* img, const float* w, const float* bias, const float* shift, const float* scale, int normalize, int B, int H, int W, enh_h16* out, int dtype, void* stream);
int enh_vgg_conv1_backward(const enh_h16* gpre, const float* w, const float* scale, int normalize, int B, int H, int W, float* dimg, int dtype, void* stream);
/* 2x2 / stride-2 max-pool ; backward gx = (x > 0) * (gy routed to the first maximum of each window + add)  (add optional) */
int enh_maxpool2_nhwc_h16(const enh_h16* x, int B, int H, int W, int C, enh_h16* y, int dtype, void* stream);
int enh_maxpool2_nhwc_h16_backward(const enh_h16* x, const enh_h16* gy, const enh_h16* add, int B, int H, int W, int C, enh_h16* gx, int dtype, void* stream);
/* LPIPS head of one slice: feat [2B,h,w,C] (images 0..B-1 = references, B..2B-1 = reconstructions), lin [C] = the slice's 1x1 "lin" weights;
 * out[b] (+)= mean over pixels of sum_c lin[c] (f0/(|f0|+1e-10) - f1/(|f1|+1e-10))_c^2 ; val_ws [B*h*w] f32 scratch (deterministic two-stage sum).
 * Backward: gradient w.r.t. the reconstruction features only, dfeat1 [B,h,w,C] bf16, given gout[B] */
int enh_lpips_head(const enh_h16* feat, const float* lin, int B, int64_t HW, int C, float* val_ws, float* out, int accumulate, int dtype, void* stream);
int enh_lpips_head_backward(const enh_h16* feat, const float* lin, const float* gout, int B, int64_t HW, int C, enh_h16* dfeat1, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ENH_HIP_H */
