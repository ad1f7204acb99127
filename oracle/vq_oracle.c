/*
 * vq_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the quantizer forward of thuanz123/enhancing-transformers
 *   VectorQuantizer.quantize   enhancing/modules/stage1/quantizers.py:74-92
 *   BaseQuantizer.forward      enhancing/modules/stage1/quantizers.py:38-63
 * using the *explicit summation orders* of the arithmetic contract in include/enh_hip.h, so that the
 * HIP kernel can be compared BIT-FOR-BIT (indices, z_q bits) while the plain-PyTorch oracle
 * (oracle/vitvq_oracle.py, pinned against the reference itself) bounds how far those orders can move
 * an argmin: only across fp32 near-ties.
 *
 *   S(x)   = chain(x[0..15]) + chain(x[16..31]),  chain = ascending fmaf chain from 0.0f
 *   n(x)   = x / max(sqrtf(S(x)), 1e-12f)                                  (quantizers.py:24)
 *   dot    = fmaf chain in the order k = 0,16,1,17,...,15,31
 *   d_k    = (S(zn) + S(en_k)) - 2*dot_k                                    (quantizers.py:78-80)
 *   idx    = first k with minimal d_k                                       (quantizers.py:82)
 * Build: gcc -O2 -fopenmp -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D 32

static float chain16_sq(const float* x) {
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s = fmaf(x[j], x[j], s);
  return s;
}
static float S32(const float* x) { return chain16_sq(x) + chain16_sq(x + 16); }

static void normalise(const float* x, float* out, int use_norm) {
  if (!use_norm) { memcpy(out, x, D * sizeof(float)); return; }
  float den = fmaxf(sqrtf(S32(x)), 1e-12f);
  for (int j = 0; j < D; ++j) out[j] = x[j] / den;
}

/* returns 0 on success.  zq_out [M,32] straight-through value z + (sum_i en_i - z); idx_out [M,depth];
 * loss_out[0] = mean_i( beta*m_i + m_i ), m_i = mean((en_i - zn_i)^2) with the sums carried in double. */
int vq_oracle_forward(const float* z, const float* E, int64_t M, int K, float beta, int depth, int use_norm,
                      float* zq_out, int64_t* idx_out, float* loss_out, int nthreads) {
  float* en = (float*)malloc((size_t)K * D * sizeof(float));
  float* ee = (float*)malloc((size_t)K * sizeof(float));
  if (!en || !ee) return -1;
  for (int k = 0; k < K; ++k) {
    normalise(E + (size_t)k * D, en + (size_t)k * D, use_norm);
    ee[k] = S32(en + (size_t)k * D);
  }
  double* lsum = (double*)calloc((size_t)depth, sizeof(double));
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    double* lloc = (double*)calloc((size_t)depth, sizeof(double));
#pragma omp for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      float r[D], zn[D], acc[D];
      memcpy(r, z + (size_t)m * D, sizeof(r));
      memset(acc, 0, sizeof(acc));
      for (int i = 0; i < depth; ++i) {
        normalise(r, zn, use_norm);
        const float zz = S32(zn);
        float best_d = INFINITY;
        int best_i = 0;
        for (int k = 0; k < K; ++k) {
          const float* e = en + (size_t)k * D;
          float dot = 0.f;
          for (int kk = 0; kk < 16; ++kk) {
            dot = fmaf(e[kk], zn[kk], dot);
            dot = fmaf(e[16 + kk], zn[16 + kk], dot);
          }
          const float d = (zz + ee[k]) - 2.0f * dot;
          if (d < best_d) { best_d = d; best_i = k; }
        }
        idx_out[(size_t)m * depth + i] = best_i;
        const float* e = en + (size_t)best_i * D;
        for (int j = 0; j < D; ++j) {
          const float df = e[j] - zn[j];
          lloc[i] += (double)(df * df);
          acc[j] = acc[j] + e[j];
          r[j] = r[j] - e[j];
        }
      }
      const float* z0 = z + (size_t)m * D;
      for (int j = 0; j < D; ++j) zq_out[(size_t)m * D + j] = z0[j] + (acc[j] - z0[j]);
    }
#pragma omp critical
    for (int i = 0; i < depth; ++i) lsum[i] += lloc[i];
    free(lloc);
  }
  float total = 0.f;
  for (int i = 0; i < depth; ++i) {
    const float mi = (float)(lsum[i] / ((double)M * D));
    total += beta * mi + mi;
  }
  loss_out[0] = total / (float)depth;
  free(lsum); free(en); free(ee);
  return 0;
}

int vq_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
