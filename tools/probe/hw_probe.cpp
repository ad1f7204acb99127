// Hardware-semantics probe for gfx950 (run once on the GPU box; torch-free).
// Pins down: ds_read_b64_tr_b16 shuffle, MFMA fragment layouts, global_load_lds
// destination layout, permlane32_swap, f32-MFMA == fmaf chain bit-exactness.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- P1: tr-read ----
__global__ void k_trread(const int* __restrict__ addr_elems, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int a = addr_elems[threadIdx.x];
  auto p = (LDS_AS bf16x4_t*)(lds + a);
  bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
  uint16_t r[4]; memcpy(r, &v, 8);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}

// ---- P2: MFMA layout checks ----
// 16x16x32 bf16: assumed A: lane l row l&15, k=(l>>4)*8+j ; B: k=(l>>4)*8+j, col l&15 ; D: col=l&15,row=(l>>4)*4+r
__global__ void k_mfma16(const uint16_t* A /*16x32 row-major*/, const uint16_t* B /*32x16 row-major [k][n]*/, float* D /*16x16*/) {
  int l = threadIdx.x;
  short8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)A[(l & 15) * 32 + (l >> 4) * 8 + j]; b[j] = (short)B[((l >> 4) * 8 + j) * 16 + (l & 15)]; }
  f32x4 c = {0, 0, 0, 0};
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// 32x32x16 bf16: A: row l&31, k=(l>>5)*8+j ; B: k=(l>>5)*8+j, col l&31 ; D: col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5)
__global__ void k_mfma32(const uint16_t* A /*32x16*/, const uint16_t* B /*16x32 [k][n]*/, float* D /*32x32*/) {
  int l = threadIdx.x;
  short8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)A[(l & 31) * 16 + (l >> 5) * 8 + j]; b[j] = (short)B[((l >> 5) * 8 + j) * 32 + (l & 31)]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// f32 16x16x4: A[l&15][k=l>>4], B[k=l>>4][l&15]; chained over K=32 (8 instrs), k-order ascending
__global__ void k_mfma16f32(const float* A /*16x32*/, const float* B /*32x16 [k][n]*/, float* D) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  for (int kk = 0; kk < 8; ++kk) {
    float a = A[(l & 15) * 32 + kk * 4 + (l >> 4)];
    float b = B[(kk * 4 + (l >> 4)) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// f32 32x32x2: A[l&31][k=l>>5], B[k=l>>5][l&31]; chained over K=32 (16 instrs)
__global__ void k_mfma32f32(const float* A /*32x32*/, const float* B /*32x32 [k][n]*/, float* D) {
  int l = threadIdx.x;
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  for (int kk = 0; kk < 16; ++kk) {
    float a = A[(l & 31) * 32 + kk * 2 + (l >> 5)];
    float b = B[(kk * 2 + (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ---- P3: global_load_lds (16 B) ----
__global__ void k_glds(const uint32_t* __restrict__ src, const int* __restrict__ src_off_dw, uint32_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t* g = src + src_off_dw[threadIdx.x];
  // LDS base: wave-uniform (lds + 256 dwords); hardware adds lane*16 B
  __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)(lds + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

// ---- P4: permlane32_swap ----
__global__ void k_permswap(uint32_t* out) {
  uint32_t a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz mem=%.1f GB lds/blk=%zu\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1e9, prop.sharedMemPerBlock);
  // P1
  {
    int h_addr[64]; uint16_t h_out[256]; int* d_addr; uint16_t* d_out;
    CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 512));
    for (int variant = 0; variant < 3; ++variant) {
      for (int l = 0; l < 64; ++l) {
        if (variant == 0) h_addr[l] = l * 4;                          // linear: lane l reads elems 4l..4l+3
        else if (variant == 1) h_addr[l] = ((l * 37) % 512) * 4 + 1024; // scattered (8B aligned)
        else h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;                // lane-in-group strided rows
      }
      CK(hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice));
      k_trread<<<1, 64>>>(d_addr, d_out); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost));
      printf("P1 trread variant %d (result as 'srcLane.srcElem' per lane, elem j):\n", variant);
      // decode: find which lane/elem supplied each value
      int hyp_ok = 1;
      for (int l = 0; l < 64; ++l) {
        printf("  lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
          int v = h_out[l * 4 + j]; int sl = -1, se = -1;
          for (int s = 0; s < 64 && sl < 0; ++s) for (int e = 0; e < 4; ++e) if (h_addr[s] + e == v) { sl = s; se = e; break; }
          printf(" %2d.%d", sl, se);
          // hypothesis: within 16-lane group g, i=l&15: src lane = g*16 + j*4 + i/4, src elem = i%4
          int g = l >> 4, i = l & 15;
          if (!(sl == g * 16 + j * 4 + i / 4 && se == i % 4)) hyp_ok = 0;
        }
        printf("\n");
      }
      printf("P1 variant %d hypothesis(src lane=g*16+j*4+i/4, elem=i%%4): %s\n", variant, hyp_ok ? "OK" : "MISMATCH");
    }
  }
  // P2
  {
    srand(7);
    auto run = [&](const char* name, int M, int N, int K, int which) {
      std::vector<float> Af(M * K), Bf(K * N), Dref(M * N), D(M * N);
      for (auto& x : Af) x = (float)((rand() % 17) - 8);
      for (auto& x : Bf) x = (float)((rand() % 13) - 6) * 0.5f;
      for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(Af[m * K + k], Bf[k * N + n], s); Dref[m * N + n] = s; }
      float* dD; CK(hipMalloc(&dD, M * N * 4));
      if (which < 2) {
        std::vector<uint16_t> Ah(M * K), Bh(K * N);
        for (int i = 0; i < M * K; ++i) Ah[i] = f2bf(Af[i]);
        for (int i = 0; i < K * N; ++i) Bh[i] = f2bf(Bf[i]);
        uint16_t *dA, *dB; CK(hipMalloc(&dA, M * K * 2)); CK(hipMalloc(&dB, K * N * 2));
        CK(hipMemcpy(dA, Ah.data(), M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bh.data(), K * N * 2, hipMemcpyHostToDevice));
        if (which == 0) k_mfma16<<<1, 64>>>(dA, dB, dD); else k_mfma32<<<1, 64>>>(dA, dB, dD);
      } else {
        float *dA, *dB; CK(hipMalloc(&dA, M * K * 4)); CK(hipMalloc(&dB, K * N * 4));
        CK(hipMemcpy(dA, Af.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bf.data(), K * N * 4, hipMemcpyHostToDevice));
        if (which == 2) k_mfma16f32<<<1, 64>>>(dA, dB, dD); else k_mfma32f32<<<1, 64>>>(dA, dB, dD);
      }
      CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost));
      double me = 0; for (int i = 0; i < M * N; ++i) me = fmax(me, fabs(D[i] - Dref[i]));
      printf("P2 %s layout check: max abs err = %g -> %s\n", name, me, me == 0 ? "OK" : "MISMATCH");
      if (me != 0) { printf("  D[0..3][0..3] got/ref:"); for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) printf(" %g/%g", D[m * N + n], Dref[m * N + n]); printf("\n"); }
    };
    run("mfma_f32_16x16x32_bf16", 16, 16, 32, 0);
    run("mfma_f32_32x32x16_bf16", 32, 32, 16, 1);
    run("mfma_f32_16x16x4f32 (K=32 chain)", 16, 16, 32, 2);
    run("mfma_f32_32x32x2f32 (K=32 chain)", 32, 32, 32, 3);
    // P5: bit-exactness of f32 MFMA vs fmaf chain with random (non-integer) floats
    for (int which = 2; which <= 3; ++which) {
      int M = which == 2 ? 16 : 32, N = M, K = 32;
      std::vector<float> Af(M * K), Bf(K * N), D(M * N);
      for (auto& x : Af) x = (float)(rand() % 2000001) / 1000000.f - 1.f;
      for (auto& x : Bf) x = (float)(rand() % 2000001) / 1000000.f - 1.f;
      float *dA, *dB, *dD; CK(hipMalloc(&dA, M * K * 4)); CK(hipMalloc(&dB, K * N * 4)); CK(hipMalloc(&dD, M * N * 4));
      CK(hipMemcpy(dA, Af.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bf.data(), K * N * 4, hipMemcpyHostToDevice));
      if (which == 2) k_mfma16f32<<<1, 64>>>(dA, dB, dD); else k_mfma32f32<<<1, 64>>>(dA, dB, dD);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost));
      int nbit = 0; for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(Af[m * K + k], Bf[k * N + n], s); if (memcmp(&s, &D[m * N + n], 4) != 0) ++nbit; }
      printf("P5 f32 MFMA (%s) vs ascending-k fmaf chain: %d / %d elements differ bitwise\n", which == 2 ? "16x16x4" : "32x32x2", nbit, M * N);
    }
  }
  // P3
  {
    std::vector<uint32_t> src(4096); for (int i = 0; i < 4096; ++i) src[i] = i;
    uint32_t *dsrc, *dout; int* doff; int hoff[64]; uint32_t hout[1024];
    CK(hipMalloc(&dsrc, 16384)); CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&doff, 256));
    CK(hipMemcpy(dsrc, src.data(), 16384, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 2; ++variant) {
      for (int l = 0; l < 64; ++l) hoff[l] = variant == 0 ? l * 4 : ((l * 29) % 64) * 4 + 1024;
      CK(hipMemcpy(doff, hoff, 256, hipMemcpyHostToDevice));
      k_glds<<<1, 64>>>(dsrc, doff, dout); CK(hipDeviceSynchronize());
      CK(hipMemcpy(hout, dout, 4096, hipMemcpyDeviceToHost));
      int ok = 1, touched_outside = 0;
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (hout[256 + l * 4 + e] != (uint32_t)(hoff[l] + e)) ok = 0;
      for (int i = 0; i < 1024; ++i) if ((i < 256 || i >= 512) && hout[i] != 0xdeadbeefu) touched_outside++;
      printf("P3 global_load_lds b128 variant %d: dest = base + lane*16B with per-lane source: %s (outside touched: %d)\n", variant, ok ? "OK" : "MISMATCH", touched_outside);
      if (!ok) { printf("  lds[256..271]:"); for (int i = 256; i < 272; ++i) printf(" %u", hout[i]); printf("\n"); }
    }
  }
  // P4
  {
    uint32_t* dout; uint32_t hout[128]; CK(hipMalloc(&dout, 512));
    k_permswap<<<1, 64>>>(dout); CK(hipDeviceSynchronize()); CK(hipMemcpy(hout, dout, 512, hipMemcpyDeviceToHost));
    // expected (guide T21): lanes 32-63 of vdst(a) swap with lanes 0-31 of src(b)
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      uint32_t ea = l < 32 ? 1000 + l : 2000 + (l - 32); // r[0] (new a): low half keeps a; high half gets b's low half
      uint32_t eb = l < 32 ? 1000 + (l + 32) : 2000 + l; // r[1] (new b): low half gets a's high half; high half keeps b
      if (hout[l * 2] != ea || hout[l * 2 + 1] != eb) ok = 0;
    }
    printf("P4 permlane32_swap hypothesis: %s ; lane0=(%u,%u) lane31=(%u,%u) lane32=(%u,%u) lane63=(%u,%u)\n", ok ? "OK" : "MISMATCH", hout[0], hout[1], hout[62], hout[63], hout[64], hout[65], hout[126], hout[127]);
  }
  printf("PROBE DONE\n");
  return 0;
}
