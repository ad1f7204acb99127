// resize.hip — the input pipeline's Resize on the device: PIL.Image.resize(size, BILINEAR) of 8-bit RGB images, bit-exact.
// Replaces torchvision's T.Resize in the reference's dataset transforms (enhancing/dataloader/imagenet.py:31,49), which runs Pillow's antialiased separable
// resampler in the DataLoader workers.  Here the workers only decode; the decoded uint8 images (ragged sizes, padded into common slots) are resized
// by two integer passes that follow Pillow's src/libImaging/Resample.c exactly: horizontal pass first (ImagingResampleHorizontal_8bpc), its uint8
// result feeds the vertical pass (ImagingResampleVertical_8bpc); per output index a window [first, first + n) and 22-bit fixed-point weights
// (precompute_coeffs / normalize_coeffs_8bpc — computed on the host in double, enhancing/dataloader/resize.py), ss = 2^21 + sum pixel * k,
// out = clip8(ss >> 22).  HBM-bound byte work: one thread per output pixel (3 channels), coalesced along x.
#include "common.h"

struct ResizeImage {     // per image of the batch
  int hin, win, hout, wout;
  int hb_off, hk_off, hks;     // horizontal tables: bounds at hb[hb_off + 2 x], weights at hk[hk_off + x * hks]
  int vb_off, vk_off, vks;     // vertical tables
};

#define RS_PRECISION_BITS 22

__device__ __forceinline__ uint8_t rs_clip8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// pass 1: src [B][HS][WS][3] -> tmp [B][HS][WT][3] (rows < hin, columns < wout of each image)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int HS, int WS, const ResizeImage* __restrict__ meta,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, uint8_t* __restrict__ tmp, int WT) {
  const int b = blockIdx.z, y = blockIdx.y;
  const ResizeImage im = meta[b];
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (y >= im.hin || xo >= im.wout) return;
  const int x0 = bounds[im.hb_off + 2 * xo], n = bounds[im.hb_off + 2 * xo + 1];
  const int* k = kk + im.hk_off + (int64_t)xo * im.hks;
  const uint8_t* row = src + (((int64_t)b * HS + y) * WS + x0) * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int w = k[x];
    s0 += row[3 * x] * w; s1 += row[3 * x + 1] * w; s2 += row[3 * x + 2] * w;
  }
  uint8_t* o = tmp + (((int64_t)b * HS + y) * WT + xo) * 3;
  o[0] = rs_clip8(s0 >> RS_PRECISION_BITS); o[1] = rs_clip8(s1 >> RS_PRECISION_BITS); o[2] = rs_clip8(s2 >> RS_PRECISION_BITS);
}

// pass 2: tmp [B][HS][WT][3] -> dst [B][HD][WD][3] (rows < hout, columns < wout; the rest of a slot is left untouched)
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, int HS, int WT, const ResizeImage* __restrict__ meta,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, uint8_t* __restrict__ dst, int HD, int WD) {
  const int b = blockIdx.z, yo = blockIdx.y;
  const ResizeImage im = meta[b];
  const int xo = blockIdx.x * 256 + threadIdx.x;
  if (yo >= im.hout || xo >= im.wout) return;
  const int y0 = bounds[im.vb_off + 2 * yo], n = bounds[im.vb_off + 2 * yo + 1];
  const int* k = kk + im.vk_off + (int64_t)yo * im.vks;
  const uint8_t* col = tmp + (((int64_t)b * HS + y0) * WT + xo) * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int w = k[y];
    const uint8_t* p = col + (int64_t)y * WT * 3;
    s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
  }
  uint8_t* o = dst + (((int64_t)b * HD + yo) * WD + xo) * 3;
  o[0] = rs_clip8(s0 >> RS_PRECISION_BITS); o[1] = rs_clip8(s1 >> RS_PRECISION_BITS); o[2] = rs_clip8(s2 >> RS_PRECISION_BITS);
}

extern "C" size_t enh_resize_u8_workspace_bytes(int B, int HS, int WD) { return (size_t)B * (size_t)HS * (size_t)WD * 3; }

extern "C" int enh_resize_u8(const uint8_t* src, int B, int HS, int WS, const int* meta /* [B][10] */, const int* bounds, const int* weights, uint8_t* dst,
                             int HD, int WD, void* workspace, size_t workspace_bytes, void* stream) {
  ENH_REQUIRE(src && meta && bounds && weights && dst && workspace, ENH_E_BADARG, "enh_resize_u8: null pointer");
  ENH_REQUIRE(B > 0 && HS > 0 && WS > 0 && HD > 0 && WD > 0 && B < 65536 && HS < 65536 && HD < 65536, ENH_E_SHAPE, "enh_resize_u8: bad sizes");
  ENH_REQUIRE(workspace_bytes >= enh_resize_u8_workspace_bytes(B, HS, WD), ENH_E_WORKSPACE, "enh_resize_u8: workspace of %zu bytes needed, %zu given",
              enh_resize_u8_workspace_bytes(B, HS, WD), workspace_bytes);
  static_assert(sizeof(ResizeImage) == 10 * sizeof(int), "meta layout");
  hipStream_t s = (hipStream_t)stream;
  const ResizeImage* m = reinterpret_cast<const ResizeImage*>(meta);
  resize_h_kernel<<<dim3((unsigned)((WD + 255) / 256), (unsigned)HS, (unsigned)B), 256, 0, s>>>(src, HS, WS, m, bounds, weights, (uint8_t*)workspace, WD);
  resize_v_kernel<<<dim3((unsigned)((WD + 255) / 256), (unsigned)HD, (unsigned)B), 256, 0, s>>>((const uint8_t*)workspace, HS, WD, m, bounds, weights, dst, HD, WD);
  return enh_check_launch("enh_resize_u8");
}
