"""TEST-ONLY torch stand-ins for the HIP primitives the discriminator is assembled from, so that its autograd wiring (every first- and
second-order formula in losses/op/conv2d_gradfix.py, the channel-major layout handling in losses/layers.py) can be checked on a machine
without a GPU.  With exact=True every bf16 container is replaced by float32, which makes the assembled discriminator comparable with
the fp32 oracle to ~1e-6.  The product path never imports this file; without libenh_hip.so and a GPU it raises."""
import torch
import torch.nn.functional as F

import disc_ops_oracle as DO


def install(monkeypatch, exact: bool = True):
    from enhancing import _C
    from enhancing.losses.op import conv2d_gradfix as cg
    low = torch.float32 if exact else torch.bfloat16

    def gemm(a, b, M, N, K, trans_a=False, trans_b=False, bias=None, act=0, aux=None, res=None, res_rows=0, accumulate=False,
             out_f32=None, out_bf16=None, lda=None, ldb=None, ldc=None):
        A = a.float().t() if trans_a else a.float()
        B = b.float().t() if trans_b else b.float()
        assert A.shape == (M, K) and B.shape == (N, K), (A.shape, B.shape, M, N, K)
        assert M % 8 == 0 and N % 8 == 0 and K % 8 == 0
        C = A @ B.t()
        if out_f32 is not None:
            out_f32.copy_(out_f32 + C if accumulate else C)
        else:
            out_bf16.copy_(C.to(low))

    def cast_bf16(x, y):
        y.copy_(x.to(low))

    def _idx(B, C, H, W, sb, sc):
        ar = torch.arange
        return ar(B)[:, None, None, None] * sb + ar(C)[None, :, None, None] * sc + ar(H)[None, None, :, None] * W + ar(W)[None, None, None, :]

    def im2col(x, sb, sc, B, C, H, W, k, s, p):
        img = x.reshape(-1)[_idx(B, C, H, W, sb, sc)]
        Ho, Wo = _C.conv_out_size(H, k, s, p), _C.conv_out_size(W, k, s, p)
        cols = F.unfold(img, k, padding=p, stride=s).permute(0, 2, 1).reshape(B * Ho * Wo, C * k * k)
        out = torch.zeros(B * Ho * Wo, (C * k * k + 7) // 8 * 8, dtype=low)
        out[:, :C * k * k] = cols.to(low)
        return out

    def col2im(dcols, B, C, H, W, k, s, p, out, sb, sc):
        Ho, Wo = _C.conv_out_size(H, k, s, p), _C.conv_out_size(W, k, s, p)
        d = dcols[:, :C * k * k].float().reshape(B, Ho * Wo, C * k * k).permute(0, 2, 1)
        out.reshape(-1)[_idx(B, C, H, W, sb, sc).reshape(-1)] = F.fold(d, (H, W), k, padding=p, stride=s).reshape(-1)
        return out

    def fused_bias_act(x, bias, ref, grad, alpha, scale):
        step = 1
        for d in x.shape[2:]:
            step *= d
        v = x
        if bias is not None and bias.numel():
            v = x + bias[(torch.arange(x.numel()).reshape(x.shape) // step) % bias.numel()]
        if grad == 0:
            return torch.where(v > 0, v, alpha * v) * scale
        return v * torch.where(ref > 0, torch.ones_like(v), torch.full_like(v, alpha)) * scale

    def channel_sum(x):
        return x.sum(dim=[d for d in range(x.ndim) if d != 1])

    def upfirdn2d(x, kernel, ux, uy, dx, dy, p0, p1, p2, p3):
        return DO.upfirdn2d(x[None], kernel, (ux, uy), (dx, dy), (p0, p1, p2, p3))[0]

    for name, fn in dict(gemm=gemm, cast_bf16=cast_bf16, im2col=im2col, col2im=col2im, fused_bias_act=fused_bias_act,
                         channel_sum=channel_sum, upfirdn2d=upfirdn2d).items():
        monkeypatch.setattr(_C, name, fn)
    if exact:
        class _Torch:   # conv2d_gradfix allocates its low-precision containers as torch.bfloat16
            bfloat16 = torch.float32

            def __getattr__(self, n):
                return getattr(torch, n)
        monkeypatch.setattr(cg, "torch", _Torch())
