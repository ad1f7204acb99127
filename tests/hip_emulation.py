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

    def mm(a, b, M, N, K, out, trans_a=False, trans_b=False, **kw):      # the exact-f32 GEMM the fp32 instrument of conv2d_gradfix runs on
        assert not kw
        gemm(a, b, M, N, K, trans_a, trans_b, out_f32=out)

    def im2col(x, sb, sc, B, C, H, W, k, s, p, dtype=None):
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


    # ---- channels-last implicit-GEMM path (op/conv_nhwc.py): emulated from the GEOMETRY the host code builds, tap by tap ----
    def _g(geom):
        return geom if isinstance(geom, dict) else {n: getattr(geom, n) for n, _ in _C.ConvGeom._fields_}

    def _gathered(src, g):
        """[B, Hm, Wm, taps, C] f32: src[b, y*gs + oy0 + jy*sty, x*gs + ox0 + jx*stx, c], zero outside"""
        B, Hs, Ws, C = src.shape
        assert (B, Hs, Ws, C) == (g["B"], g["Hs"], g["Ws"], g["C"]) and C % 8 == 0
        out = torch.zeros(B, g["Hm"], g["Wm"], g["nty"] * g["ntx"], C)
        ys, xs = torch.arange(g["Hm"]) * g["gs"], torch.arange(g["Wm"]) * g["gs"]
        for jy in range(g["nty"]):
            for jx in range(g["ntx"]):
                sy, sx = ys + g["oy0"] + jy * g["sty"], xs + g["ox0"] + jx * g["stx"]
                oky, okx = (sy >= 0) & (sy < Hs), (sx >= 0) & (sx < Ws)
                v = src.float()[:, sy.clamp(0, Hs - 1)][:, :, sx.clamp(0, Ws - 1)]
                out[:, :, :, jy * g["ntx"] + jx] = v * (oky[:, None] & okx[None, :])[None, :, :, None]
        return out

    def conv_nhwc(src, wt, geom, mode, bias=None, aux=None, add=None, p0=0.0, p1=1.0, out=None):
        g = _g(geom)
        taps = g["nty"] * g["ntx"]
        if out is None:
            out = torch.full((g["B"], g["HO"], g["WO"], g["N"]), float("nan"), dtype=low)
        A = _gathered(src, g).reshape(g["B"], g["Hm"], g["Wm"], taps * g["C"])
        acc = A @ wt[:, :taps * g["C"]].float().t() if taps else torch.zeros(g["B"], g["Hm"], g["Wm"], g["N"])
        assert wt.shape[0] == g["N"]
        sl = (slice(None), slice(g["oph"], g["oph"] + (g["Hm"] - 1) * g["os"] + 1, g["os"]), slice(g["opw"], g["opw"] + (g["Wm"] - 1) * g["os"] + 1, g["os"]))
        if mode == 0:
            acc = torch.relu(acc + bias)
        elif mode == 1:
            acc = (acc + (add[sl].float() if add is not None else 0)) * (aux[sl] > 0)
        elif mode == 3:
            v = acc + bias if bias is not None else acc
            acc = torch.where(v > 0, v, v * p0) * p1
        elif mode == 4:
            acc = acc + p0 * add[sl].float()
        out[sl] = acc.to(low)
        return out

    def conv_wgrad_nhwc(src, dy, geom):
        g = _g(geom)
        A = _gathered(src, g).reshape(-1, g["nty"] * g["ntx"] * g["C"])
        assert dy.shape == (g["B"], g["Hm"], g["Wm"], g["N"])
        return dy.float().reshape(-1, g["N"]).t() @ A

    def conv_pack_weight(w, scale, transposed, kh0, kw0, kstep, nty, ntx, rows_padded, cols_padded, dtype=None):
        Cout, Cin, k, _ = w.shape
        out = torch.zeros(rows_padded, max(nty * ntx * cols_padded, 8))
        if nty * ntx:
            sel = (w * scale)[:, :, kh0::kstep, kw0::kstep][:, :, :nty, :ntx]          # [Cout, Cin, nty, ntx]
            m = sel.permute(1, 2, 3, 0) if transposed else sel.permute(0, 2, 3, 1)    # [rows, nty, ntx, cols]
            o = torch.zeros(rows_padded, nty, ntx, cols_padded)
            o[:m.shape[0], :, :, :m.shape[3]] = m
            out = o.reshape(rows_padded, -1)
        return out.to(low)

    def conv_unpack_wgrad(dwp, Cout, Cin, cin_padded, k, scale):
        return scale * dwp.reshape(Cout, k, k, cin_padded)[:, :, :, :Cin].permute(0, 3, 1, 2).contiguous()

    def blur_nhwc(x, kernel, pad0, pad1, flip):
        assert x.shape[3] % 8 == 0
        kq = kernel if not flip else torch.flip(kernel, [0, 1])       # DO.upfirdn2d applies the (flipped) kernel of the reference's convention
        xp = x.float().permute(0, 3, 1, 2)
        neg0, neg1 = max(-pad0, 0), max(-pad1, 0)
        if neg0 or neg1:
            xp = xp[:, :, neg0:xp.shape[2] - neg1, neg0:xp.shape[3] - neg1]
        y = DO.upfirdn2d(xp, kq, pad=(max(pad0, 0), max(pad1, 0)))
        return y.permute(0, 2, 3, 1).contiguous().to(low)

    def lrelu_gate(g_, ref, slope, scale):
        assert g_.numel() % 8 == 0
        v = g_.float()
        if ref is not None:
            v = v * torch.where(ref > 0, torch.ones_like(v), torch.full_like(v, slope))
        return (v * scale).to(low)

    def img_to_nhwc8(img, dtype=None):
        B, C, H, W = img.shape
        out = torch.zeros(B, H, W, 8)
        out[..., :C] = img.permute(0, 2, 3, 1)
        return out.to(low)

    def nhwc8_to_img(src, C):
        return src.float()[..., :C].permute(0, 3, 1, 2).contiguous()

    def colsum_nhwc(x):
        return x.float().reshape(-1, x.shape[-1]).sum(0)

    def minibatch_stddev_nhwc(x, group, Cp):
        from enhancing.losses.op.conv_nhwc import _stddev_torch
        return _stddev_torch(x.float(), group, Cp).to(low)

    def minibatch_stddev_nhwc_backward(x, g, group):
        from enhancing.losses.op.conv_nhwc import _stddev_torch
        with torch.enable_grad():
            x32 = x.detach().float().requires_grad_(True)
            dx, = torch.autograd.grad(_stddev_torch(x32, group, g.shape[3]), x32, g.float())
        return dx.to(low)

    for name, fn in dict(gemm=gemm, mm=mm, cast_bf16=cast_bf16, im2col=im2col, col2im=col2im, fused_bias_act=fused_bias_act,
                         channel_sum=channel_sum, upfirdn2d=upfirdn2d, conv_nhwc=conv_nhwc, conv_wgrad_nhwc=conv_wgrad_nhwc,
                         conv_pack_weight=conv_pack_weight, conv_unpack_wgrad=conv_unpack_wgrad, blur_nhwc=blur_nhwc, lrelu_gate=lrelu_gate,
                         img_to_nhwc8=img_to_nhwc8, nhwc8_to_img=nhwc8_to_img, colsum_nhwc=colsum_nhwc,
                         minibatch_stddev_nhwc=minibatch_stddev_nhwc, minibatch_stddev_nhwc_backward=minibatch_stddev_nhwc_backward).items():
        monkeypatch.setattr(_C, name, fn)
    if exact:
        class _Torch:   # conv2d_gradfix allocates its low-precision containers as torch.bfloat16
            bfloat16 = torch.float32

            def __getattr__(self, n):
                return getattr(torch, n)
        monkeypatch.setattr(cg, "torch", _Torch())
        monkeypatch.setattr(cg, "_OPERAND", torch.float32)      # = conv2d_gradfix.operand_dtype("fp32"): the fp32 instrument's code path
