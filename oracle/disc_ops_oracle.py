"""CPU ORACLE (test infrastructure) for the discriminator's two native ops: plain-PyTorch restatements of what the reference's
CUDA kernels compute (reference enhancing/losses/op/fused_bias_act_kernel.cu:40-61, upfirdn2d.py:168-209), differentiable to any
order through torch.autograd.  Pinned in oracle/make_golden.py against the reference's own pure-Python fallbacks
(fused_act.py:111-122 — which hard-code slope 0.2, so the pin uses 0.2 — and upfirdn2d_native, upfirdn2d.py:168-209)."""
import torch
import torch.nn.functional as F


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if bias is not None:
        x = x + bias.view(1, bias.shape[0], *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope=negative_slope) * scale


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x [B,C,H,W]; zero-insertion upsample, pad, correlate with the FLIPPED kernel, stride-downsample."""
    up_x, up_y = (up, up) if isinstance(up, int) else up
    down_x, down_y = (down, down) if isinstance(down, int) else down
    px0, px1, py0, py1 = (pad[0], pad[1], pad[0], pad[1]) if len(pad) == 2 else pad
    B, C, H, W = x.shape
    u = x.new_zeros(B, C, H * up_y, W * up_x)
    u[:, :, ::up_y, ::up_x] = x
    u = F.pad(u, [px0, px1, py0, py1])
    w = torch.flip(kernel, [0, 1])[None, None].to(x)
    out = F.conv2d(u.reshape(B * C, 1, u.shape[2], u.shape[3]), w)
    return out[:, :, ::down_y, ::down_x].reshape(B, C, (out.shape[2] + down_y - 1) // down_y, (out.shape[3] + down_x - 1) // down_x)
