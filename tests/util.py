import torch


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    """relative Frobenius error ||a-b|| / ||b|| in float64 (the parity metric of SURVEY.md §8d)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def bf16r(x: torch.Tensor) -> torch.Tensor:
    """round to bf16 and back: test inputs are made bf16-representable so that the fp32 oracle and the bf16
    MFMA path consume IDENTICAL operand values."""
    return x.to(torch.bfloat16).to(torch.float32)


def bf16_floor(ref: torch.Tensor) -> float:
    """relative Frobenius error of merely ROUNDING the exact result to bfloat16 — the best any bf16-stored output can do
    (1.66e-3 for Gaussian-like data).  bf16-output kernels are required to stay within 15 % of it."""
    return rel(bf16r(ref.float()), ref)


def disc_case(golden_npz):
    """the discriminator golden case of oracle/make_golden_disc.py: module (CPU parameters regenerated from the seed), real, fake"""
    from enhancing.losses.layers import StyleDiscriminator
    size, B = int(golden_npz["size"]), int(golden_npz["B"])
    torch.manual_seed(int(golden_npz["param_seed"]))
    D = StyleDiscriminator(size=size)
    with torch.no_grad():
        for n, p in D.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape))
    g = torch.Generator().manual_seed(int(golden_npz["data_seed"]))
    real = torch.rand(B, 3, size, size, generator=g)
    fake = (real + 0.1 * torch.randn(B, 3, size, size, generator=g)).clamp(0, 1)
    return D, real, fake
