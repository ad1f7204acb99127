"""Pins the seed-for-seed initialisation against the REAL reference and writes tests/golden/init_seed.npz.

Run in the build container (needs /root/reference):  python oracle/make_golden_init.py
Constructs the reference's own ViTEncoder / ViTDecoder / VectorQuantizer (+ the two stock nn.Linear of vitvqgan.py:38-39) in the
reference's construction order (vitvqgan.py:34-39) under torch.manual_seed(seed) and stores, per tensor, its float64 sum, its sum of
squares and its first 4 values — enough to tell a different RNG stream apart with certainty, small enough to commit.
TEST INFRASTRUCTURE ONLY."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_loader as RL  # noqa: E402

CASES = {"tiny": dict(image_size=64, patch_size=8, enc=dict(dim=128, depth=2, heads=2, mlp_dim=256), dec=dict(dim=192, depth=3, heads=4, mlp_dim=320),
                      q=dict(embed_dim=32, n_embed=512)),
         "small": dict(image_size=256, patch_size=8, enc=dict(dim=512, depth=8, heads=8, mlp_dim=2048), dec=dict(dim=512, depth=8, heads=8, mlp_dim=2048),
                       q=dict(embed_dim=32, n_embed=8192))}


def build(L, Q, case, seed):
    """the construction order of the reference's ViTVQ.__init__ (vitvqgan.py:34-39), loss excluded"""
    c = CASES[case]
    torch.manual_seed(seed)
    mods = dict(encoder=L.ViTEncoder(image_size=c["image_size"], patch_size=c["patch_size"], **c["enc"]),
                decoder=L.ViTDecoder(image_size=c["image_size"], patch_size=c["patch_size"], **c["dec"]),
                quantizer=Q.VectorQuantizer(**c["q"]),
                pre_quant=torch.nn.Linear(c["enc"]["dim"], c["q"]["embed_dim"]),
                post_quant=torch.nn.Linear(c["q"]["embed_dim"], c["dec"]["dim"]))
    sd = {f"{k}.{n}": t for k, m in mods.items() for n, t in m.state_dict().items()}
    return sd, torch.rand(4)   # the tail proves the RNG stream ends in the same state


def fingerprint(sd):
    names = sorted(sd)
    fp = np.array([[sd[n].double().sum().item(), (sd[n].double() ** 2).sum().item(), *(sd[n].flatten()[:4].double().tolist() + [0.0] * 4)[:4]] for n in names])
    return names, fp


if __name__ == "__main__":
    L, Q = RL.load_layers(), RL.load_quantizers()
    out = {}
    for case in CASES:
        sd, tail = build(L, Q, case, 0)
        names, fp = fingerprint(sd)
        out[f"{case}_names"], out[f"{case}_fp"], out[f"{case}_tail"] = np.array(names), fp, tail.numpy()
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "init_seed.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
