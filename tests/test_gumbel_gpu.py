"""ViTVQGumbel (reference vitvqgan.py:191-212) on the split HIP schedule: encoder half -> GumbelQuantizer in plain torch under autograd -> decoder half.
Not a hot path (no shipped stage-1 config names the class); the test pins the WIRING: one training step's loss and parameter gradients against the CPU
oracle's towers with the same quantizer module in between (the Gumbel noise is replaced by its noiseless limit on both sides so that the GPU and CPU
random streams do not enter)."""
import numpy as np
import pytest
import torch

from util import rel

pytestmark = pytest.mark.gpu


def _noiseless(logits, tau=1.0, hard=False, dim=-1, **_):
    y = torch.softmax(logits / tau, dim=dim)
    if hard:
        one = torch.zeros_like(y).scatter_(dim, y.argmax(dim, keepdim=True), 1.0)
        y = one - y.detach() + y
    return y


def test_vitvq_gumbel_training_step_vs_oracle(monkeypatch):
    import vitvq_oracle as O
    from enhancing.modules.stage1.quantizers import GumbelQuantizer
    from enhancing.modules.stage1.vitvqgan import ViTVQGumbel
    from enhancing.utils.general import AttrDict
    monkeypatch.setattr(torch.nn.functional, "gumbel_softmax", _noiseless)
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"])
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS",
            "params": dict(codebook_weight=0.5, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}
    qcfg = dict(embed_dim=32, n_embed=512, temp_init=0.9)
    tsched = {"target": "enhancing.utils.scheduler.ExponentialDecayScheduler", "params": dict(start=0.9, end=0.1, decay_every_step=1, scale_factor=1e-3)}
    m = ViTVQGumbel("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(qcfg),
                    AttrDict.wrap(loss), temperature_scheduler=AttrDict.wrap(tsched))
    assert isinstance(m.quantizer, GumbelQuantizer)
    m.load_state_dict(P, strict=True)
    m.train()
    out = m.training_step({"image": x}, 0, 0)
    torch.cuda.synchronize()
    assert abs(m.logged["temperature"] - 0.9) < 1e-6
    # oracle: the CPU towers with the same quantizer class (plain torch) in between
    leaves = {k: v.detach().clone().requires_grad_(not k.endswith("pos_embedding")) for k, v in P.items()}
    q = GumbelQuantizer(**qcfg)
    q.temperature = 0.9
    q.train()
    q.embedding.weight = torch.nn.Parameter(leaves["quantizer.embedding.weight"])
    h = O.encoder(x, leaves, cfg) @ leaves["pre_quant.weight"].t() + leaves["pre_quant.bias"]
    quant, qloss, idx = q(h)
    xrec = O.decoder(quant @ leaves["post_quant.weight"].t() + leaves["post_quant.bias"], leaves, cfg)
    o_loss = (xrec - x).pow(2).mean() + 0.5 * qloss
    o_loss.backward()
    grads = {k: (q.embedding.weight.grad if k == "quantizer.embedding.weight" else v.grad) for k, v in leaves.items()}
    grads = {k: g for k, g in grads.items() if g is not None}
    assert abs(float(out) - float(o_loss)) <= 1e-2 * abs(float(o_loss)), (float(out), float(o_loss))
    m.engine.unscale_grads()      # fp16 engine: param.grad carries the loss scale until the step (or this call)
    errs = {k: rel(p.grad, grads[k]) for k, p in m.named_parameters() if k in grads}
    worst = max(errs, key=errs.get)
    print(f"ViTVQGumbel train step: loss {float(out):.5f} vs oracle {float(o_loss):.5f}; grads median rel {np.median(list(errs.values())):.2e}, worst {worst} {errs[worst]:.2e}")
    assert set(errs) == set(grads) and errs[worst] <= 3e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # inference API
    m.eval()
    codes = m.encode_codes(x, precision="bf16")      # the same encoder arithmetic as m(x) below (the default for codes is the x3 encoder)
    assert codes.shape == (2, 64) and codes.dtype == torch.int64
    rec = m.decode_codes(codes)
    with torch.no_grad():
        xr, _ = m(x)
    assert rec.shape == xr.shape == x.shape
    assert rel(rec, xr) <= 1e-2          # eval mode quantises hard: decode(codes) is the model's own reconstruction


def test_gumbel_temperature_anneals_under_graph_replay_mode(monkeypatch):
    """with engine.use_graphs the loss-module path replays from HIP graphs; the Gumbel temperature is a host scalar inside the step, so
    ViTVQGumbel must not take that path — the step's loss has to follow the scheduled temperature, and no step graph may be captured."""
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQGumbel
    from enhancing.utils.general import AttrDict
    monkeypatch.setattr(torch.nn.functional, "gumbel_softmax", _noiseless)
    cfg = O.TINY_CFG
    P = O.make_params(cfg, seed=11)
    x = O.make_images(5, 2, cfg["image_size"])
    loss = {"target": "enhancing.losses.vqperceptual.VQLPIPS",
            "params": dict(codebook_weight=0.5, loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0)}
    qcfg = dict(embed_dim=32, n_embed=512, temp_init=0.9)

    def run(temps, graphs):
        m = ViTVQGumbel("image", cfg["image_size"], cfg["patch_size"], AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(qcfg),
                        AttrDict.wrap(loss))
        m.load_state_dict(P, strict=True)
        m.train()
        m.engine.use_graphs = graphs
        it = iter(temps)
        m.temperature_scheduler = lambda step: next(it)
        out = [float(m.training_step({"image": x}, i, 0)) for i in range(len(temps))]
        torch.cuda.synchronize()
        return out, m
    eager, _ = run([0.9, 0.9, 0.2], graphs=False)
    replay, m = run([0.9, 0.9, 0.2], graphs=True)
    assert replay == eager, (replay, eager)                   # same weights every step (no optimizer step): the loss is a function of tau alone
    assert eager[0] == eager[1] and eager[2] != eager[1]      # ... and it moved when tau did
    assert not m.__dict__.get("_step_graphs")
