"""CPU suite: the C-ABI library loads and exports every symbol include/enh_hip.h declares (no compute calls without a
GPU), and the host-side mirror of the reference interface behaves (config factory, module tree, data contract)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "enh_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(enh_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from enhancing import _C
    if not os.path.exists(_C.LIB_PATH):
        import __graft_entry__ as G
        G.build()
    L = _C.lib()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/enh_hip.h but not exported"
        assert name in _C.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_C.SIGNATURES) == declared
    import re
    hdr = open(os.path.join(ROOT, "include", "enh_hip.h")).read()
    assert L.enh_abi_version() == _C.ABI_VERSION == int(re.search(r"#define ENH_ABI_VERSION (\d+)", hdr).group(1))
    assert L.enh_vq_workspace_bytes(131072, 8192, 4) > 8192 * 32 * 4


def test_host_tensors_are_rejected_loudly():
    from enhancing import _C
    x = torch.zeros(4, 32)
    with pytest.raises(RuntimeError, match="ROCm device"):
        _C.vq_forward(x, torch.zeros(8, 32), 0.25, 1, True)


def test_config_factory_and_module_tree():
    import vitvq_oracle as O
    from enhancing.utils.general import get_config_from_file, initialize_from_config
    for name, (n_train, enc_dim) in {"imagenet_vitvq_small": (50.908e6, 512), "imagenet_vitvq_base": (170.664e6, 768)}.items():
        cfg = get_config_from_file(os.path.join(ROOT, "configs", name + ".yaml"))
        assert cfg.model.target == "enhancing.modules.stage1.vitvqgan.ViTVQ" and cfg.model.params.encoder.dim == enc_dim
        model = initialize_from_config(cfg.model)
        n = sum(p.numel() for p in model.parameters() if p.requires_grad)
        assert abs(n - n_train) < 2e3, (name, n)          # SURVEY.md §A.3 parameter counts
        oc = dict(image_size=256, patch_size=8, encoder=dict(cfg.model.params.encoder), decoder=dict(cfg.model.params.decoder),
                  quantizer=dict(cfg.model.params.quantizer))
        sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert sd == {k: tuple(s) for k, s in O.param_shapes(oc).items()}  # reference state-dict contract, SURVEY.md §8b
        assert not model.encoder.en_pos_embedding.requires_grad
        assert torch.equal(model.encoder.en_pos_embedding, O.pos_embedding(enc_dim, 32, 32))
    rq = get_config_from_file(os.path.join(ROOT, "configs", "imagenet_rqvae_base.yaml"))
    q = initialize_from_config(rq.model).quantizer
    assert q.use_residual and q.num_quantizers == 4 and q.depth == 4


def test_missing_loss_terms_fail_loudly(monkeypatch):
    from enhancing.losses.vqperceptual import VQLPIPS, VQLPIPSWithDiscriminator
    monkeypatch.delenv("ENH_ALLOW_MISSING_TERMS", raising=False)
    with pytest.raises(NotImplementedError):
        VQLPIPS(perceptual_weight=0.1)
    with pytest.raises(NotImplementedError):
        VQLPIPSWithDiscriminator(perceptual_weight=0.1, adversarial_weight=0.0)
    with pytest.raises(NotImplementedError):
        VQLPIPSWithDiscriminator(perceptual_weight=0.0, adversarial_weight=0.1, use_adaptive_adv=True)
    assert hasattr(VQLPIPSWithDiscriminator(perceptual_weight=0.0, adversarial_weight=0.1, disc_params={"size": 16}), "discriminator")
    L = VQLPIPSWithDiscriminator(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0, adversarial_weight=0.0)
    x, r = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 8, 8)
    loss, log = L(torch.tensor(0.5), x, r, 0, 0, 0, split="val")
    assert set(log) == {"val/total_loss", "val/quant_loss", "val/rec_loss", "val/loglaplace_loss", "val/loggaussian_loss", "val/perceptual_loss"}
    assert abs(loss.item() - (((r - x) ** 2).mean().item() + 0.5)) < 1e-6


def test_engine_requires_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vitvq_oracle as O
    from enhancing.modules.stage1.vitvqgan import ViTVQ
    from enhancing.utils.general import AttrDict
    cfg = O.TINY_CFG
    m = ViTVQ("image", 64, 8, AttrDict.wrap(cfg["encoder"]), AttrDict.wrap(cfg["decoder"]), AttrDict.wrap(cfg["quantizer"]),
              AttrDict.wrap({"target": "enhancing.losses.vqperceptual.DummyLoss"}))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 64, 64))


def test_synthetic_data_contract():
    from enhancing.dataloader import DataModuleFromConfig
    dm = DataModuleFromConfig(batch_size=3, train={"target": "enhancing.dataloader.synthetic.SyntheticImages", "params": {"resolution": 32, "length": 12}})
    dm.setup(rank=1, world=2)
    b = next(iter(dm.train_dataloader()))
    assert b["image"].shape == (3, 3, 32, 32) and b["image"].dtype == torch.float32 and b["class"].shape == (3, 1)
    assert 0.0 <= b["image"].min() and b["image"].max() <= 1.0
    dm0 = DataModuleFromConfig(batch_size=3, train={"target": "enhancing.dataloader.synthetic.SyntheticImages", "params": {"resolution": 32, "length": 12}})
    dm0.setup(rank=0, world=2)
    assert not torch.equal(next(iter(dm0.train_dataloader()))["image"], b["image"])  # ranks see different data


def test_schedulers():
    from enhancing.utils.scheduler import ExponentialDecayScheduler, LambdaWarmUpCosineScheduler
    s = LambdaWarmUpCosineScheduler(10, 100, 1e-6, 1e-4, 1e-5)
    assert abs(s(0) - 1.0) < 1e-9 and abs(s(10) - 10.0) < 1e-6 and abs(s(100) - 0.1) < 1e-6
    assert ExponentialDecayScheduler(0.1, 1, 1.0, 0.5)(100) == 0.5
