"""CPU suite: the C-ABI library loads and exports every symbol include/enh_hip.h declares (no compute calls without a
GPU), and the host-side mirror of the reference interface behaves (config factory, module tree, data contract)."""
import json
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


def test_loss_modules_construct_every_term():
    """every shipped reference config sets perceptual_weight 0.1 and adversarial_weight 0.1 (configs/imagenet_vitvq_*.yaml:20-26): both terms must
    construct; the LPIPS module carries lpips 0.1.4's state-dict keys; pixel + codebook arithmetic and log keys as vqperceptual.py:41-56"""
    import warnings
    from enhancing.losses.vqperceptual import VQLPIPS, VQLPIPSWithDiscriminator
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        P = VQLPIPS(perceptual_weight=0.1)
        D = VQLPIPSWithDiscriminator(perceptual_weight=0.1, adversarial_weight=0.1, disc_params={"size": 16})
    keys = set(P.state_dict())
    assert {"perceptual_loss.scaling_layer.shift", "perceptual_loss.net.slice1.0.weight", "perceptual_loss.net.slice5.28.bias",
            "perceptual_loss.lin0.model.1.weight", "perceptual_loss.lins.4.model.1.weight"} <= keys
    assert hasattr(D, "discriminator") and hasattr(D, "perceptual_loss")
    assert not any(p.requires_grad for p in P.perceptual_loss.parameters())       # frozen, as lpips' (pnet_tune=False, lin layers in eval)
    L = VQLPIPSWithDiscriminator(loglaplace_weight=0.0, loggaussian_weight=1.0, perceptual_weight=0.0, adversarial_weight=0.0)
    x, r = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 8, 8)
    loss, log = L(torch.tensor(0.5), x, r, 0, 0, 0, split="val")
    assert set(log) == {"val/total_loss", "val/quant_loss", "val/rec_loss", "val/loglaplace_loss", "val/loggaussian_loss", "val/perceptual_loss"}
    assert abs(loss.item() - (((r - x) ** 2).mean().item() + 0.5)) < 1e-6


def test_loss_network_operand_format_resolution(monkeypatch):
    """Round 6: the loss networks' 16-bit operand format — module attribute > ENH_LOSS_OPERANDS > the engine that produced the reconstruction (fp16 engine ->
    fp16, as the reference's --use_amp autocast does, main.py:52; bf16 / fp32 engine -> bf16) > the conv_nhwc module default; a bf16 engine with fp16 loss
    networks is refused for a differentiable generator-side call (their gradients would run unscaled in fp16)."""
    import types
    from enhancing.losses.op import conv_nhwc
    from enhancing.losses.vqperceptual import VQLPIPS
    monkeypatch.delenv("ENH_LOSS_OPERANDS", raising=False)
    L = VQLPIPS(perceptual_weight=0.0)
    fp16_layer = types.SimpleNamespace(_enh_engine=types.SimpleNamespace(scaled=True))
    bf16_layer = types.SimpleNamespace(_enh_engine=types.SimpleNamespace(scaled=False))
    assert L.loss_operands(fp16_layer) == "fp16" and L.loss_operands(bf16_layer) == "bf16"
    assert L.loss_operands(None) == ("fp16" if conv_nhwc.OPERAND_DTYPE == torch.float16 else "bf16")
    with conv_nhwc.operand_dtype("fp16"):
        assert L.loss_operands(None) == "fp16" and L.loss_operands(bf16_layer) == "bf16"      # (an engine, when known, decides)
    monkeypatch.setenv("ENH_LOSS_OPERANDS", "bf16")
    assert L.loss_operands(fp16_layer) == "bf16"
    L.operands = "fp16"
    assert L.loss_operands(bf16_layer) == "fp16"
    L.operands = "fp8"
    with pytest.raises(ValueError, match="'bf16' or 'fp16'"):
        L.loss_operands(None)
    # the refusal: fp16 loss networks behind an unscaled engine, differentiable generator-side call with a loss network present
    L2 = VQLPIPS(perceptual_weight=0.0)
    L2.operands = "fp16"
    L2.perceptual_loss = torch.nn.Identity()
    x = torch.zeros(1, 3, 8, 8, requires_grad=True)
    with pytest.raises(ValueError, match="loss-scaled backward"):
        L2(torch.zeros(()), torch.zeros(1, 3, 8, 8), x, 0, 0, 0, last_layer=bf16_layer)


def test_lpips_without_weights_raises_and_parent_load_supplies_them(monkeypatch):
    """(1) no silent random perceptual term — without weights the module constructs (so a checkpoint can be loaded into it) but its forward
    raises unless random init was an explicit opt-in; (2) a PARENT's load_state_dict (a reference checkpoint carrying loss.perceptual_loss.*) marks
    the weights as loaded and invalidates the packed operand cache, and so does an in-place write to a parameter."""
    from enhancing.losses.lpips import LPIPS
    from enhancing.losses.vqperceptual import VQLPIPS
    monkeypatch.delenv("ENH_LPIPS_RANDOM_INIT", raising=False)
    monkeypatch.delenv("ENH_LPIPS_WEIGHTS", raising=False)
    P = VQLPIPS(perceptual_weight=0.1)
    m = P.perceptual_loss
    assert not m.weights_loaded and not m.random_init
    with pytest.raises(RuntimeError, match="LPIPS has no weights"):
        m(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        src = LPIPS(net="vgg", pretrained=False)                     # explicit opt-in
    assert src.random_init and not src.weights_loaded
    cpu = torch.device("cpu")
    before = m._device_weights(cpu)["fwd"][2].clone()
    # a random trunk never reaches a checkpoint (its tensors are omitted from state dicts), and a PARTIAL dict does not count as weights
    assert not [k for k in src.state_dict()] and not [k for k in VQLPIPS(perceptual_weight=0.1, **{}).state_dict() if False]
    real = {n: t.detach().clone() for n, t in list(src.named_parameters(remove_duplicate=False)) + list(src.named_buffers(remove_duplicate=False))}       # stands in for an lpips-format checkpoint
    partial = {"perceptual_loss." + k: v for k, v in real.items() if not k.startswith("lin")}
    P.load_state_dict(partial, strict=False)
    assert not m.weights_loaded
    with pytest.raises(RuntimeError, match="LPIPS has no weights"):
        m(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))
    with torch.no_grad():
        for p_ in m.parameters():
            p_.zero_()
    m._dev.clear()
    P.load_state_dict({"perceptual_loss." + k: v for k, v in real.items()}, strict=True)      # parent load: child override never runs
    assert m.weights_loaded and not m.random_init
    assert set(P.state_dict()) == {"perceptual_loss." + k for k in real}                    # loaded weights ARE saved
    after = m._device_weights(cpu)["fwd"][2]
    assert before.abs().max() == 0 and after.abs().max() > 0       # stale (zero) operands were not re-used
    with torch.no_grad():
        m._conv(2).weight.mul_(2.0)                             # in-place write: version bump -> cache rebuilt
    assert torch.equal(m._device_weights(cpu)["fwd"][2].float(), (after.float() * 2).to(torch.bfloat16).float())
    # ONE lin-key family is a complete lpips state dict (lin{k} and lins.{k} are the same Parameters) — nested inside a parent checkpoint too
    for fam_drop in ("lins.", "lin"):
        P2 = VQLPIPS(perceptual_weight=0.1)
        one = {"perceptual_loss." + k: v for k, v in real.items()
               if not (k.startswith("lins.") if fam_drop == "lins." else (k.startswith("lin") and not k.startswith("lins.")))}
        assert len(one) == len(real) - 5
        missing = P2.load_state_dict(one, strict=False)
        assert P2.perceptual_loss.weights_loaded and len(missing.missing_keys) == 5
        assert torch.equal(P2.perceptual_loss.lin3.model[1].weight, real["lin3.model.1.weight"])
    monkeypatch.setenv("ENH_LPIPS_RANDOM_INIT", "1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert LPIPS(net="vgg").random_init


def test_reference_yaml_files_load_unchanged():
    """the reference's OWN configs/imagenet_vitvq_{small,base,large}.yaml (not this repo's edited copies) resolve through the
    reflection factory byte for byte — ViTVQ with VQLPIPSWithDiscriminator (LPIPS 0.1 + StyleGAN discriminator 0.1) and the ImageNet data module."""
    import warnings
    from enhancing.utils.general import get_config_from_file, get_obj_from_str, initialize_from_config
    ref = os.environ.get("ENH_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "configs")):
        pytest.skip("reference checkout not present on this machine")
    for name, n_train in (("imagenet_vitvq_small", 50.908e6), ("imagenet_vitvq_base", 170.664e6)):
        cfg = get_config_from_file(os.path.join(ref, "configs", name + ".yaml"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = initialize_from_config(cfg.model)
        ae = sum(p.numel() for n, p in model.named_parameters() if p.requires_grad and not n.startswith("loss."))
        assert abs(ae - n_train) < 2e3
        assert model.loss.perceptual_weight == 0.1 and model.loss.adversarial_weight == 0.1
        assert hasattr(model.loss, "discriminator") and hasattr(model.loss, "perceptual_loss")
        assert get_obj_from_str(cfg.dataset.target).__name__ == "DataModuleFromConfig"
        assert get_obj_from_str(cfg.dataset.params.train.target).__name__ == "ImageNetTrain"
    large = get_config_from_file(os.path.join(ref, "configs", "imagenet_vitvq_large.yaml"))
    assert large.model.params.decoder.dim == 1280 and get_obj_from_str(large.model.target).__name__ == "ViTVQ"


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
    """reference scheduler.py API: constructor names, schedule() = multiplier of start (LambdaLR), __call__ = absolute value"""
    from enhancing.utils.scheduler import ExponentialDecayScheduler, LambdaWarmUpCosineScheduler, LambdaWarmUpLinearScheduler
    s = LambdaWarmUpCosineScheduler(warm_up_steps=10, max_decay_steps=100, min_=1e-6, max_=1e-4, start=1e-5)
    assert abs(s.schedule(0) - 1.0) < 1e-9 and abs(s.schedule(10) - 10.0) < 1e-6 and abs(s.schedule(100) - 0.1) < 1e-6
    assert abs(s(10) - 1e-4) < 1e-12                                  # __call__ multiplies by start
    e = ExponentialDecayScheduler(start=1.0, end=0.5, decay_every_step=10, scale_factor=0.01)
    assert e.schedule(0) == 1.0 and abs(e.schedule(10) - 0.9048374) < 1e-6 and e.schedule(15) == e.schedule(10)   # held between multiples
    assert e.schedule(1000) == 0.5 and e(1000) == 0.5
    lin = LambdaWarmUpLinearScheduler(warm_up_steps=10, max_decay_steps=100, min_=0.0, max_=1e-4, start=1e-5)
    assert abs(lin.schedule(50) - 5.0) < 1e-9
    with pytest.raises(TypeError):
        LambdaWarmUpCosineScheduler(warmup_steps=10, max_decay_steps=100, min_=0, max_=1, start=1)      # the reference's spelling is warm_up_steps


def test_real_datasets_are_sharded_across_ranks():
    """a dataset without set_shard gets a DistributedSampler when world > 1 -> disjoint per-rank index sets that cover the data"""
    import torch
    from enhancing.dataloader import DataModuleFromConfig

    class Plain(torch.utils.data.Dataset):
        def __len__(self): return 40
        def __getitem__(self, i): return {"image": torch.full((3, 2, 2), float(i)), "class": torch.tensor([i])}

    seen = []
    for rank in range(2):
        dm = DataModuleFromConfig(batch_size=4, num_workers=0)
        dm.dataset_configs["train"] = None
        dm.datasets, dm.rank, dm.world = {"train": Plain()}, rank, 2
        loader = dm.train_dataloader()
        dm.set_epoch(3)
        seen.append(sorted(int(c) for b in loader for c in b["class"].view(-1)))
    assert len(seen[0]) == len(seen[1]) == 20 and not set(seen[0]) & set(seen[1]) and sorted(seen[0] + seen[1]) == list(range(40))


def test_deep_rq_fails_loudly():
    from enhancing.modules.stage1.quantizers import VectorQuantizer
    with pytest.raises(ValueError, match="num_quantizers <= 8"):
        VectorQuantizer(32, 1024, use_residual=True, num_quantizers=9)


@pytest.mark.parametrize("residual", [False, True])
def test_gumbel_quantizer_equals_the_reference_class(residual):
    """GumbelQuantizer (plain torch, outside the HIP hot path): same constructor, same RNG consumption, same outputs and gradients as the reference's
    class (quantizers.py:95-126 + the residual loop of BaseQuantizer.forward) for one seed, in training (soft) and eval (hard) mode — checked against
    the reference itself where /root/reference is present, and for its invariants everywhere."""
    from enhancing.modules.stage1.quantizers import GumbelQuantizer
    import _reference_loader as RL
    kw = dict(embed_dim=32, n_embed=64, temp_init=0.7, use_residual=residual, num_quantizers=3 if residual else None)
    torch.manual_seed(3)
    ours = GumbelQuantizer(**kw)
    z = torch.randn(2, 16, 32, requires_grad=True)
    ref = None
    if RL.available():
        torch.manual_seed(3)
        ref = RL.load_quantizers().GumbelQuantizer(**kw)
        assert torch.equal(ref.embedding.weight, ours.embedding.weight)
    for training in (True, False):
        ours.train(training)
        torch.manual_seed(11)
        zq, loss, idx = ours(z)
        assert zq.shape == z.shape and idx.shape == ((2, 16, 3) if residual else (2, 16)) and idx.dtype == torch.int64 and float(loss) >= 0
        g = torch.autograd.grad(zq.pow(2).sum() + loss, [z, ours.embedding.weight], allow_unused=True)
        if not training and not residual:       # hard one-hot: the output IS a normalised code
            en = torch.nn.functional.normalize(ours.embedding.weight, dim=-1)
            assert torch.allclose(zq, en[idx], atol=1e-6)
        if ref is not None:
            ref.train(training)
            z2 = z.detach().clone().requires_grad_(True)
            torch.manual_seed(11)
            rq, rl, ri = ref(z2)
            assert torch.equal(ri, idx) and torch.allclose(rq, zq, atol=1e-6) and torch.allclose(rl, loss, atol=1e-6)
            rg = torch.autograd.grad(rq.pow(2).sum() + rl, [z2, ref.embedding.weight], allow_unused=True)
            for a, b in zip(g, rg):
                assert (a is None) == (b is None) and (a is None or torch.allclose(a, b, atol=1e-5))


def test_trainer_drives_the_lightning_protocol_in_order(monkeypatch, tmp_path):
    """Trainer.fit with a recording stand-in model: per batch and per optimizer `training_step -> step` (Lightning 1.5 order for the
    two-optimizer GAN setup, reference main.py:51-61), gradient accumulation windows, the step-wise LR schedule, validation, and a
    {"state_dict": ...} checkpoint per epoch (general.py:49-55)."""
    from enhancing.engine import trainer as T
    calls, syncs = [], []

    class Opt:
        def __init__(self, name):
            self.name, self.param_groups, self.grad_scale = name, [dict(lr=1.0)], 1.0

        def step(self):
            calls.append((self.name + ".step", self.param_groups[0]["lr"], self.grad_scale))

        def state_dict(self):
            return {"name": self.name}

    class Model:
        logged, global_step = {}, 0
        engine = type("E", (), {"store": None})()

        def configure_optimizers(self):
            return [Opt("ae"), Opt("disc")], [{"scheduler": lambda s: 1.0 / (1 + s)}, {"scheduler": lambda s: 1.0 / (1 + s)}]

        def training_step(self, batch, batch_idx, optimizer_idx, zero_grad=True):
            calls.append(("ts", batch_idx, optimizer_idx, zero_grad))
            syncs.append(self.engine.sync_grads)
            self.logged["train/total_loss"] = torch.tensor(0.5)

        def validation_step(self, batch, batch_idx):
            calls.append(("val", batch_idx))
            self.logged["val/rec_loss"] = torch.tensor(0.25)

        def state_dict(self):
            return {"w": torch.ones(2)}

    class Data:
        dataset_configs = {"train": 1, "validation": 1}

        def setup(self, rank, world):
            calls.append(("setup", rank, world))

        def train_dataloader(self):
            return [{"image": torch.zeros(3, 3, 8, 8)} for _ in range(4)]

        def val_dataloader(self):
            return [{"image": torch.zeros(3, 3, 8, 8)} for _ in range(3)]

    monkeypatch.setattr(T, "init_process_group_from_env", lambda *a, **k: (0, 0, 1))
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: None)
    tr = T.Trainer(max_epochs=1, accumulate_grad_batches=2, default_root_dir=str(tmp_path), log_every_n_steps=1, limit_val_batches=2)
    tr.fit(Model(), Data())
    train = [c for c in calls if c[0] in ("ts", "ae.step", "disc.step")]
    assert train == [
        ("ts", 0, 0, True), ("ts", 0, 1, True),                                                      # first half of the window: no steps
        ("ts", 1, 0, False), ("ae.step", 1.0, 0.5), ("ts", 1, 1, False), ("disc.step", 1.0, 0.5),      # optimizer 0 steps before optimizer 1 runs
        ("ts", 2, 0, True), ("ts", 2, 1, True),
        ("ts", 3, 0, False), ("ae.step", 0.5, 0.5), ("ts", 3, 1, False), ("disc.step", 0.5, 0.5)]      # lr = base * schedule(global_step = 1)
    assert syncs == [False, False, True, True] * 2     # gradients are all-reduced only on the last micro-batch of a window (no_sync)
    assert [c for c in calls if c[0] == "val"] == [("val", 0), ("val", 1)] and tr.global_step == 2
    ck = torch.load(os.path.join(str(tmp_path), "ckpt", "epoch=00.ckpt"))
    assert set(ck) >= {"state_dict", "epoch", "global_step", "optimizer", "optimizer_states"} and len(ck["optimizer_states"]) == 2
    rows = [json.loads(l) for l in open(os.path.join(str(tmp_path), "metrics.jsonl"))]
    assert rows[0]["step"] == 1 and rows[0]["train/total_loss"] == 0.5 and rows[-1]["val/rec_loss"] == 0.25


def test_imagenet_folder_dataset_contract(tmp_path):
    """reference dataloader/imagenet.py:15-54 on a miniature folder tree: class index = sorted folder order, sample = {'image' float [3,R,R]
    in [0,1], 'class' [1]}, train = resize + random crop (+flip), validation = resize + centre crop (deterministic)"""
    import numpy as np
    from PIL import Image
    from enhancing.dataloader import DataModuleFromConfig
    rng = np.random.default_rng(0)
    for split in ("train", "val"):
        for ci, wnid in enumerate(("n02", "n01")):
            d = tmp_path / split / wnid
            d.mkdir(parents=True)
            for k in range(3):
                Image.fromarray(rng.integers(0, 256, (40 + 7 * k, 56 - 5 * ci, 3), dtype=np.uint8)).save(d / f"img{k}.png")
            (d / "notes.txt").write_text("ignored")
    node = lambda cls: {"target": f"enhancing.dataloader.imagenet.{cls}", "params": {"root": str(tmp_path), "resolution": 32}}
    dm = DataModuleFromConfig(batch_size=4, num_workers=0, train=node("ImageNetTrain"), validation=node("ImageNetValidation"))
    dm.setup()
    tr, va = dm.datasets["train"], dm.datasets["validation"]
    assert len(tr) == 6 and len(va) == 6 and tr.labels == [0, 0, 0, 1, 1, 1] and "n01" in tr.paths[0]
    s = va[4]
    assert s["image"].shape == (3, 32, 32) and s["image"].dtype == torch.float32 and 0.0 <= s["image"].min() and s["image"].max() <= 1.0
    assert s["class"].shape == (1,) and int(s["class"]) == 1 and torch.equal(va[4]["image"], s["image"])
    batch = next(iter(dm.val_dataloader()))
    assert batch["image"].shape == (4, 3, 32, 32) and batch["class"].shape == (4, 1)
    assert tr[0]["image"].shape == (3, 32, 32)
    with pytest.raises(FileNotFoundError):
        type(tr)(str(tmp_path / "nope"))


def test_device_transform_paths_equal_the_host_path(tmp_path, monkeypatch):
    """device_transform=True: workers hand over uint8 pixels + the crop window / flip they drew; crop + flip + ToTensor then run in ONE kernel
    (enh_crop_flip_u8, replaced here by its numpy statement).  device_resize=True: the workers only decode and the antialiased bilinear resize runs on
    the device as well (enh_resize_u8, replaced here by the oracle's statement of Pillow's two passes, driven by the PRODUCT's coefficient tables).
    Same seed -> bit-identical batches on all three paths, for the training transform (shorter side -> R, random crop, flip) and the validation
    transform (exact (R, R) resize, reference imagenet.py:44-49)."""
    import numpy as np
    from PIL import Image
    from enhancing import _C
    from enhancing.dataloader import DataModuleFromConfig
    rng = np.random.default_rng(1)
    for split in ("train", "val"):
        for ci, wnid in enumerate(("a", "b")):
            d = tmp_path / split / wnid
            d.mkdir(parents=True)
            for k in range(3):
                Image.fromarray(rng.integers(0, 256, (41 + 9 * k, 60 - 11 * ci, 3), dtype=np.uint8)).save(d / f"i{k}.png")

    def crop_flip_u8(src, meta, R):      # the kernel's contract, stated in numpy
        out = np.zeros((src.shape[0], 3, R, R), np.float32)
        for b in range(src.shape[0]):
            y0, x0, flip = (int(v) for v in meta[b])
            w = src[b, y0:y0 + R, x0:x0 + R].numpy()
            w = w[:, ::-1] if flip else w
            out[b] = w.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
        return torch.from_numpy(out)

    def resize_u8(src, meta, bounds, weights, dst):      # enh_resize_u8's contract: the two integer passes over the tables the product built
        src, bounds, weights = src.numpy().astype(np.int64), bounds.numpy(), weights.numpy().astype(np.int64)
        for b, (hin, win, hout, wout, hbo, hko, hks, vbo, vko, vks) in enumerate(meta.tolist()):
            t = np.zeros((hin, wout, 3), np.int64)
            for x in range(wout):
                x0, n = bounds[hbo + 2 * x], bounds[hbo + 2 * x + 1]
                t[:, x] = (1 << 21) + (src[b, :hin, x0:x0 + n] * weights[hko + x * hks:hko + x * hks + n][None, :, None]).sum(1)
            t = np.clip(t >> 22, 0, 255)
            o = np.zeros((hout, wout, 3), np.int64)
            for y in range(hout):
                y0, n = bounds[vbo + 2 * y], bounds[vbo + 2 * y + 1]
                o[y] = (1 << 21) + (t[y0:y0 + n] * weights[vko + y * vks:vko + y * vks + n][:, None, None]).sum(0)
            dst[b, :hout, :wout] = torch.from_numpy(np.clip(o >> 22, 0, 255).astype(np.uint8))
        return dst
    monkeypatch.setattr(_C, "crop_flip_u8", crop_flip_u8)
    monkeypatch.setattr(_C, "resize_u8", resize_u8)
    for split, target in (("train", "ImageNetTrain"), ("validation", "ImageNetValidation")):
        batches = {}
        for mode, params in (("host", {}), ("device_tail", {"device_transform": True}), ("device_resize", {"device_resize": True})):
            node = {"target": f"enhancing.dataloader.imagenet.{target}", "params": dict(root=str(tmp_path), resolution=32, **params)}
            dm = DataModuleFromConfig(batch_size=3, num_workers=0, **{split: node})
            dm.setup()
            loader = dm.train_dataloader() if split == "train" else dm.val_dataloader()
            if mode != "host":
                loader.fn.device = torch.device("cpu")
            np.random.seed(7)
            torch.manual_seed(7)
            batches[mode] = list(loader)
        assert len(batches["device_resize"]) == 2 and set(batches["device_resize"][0]) == {"image", "class"}
        for a, b, c in zip(batches["host"], batches["device_tail"], batches["device_resize"]):
            assert torch.equal(a["image"], b["image"]) and torch.equal(a["class"], b["class"])
            assert torch.equal(a["image"], c["image"]) and torch.equal(a["class"], c["class"]), split


def test_resize_tables_are_built_by_the_collate_and_outliers_can_be_prereduced(tmp_path):
    """device_resize batches leave collate_u8 (i.e. the DataLoader worker) with the enh_resize_u8 tables already built — identical to what the consumer
    would build — so DeviceTransform only copies them.  host_prereduce=k bounds the slot of a decoded outlier: shorter side lands in [k, 2k); images
    below 2k are untouched (bit-identical samples)."""
    import numpy as np
    from PIL import Image
    from enhancing.dataloader.imagenet import ImageNetTrain, collate_u8
    from enhancing.dataloader.resize import build_tables, output_size
    rng = np.random.default_rng(3)
    d = tmp_path / "train" / "a"
    d.mkdir(parents=True)
    sizes = [(50, 70), (45, 33), (400, 610)]                        # (h, w); the last one is the "outlier"
    for k, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"i{k}.png")
    ds = ImageNetTrain(str(tmp_path), resolution=32, device_resize=True)
    np.random.seed(0)
    samples = [ds[i] for i in range(3)]
    batch = collate_u8(samples)
    assert batch["in_size"].tolist() == [list(s) for s in sizes] and tuple(batch["pixels_u8"].shape) == (3, 400, 610, 3)
    outs = []
    for h, w in sizes:
        wo, ho = output_size(w, h, 32)
        outs.append((ho, wo))
    assert batch["out_size"].tolist() == [list(o) for o in outs]
    meta, bnd, wts = build_tables(sizes, outs)
    assert torch.equal(batch["resize_meta"], meta) and torch.equal(batch["resize_bounds"], bnd) and torch.equal(batch["resize_weights"], wts)
    assert batch["resize_meta"].dtype == torch.int32 and batch["resize_bounds"].dtype == torch.int32 and batch["resize_weights"].dtype == torch.int32

    capped = ImageNetTrain(str(tmp_path), resolution=32, device_resize=True, host_prereduce=64)
    np.random.seed(0)
    csamples = [capped[i] for i in range(3)]
    for a, b in zip(samples[:2], csamples[:2]):                     # below 2k: untouched, same random draws
        assert torch.equal(a["pixels_u8"], b["pixels_u8"]) and torch.equal(a["window"], b["window"])
    h, w = csamples[2]["in_size"].tolist()
    assert 64 <= min(h, w) < 128 and tuple(collate_u8(csamples)["pixels_u8"].shape[1:3]) == (h, w)
    ref = np.asarray(Image.open(d / "i2.png").convert("RGB").reduce(400 // 64))
    assert np.array_equal(csamples[2]["pixels_u8"].numpy(), ref)
    assert csamples[2]["out_size"].tolist() == [32, int(32 * w / h)]


def test_image_logger_and_setup_callbacks(tmp_path, monkeypatch):
    """reference utils/callback.py:21-141 + general.py:43-60: frequency rule (every batch_frequency batches and the early 2, 4, ... steps), max_images,
    clamp, file naming, torchvision's grid geometry; the Trainer calls the Lightning hook names"""
    import numpy as np
    from PIL import Image
    from enhancing.utils.callback import ImageLogger, SetupCallback, make_grid
    from enhancing.utils.general import AttrDict, setup_callbacks
    g = make_grid(torch.ones(5, 3, 8, 6), nrow=4)
    assert g.shape == (3, 2 * 10 + 2, 4 * 8 + 2) and g[:, :2].sum() == 0 and g[:, 2:10, 2:8].min() == 1 and g[:, 12:20, 10:16].sum() == 0
    assert make_grid(torch.ones(1, 1, 4, 4)).shape == (3, 4, 4)

    class Mod:
        training, global_step, calls = True, 3, 0

        def eval(self): self.training = False
        def train(self): self.training = True

        def log_images(self, batch, split="train", pl_module=None):
            assert not self.training
            self.calls += 1
            return {"originals": batch["image"] * 2 - 0.5, "reconstructions": batch["image"]}

    class Tr:
        rank, root, current_epoch = 0, str(tmp_path), 1
    lg = ImageLogger(batch_frequency=8, max_images=2)
    assert lg.log_steps == [1, 2, 4, 8]
    mod, batch = Mod(), {"image": torch.rand(6, 3, 16, 16)}
    fired = []
    for bi in range(20):
        before = mod.calls
        lg.on_train_batch_end(Tr(), mod, None, batch, bi)
        if mod.calls > before:
            fired.append(bi)
    assert fired == [0, 2, 4, 8, 16] and mod.training      # the reference's rule: firing at 0 pops the '1' entry (callback.py:117-124)
    f = tmp_path / "results" / "train" / "originals_gs-000003_e-000001_b-000016.png"
    assert f.exists() and (tmp_path / "results" / "train" / "reconstructions_gs-000003_e-000001_b-000000.png").exists()
    im = np.asarray(Image.open(f))
    assert im.shape == (16 + 4, 2 * 18 + 2, 3)                                  # max_images = 2 -> one row of the nrow=4 grid, 2 px padding
    Tr.rank = 1
    lg.on_validation_batch_end(Tr(), mod, None, batch, 0, 0)                   # rank_zero_only
    assert not (tmp_path / "results" / "val").exists()
    monkeypatch.chdir(tmp_path)
    cbs, logger = setup_callbacks(AttrDict(name="exp", batch_frequency=750, max_images=4), AttrDict(model={}))
    assert isinstance(cbs[0], SetupCallback) and isinstance(cbs[1], ImageLogger) and logger is None
    Tr.rank = 0
    cbs[0].on_pretrain_routine_start(Tr(), mod)
    assert cbs[0].logdir.is_dir() and cbs[0].ckptdir.is_dir()


def _build_ours(case, seed):
    """this package's modules in the reference's construction order (vitvqgan.py:34-39)"""
    sys_path_oracle = os.path.join(ROOT, "oracle")
    import sys
    if sys_path_oracle not in sys.path:
        sys.path.insert(0, sys_path_oracle)
    import make_golden_init as G
    from enhancing.modules.stage1 import layers as L, quantizers as Q
    return G, G.build(L, Q, case, seed)


@pytest.mark.parametrize("case", ["tiny", "small"])
def test_init_is_seed_for_seed_with_the_reference(case):
    """torch.manual_seed(0) then construction -> the REFERENCE's tensors, bit for bit (the reference constructs stock
    nn.Linear / nn.Conv2d, whose default init consumes the RNG, then re-draws in apply() order; layers.py:71-82,175,207).
    Checked against fingerprints of the reference's own state dict (tests/golden/init_seed.npz, oracle/make_golden_init.py) and,
    where /root/reference is present (this container), against the reference modules tensor by tensor."""
    import numpy as np
    G, (sd, tail) = _build_ours(case, 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "init_seed.npz"))
    names, fp = G.fingerprint(sd)
    assert names == list(g[f"{case}_names"])
    assert np.array_equal(fp, g[f"{case}_fp"]), [n for n, a, b in zip(names, fp, g[f"{case}_fp"]) if not np.array_equal(a, b)]
    assert np.array_equal(tail.numpy(), g[f"{case}_tail"]), "the RNG stream must end in the same state"
    import _reference_loader as RL
    if RL.available():
        ref_sd, ref_tail = G.build(RL.load_layers(), RL.load_quantizers(), case, 0)
        assert list(ref_sd) == list(sd)
        assert all(torch.equal(ref_sd[k], sd[k]) for k in sd) and torch.equal(ref_tail, tail)
