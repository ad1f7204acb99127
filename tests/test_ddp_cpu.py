"""world_size-2 gloo test of the bucketed gradient all-reduce (the N > 1 path of bench.py / main.py) on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeStore:
    """what GradSync needs from enhancing.engine.stage1.ParamStore: flat p / g and slice_of(prefix)."""

    def __init__(self, rank):
        self.sizes = {"encoder.a.": 1000, "encoder.b.": 3000, "decoder.a.": 500, "decoder.b.": 2500, "quantizer.": 64}
        self.off, o = {}, 0
        for k, n in self.sizes.items():
            self.off[k] = (o, o + n); o += n
        g = torch.Generator().manual_seed(100 + rank)
        self.g = torch.randn(o, generator=g)
        self.p = torch.full((o,), float(rank))

    def slice_of(self, prefix):
        return self.off[prefix]


def _worker(rank, world, port, q, algo="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enhancing-transformers_amd"))
    from enhancing.engine.ddp import GradSync
    store = FakeStore(rank)
    local = store.g.clone()
    sync = GradSync(store, min_bucket_elems=2000, algo=algo)
    sync.broadcast_parameters(0)
    assert torch.all(store.p == 0.0)
    for step in range(2):  # two steps: state must reset between them
        store.g.copy_(local)
        for prefix in ["decoder.b.", "decoder.a.", "quantizer.", "encoder.b.", "encoder.a."]:  # backward order
            sync.layer_done(prefix)
        sync.finish()
        # the per-bucket issue log (bench.py's comm block): the finished step's collectives in issue order, covering every element exactly once
        tl = sync.issue_timeline(0.0)
        assert tl and sum(r["elems"] for r in tl) == store.g.numel() and all(a["t_ms"] <= b["t_ms"] for a, b in zip(tl, tl[1:])), tl
    q.put((rank, store.g.numpy().copy(), local.numpy().copy()))   # numpy, not tensors: a tensor travels as a shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,algo", [(2, "allreduce"), (2, "rs_ag"), (3, "rs_ag"), (4, "rs_ag"), (4, "allreduce")])
def test_gradsync_gloo_world2(world, algo):
    """bucketed gradient sum, as one all-reduce per bucket or as reduce-scatter + all-gather (world 3: bucket sizes are not multiples of the world size,
    so the remainder path runs too)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    got = [(r, torch.from_numpy(a), torch.from_numpy(b)) for r, a, b in (q.get(timeout=120) for _ in range(world))]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    total = sum(g[2] for g in got)
    for _, reduced, _ in got:
        assert torch.allclose(reduced, total, atol=1e-5)   # SUM; the 1/world mean is folded into AdamW's grad_scale
        assert torch.equal(reduced, got[0][1])             # every rank ends with the same bits


def test_gradsync_reduces_unannounced_slices():
    """single-process group: if the schedule forgets a unit, finish() must still synchronise it (and record the gap)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from enhancing.engine.ddp import GradSync
        store = FakeStore(0)
        sync = GradSync(store)
        sync.layer_done("decoder.b.")
        sync.finish()
        assert sync.gap_elems == store.g.numel() - 2500
        sync.gap_elems = 0
        for prefix in store.sizes:
            sync.layer_done(prefix)
        sync.finish()
        assert sync.gap_elems == 0
    finally:
        dist.destroy_process_group()


def _disc_worker(rank, world, port, q):
    """the discriminator optimizer under DDP: per-rank gradients -> ONE all-reduce of the flat buffer -> fused AdamW with the mean"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("enhancing-transformers_amd", "oracle"):
        sys.path.insert(0, os.path.join(root, p))
    import vitvq_oracle as O
    from enhancing import _C
    from enhancing.engine.optim import FlatAdamW

    def adamw_cpu(p, g, m, v, p16, step, lr, b1, b2, eps, wd, grad_scale):   # test-only stand-in for enh_adamw_step
        O.adamw_step(p, g * grad_scale, m, v, step, lr, b1, b2, eps, wd)
    _C.adamw_step = adamw_cpu

    class Store:
        def __init__(self):
            self.p = torch.linspace(-1, 1, 4096)
            self.g = torch.randn(4096, generator=torch.Generator().manual_seed(7 + rank))
            self.m, self.v, self.step_count = torch.zeros(4096), torch.zeros(4096), 0

        def zero_grad(self):
            self.g.zero_()
    s = Store()
    local = s.g.clone()
    opt = FlatAdamW(s, lr=1e-3)
    opt.step()
    q.put((rank, s.p.numpy().copy(), local.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_discriminator_optimizer_gloo_world2():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import vitvq_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_disc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    p_ref, m, v = torch.linspace(-1, 1, 4096), torch.zeros(4096), torch.zeros(4096)
    O.adamw_step(p_ref, (torch.from_numpy(got[0][2]) + torch.from_numpy(got[1][2])) / 2, m, v, 1, 1e-3, 0.9, 0.99, 1e-8, 1e-4)
    for _, p_new, _ in got:
        assert torch.allclose(torch.from_numpy(p_new), p_ref, atol=1e-7)   # both ranks applied the MEAN gradient


def _autograd_sync_worker(rank, world, port, q):
    """AutogradGradSync: buckets fire from post-accumulate-grad hooks DURING a real autograd backward (the discriminator's path), twice-used
    parameters (real + fake pass) included, an unused parameter left to the gap pass, a no_sync micro-batch in front, and FlatAdamW waits in step()"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in ("enhancing-transformers_amd", "oracle"):
        sys.path.insert(0, os.path.join(root, p))
    import vitvq_oracle as O
    from enhancing import _C
    from enhancing.engine.optim import FlatAdamW

    def adamw_cpu(p, g, m, v, p16, step, lr, b1, b2, eps, wd, grad_scale):
        O.adamw_step(p, g * grad_scale, m, v, step, lr, b1, b2, eps, wd)
    _C.adamw_step = adamw_cpu
    torch.manual_seed(3)
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1))
            self.unused = torch.nn.Linear(8, 8)      # never reached by the loss: its slice must still be synchronised (zeros)

        def forward(self, x):
            return self.body(x)
    net = Net()

    class Store:    # ParamStore's contract on CPU: flat p / g, parameters and .grad are views, 64-element aligned offsets in registration order
        def __init__(self, module):
            params = [(n, p) for n, p in module.named_parameters()]
            self.offsets, tot = {}, 0
            for n, p in params:
                self.offsets[n] = (tot, p.numel(), p.shape); tot += (p.numel() + 63) // 64 * 64
            self.p, self.g = torch.zeros(tot), torch.zeros(tot)
            self.m, self.v, self.step_count = torch.zeros(tot), torch.zeros(tot), 0
            for n, p in params:
                off, cnt, shp = self.offsets[n]
                self.p[off:off + cnt].view(shp).copy_(p.data); p.data = self.p[off:off + cnt].view(shp); p.grad = self.g[off:off + cnt].view(shp)

        def zero_grad(self):
            self.g.zero_()
    st = Store(net)
    opt = FlatAdamW(st, lr=1e-3)
    first = opt.attach_sync(net, min_bucket_elems=128)
    sync = opt.attach_sync(net, min_bucket_elems=128)      # a re-attach (Trainer.fit called twice) must REPLACE the hooks: every bucket travels once (bytes check below)
    assert first is not sync and not first._hooks and st._grad_sync is sync
    assert len(sync.buckets) >= 3 and sync.buckets[0][1] == st.g.numel() and sync.buckets[-1][0] == 0
    for (b0, e0), (b1, e1) in zip(sync.buckets, sync.buckets[1:]):
        assert b0 == e1                                           # contiguous, descending: the buffer is tiled exactly once
    g = torch.Generator().manual_seed(50 + rank)
    xs = [torch.randn(8, 16, generator=g) for _ in range(4)]

    def loss_of(xa, xb):
        return torch.nn.functional.softplus(net(xa)).mean() + torch.nn.functional.softplus(-net(xb)).mean()    # two passes: every weight is used twice
    local = torch.zeros_like(st.g)
    # window of two micro-batches: the first under no_sync, the second fires the buckets
    sync.enabled = False
    loss_of(xs[0], xs[1]).backward()
    assert sync.bytes_reduced == 0 and not sync._handles
    sync.enabled = True
    snap = st.g.clone()
    loss_of(xs[2], xs[3]).backward()
    fired = len(sync._done)
    # what this rank contributed = its two micro-batch gradients (recomputed without the sync)
    ref = [torch.autograd.grad(loss_of(xs[0], xs[1]) + loss_of(xs[2], xs[3]), [p for _, p in net.named_parameters() if "unused" not in _])]
    i = 0
    for n, p in net.named_parameters():
        off, cnt, shp = st.offsets[n]
        if "unused" not in n:
            local[off:off + cnt] = ref[0][i].reshape(-1); i += 1
    p_before = st.p.clone()
    opt.step()
    q.put((rank, st.p.numpy().copy(), p_before.numpy().copy(), local.numpy().copy(), fired, len(sync.buckets), sync.gap_elems, sync.bytes_reduced, st.g.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_autograd_hooked_bucket_sync_gloo_world2():
    """the discriminator's gradient all-reduce is bucketed and issued from autograd hooks behind its backward, not one blocking
    all-reduce in front of AdamW; the result is the mean-gradient AdamW step on both ranks."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import vitvq_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_autograd_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=90) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = got[0][8]
    mean = (torch.from_numpy(got[0][3]) + torch.from_numpy(got[1][3])) / 2
    p_ref, m, v = torch.from_numpy(got[0][2]).clone(), torch.zeros(n), torch.zeros(n)
    O.adamw_step(p_ref, mean, m, v, 1, 1e-3, 0.9, 0.99, 1e-8, 1e-4)
    for _, p_new, _, _, fired, nb, gap, nbytes, _ in got:
        assert torch.allclose(torch.from_numpy(p_new), p_ref, atol=2e-7)
        assert 1 <= fired < nb            # the bucket holding the unused layer never fires by itself ...
        assert 0 < gap < n                # ... finish() reduces it, and counts it
        assert nbytes == 4 * n            # every element of the flat buffer travelled exactly once in the window
    assert torch.equal(torch.from_numpy(got[0][1]), torch.from_numpy(got[1][1]))


def test_bucket_cover_at_base_config():
    """BASELINE config 3 (base towers, DDP): the prefixes the backward schedule announces are contiguous slices of the flat gradient buffer that
    (a) arrive in reverse layer order, (b) cover the buffer exactly once (gap_elems == 0 by construction), (c) give ~28 MB fp32 buckets per
    transformer layer — checked on the REAL base-size module tree (host arithmetic only; no device needed)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "enhancing-transformers_amd"))
    from enhancing.engine.stage1 import ParamStore, backward_unit_order
    from enhancing.modules.stage1.layers import ViTDecoder, ViTEncoder
    from enhancing.modules.stage1.quantizers import VectorQuantizer
    kw = dict(image_size=256, patch_size=8, dim=768, depth=12, heads=12, mlp_dim=3072)
    mods = torch.nn.ModuleDict(dict(encoder=ViTEncoder(**kw), decoder=ViTDecoder(**kw), pre_quant=torch.nn.Linear(768, 32), post_quant=torch.nn.Linear(32, 768),
                                    quantizer=VectorQuantizer(32, 8192)))
    names, offsets, total = ParamStore.layout([(n, p) for n, p in mods.named_parameters() if p.requires_grad])
    assert abs(total - 170.664e6) < 1e5                                   # 170.66 M trainable parameters (+ alignment padding)
    order = backward_unit_order(12, 12)
    slices = [ParamStore.slice_from(names, offsets, p) for p in order]
    dec_layers = [s for p, s in zip(order, slices) if p.startswith("decoder.transformer.layers.")]
    assert [b for b, _ in dec_layers] == sorted((b for b, _ in dec_layers), reverse=True)      # reverse layer order = descending offsets
    cover = sorted(slices)
    assert cover[0][0] == 0 and cover[-1][1] == total
    assert all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1)), "slices must tile the buffer: no gap, no overlap"
    per_layer = {e - b for b, e in dec_layers}
    assert len(per_layer) == 1 and abs(per_layer.pop() * 4 - 28.3e6) < 0.5e6                  # 7.09 M parameters = 28.3 MB fp32 per layer bucket


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enhancing-transformers_amd"))
    from enhancing.engine.ddp import GradSync
    store = FakeStore(rank)
    local = store.g.clone()
    sync = GradSync(store, min_bucket_elems=2000, compress="bf16")
    for prefix in ["decoder.b.", "decoder.a.", "quantizer.", "encoder.b.", "encoder.a."]:
        sync.layer_done(prefix)
    sync.finish()
    q.put((rank, store.g.numpy().copy(), local.numpy().copy(), sync.bytes_reduced))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_bf16_buckets_world2():
    """optional bf16 gradient buckets: half the bytes on the wire, sum accurate to bf16 rounding"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(((r, torch.from_numpy(a), torch.from_numpy(b), n) for r, a, b, n in (q.get(timeout=120) for _ in range(2))), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = got[0][2] + got[1][2]
    for _, reduced, _, nbytes in got:
        assert nbytes == total.numel() * 2
        assert ((reduced - total).norm() / total.norm()).item() <= 6e-3
    assert torch.equal(got[0][1], got[1][1])


def _val_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enhancing-transformers_amd"))
    from enhancing.engine.trainer import Trainer
    from enhancing.modules.stage1.vitvqgan import ViTVQ

    class Stub:                      # the logging half of ViTVQ without an engine: its real `log` (the sync_dist all-reduce) on a CPU object
        def __init__(self):
            self.logged = {}
        log = ViTVQ.log
        log_dict = ViTVQ.log_dict

        def validation_step(self, batch, batch_idx):
            v = batch["image"].float().mean()                                        # a per-rank, per-batch value
            self.log("val/rec_loss", v, sync_dist=True)                              # vitvqgan.py:137
            self.log("val/total_loss", v * 2, sync_dist=True)                        # vitvqgan.py:138
            self.log_dict({"val/quant_loss": v * 3})                                 # vitvqgan.py:142: no sync_dist -> rank-local
            return self.logged

    # rank r sees batches of sizes 4, 4, 2 whose pixels equal 10*r + batch index
    batches = [{"image": torch.full((n, 3, 2, 2), 10.0 * rank + i)} for i, n in enumerate((4, 4, 2))]
    data = types.SimpleNamespace(val_dataloader=lambda: batches)
    tr = Trainer()
    tr.rank, tr.world = rank, world
    out = tr.validate(Stub(), data)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_sync_dist_mean_and_epoch_aggregation_gloo_world2():
    """reference vitvqgan.py:137-138 (`sync_dist=True`, on_epoch=True): val/rec_loss and val/total_loss are cross-rank means per batch and batch-size
    weighted means over the epoch — the same number on every rank, equal to the mean over the whole validation set; the keys logged without sync_dist
    stay rank-local (vitvqgan.py:142)."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = [4, 4, 2]
    local = lambda r: sum((10.0 * r + i) * n for i, n in enumerate(w)) / sum(w)
    glob = (local(0) + local(1)) / 2
    for r in range(world):
        assert abs(got[r]["val/rec_loss"] - glob) < 1e-5 and abs(got[r]["val/total_loss"] - 2 * glob) < 1e-5
        assert abs(got[r]["val/quant_loss"] - 3 * local(r)) < 1e-5
        assert got[r]["val_images_per_rank"] == 10
