"""Data-parallel gradient synchronisation: one process per GPU, RCCL over xGMI (torch.distributed backend
"nccl" on ROCm), bucketed and overlapped with the backward pass.

The reference's only parallelism is Lightning's ``strategy="ddp"`` (reference main.py:56): torch DDP's bucketed
gradient all-reduce (mean) per backward plus an initial parameter broadcast.  Here the gradients already live
in ONE flat fp32 buffer, so a bucket is a contiguous slice: as the static backward schedule finishes a unit
(a transformer layer, the pixel head, the quantizer ...) it calls ``layer_done(prefix)`` and that slice is
all-reduced asynchronously on RCCL's stream while the remaining backward kernels keep the compute stream
busy.  ``finish()`` (called by the optimizer step) waits for the outstanding reductions; the 1/world mean is
folded into the fused AdamW launch (``grad_scale``), so no extra pass over the gradients is needed.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a transformer-layer bucket (7.1 M params = 28 MB fp32 at
base) is large enough to run at link bandwidth and small enough that ~24 of them pipeline behind backward.

The class only needs an object with ``g`` (flat grad tensor), ``p`` (flat params) and ``slice_of(prefix)``,
so it is exercised on CPU with the gloo backend in tests/test_ddp_cpu.py.
"""
from __future__ import annotations

import time
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, store, process_group=None, min_bucket_elems: int = 1 << 20, compress: Optional[str] = None, algo: str = "allreduce",
                 comm_cus: int = 0) -> None:
        """algo = "allreduce": one all-reduce per bucket (RCCL chooses the schedule).  algo = "rs_ag": the bucket is reduce-scattered (each rank sums
        1/world of it) and all-gathered back — the direct exchange SURVEY.md §8e prefers on a fully connected xGMI node (every rank talks to all 7
        peers at once, (world-1)/world of the bucket each way, instead of a ring bounded by one link); the elements beyond a multiple of `world`
        go through a small all-reduce.  Both give the same sums.  Which is faster on 8 x MI355X is a measurement this build has not been able to make.
        compress = "bf16": buckets travel as bfloat16 (half the xGMI bytes: 341 MB instead of 683 MB per step at base); the sum is
        formed in bf16 by the collective, so this trades ~3 significant digits of the summed gradient for bandwidth — off by default.
        comm_cus > 0: while gradient buckets are in flight (first bucket of a step .. finish()) the compute library plans its CU-count-sized launches
        for (CUs - comm_cus) — the one-round split-K weight-gradient plans and the LayerNorm backward's one-workgroup-per-CU grid, which otherwise
        wait for a second round when RCCL's channel workgroups hold CUs (profiles/r04_comm_contention.txt: +45 % on those launches; the persistent
        GEMMs claim their tiles dynamically and need no budget).  Off by default: the collectives of the base config are in flight for a few per cent
        of the step, a standing reservation costs more than the occasional second round."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.store = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.min_bucket = min_bucket_elems
        self._pending: List[Tuple[int, int]] = []   # finished slices not yet flushed (coalesced when adjacent)
        self._handles = []
        self._done: List[Tuple[int, int]] = []       # every slice announced in this step
        self.bytes_reduced = 0
        self.gap_elems = 0                           # elements finish() had to reduce because nobody announced them
        self.announced: List[str] = []               # prefixes in announce order (last step), for the schedule tests
        if algo not in ("allreduce", "rs_ag"):
            raise ValueError("algo must be 'allreduce' or 'rs_ag'")
        self.algo = algo
        self._ordered = dist.get_backend(process_group) == "nccl"   # RCCL runs a group's collectives in issue order on its own stream; gloo does not promise it
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self.comm_cus = int(comm_cus)
        self._budget_on = False
        self._g16 = torch.empty_like(store.g, dtype=torch.bfloat16) if compress == "bf16" else None
        self._copyback: List[Tuple[int, int]] = []
        self._shards: List[torch.Tensor] = []
        # exposed communication per step: time finish() blocked the host (gloo waits on the host) and time the COMPUTE STREAM stood waiting for the
        # collectives' stream (RCCL: wait() is a stream dependency, the host does not block) — HIP events around the waits, read by comm_wait_ms()
        self.host_wait_ms: List[float] = []
        self._wait_events: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self.buckets_last_step = 0
        # per-bucket issue log of the step in flight / of the last finished step: (host time of the issue in ms on the process-wide time.perf_counter
        # clock, first element, elements).  Two synchronisers of one process (the autoencoder's GradSync and the discriminator's AutogradGradSync share
        # RCCL's one stream per process group) log on the SAME clock, so their lists interleave into the order the collectives were queued in —
        # bench.py's `comm` block prints them side by side (does a discriminator bucket wait behind autoencoder buckets?).
        self._issue_log: List[Tuple[float, int, int]] = []
        self.last_issue_log: List[Tuple[float, int, int]] = []

    def broadcast_parameters(self, src: int = 0) -> None:
        """DDP's initial parameter broadcast: every rank starts from rank `src`'s weights."""
        dist.broadcast(self.store.p, src=src, group=self.pg)

    # ---- called by the backward schedule -------------------------------------------------------
    def layer_done(self, prefix: str) -> None:
        b, e = self.store.slice_of(prefix)
        self.announced.append(prefix)
        self._pending.append((b, e))
        self._done.append((b, e))
        if sum(y - x for x, y in self._pending) >= self.min_bucket:
            self._flush()

    def _flush(self) -> None:
        if not self._pending:
            return
        self._pending.sort()
        merged = [list(self._pending[0])]
        for b, e in self._pending[1:]:
            if b <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([b, e])
        for b, e in merged:
            self._reduce(b, e)
        self._pending = []

    def _set_budget(self, on: bool) -> None:
        if self.comm_cus > 0 and self.store.g.is_cuda and on != self._budget_on:
            from .. import _C
            _C.set_cu_budget(max(_C.device_cus() - self.comm_cus, 8) if on else 0)
            self._budget_on = on

    def _reduce(self, b: int, e: int) -> None:
        self._issue_log.append((time.perf_counter() * 1e3, b, e - b))
        self._set_budget(True)
        view = self.store.g[b:e]
        if self._g16 is not None:
            view = self._g16[b:e]
            view.copy_(self.store.g[b:e])             # f32 -> bf16 on the compute stream, ordered before the collective
            self._copyback.append((b, e))
        self.bytes_reduced += view.numel() * view.element_size()
        main = (view.numel() // self.world) * self.world if self.algo == "rs_ag" and self.world > 1 else 0
        if main:
            shard = torch.empty(main // self.world, dtype=view.dtype, device=view.device)
            rs = dist.reduce_scatter_tensor(shard, view[:main], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if not self._ordered:
                rs.wait()
            self._handles.append(rs)
            self._handles.append(dist.all_gather_into_tensor(view[:main], shard, group=self.pg, async_op=True))
            self._shards.append(shard)            # kept alive until finish()
        if main < view.numel():
            self._handles.append(dist.all_reduce(view[main:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> None:
        """Flush what is left, all-reduce any range of the flat gradient nobody announced (safety net: a missed unit must
        never leave un-synchronised gradients; `gap_elems` records it so tests can insist on zero), and wait."""
        self._flush()
        n = self.store.g.numel()
        cur, gaps = 0, []
        for b, e in sorted(self._done):
            if b > cur:
                gaps.append((cur, b))
            cur = max(cur, e)
        if cur < n:
            gaps.append((cur, n))
        for b, e in gaps:
            self.gap_elems += e - b
            self._reduce(b, e)
        on_gpu = self.store.g.is_cuda
        if on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t0 = time.perf_counter()
        for h in self._handles:
            h.wait()
        self.host_wait_ms.append((time.perf_counter() - t0) * 1e3)
        if on_gpu:
            ev[1].record()
            self._wait_events.append(ev)
        if len(self.host_wait_ms) > 4096:        # a long training run keeps the most recent window only
            del self.host_wait_ms[:2048]
            del self._wait_events[:max(len(self._wait_events) - 2048, 0)]
        self.buckets_last_step = len(self._handles)
        self._set_budget(False)
        for b, e in self._copyback:
            self.store.g[b:e].copy_(self._g16[b:e])
        self._copyback = []
        self._handles = []
        self._shards = []
        self._done = []
        self.last_issue_log, self._issue_log = self._issue_log, []

    def issue_timeline(self, t0_ms: float = 0.0) -> List[dict]:
        """the last finished step's collectives in issue order: offset of the issue from t0_ms (host clock, time.perf_counter() * 1e3), first element, elements"""
        return [dict(t_ms=round(t - t0_ms, 3), begin=b, elems=n) for t, b, n in self.last_issue_log]

    def begin_step(self) -> None:
        self.announced = []

    def comm_wait_ms(self, last: Optional[int] = None) -> dict:
        """per-step exposed communication of the last `last` steps (all if None): `stream_ms` = how long the compute stream stood still between the end
        of backward and the last collective's completion (HIP events; what the step actually pays for the all-reduce under RCCL), `host_ms` = host
        time inside the waits (the whole cost under gloo).  Synchronises the device."""
        host = self.host_wait_ms[-last:] if last else list(self.host_wait_ms)
        stream = []
        if self._wait_events:
            torch.cuda.synchronize()
            evs = self._wait_events[-last:] if last else self._wait_events
            stream = [a.elapsed_time(b) for a, b in evs]
        mean = lambda v: sum(v) / len(v) if v else None
        return {"host_ms": mean(host), "stream_ms": mean(stream), "steps": len(host), "buckets_per_step": self.buckets_last_step}


class AutogradGradSync(GradSync):
    """The same bucketed, overlapped all-reduce for a store whose gradients are written by a torch AUTOGRAD backward (the StyleGAN discriminator, the
    second optimizer of reference vitvqgan.py:163-164, whose DDP wrap all-reduces its 115 MB of gradients bucket by bucket behind ITS backward —
    reference main.py:54-57).  There is no static schedule to announce units, so every parameter gets a post-accumulate-grad hook: the flat buffer is cut
    into contiguous buckets of >= `min_bucket_elems` elements in REVERSE registration order (the order autograd finishes them: the last layers first) and a
    bucket is handed to RCCL the moment its last parameter's gradient has been accumulated — while autograd is still differentiating the earlier blocks.
    `enabled = False` is DDP's no_sync (all but the last micro-batch of an accumulation window); parameters that received no gradient in a step are
    covered by finish()'s gap pass, as in GradSync."""

    def __init__(self, store, named_params, **kw) -> None:
        super().__init__(store, **kw)
        self.enabled = True
        order = [(n, p) for n, p in named_params if p.requires_grad and n in store.offsets]
        order.sort(key=lambda np_: store.offsets[np_[0]][0], reverse=True)
        self.buckets: List[Tuple[int, int]] = []          # [begin, end) of the flat buffer, in firing order
        self._members: List[int] = []                     # parameters per bucket
        self._hooks = []
        hi = None
        cnt = 0
        pending = []
        for i, (n, p) in enumerate(order):
            off, numel, _ = store.offsets[n]
            if hi is None:
                hi = store.g.numel()          # the highest parameter's slice runs to the end of the flat buffer (alignment padding included)
            pending.append(p)
            cnt += 1
            if hi - off >= self.min_bucket or i == len(order) - 1:
                bid = len(self.buckets)
                self.buckets.append((off, hi))
                self._members.append(cnt)
                for q in pending:
                    self._hooks.append(q.register_post_accumulate_grad_hook(self._make_hook(bid)))
                hi, cnt, pending = off, 0, []
        self._left = list(self._members)

    def _make_hook(self, bid: int):
        def hook(_param) -> None:
            if not self.enabled:
                return
            self._left[bid] -= 1
            if self._left[bid] == 0:
                b, e = self.buckets[bid]
                self._done.append((b, e))
                self._reduce(b, e)
        return hook

    def finish(self) -> None:
        super().finish()
        self._left = list(self._members)

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def init_process_group_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). -> (rank, local_rank, world)"""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("ENH_DIST_BACKEND", backend)  # test hook: gloo lets 2 ranks share one GPU (RCCL refuses)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world
