"""ViT-VQGAN / RQ-VAE stage-1 tokenizer with the reference's constructor, method set and state-dict layout
(reference enhancing/modules/stage1/vitvqgan.py:25-188), running on the MI355X HIP engine.

Differences from the reference, all forced by the environment or by the tier's scope:
  * pytorch-lightning is not installable here, so this is a plain ``nn.Module`` that implements the Lightning
    *protocol* the reference relies on (``training_step(batch, batch_idx, optimizer_idx)``,
    ``validation_step``, ``configure_optimizers``, ``log`` / ``log_dict``, ``global_step``, ``learning_rate``);
    the in-repo trainer (``enhancing.engine.trainer``) drives it the way ``pl.Trainer.fit`` does.
  * ``training_step`` runs forward AND backward and leaves the gradients in ``param.grad`` (views of one flat buffer per
    optimizer); it returns the detached loss.  optimizer_idx 0 with a pixel + codebook loss is the fused schedule; with a
    discriminator in the loss it is forward -> loss module -> autograd, and optimizer_idx 1 is the discriminator step (with the
    frozen / trained parameter sets Lightning's ``toggle_optimizer`` would produce).  A loss that needs LPIPS raises on construction
    (SURVEY.md §8f rank 2).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ...utils.general import initialize_from_config
from .layers import ViTDecoder as Decoder
from .layers import ViTEncoder as Encoder
from .quantizers import GumbelQuantizer, VectorQuantizer


def _get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


class ViTVQ(nn.Module):
    def __init__(self, image_key: str, image_size: int, patch_size: int, encoder, decoder, quantizer, loss,
                 path: Optional[str] = None, ignore_keys: List[str] = list(), scheduler=None) -> None:
        super().__init__()
        self.path = path
        self.ignore_keys = ignore_keys
        self.image_key = image_key
        self.scheduler = scheduler
        self.learning_rate = 4.5e-6  # main.py:24,41 overwrite this with --base_lr
        self.global_step = 0
        self.logged: Dict[str, Any] = {}

        self.loss = initialize_from_config(loss)
        self.encoder = Encoder(image_size=image_size, patch_size=patch_size, **encoder)
        self.decoder = Decoder(image_size=image_size, patch_size=patch_size, **decoder)
        self.quantizer = VectorQuantizer(**quantizer)
        # nn.Linear keeps torch's default init, as the reference does (vitvqgan.py:38-39); used as containers
        self.pre_quant = nn.Linear(_get(encoder, "dim"), _get(quantizer, "embed_dim"))
        self.post_quant = nn.Linear(_get(quantizer, "embed_dim"), _get(decoder, "dim"))
        self._engine = None
        self.precision = None  # None -> ENH_PRECISION or "bf16"; set to "fp32" BEFORE first use for the exact (parity) mode
        # encoder forward of the bf16 product path: "bf16" | "x3" (split-bf16 operands, ~1e-5: codes follow the fp32 reference).  None -> the engine's
        # defaults (ENH_ENCODER_PRECISION / "bf16" for training and reconstruction; ENH_CODES_PRECISION / "x3" for encode_codes)
        self.encoder_precision = None
        self.codes_precision = None
        self.decoder_precision = None     # "bf16" | "x3": post_quant .. to_pixel forward (None -> ENH_DECODER_PRECISION / "bf16")

        if path is not None:
            self.init_from_ckpt(path, ignore_keys)

    # ---- engine binding ----------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from ...engine.stage1 import Stage1Engine
            self._engine = Stage1Engine(self, precision=self.precision, encoder_precision=self.encoder_precision, codes_precision=self.codes_precision,
                                        decoder_precision=self.decoder_precision)
        return self._engine

    @property
    def device(self) -> torch.device:
        return self._engine.device if self._engine is not None else torch.device("cpu")

    def init_from_ckpt(self, path: str, ignore_keys: List[str] = list()):
        """reference vitvqgan.py:50-59: torch.load(path)['state_dict'], prefix filter, strict=False."""
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            for ik in ignore_keys:
                if k.startswith(ik):
                    print("Deleting key {} from state_dict.".format(k))
                    del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._engine is not None:
            self._engine.store.refresh_shadows()  # bf16 operand copies follow the fp32 masters
        return out

    # ---- reference API -----------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(dec, diff) — reference vitvqgan.py:44-48.  Under torch.no_grad() nothing is saved; with grad enabled the outputs
        are connected to autograd (one outstanding forward per batch size) so any loss module on top can call .backward()."""
        if torch.is_grad_enabled():
            return self.engine.differentiable_forward(x)
        xrec, qloss, _ = self.engine.reconstruct(x)
        return xrec, qloss

    def _reconstruct_detached(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        xrec, qloss, _ = self.engine.reconstruct(x)
        return xrec, qloss

    def encode(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(quant, emb_loss) — reference vitvqgan.py:61-66"""
        eng = self.engine
        h = self.pre_quant_tokens(x)
        quant, emb_loss, _ = self.quantizer(h)
        return quant, emb_loss

    def pre_quant_tokens(self, x: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """h = pre_quant(encoder(x)) as f32 [B, N, embed_dim] (the quantizer's input: op-boundary parity point).  precision: "bf16" | "x3" for
        this call (default: the engine's encoder_precision)."""
        eng = self.engine
        x = eng._check_img(x)
        eng._invalidate_saved()
        with torch.no_grad():
            b = eng._encode_tokens(x, save=False, x3=(precision or eng.encoder_precision) == "x3")
            h = eng._pre_quant(b, x.shape[0])
        return h.view(x.shape[0], eng.n_tok, eng.ed).clone()

    def decode(self, quant: torch.Tensor) -> torch.Tensor:
        """reference vitvqgan.py:68-72"""
        return self.engine.decode_from_quant(quant)

    def encode_codes(self, x: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        """reference vitvqgan.py:74-79 -> int64 [B, N] or [B, N, D].  The encoder runs on split-bf16 ("x3") operands by default, so the codes are the
        fp32 reference's up to its own near-ties; precision="bf16" selects the faster single-pass encoder (~2 % of the codes differ)."""
        return self.engine.encode_codes(x, precision)

    def decode_codes(self, code: torch.Tensor) -> torch.Tensor:
        """reference vitvqgan.py:81-90"""
        self.engine  # make sure parameters live on the device
        quant = self.quantizer.lookup(code.to(self.device))
        return self.decode(quant)

    def get_input(self, batch, key: str = 'image') -> Any:
        """reference vitvqgan.py:92-99"""
        x = batch[key]
        if len(x.shape) == 3:
            x = x[..., None]
        if x.dtype == torch.double:
            x = x.float()
        return x.contiguous()

    # ---- Lightning protocol ------------------------------------------------------------------
    def log(self, name: str, value, sync_dist: bool = False, **_) -> None:
        """Lightning's ``self.log``.  ``sync_dist=True`` (reference vitvqgan.py:137-138, the only collective outside DDP's gradient all-reduce,
        SURVEY.md §8e): the value is replaced by its MEAN over the ranks of the default process group before it is recorded."""
        if sync_dist and torch.is_tensor(value):
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                value = value.detach().clone().float()
                dist.all_reduce(value, op=dist.ReduceOp.SUM)
                value /= dist.get_world_size()
        self.logged[name] = value

    def log_dict(self, d: Dict[str, Any], **_) -> None:
        self.logged.update(d)

    def _fusable_loss(self) -> bool:
        """the pixel + codebook losses of this package are fused into the unpatchify kernel; anything else goes through autograd"""
        from ...losses.vqperceptual import VQLPIPS
        return isinstance(self.loss, VQLPIPS) and float(getattr(self.loss, "perceptual_weight", 0.0)) == 0.0 and \
            not hasattr(self.loss, "discriminator")

    def _loss_weights(self) -> Tuple[float, float, float]:
        L = self.loss
        return float(getattr(L, "loglaplace_weight", 0.0)), float(getattr(L, "loggaussian_weight", 1.0)), float(getattr(L, "codebook_weight", 1.0))

    # ---- HIP-graph replay of the loss-module path (both optimizers of the adversarial protocol) ---------------------------------------------
    def _step_variant(self, batch_idx: int, optimizer_idx: int) -> tuple:
        """what the HOST decides inside one training_step of the loss-module path (everything else is device work): whether the discriminator is
        active yet (vqperceptual.py:109) and whether this discriminator step carries the lazy R1 penalty (vqperceptual.py:157)"""
        L = self.loss
        disc_on = self.global_step >= int(getattr(L, "discriminator_iter_start", 0))
        r1 = bool(optimizer_idx == 1 and L.training and disc_on and hasattr(L, "discriminator") and batch_idx % int(getattr(L, "do_r1_every", 16)) == 0)
        return bool(disc_on), r1

    def _graphed_training_step(self, batch, batch_idx: int, optimizer_idx: int, zero_grad: bool):
        """training_step for the loss-module path (forward -> loss module -> autograd: the protocol every shipped reference config runs,
        configs/imagenet_vitvq_*.yaml) captured ONCE per (optimizer, batch shape, host-side variant) into a HIP graph and replayed: at the reference
        yaml's 2 images per GPU the step is ~2000 launches plus the autograd engine's bookkeeping and the host cannot issue them as fast as the GPU
        retires them.  The captured region is the eager code itself (warmed up twice on a side stream first), so the results are bit-identical; the
        returned loss and the logged tensors are the graph's static outputs (valid until the next replay)."""
        from ... import _C
        x = self.get_input(batch, self.image_key)
        eng = self.engine
        key = (optimizer_idx, tuple(x.shape), bool(zero_grad), bool(self.loss.training)) + self._step_variant(batch_idx, optimizer_idx)
        graphs = self.__dict__.setdefault("_step_graphs", {})
        entry = graphs.get(key)
        if entry is None and len(graphs) >= 12:
            return self._training_step_eager(batch, batch_idx, optimizer_idx, zero_grad)
        if entry is None:
            static_x = torch.empty(x.shape, dtype=torch.float32, device=eng.device)
            static_x.copy_(x)
            sb = {self.image_key: static_x}
            stores = [eng.store] + ([self.loss.disc_store(eng.device)] if hasattr(self.loss, "discriminator") else [])
            saved = [st.g.clone() for st in stores]            # the warm-up passes must not leak into an accumulation window
            # Operand images DERIVED from the parameters on the host side of the eager code (the discriminator's packed convolution weights, cached
            # between optimizer steps: losses/op/conv_nhwc.py) must be rebuilt INSIDE every graph: a graph that found the cache warm would read, on every
            # replay, the image of the weights as they were at capture (caught by the bit-identity test in the full suite: optimizer 1's no-R1 graph,
            # captured in a step whose optimizer-0 pass was already a replay).  So the cache is dropped before the warm-up, before the capture and after it.
            from ...losses.op.conv_nhwc import invalidate_packed_weights
            cur, side = torch.cuda.current_stream(), torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    invalidate_packed_weights()
                    self._training_step_eager(sb, batch_idx, optimizer_idx, zero_grad)
            cur.wait_stream(side)
            for st, g0 in zip(stores, saved):
                st.g.copy_(g0)
            before = dict(self.logged)
            invalidate_packed_weights()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._training_step_eager(sb, batch_idx, optimizer_idx, zero_grad)
            invalidate_packed_weights()       # (what the capture cached lives in the graph's memory pool and is rewritten by every replay)
            for st, g0 in zip(stores, saved):
                st.g.copy_(g0)
            logged = {k: v for k, v in self.logged.items() if k not in before or before[k] is not v}
            keep = (list(_C._GEMM_WS.values()), list(_C._WS.values()))      # workspaces whose addresses the graph has baked in
            entry = graphs[key] = (graph, static_x, out, logged, keep)
        graph, static_x, out, logged, _ = entry
        static_x.copy_(x, non_blocking=True)
        graph.replay()
        eng._invalidate_saved()
        self.logged.update(logged)
        return out

    def training_step(self, batch, batch_idx: int, optimizer_idx: int = 0, zero_grad: bool = True):
        """reference vitvqgan.py:101-127 (see _training_step_eager).  With engine.use_graphs (ENH_GRAPHS=1 / bench.py --graphs) the loss-module path
        replays from HIP graphs; the fused pixel + codebook step has its own graph inside the engine (forward_backward_graphed)."""
        from ... import _C
        eng = self.engine
        if eng.use_graphs and _C.TIMER is None and eng.comm is None and not self._fusable_loss() and (optimizer_idx == 0 or hasattr(self.loss, "discriminator")):
            return self._graphed_training_step(batch, batch_idx, optimizer_idx, zero_grad)
        return self._training_step_eager(batch, batch_idx, optimizer_idx, zero_grad)

    def _training_step_eager(self, batch, batch_idx: int, optimizer_idx: int = 0, zero_grad: bool = True):
        """reference vitvqgan.py:101-127.  optimizer_idx 0 = autoencoder: forward + backward run fused on the engine
        and the returned loss is detached (the gradients are already in param.grad)."""
        x = self.get_input(batch, self.image_key)
        if optimizer_idx == 0 and not self._fusable_loss():
            # generic path, exactly the reference's: forward -> loss module -> autograd backward (vitvqgan.py:103-115)
            if zero_grad:
                self.engine.store.zero_grad()
            xrec, qloss = self(x)
            frozen = [p for p in self.loss.parameters() if p.requires_grad]   # Lightning's toggle_optimizer: only optimizer 0's parameters train
            for p in frozen:
                p.requires_grad_(False)
            try:
                aeloss, log_dict_ae = self.loss(qloss, x.to(xrec.device), xrec, optimizer_idx, self.global_step, batch_idx,
                                                last_layer=self.decoder.get_last_layer(), split="train")
                self.engine.scale_loss(aeloss).backward()      # (fp16 engine: the loss scale of the 16-bit backward; the identity for bf16 / fp32)
            finally:
                for p in frozen:
                    p.requires_grad_(True)
            self.log("train/total_loss", aeloss.detach())
            self.log_dict({k: v for k, v in log_dict_ae.items() if k != "train/total_loss"})
            return aeloss.detach()
        if optimizer_idx == 0:
            w1, w2, cw = self._loss_weights()
            out = self.engine.forward_backward_graphed(x, w_l1=w1, w_l2=w2, codebook_weight=cw, zero_grad=zero_grad)   # eager unless engine.use_graphs
            log = {"train/total_loss": out["loss"], "train/quant_loss": out["quant_loss"], "train/rec_loss": out["rec_loss"],
                   "train/loglaplace_loss": out["loglaplace_loss"], "train/loggaussian_loss": out["loggaussian_loss"],
                   "train/perceptual_loss": torch.zeros((), device=out["loss"].device)}
            self.log("train/total_loss", out["loss"])
            self.log_dict({k: v for k, v in log.items() if k != "train/total_loss"})
            return out["loss"]
        if optimizer_idx == 1:
            if not hasattr(self.loss, "discriminator"):
                return None
            # reference vitvqgan.py:117-127; the reconstruction enters the discriminator loss detached, so no autoencoder graph is kept
            xrec, qloss = self._reconstruct_detached(x)
            if zero_grad:
                self.loss.disc_store(xrec.device).zero_grad()
            discloss, log_dict_disc = self.loss(qloss, x.to(xrec.device), xrec, optimizer_idx, self.global_step, batch_idx,
                                                last_layer=self.decoder.get_last_layer(), split="train")
            if torch.is_tensor(discloss) and discloss.requires_grad:
                self.loss.scale_disc_loss(discloss).backward()      # (fp16 loss networks: scaled backward, unscaled inside the discriminator's AdamW launch)
            self.log("train/disc_loss", log_dict_disc["train/disc_loss"])
            self.log_dict({k: v for k, v in log_dict_disc.items() if k != "train/disc_loss"})
            return log_dict_disc["train/disc_loss"]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int) -> Dict:
        """reference vitvqgan.py:129-150"""
        x = self.get_input(batch, self.image_key)
        xrec, qloss = self(x)
        aeloss, log = self.loss(qloss, x.to(xrec.device), xrec, 0, self.global_step, batch_idx, last_layer=self.decoder.get_last_layer(), split="val")
        self.log("val/rec_loss", log["val/rec_loss"], sync_dist=True)     # vitvqgan.py:137-138: on_step + on_epoch, sync_dist=True
        self.log("val/total_loss", aeloss, sync_dist=True)
        self.log_dict({k: v for k, v in log.items() if k not in ("val/rec_loss", "val/total_loss")})
        if hasattr(self.loss, "discriminator"):   # vitvqgan.py:144-148
            _, log_disc = self.loss(qloss, x.to(xrec.device), xrec, 1, self.global_step, batch_idx, last_layer=self.decoder.get_last_layer(), split="val")
            self.log_dict(log_disc)
        return self.logged

    def configure_optimizers(self):
        """reference vitvqgan.py:152-178: one AdamW(lr, betas=(0.9, 0.99), weight_decay=1e-4) over encoder + decoder +
        pre/post_quant + quantizer as a single group -> here ONE fused launch over the flat buffer."""
        from ...engine.optim import FlatAdamW, FusedAdamW
        optimizers = [FusedAdamW(self.engine, lr=self.learning_rate, betas=(0.9, 0.99), weight_decay=1e-4)]
        if hasattr(self.loss, "discriminator"):   # vitvqgan.py:163-164: a second AdamW with the same hyper-parameters
            optimizers.append(FlatAdamW(self.loss.disc_store(self.engine.device), lr=self.learning_rate, betas=(0.9, 0.99), weight_decay=1e-4))
        schedulers = []
        if self.scheduler is not None:
            self.scheduler.params.start = self.learning_rate
            sched = initialize_from_config(self.scheduler)
            schedulers = [{"scheduler": sched, "interval": "step", "frequency": 1} for _ in optimizers]
        return optimizers, schedulers

    @torch.no_grad()
    def log_images(self, batch, *args, **kwargs) -> Dict:
        """reference vitvqgan.py:180-188"""
        x = self.get_input(batch, self.image_key)
        xrec, _ = self(x)
        return {"originals": x, "reconstructions": xrec}


class ViTVQGumbel(ViTVQ):
    """reference vitvqgan.py:191-212: ViTVQ with a GumbelQuantizer and an optional temperature schedule.  The two towers run on the HIP schedule; the
    quantizer between them is plain torch under autograd (not a hot path of this build — no shipped stage-1 config names this class)."""

    def __init__(self, image_key: str, image_size: int, patch_size: int, encoder, decoder, quantizer, loss, path: Optional[str] = None,
                 ignore_keys: List[str] = list(), temperature_scheduler=None, scheduler=None) -> None:
        super().__init__(image_key, image_size, patch_size, encoder, decoder, quantizer, loss, None, list(), scheduler)
        self.temperature_scheduler = initialize_from_config(temperature_scheduler) if temperature_scheduler else None
        self.quantizer = GumbelQuantizer(**quantizer)
        if path is not None:
            self.init_from_ckpt(path, ignore_keys)

    def _fusable_loss(self) -> bool:
        return False       # the fused training step contains the VectorQuantizer kernel

    def _quantize(self, h: torch.Tensor):
        self.quantizer.to(h.device)
        return self.quantizer(h)

    def _reconstruct_detached(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        with torch.no_grad():
            return self(x)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        eng = self.engine
        if torch.is_grad_enabled():
            quant, diff, _ = self._quantize(eng.differentiable_encode(x))
            return eng.differentiable_decode(quant), diff
        quant, diff, _ = self._quantize(self.pre_quant_tokens(x))
        return self.decode(quant), diff

    def encode(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        quant, emb_loss, _ = self._quantize(self.pre_quant_tokens(x))
        return quant, emb_loss

    @torch.no_grad()
    def encode_codes(self, x: torch.Tensor, precision: Optional[str] = None) -> torch.Tensor:
        return self._quantize(self.pre_quant_tokens(x, precision or self.engine.codes_precision))[2]

    def training_step(self, batch, batch_idx: int, optimizer_idx: int = 0, zero_grad: bool = True):
        if self.temperature_scheduler:
            self.quantizer.temperature = self.temperature_scheduler(self.global_step)
        # always the eager sequence: the Gumbel temperature is a HOST scalar read inside the step (gumbel_softmax(tau=...)), a captured HIP graph would
        # replay the capture-time tau while the logged 'temperature' keeps annealing; the quantizer is plain torch between the two HIP halves
        loss = self._training_step_eager(batch, batch_idx, optimizer_idx, zero_grad)
        if optimizer_idx == 0:
            self.log("temperature", self.quantizer.temperature)
        return loss
