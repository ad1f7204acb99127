"""One-GPU contention experiment for the data-parallel path: what does a collective's kernel that HOLDS k CUs cost the training step?

A side stream runs enh_debug_occupy_cus (k workgroups, each owning a CU's whole LDS, asleep on the wall clock) for the duration of the measured steps —
the stand-in for RCCL's channel workgroups during an all-reduce.  Measured at the headline configuration (base, 128 images): the AE step under
  * the STATIC tile partition of the persistent GEMMs (round 3: grid = CU count, workgroup b walks b, b + grid, ...),
  * the DYNAMIC schedule (tiles claimed from per-XCD queues: a workgroup that gets its CU late finds the queue empty),
  * the dynamic schedule + a CU budget of (CUs - k) for the one-round split-K weight-gradient plans,
and per kernel role (one GEMM call each).   python tools/comm_contention.py [--batch 128] [--steps 3]  -> table on stdout (+ gpurun_out/r04_comm_contention.txt)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing import _C  # noqa: E402
from enhancing.utils.general import get_config_from_file, initialize_from_config, set_seed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--ks", type=str, default="0,4,8,16,32")
    a = ap.parse_args()
    ks = [int(k) for k in a.ks.split(",")]
    dev = torch.device("cuda:0")
    set_seed(0)
    cfg = get_config_from_file(os.path.join(ROOT, "configs", "imagenet_vitvq_base.yaml"))
    model = initialize_from_config(cfg.model)
    eng = model.engine
    B = a.batch
    x = torch.rand(B, 3, 256, 256, device=dev)
    side = torch.cuda.Stream()
    n_cu = _C.get_cu_budget()
    lines = [f"# one MI355X ({n_cu} CUs), imagenet_vitvq_base AE step at {B} images, {a.steps} timed steps per cell; k = CUs held by a side-stream kernel for the whole window"]

    def step():
        eng.forward_backward(x, w_l1=0.0, w_l2=1.0, codebook_weight=1.0)
        eng.optimizer_step(4.5e-6)

    def timed(fn, n, k, est_ms):
        torch.cuda.synchronize()
        if k:
            with torch.cuda.stream(side):
                _C.occupy_cus(k, min(est_ms * 1.5 + 50.0, 2000.0), side)
            time.sleep(0.01)          # the holder is resident before the first launch of the window
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    for _ in range(2):
        step()
    modes = [("static partition", 0, False), ("dynamic queues", 1, False), ("dynamic + CU budget", 1, True)]
    base = {}
    lines.append(f"{'schedule':<22}" + "".join(f"{'k=' + str(k):>16}" for k in ks))
    for name, dyn, budget in modes:
        _C.gemm_set_scheduler(bool(dyn))
        row = f"{name:<22}"
        for k in ks:
            _C.set_cu_budget(n_cu - k if (budget and k) else 0)
            step()
            ms = timed(step, a.steps, k, 230.0 * a.steps * (2 if k else 1))
            if k == 0:
                base[name] = ms
            row += f"{ms:9.1f} ms {100 * (ms / base[name] - 1):+5.1f}%"[:16].rjust(16)
        _C.set_cu_budget(0)
        lines.append(row)
    # per kernel role: one call each, M = B * 1024 tokens
    M, D, H3, MLP = B * 1024, 768, 2304, 3072
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    a_d, a_mlp, w_qkv, w_fc1, w_fc2 = bf(M, D), bf(M, MLP), bf(H3, D), bf(MLP, D), bf(D, MLP)
    o_qkv, o_hid, o_x = torch.empty(M, H3, dtype=torch.bfloat16, device=dev), torch.empty(M, MLP, dtype=torch.bfloat16, device=dev), torch.empty(M, D, device=dev)
    res, bias_d, bias_m = torch.randn(M, D, device=dev), torch.randn(D, device=dev), torch.randn(MLP, device=dev)
    gw = torch.zeros(MLP, D, device=dev)
    roles = [
        ("qkv forward (w256r bf16)", lambda: _C.gemm(a_d, w_qkv, M, H3, D, out_bf16=o_qkv)),
        ("fc1 + tanh (w256r)", lambda: _C.gemm(a_d, w_fc1, M, MLP, D, bias=bias_m, act=_C.ACT_TANH, out_bf16=o_hid)),
        ("fc2 + bias + res (w256p f32)", lambda: _C.gemm(a_mlp, w_fc2, M, D, MLP, bias=bias_d, res=res, res_rows=M, out_f32=o_x)),
        ("fc2 dgrad * tanh' (w256p)", lambda: _C.gemm(a_d, w_fc2, M, MLP, D, trans_b=True, act=_C.ACT_DTANH, aux=a_mlp, out_bf16=o_hid)),
        ("fc1 wgrad (split-K w256)", lambda: _C.gemm(a_mlp, a_d, MLP, D, M, trans_a=True, trans_b=True, accumulate=True, out_f32=gw)),
    ]
    lines.append("")
    lines.append("# per kernel role, ms per launch (10 launches per cell)")
    lines.append(f"{'role / schedule':<44}" + "".join(f"{'k=' + str(k):>10}" for k in ks))
    for rname, fn in roles:
        for name, dyn, budget in modes:
            _C.gemm_set_scheduler(bool(dyn))
            row = f"{rname + ' / ' + name:<44}"
            for k in ks:
                _C.set_cu_budget(n_cu - k if (budget and k) else 0)
                fn(); fn()
                row += f"{timed(fn, 10, k, 25.0):10.3f}"
            _C.set_cu_budget(0)
            lines.append(row)
    _C.gemm_set_scheduler(True)
    out = "\n".join(lines)
    print(out)
    od = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(od):
        open(os.path.join(od, "r04_comm_contention.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
