"""Launch plan for the counter passes of the landing lab (tools/r5_pmc_landing.sh): a few GEMM roles of the base step, each dense / padded / in two tile
orders, in a FIXED dispatch order printed as JSON (the summariser matches counter rows to labels by dispatch order of the gemm_bf16 kernels)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("LAB_ROUNDS", "1")
import gemm_ld_lab as G  # noqa: E402  (does not run anything at import)

M = G.M
ROLES = [("fwd qkv", "fwd", M, 2304, 768), ("fwd fc2 +res", "fwd_res", M, 768, 3072), ("dgrad fc1", "dgrad", M, 768, 3072), ("dgrad dtanh", "dgrad_dtanh", M, 3072, 768),
         ("wgrad fc1", "wgrad", 3072, 768, M)]
plan = []
for name, kind, m, n, k in ROLES:
    variants = [("dense", 0, (0, 0)), ("padded", G.PAD, (0, 0))]
    if kind != "wgrad":
        variants += [("dense order(8,0)", 0, (8, 0)), ("dense order(4,1)", 0, (4, 1))]
    for label, pad, order in variants:
        fn, _ = G.make_case(kind, m, n, k, pad)
        G.L.enh_debug_gemm_order(*order)
        torch.cuda.synchronize()
        for _ in range(3):
            fn()
            plan.append(f"{name} | {label}")
        torch.cuda.synchronize()
        del fn
        torch.cuda.empty_cache()
G.L.enh_debug_gemm_order(0, 0)
print("PLAN " + json.dumps(plan))
