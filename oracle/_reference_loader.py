"""Imports the REAL reference modules from /root/reference (this container only).

TEST INFRASTRUCTURE ONLY. Used by oracle/make_golden.py to (a) validate the CPU
restatement in oracle/vitvq_oracle.py against the reference's own PyTorch code and
(b) generate the golden vectors committed under tests/golden/.  Nothing on the GPU
box may import this file: /root/reference does not exist there.

Recipe follows SURVEY.md §A.2: quantizers.py imports as-is; layers.py needs the
shim ``numpy.float = float`` (reference bug at enhancing/modules/stage1/layers.py:57
under NumPy >= 1.24).
"""
import importlib.util
import os
import sys

REF_ROOT = os.environ.get("ENH_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "enhancing/modules/stage1/quantizers.py"))


def _load(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_quantizers():
    return _load("_ref_quantizers", "enhancing/modules/stage1/quantizers.py")


def load_layers():
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float  # shim for layers.py:57
    return _load("_ref_layers", "enhancing/modules/stage1/layers.py")
