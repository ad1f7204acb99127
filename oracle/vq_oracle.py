"""ctypes front-end of oracle/vq_oracle.c (CPU ORACLE — test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    so = os.path.join(_HERE, "libvq_oracle.so")
    src = os.path.join(_HERE, "vq_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.vq_oracle_forward.restype = ctypes.c_int
        _LIB.vq_oracle_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB.vq_oracle_max_threads.restype = ctypes.c_int
    return _LIB


def forward(z: np.ndarray, E: np.ndarray, beta: float = 0.25, depth: int = 1, use_norm: bool = True, nthreads: int = 0):
    """z [M,32] f32, E [K,32] f32 -> (zq [M,32] f32, idx [M,depth] i64, loss float32)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    E = np.ascontiguousarray(E, dtype=np.float32)
    M, K = z.shape[0], E.shape[0]
    assert z.shape[1] == 32 and E.shape[1] == 32
    zq = np.empty_like(z)
    idx = np.empty((M, depth), dtype=np.int64)
    loss = np.zeros(1, dtype=np.float32)
    rc = lib().vq_oracle_forward(z.ctypes.data, E.ctypes.data, M, K, beta, depth, int(use_norm),
                                 zq.ctypes.data, idx.ctypes.data, loss.ctypes.data, nthreads)
    assert rc == 0
    return zq, idx, loss[0]
