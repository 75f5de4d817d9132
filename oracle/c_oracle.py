"""ctypes loader of oracle/libpair_oracle.so (TEST INFRASTRUCTURE ONLY; see pair_oracle.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libpair_oracle.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        pd = ctypes.POINTER(ctypes.c_double)
        _lib.pair_oracle_f64.argtypes = [pd, pd, pd, ctypes.c_size_t, pd, pd, ctypes.c_size_t, ctypes.c_double, pd, pd]
        _lib.pair_oracle_f64.restype = None
        _lib.pair_oracle_threads.restype = ctypes.c_int
        _lib.pair_oracle_set_threads.argtypes = [ctypes.c_int]
        _lib.pair_oracle_set_threads.restype = None
    return _lib


def threads():
    return _load().pair_oracle_threads()


def set_threads(n):
    """OpenMP threads of the following calls (rows are independent: same numbers, other wall time)."""
    _load().pair_oracle_set_threads(int(n))


def induced_velocity(circulation, xw, zw, xp, zp, v_core):
    lib = _load()
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (circulation, xw, zw, xp, zp)]
    u, w = np.empty(len(a[3])), np.empty(len(a[3]))
    p = lambda v: v.ctypes.data_as(ctypes.POINTER(ctypes.c_double))  # noqa: E731
    lib.pair_oracle_f64(p(a[0]), p(a[1]), p(a[2]), len(a[1]), p(a[3]), p(a[4]), len(a[3]), float(v_core), p(u), p(w))
    return u, w
