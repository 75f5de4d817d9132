import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def grouped(npz):
    """'case/key' -> {case: {key: array}}"""
    out = {}
    for k in npz.files:
        if "/" in k:
            case, key = k.split("/", 1)
            out.setdefault(case, {})[key] = npz[k]
    return out


@pytest.fixture(scope="session")
def g1_cases():
    return grouped(load_golden("g1_kernel_kats.npz"))


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_config1.npz")


CONFIG1 = dict(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30, LESPcrit=0.2, Naca="0012")


@pytest.fixture(scope="module")
def exp_eng():
    """An engine on the MEASUREMENT build of the library (libludvm_hip_exp.so, -DLUDVM_EXPERIMENTS): the same kernels, plus
    the codes and environment switches that force a kernel variant at sizes where the product's rule would not pick it.
    The product build refuses them (tests/test_cabi.py)."""
    from ludvm_amd import Engine, _ffi
    e = Engine(0, lib_path=_ffi.EXP_LIB_PATH)
    yield e
    e.close()
