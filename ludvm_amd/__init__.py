"""ludvm_amd -- MI355X (gfx950) all-pairs vortex-induction engine behind the LUDVM method surface.

    from ludvm_amd import LUDVM
    sim = LUDVM(t0=0, tf=20, dt=5e-2, chord=1, rho=1.225, Uinf=1, Npoints=81, Ncoeffs=30,
                LESPcrit=0.2, Naca='0012')

The package holds only what the hot path needs: csrc/ (HIP kernels + the C ABI of
include/ludvm_hip.h), the ctypes binding, and the host-side mirror of the reference class.
"""
from .comm import prepare_ipc_environment as _prepare_ipc_environment

_prepare_ipc_environment()      # (a launcher announced several ranks: the IPC mode RCCL needs, before any HIP call -- comm.py)

from ._ffi import LudvmHipError  # noqa: F401,E402
from .engine import Engine  # noqa: F401,E402
from .ludvm import LUDVM, SparseHistory  # noqa: F401,E402
from .freevort import (generate_free_vortices, generate_free_single_vortex, generate_flowfield_vortices,  # noqa: F401,E402
                       generate_flowfield_turbulence)

__version__ = "0.1.0"
