"""ctypes binding of libludvm_hip.so (the C ABI declared in include/ludvm_hip.h).

There is no CPU implementation behind this module: if the shared library is missing or cannot be
loaded, `load()` raises, and so does every product entry point that needs it.
"""
import ctypes
import importlib.util
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libludvm_hip.so")
# the measurement build (-DLUDVM_EXPERIMENTS: environment switches, forced kernel variants); Engine(lib_path=EXP_LIB_PATH)
EXP_LIB_PATH = os.path.join(_HERE, "csrc", "libludvm_hip_exp.so")

OK, E_ARG, E_HIP, E_NOMEM, E_NODEVICE, E_STATE, E_COMM = range(7)
PREC_F32, PREC_F32X2, PREC_F64 = 0, 1, 2
SYM_TILE = 512
SYM_OWNER_ALIGN = 4      # tiles: an owner's block of the tile ring is a whole number of these quads (include/ludvm_hip.h)
ABI_VERSION = 5
COMM_ID_BYTES = 128
SYM_SCALE_BYTES = 32

_pd, _pf = POINTER(c_double), POINTER(c_float)

# name -> argtypes; every function returns int except ludvm_last_error (const char*).  Keep in step
# with include/ludvm_hip.h (tests/test_cabi.py checks the header, this table and the library agree).
SIGNATURES = {
    "ludvm_abi_version": [],
    "ludvm_create": [c_int, POINTER(c_void_p)],
    "ludvm_destroy": [c_void_p],
    "ludvm_last_error": [c_void_p],
    "ludvm_device_info": [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_longlong), c_char_p, c_int],
    "ludvm_set_stream": [c_void_p, c_void_p, c_int],
    "ludvm_synchronize": [c_void_p],
    "ludvm_set_tuning": [c_void_p, c_int, c_int],
    "ludvm_set_symmetric": [c_void_p, c_int],
    "ludvm_set_shard": [c_void_p, c_int, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t],
    "ludvm_set_sym_tuning": [c_void_p, c_int, c_int],
    "ludvm_comm_unique_id": [c_void_p, c_size_t],
    "ludvm_comm_init": [c_void_p, c_int, c_int, c_void_p, c_size_t, c_size_t],
    "ludvm_comm_init_all": [POINTER(c_void_p), c_int, c_size_t],
    "ludvm_comm_destroy": [c_void_p],
    "ludvm_comm_info": [c_void_p, POINTER(c_int), POINTER(c_int)],
    "ludvm_comm_allreduce_i64_dev": [c_void_p, c_void_p, c_size_t],
    "ludvm_comm_allgather_dev": [c_void_p, c_void_p, c_void_p, c_size_t],
    "ludvm_comm_allgather_host": [c_void_p, c_void_p, c_void_p, c_size_t],
    "ludvm_induce_f64": [c_void_p, _pd, _pd, _pd, c_size_t, _pd, _pd, c_size_t, c_double, c_int, _pd, _pd],
    "ludvm_spatial_order": [c_void_p, _pd, _pd, c_size_t, POINTER(ctypes.c_uint), POINTER(c_int), _pd],
    "ludvm_induce_f32": [c_void_p, _pf, _pf, _pf, c_size_t, _pf, _pf, c_size_t, c_float, _pf, _pf],
    "ludvm_induce_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t,
                             c_float, c_void_p, c_void_p],
    "ludvm_advect_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_float,
                             c_float, c_void_p, c_void_p],
    "ludvm_sym_scale_dev_f32": [c_void_p, c_void_p, c_size_t, c_float, c_void_p],
    "ludvm_sym_accumulate_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_float,
                                     c_void_p, c_void_p, c_void_p, c_void_p],
    "ludvm_advect_from_sums_dev_f32": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_size_t, c_float, c_void_p, c_void_p],
    "ludvm_wake_reserve": [c_void_p, c_size_t],
    "ludvm_wake_clear": [c_void_p],
    "ludvm_wake_size": [c_void_p, POINTER(c_size_t)],
    "ludvm_wake_truncate": [c_void_p, c_size_t],
    "ludvm_wake_append": [c_void_p, _pd, _pd, _pd, c_size_t],
    "ludvm_wake_write": [c_void_p, c_size_t, c_size_t, _pd, _pd, _pd],
    "ludvm_wake_read": [c_void_p, c_size_t, c_size_t, _pd, _pd, _pd],
    "ludvm_wake_induce_on_points": [c_void_p, c_size_t, c_size_t, _pd, _pd, c_size_t, c_double, _pd, _pd],
    "ludvm_wake_chord_sums": [c_void_p, c_size_t, c_size_t, _pd, _pd, c_size_t, _pd, _pd, c_size_t, c_double, _pd, _pd,
                              _pd, _pd],
    "ludvm_wake_advect": [c_void_p, c_double, _pd, _pd, _pd, c_size_t, c_double, c_int, _pd, _pd],
    "ludvm_wake_advect_tail": [c_void_p, c_double, _pd, _pd, _pd, c_size_t, c_double, c_int, c_size_t, _pd, _pd],
    "ludvm_wake_step": [c_void_p, _pd, _pd, _pd, c_size_t, c_double, _pd, _pd, _pd, c_size_t, c_double, c_int, _pd, _pd,
                        c_int, c_size_t, _pd, _pd, c_size_t, _pd, _pd, _pd, _pd, _pd, _pd, _pd, _pd],
    "ludvm_march_setup": [c_void_p, c_int, c_int, _pd, _pd, _pd, c_size_t],
    "ludvm_march_run": [c_void_p, c_longlong, c_longlong, c_int, _pd, _pd, _pd, c_size_t, POINTER(c_longlong)],
    "ludvm_flowfield_f32": [c_void_p, c_double, c_double, c_double, c_size_t, c_size_t, _pd, _pd, _pd, c_size_t,
                            c_double, _pf, _pf],
    "ludvm_flowfield_vorticity_f32": [c_void_p, c_double, c_double, c_double, c_size_t, c_size_t, _pd, _pd, _pd, c_size_t,
                                      c_double, _pf, _pf, _pf],
    "ludvm_flowfield_rows_f32": [c_void_p, c_double, c_double, c_double, c_size_t, c_size_t, c_size_t, c_size_t, _pd, _pd, _pd,
                                 c_size_t, c_double, _pf, _pf, _pf],
    "ludvm_flowfield_rows_f64": [c_void_p, c_double, c_double, c_double, c_size_t, c_size_t, c_size_t, c_size_t, _pd, _pd, _pd,
                                 c_size_t, c_double, _pd, _pd, _pd],
    "ludvm_flowfield_dev_f32": [c_void_p, c_double, c_double, c_double, c_size_t, c_size_t, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_float, c_void_p, c_void_p],
    "ludvm_vorticity_f32": [c_void_p, _pf, _pf, c_size_t, c_size_t, c_double, _pf],
    "ludvm_vorticity_dev_f32": [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_void_p],
    "ludvm_fixed_point_probe": [c_void_p, _pf, c_size_t, c_int, POINTER(c_longlong)],
    "ludvm_kernel_timing": [c_void_p, c_int],
    "ludvm_kernel_time_ms": [c_void_p, c_int, POINTER(c_double), POINTER(c_longlong)],
}

ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_void_p)

_lib = None


class LudvmHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libludvm_hip error {code}: {message}")
        self.code = code


def _pin_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch-ROCm wheels carry their own libamdhip64.so /
    libhsa-runtime64.so (SONAME libamdhip64.so.7, like /opt/rocm's).  If torch is loaded first the
    engine's DT_NEEDED entry resolves to torch's copy and everything shares one runtime; if the engine
    were loaded first it would bring in /opt/rocm's copy, torch would later map its own next to it, and
    stream handles / device init would cross two runtimes (torch then reports "No HIP GPUs are
    available").  So when a torch install is present its runtime is mapped before the engine, whatever
    the import order.  Without torch the engine uses the system ROCm runtime."""
    if "torch" in sys.modules:
        return
    with open("/proc/self/maps") as f:
        if any("libamdhip64" in line for line in f):
            return                              # a runtime is already mapped: the loader will reuse it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        try:
            ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass                                # unusable torch install: fall back to the system runtime


def prefer_matching_rccl():
    """The library opens librccl itself when a communicator is first created (ludvm_comm_*; LUDVM_RCCL_LIB names the file,
    default librccl.so.1 from the loader's path).  When the process runs on the HIP runtime a PyTorch-ROCm install ships
    (see _pin_hip_runtime) the RCCL built against THAT runtime is the one to open -- it sits next to it."""
    if os.environ.get("LUDVM_RCCL_LIB"):
        return os.environ["LUDVM_RCCL_LIB"]
    try:
        with open("/proc/self/maps") as f:
            rts = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        rts = set()
    for rt in rts:
        cand = os.path.join(os.path.dirname(rt), "librccl.so")
        if os.path.exists(cand):
            os.environ["LUDVM_RCCL_LIB"] = cand
            return cand
    return "librccl.so.1"


def load(path=None):
    """dlopen the engine and set the prototypes.  Raises OSError when the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("LUDVM_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `make -C ludvm_amd/csrc` (or __graft_entry__.build()); "
                      "there is no CPU fallback")
    _pin_hip_runtime()
    lib = ctypes.CDLL(p)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_char_p if name == "ludvm_last_error" else c_int
    if lib.ludvm_abi_version() != ABI_VERSION:
        raise OSError(f"{p}: ABI version {lib.ludvm_abi_version()} != {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib
