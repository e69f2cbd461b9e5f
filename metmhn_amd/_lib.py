"""Build and load the native library (libmetmhn_amd.so) through ctypes.

The C ABI is declared in include/metmhn_amd.h.  There is no CPU fallback: if the
library cannot be built/loaded, or no GPU is visible when an engine is created,
the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
import glob
_SRC = [os.path.join(_HERE, "csrc", "engine.hip")] + sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")))
_HDR = os.path.join(_ROOT, "include", "metmhn_amd.h")
LIB_PATH = os.environ.get("MMHN_LIB", os.path.join(_HERE, "libmetmhn_amd.so"))   # MMHN_LIB: A/B builds

f64p = C.POINTER(C.c_double)
i8p = C.POINTER(C.c_int8)
i64p = C.POINTER(C.c_int64)


KERNEL_CLASSES = ("other_solve", "psolve_fwd", "psolve_adj", "pclass", "csolve_fwd", "csolve_adj")      # MMHN_K_* of include/metmhn_amd.h


class KernelCounter(C.Structure):
    _fields_ = [("ms", C.c_double), ("launches", C.c_int64), ("alg_bytes", C.c_double)]


class Counters(C.Structure):
    _fields_ = [("kernel", KernelCounter * len(KERNEL_CLASSES)), ("eval_ms", C.c_double), ("evals", C.c_int64),
                ("comm_ranks", C.c_int32), ("comm_rank", C.c_int32)]


# name -> argtypes (every function returns int status unless noted)
SIGNATURES = {
    "mmhn_create": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "mmhn_set_workspace_limit": [C.c_void_p, C.c_size_t],
    "mmhn_set_cohort": [C.c_void_p, i8p, C.c_int64, C.c_int],
    "mmhn_score": [C.c_void_p, f64p, f64p, f64p, C.c_double, f64p],
    "mmhn_score_and_grad": [C.c_void_p, f64p, f64p, f64p, C.c_double, f64p, f64p, f64p, f64p],
    "mmhn_cohort_sums": [C.c_void_p, f64p, f64p, f64p, C.c_int, f64p],
    "mmhn_cohort_sums_begin": [C.c_void_p, f64p, f64p, f64p, C.c_int],
    "mmhn_cohort_sums_end": [C.c_void_p, f64p],
    "mmhn_cohort_wsums_begin": [C.c_void_p, f64p, f64p, f64p, C.c_int, C.c_double],
    "mmhn_cohort_wsums_end": [C.c_void_p, f64p],
    "mmhn_set_reduce_flag": [C.c_void_p, C.c_double],
    "mmhn_get_reduce_flag": [C.c_void_p, f64p],
    "mmhn_patient_grads": [C.c_void_p, f64p, f64p, f64p, f64p, f64p, f64p, f64p],
    "mmhn_kronvec": [C.c_void_p, f64p, i8p, f64p, f64p, C.c_int, C.c_int],
    "mmhn_kronvec_batched": [C.c_void_p, f64p, i8p, C.c_int64, f64p, f64p, C.c_int, C.c_int],
    "mmhn_jacobi_step_batched": [C.c_void_p, f64p, f64p, f64p, i8p, C.c_int64, f64p, f64p, f64p, C.c_int],
    "mmhn_kron_diag": [C.c_void_p, f64p, i8p, f64p],
    "mmhn_diag_scal": [C.c_void_p, f64p, i8p, f64p, f64p, C.c_int],
    "mmhn_obs_states": [C.c_void_p, i8p, C.c_int, i64p, i64p],
    "mmhn_resolvent": [C.c_void_p, f64p, f64p, f64p, i8p, f64p, f64p, C.c_int],
    "mmhn_x_partial_Q_y": [C.c_void_p, f64p, i8p, f64p, f64p, f64p],
    "mmhn_x_partial_D_y": [C.c_void_p, f64p, f64p, i8p, f64p, f64p, f64p, f64p],
    "mmhn_partial_diag_scal": [C.c_void_p, f64p, i8p, f64p, C.c_int, C.c_int, f64p],
    "mmhn_v_kronvec": [C.c_void_p, f64p, i8p, f64p, f64p, C.c_int, C.c_int],
    "mmhn_v_resolvent": [C.c_void_p, f64p, i8p, f64p, f64p, f64p, C.c_int],
    "mmhn_v_x_partial_Q_y": [C.c_void_p, f64p, i8p, f64p, f64p, f64p, f64p],
    "mmhn_v_kron_diag": [C.c_void_p, f64p, i8p, f64p, f64p],
    "mmhn_v_scal_d_pt": [C.c_void_p, f64p, f64p, i8p, f64p, f64p, f64p],
    "mmhn_v_d_scal_d_pt": [C.c_void_p, f64p, f64p, i8p, f64p, C.c_int, f64p, f64p],
    "mmhn_v_x_partial_D_y": [C.c_void_p, f64p, f64p, i8p, f64p, f64p, f64p, f64p],
    "mmhn_comm_unique_id": [C.c_void_p],
    "mmhn_comm_init": [C.c_void_p, C.c_void_p, C.c_int, C.c_int],
    "mmhn_comm_destroy": [C.c_void_p],
    "mmhn_bench_stream": [C.c_void_p, C.c_size_t, C.c_int, C.c_int, f64p],
    "mmhn_bench_kronvec": [C.c_void_p, f64p, i8p, C.c_int64, C.c_int, C.c_int, C.c_int, f64p, i64p],
    "mmhn_simulate": [C.c_void_p, f64p, f64p, f64p, C.c_int64, C.c_uint64, i8p, i8p],
    "mmhn_get_counters": [C.c_void_p, C.POINTER(Counters)],
    "mmhn_reset_counters": [C.c_void_p],
    "mmhn_debug_lane_moves": [C.c_void_p, C.c_int, C.POINTER(C.c_int)],
}
OTHER_SYMBOLS = ("mmhn_destroy", "mmhn_last_error", "mmhn_abi_version")
ABI_VERSION = 5          # MMHN_ABI_VERSION of include/metmhn_amd.h these prototypes were written against


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in _SRC + [_HDR])


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into metmhn_amd/libmetmhn_amd.so (in-tree)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-comment",
           "-o", LIB_PATH, _SRC[0]]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr)
    return LIB_PATH


_lib = None


def load():
    """dlopen the library and attach the ctypes prototypes of include/metmhn_amd.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(metmhn_amd has no CPU fallback)")
    # PyTorch wheels bundle their own ROCm runtime (libamdhip64.so.7, libhsa-runtime64.so.1 under torch/lib).  A process
    # must not end up with two HIP runtimes: if this library came first (bound to /opt/rocm's copy), a later
    # `import torch` + torch.cuda / RCCL initialisation fails with "No HIP GPUs are available".  So when torch is
    # installed it is imported first and both sides share its runtime (same sonames; the order bench.py always had).
    # An engine issues an evaluation on four streams (joint path; the paired rows' 1024-thread small-space launch the joint adjoint
    # waits for; the own-problem rows' small-space launches; their staged chain).  The HIP runtime spreads a process's streams over
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that land on one queue run in order: on the 28-event LUAD cohort the
    # paired launch then queues behind the own-problem chain - 1.42 instead of 1.25 ms per evaluation in a fresh process (DESIGN.md
    # section 6).  Read when the runtime initialises, so it is set here, before the first HIP call of a typical process; a value the
    # user exported wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mmhn_destroy.argtypes = [C.c_void_p]
    lib.mmhn_destroy.restype = None
    lib.mmhn_last_error.argtypes = []
    lib.mmhn_last_error.restype = C.c_char_p
    lib.mmhn_abi_version.argtypes = []
    lib.mmhn_abi_version.restype = C.c_int
    if lib.mmhn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.mmhn_abi_version()}, these bindings expect {ABI_VERSION} - rebuild "
                           "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise RuntimeError("metmhn_amd: " + load().mmhn_last_error().decode(errors="replace"))
