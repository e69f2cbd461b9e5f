"""ctypes access to oracle/_build/libmetmhn_ref.so (C restatement; checker / CPU baseline only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMHN_ORACLE_DIR: load another build of the two libraries (oracle/Makefile `asan`: the sanitizer build)
_DIR = os.path.abspath(os.environ["MMHN_ORACLE_DIR"]) if os.environ.get("MMHN_ORACLE_DIR") else os.path.join(_HERE, "_build")
LIB = os.path.join(_DIR, "libmetmhn_ref.so")
_f = C.POINTER(C.c_double)
_i8 = C.POINTER(C.c_int8)
_lib = None


def _cpu_share() -> int:
    """Cores this process may really use: affinity, capped by the cgroup quota and by 16.  The GPU boxes show every core
    of the host while the container owns a fraction: an OpenMP team of the visible count spins against itself (a 1 s
    ref_kronvec then takes minutes)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(cores, 16))


def _limit_threads():
    if os.environ.get("OMP_NUM_THREADS"):
        return
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(_cpu_share())
    except Exception:
        pass


def load(build: bool = True):
    global _lib
    _limit_threads()
    if _lib is None:
        if build and not os.environ.get("MMHN_ORACLE_DIR") and (not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "metmhn_ref.c"))):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = C.CDLL(LIB)
        _lib.ref_patients.argtypes = [C.c_int, _f, _f, _f, _i8, C.c_int64, C.c_int, C.c_int, C.c_int, _f, _f, _f, _f]
        _lib.ref_kronvec.argtypes = [C.c_int, _f, _i8, _f, _f, C.c_int, C.c_int]
        _lib.ref_num_threads.restype = C.c_int
    return _lib


def num_threads() -> int:
    return load().ref_num_threads()


def patients(log_theta, log_d_p, log_d_m, dat, with_grad=True, threads=0, patient_parallel=True):
    """Per-patient (lp, d_theta, d_dp, d_dm) of the rows of `dat`.

    patient_parallel=True: OpenMP over patients (many small patients); False: patients one after the
    other with every pass over the 2^k vector split across the threads (few large patients)."""
    lib = load()
    lt = np.ascontiguousarray(log_theta, dtype=np.float64)
    dp = np.ascontiguousarray(log_d_p, dtype=np.float64)
    dm = np.ascontiguousarray(log_d_m, dtype=np.float64)
    dat = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
    n = (dat.shape[1] - 3) // 2
    N, P = n + 1, dat.shape[0]
    lp = np.zeros(P)
    g, a, b = np.zeros((P, N, N)), np.zeros((P, N)), np.zeros((P, N))
    rc = lib.ref_patients(n, lt.ctypes.data_as(_f), dp.ctypes.data_as(_f), dm.ctypes.data_as(_f),
                          dat.ctypes.data_as(_i8), P, int(with_grad), int(threads), int(patient_parallel),
                          lp.ctypes.data_as(_f),
                          g.ctypes.data_as(_f), a.ctypes.data_as(_f), b.ctypes.data_as(_f))
    if rc != 0:
        raise RuntimeError("ref_patients failed")
    return lp, g, a, b


def kronvec(log_theta, p, state, diag=True, transpose=False):
    lib = load()
    lt = np.ascontiguousarray(log_theta, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    st = np.ascontiguousarray(np.asarray(state).astype(np.int8))
    y = np.zeros_like(p)
    lib.ref_kronvec((st.shape[0] - 1) // 2, lt.ctypes.data_as(_f), st.ctypes.data_as(_i8), p.ctypes.data_as(_f),
                    y.ctypes.data_as(_f), int(diag), int(transpose))
    return y


FAST = os.path.join(_DIR, "libmetmhn_fast.so")
_fast = None


def load_fast(build: bool = True):
    """oracle/_build/libmetmhn_fast.so: optimised CPU variant (paired rows only; baseline / checker only)."""
    global _fast
    _limit_threads()
    if _fast is None:
        if build and not os.environ.get("MMHN_ORACLE_DIR") and (not os.path.exists(FAST) or os.path.getmtime(FAST) < os.path.getmtime(os.path.join(_HERE, "metmhn_fast.c"))):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _fast = C.CDLL(FAST)
        _fast.fast_patients.argtypes = [C.c_int, _f, _f, _f, _i8, C.c_int64, C.c_int, _f, _f, _f, _f]
    return _fast


def fast_patients(log_theta, log_d_p, log_d_m, dat, threads=0):
    """Per-patient (lp, d_theta, d_dp, d_dm) of PAIRED rows (type 3) by the gather formulation."""
    lib = load_fast()
    lt = np.ascontiguousarray(log_theta, dtype=np.float64)
    dp = np.ascontiguousarray(log_d_p, dtype=np.float64)
    dm = np.ascontiguousarray(log_d_m, dtype=np.float64)
    dat = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
    n = (dat.shape[1] - 3) // 2
    N, P = n + 1, dat.shape[0]
    lp = np.zeros(P)
    g, a, b = np.zeros((P, N, N)), np.zeros((P, N)), np.zeros((P, N))
    rc = lib.fast_patients(n, lt.ctypes.data_as(_f), dp.ctypes.data_as(_f), dm.ctypes.data_as(_f),
                           dat.ctypes.data_as(_i8), P, int(threads), lp.ctypes.data_as(_f),
                           g.ctypes.data_as(_f), a.ctypes.data_as(_f), b.ctypes.data_as(_f))
    if rc != 0:
        raise ValueError(f"row {rc - 1} is not a paired datapoint with seeding")
    return lp, g, a, b
