"""Drop-in for metmhn/regularized_optimization.py (same names, arguments, return values).

`score`, `score_and_grad`, `score_reg`, `score_and_grad_reg`, `learn_mhn` and the penalties
keep the reference signatures (regularized_optimization.py:11-334); the per-patient work
(metmhn/jx/*) runs on the GPU through the C ABI.  The cohort is uploaded and laid out once
per distinct `dat` (cached), parameters go up and a 484-double result comes back per call.
With an initialised torch.distributed process group every rank evaluates its own patient
shard and one all-reduce combines the partial sums (metmhn_amd.distributed).
"""
from __future__ import annotations

import zlib
from typing import Callable

import numpy as np
import scipy.optimize as opt

from . import distributed as _dist
from .engine import Engine

_CACHE: dict = {}
_MAX_CACHE = 4
_OPTIONS = {"device": None, "dtype": "f64", "shard": True, "strict_guard": False}


def configure(device: int | None = None, dtype: str = "f64", shard: bool = True, strict_guard: bool = False):
    """Pick the GPU (default: LOCAL_RANK or 0), the engine dtype and whether to shard over ranks.
    strict_guard: the cache guard hashes ALL of `dat` on every call (the reference is a pure function of `dat`);
    default: all of it up to 256 KB, 64 sampled rows above that."""
    from .engine import set_default_device
    _OPTIONS.update(device=device, dtype=dtype, shard=shard, strict_guard=bool(strict_guard))
    set_default_device(device)
    invalidate()


def invalidate(dat=None):
    """Drop the cached device cohort of `dat` (all of them if None).  The cache recognises an array by identity
    (object, buffer address, shape) plus a CRC of 64 sampled rows (`_sample_crc`): in-place edits that touch the
    sampled rows are noticed and re-uploaded, for anything finer call this after changing a cohort array IN PLACE."""
    for key in list(_CACHE):
        if dat is None or key[0] == id(dat):
            _CACHE.pop(key)[0].close()


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def _join_comm(eng: Engine, rank: int, world: int) -> bool:
    """Attach the engine to an RCCL communicator over the ranks of the torch.distributed job (backend nccl): the
    id travels through torch's object broadcast, the all-reduce itself then runs inside mmhn_cohort_sums on the
    engine's stream.  False -> the caller keeps the host-staged torch all-reduce (gloo / CPU tests).

    The decision is COLLECTIVE: rank 0 always broadcasts (a failure sentinel if it could not make an id), every rank
    reports whether its mmhn_comm_init succeeded, and the device communicator is used only if all did - otherwise
    every rank destroys its half and all of them use torch's collective (no rank is ever left alone in one of the two).
    MMHN_COMM_TIMEOUT (seconds, default 180): a rank whose ncclCommInitRank does not return fails instead of hanging."""
    import os
    import torch.distributed as dist
    from .engine import unique_id
    if world > 1 and dist.get_backend() != "nccl":
        return False
    strict = os.environ.get("MMHN_STRICT_COMM") == "1"
    err = None
    box = [None]
    if rank == 0:
        try:
            box[0] = unique_id()
        except Exception as exc:
            err = exc
    if world > 1:
        dist.broadcast_object_list(box, src=0)                # always entered by every rank
    if box[0] is None:
        all_ok, msg = False, f"rank 0 could not create an RCCL id ({err})"
    else:
        store = None
        if world > 1:
            from torch.distributed.distributed_c10d import _get_default_store
            store = _get_default_store()
        # (no collective in here: a rank that fails before ncclCommInitRank tells the others through the store, and a rank
        # still blocked in it gives up within a fraction of a second instead of after MMHN_COMM_TIMEOUT - distributed.py)
        all_ok, msg = _dist.collective_init(lambda: eng.comm_init(box[0], rank, world), rank, world, store,
                                            timeout=float(os.environ.get("MMHN_COMM_TIMEOUT", "180")))
    if all_ok:
        return True
    if getattr(eng, "comm_size", 1) > 1 or box[0] is not None:
        try:
            eng.comm_destroy()                                # some other rank failed: nobody uses the device communicator
        except Exception:                                     # noqa: BLE001
            pass
    if strict:
        raise RuntimeError(f"metmhn_amd: in-library RCCL communicator unavailable on some rank ({msg})")
    import warnings
    warnings.warn(f"metmhn_amd: in-library RCCL communicator unavailable ({msg}); using torch.distributed")
    return False


def _cache_key(dat, rank, world):
    """Identity key of a cohort array (no pass over its bytes on the per-evaluation path)."""
    if isinstance(dat, np.ndarray):
        return (id(dat), dat.__array_interface__["data"][0], dat.shape, dat.dtype.str, rank, world, _OPTIONS["dtype"])
    arr = np.ascontiguousarray(np.asarray(dat).astype(np.int8))          # lists, foreign array types: by content
    return (None, zlib.crc32(arr.tobytes()), arr.shape, "i1", rank, world, _OPTIONS["dtype"])


_CRC_ROWS = {}


def _digest(arr: np.ndarray) -> int:
    """64-bit digest of a contiguous array: xxh3 where the package is there (13 us for the 208 KB of the LUAD-reduced
    cohort - zlib's CRC takes 240 us, longer than a score-only evaluation of that cohort), else zlib.crc32."""
    if arr.size == 0:
        return 0
    flat = arr.reshape(-1).view(np.uint8)                     # (memoryview.cast refuses shapes with a zero in them)
    try:
        import xxhash                                         # optional: `pip install metmhn_amd[fast]`
    except ImportError:
        return zlib.crc32(flat)
    return xxhash.xxh3_64_intdigest(flat)


def _sample_crc(dat: np.ndarray) -> int:
    """Guard of the identity-keyed cache against in-place edits, run while the GPU evaluates: the CRC of the whole array
    up to 256 KB (every LUAD-sized cohort: ~50 us) or with configure(strict_guard=True); of 64 evenly spaced rows above
    that (a full pass over a large cohort per evaluation would be host time on the critical path) - there a single
    edited row between the samples still needs invalidate()."""
    if dat.shape[0] <= 64 or dat.nbytes <= (256 << 10) or _OPTIONS["strict_guard"]:
        return _digest(np.ascontiguousarray(dat))
    idx = _CRC_ROWS.get(dat.shape[0])
    if idx is None:
        if len(_CRC_ROWS) > 64:
            _CRC_ROWS.clear()
        idx = _CRC_ROWS[dat.shape[0]] = np.linspace(0, dat.shape[0] - 1, 64).astype(np.int64)
    return _digest(np.ascontiguousarray(dat[idx]))


def _engine_for(dat, check: bool = True) -> Engine:
    """The engine holding `dat`'s layout.  check=False: a cache hit is returned without the in-place-edit guard - the
    caller runs it next to the GPU (_result(..., guard=dat)) and evaluates again if it fires."""
    import os
    import weakref
    rank, world = _rank_world()
    if not _OPTIONS["shard"]:
        rank, world = 0, 1
    key = _cache_key(dat, rank, world)
    hit = _CACHE.get(key)
    if hit is not None and (hit[1] is None or hit[1]() is dat):
        if not check or not isinstance(dat, np.ndarray) or not _stale_anywhere(hit[0], dat, world):
            return hit[0]
        # same array object, edited in place: the SAME engine (callers may hold it; its communicator stays) gets the new rows
        _load_rows(hit[0], dat, rank, world)
        return hit[0]
    if hit is not None:                                       # the id was recycled for another array
        _CACHE.pop(key)[0].close()
    for k2 in [k2 for k2, (e2, r2) in _CACHE.items() if r2 is not None and r2() is None]:
        _CACHE.pop(k2)[0].close()                             # cohorts whose array is gone free their HBM now
    arr = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
    if len(_CACHE) >= _MAX_CACHE:
        _CACHE.pop(next(iter(_CACHE)))[0].close()
    n_mut = (arr.shape[1] - 3) // 2
    eng = Engine(n_mut, device=_OPTIONS["device"], dtype=_OPTIONS["dtype"])
    _load_rows(eng, dat, rank, world, arr)
    # MMHN_FORCE_ALLREDUCE=1: run the collective even with one rank (exercises the RCCL path on a 1-GPU box)
    eng.world_size_hint = world
    eng._sharded = world > 1 or os.environ.get("MMHN_FORCE_ALLREDUCE") == "1"
    eng._device_comm = eng._sharded and _reduce_mode() != "host_fixed_order" and _join_comm(eng, rank, world)
    ref = None
    if isinstance(dat, np.ndarray):
        try:
            ref = weakref.ref(dat)
        except TypeError:
            ref = None
    _CACHE[key] = (eng, ref)
    return eng


def _load_rows(eng: Engine, dat, rank: int, world: int, arr=None):
    """(Re)build the device layout of this rank's shard of `dat` on `eng`."""
    if arr is None:
        arr = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
    rows = arr if world == 1 else arr[_dist.shard_rows(arr, world)[rank]]
    eng.set_cohort(rows)
    # global EM / NM counts (every rank holds the whole `dat`): the weight of :121-128 is known without communication
    eng._global_counts = (float(arr[:, -3].sum()), float(arr.shape[0]))
    eng._sample_crc = _sample_crc(dat) if isinstance(dat, np.ndarray) else None


def _stale_anywhere(eng: Engine, dat, world: int) -> bool:
    """Has `dat` been edited in place since `eng` laid it out - on ANY rank (the ranks must decide together: a rebuild
    and the repeated evaluation are collective)?  Used where an engine is fetched outside an evaluation; inside one the bit
    rides in the evaluation's own all-reduce (_result)."""
    stale = isinstance(dat, np.ndarray) and eng._sample_crc != _sample_crc(dat)
    if world > 1:
        import torch
        import torch.distributed as dist
        flag = torch.tensor([1 if stale else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        stale = bool(int(flag.item()))
    return stale


def _reduce_mode():
    import os
    return os.environ.get("MMHN_REDUCE", "")


def _result(eng: Engine, log_theta, log_d_p, log_d_m, perc_met: float, with_grad: bool, meanwhile: Callable = None,
            guard=None):
    """((score, d_theta, d_dp, d_dm), meanwhile's result).  EM and NM sums are combined on the device (the weight of
    regularized_optimization.py:121-128 only needs the global counts, known when the cohort is set): 1 + N^2 + 2N
    doubles come back - and, with several ranks, cross them in ONE all-reduce, inside the library on the engine's
    stream (RCCL) when the communicator is attached, else through torch.distributed.
    `meanwhile()` (host work that does not need the result: the penalty terms) runs while the GPU evaluates, and so
    does the cache guard of `guard` (the caller's `dat`, engine from _engine_for(dat, check=False)): if the array was
    edited in place since its layout was built, the result is dropped and the evaluation repeated on a fresh layout."""
    w, n_full = _dist.em_weight(*eng._global_counts, perc_met)
    guarded = isinstance(guard, np.ndarray)                   # (anything else was keyed by content: cannot be stale)
    many = eng._sharded and eng.world_size_hint > 1
    stale = False
    if many and guarded:
        # several ranks must decide TOGETHER whether to rebuild: the bit rides in the evaluation's own all-reduce (one
        # collective per evaluation), so the guard runs before the launch here (13 us on a LUAD-sized array)
        stale = eng._sample_crc != _sample_crc(guard)
    eng.cohort_wsums_begin(log_theta, log_d_p, log_d_m, w, with_grad=with_grad, flag=(1.0 if stale else 0.0) if many and guarded else None)
    try:
        if not many:
            stale = guarded and eng._sample_crc != _sample_crc(guard)          # one rank: next to the GPU
        aside = meanwhile() if meanwhile is not None and not stale else None
    finally:
        ws = eng.cohort_wsums_end()
    if eng._sharded and not eng._device_comm:
        ext = np.append(ws, 1.0 if (stale and many and guarded) else 0.0)       # (the same bit through torch's collective)
        ext = _dist.allreduce_sums_fixed_order(ext) if _reduce_mode() == "host_fixed_order" else _dist.allreduce_sums(ext)
        ws, stale_any = ext[:-1], ext[-1] > 0
    else:
        stale_any = eng.reduce_flag > 0
    if many and guarded:
        stale = bool(stale_any)
    if stale:
        rank, world = _rank_world() if _OPTIONS["shard"] else (0, 1)
        _load_rows(eng, guard, rank, world)                   # the same engine (and communicator), the new rows
        return _result(eng, log_theta, log_d_p, log_d_m, perc_met, with_grad, meanwhile)
    if aside is None and meanwhile is not None:
        aside = meanwhile()
    return _dist.split_wsums(ws, eng.N, n_full), aside


# ---- penalties (host NumPy, as in the reference; regularized_optimization.py:11-52) ----

def L1(theta, eps: float = 1e-05):
    t = np.array(theta, dtype=np.float64, copy=True)
    if t.ndim == 2:
        np.fill_diagonal(t, 0.0)
    return np.sum(np.sqrt(t ** 2 + eps))


def L1_(theta, eps: float = 1e-05):
    t = np.array(theta, dtype=np.float64, copy=True)
    if t.ndim == 2:
        np.fill_diagonal(t, 0.0)
    return t.flatten() / np.sqrt(t.flatten() ** 2 + eps)


def sym_penal(log_theta, eps: float = 1e-05):
    t = np.array(log_theta, dtype=np.float64, copy=True)
    n = t.shape[0]
    np.fill_diagonal(t, 0.0)
    return 0.5 * (np.sum(np.sqrt(t ** 2 + t.T ** 2 - t * t.T + eps)) - n * np.sqrt(eps))


def sym_penal_(log_theta, eps: float = 1e-05):
    t = np.array(log_theta, dtype=np.float64, copy=True)
    np.fill_diagonal(t, 0.0)
    return ((2 * t - t.T) / (2 * np.sqrt(t ** 2 + t.T ** 2 - t * t.T + eps))).flatten()


def symmetric_penal(params, n_total: int, eps=1e-05):
    params = np.asarray(params, dtype=np.float64)
    log_theta = params[0:n_total ** 2].reshape((n_total, n_total))
    log_d_p = params[n_total ** 2:n_total * (n_total + 1)]
    log_d_m = params[n_total * (n_total + 1):]
    penal = np.array(sym_penal(log_theta) + L1(log_d_p) + L1(log_d_m))
    penal_ = np.concatenate((sym_penal_(log_theta), L1_(log_d_p), L1_(log_d_m)))
    return penal, penal_


# ---- objective -------------------------------------------------------------------------

def score(log_theta, log_d_p, log_d_m, dat, perc_met: float):
    """Log-likelihood of the dataset (regularized_optimization.py:55-130)."""
    eng = _engine_for(dat, check=False)
    return _result(eng, log_theta, log_d_p, log_d_m, perc_met, False, guard=dat)[0][0]


def score_and_grad(log_theta, log_d_p, log_d_m, dat, perc_met: float):
    """(score, d_theta, d_d_p, d_d_m)  (regularized_optimization.py:163-267)."""
    eng = _engine_for(dat, check=False)
    return _result(eng, log_theta, log_d_p, log_d_m, perc_met, True, guard=dat)[0]


def _unpack(params, n_total):
    params = np.asarray(params, dtype=np.float64)
    return (params[0:n_total ** 2].reshape((n_total, n_total)), params[n_total ** 2:n_total * (n_total + 1)],
            params[n_total * (n_total + 1):])


def score_reg(params, dat, perc_met: float, penal: Callable, w_penal: float):
    """regularized_optimization.py:133-160."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    eng = _engine_for(dat, check=False)
    res, (pen, _) = _result(eng, th, dp, dm, perc_met, False, meanwhile=lambda: penal(params, n_total), guard=dat)    # penalty next to the GPU
    return np.array(-res[0] + w_penal * pen)


def score_and_grad_reg(params, dat, perc_met: float, penal: Callable, w_penal: float):
    """regularized_optimization.py:270-298."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    eng = _engine_for(dat, check=False)
    (sc, d_th, d_d_p, d_d_m), (pen, pen_) = _result(eng, th, dp, dm, perc_met, True, meanwhile=lambda: penal(params, n_total), guard=dat)  # penalty next to the GPU
    grad_vec = np.concatenate((d_th.flatten(), d_d_p, d_d_m))
    return np.array(-sc + w_penal * pen), -grad_vec + w_penal * pen_


def learn_mhn(th_init, dp_init, dm_init, dat, perc_met: float, penal: Callable, w_penal: float,
              opt_iter: int = 1e05, opt_ftol: float = 1e-04, opt_v: bool = True):
    """Infer a metMHN with SciPy's L-BFGS-B (regularized_optimization.py:301-334)."""
    th_init = np.asarray(th_init, dtype=np.float64)
    n_total = th_init.shape[0]
    start = np.concatenate((th_init.flatten(), np.asarray(dp_init, float), np.asarray(dm_init, float)))
    x = opt.minimize(fun=score_and_grad_reg, jac=True, x0=start, method="L-BFGS-B",
                     args=(dat, perc_met, penal, w_penal),
                     options={"maxiter": int(opt_iter), "disp": opt_v, "ftol": opt_ftol})
    theta = np.array(x.x[:n_total ** 2]).reshape((n_total, n_total))
    d_p = np.array(x.x[n_total ** 2:n_total * (n_total + 1)])
    d_m = np.array(x.x[n_total * (n_total + 1):])
    return theta, d_p, d_m
