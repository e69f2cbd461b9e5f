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
_OPTIONS = {"device": None, "dtype": "f64", "shard": True}


def configure(device: int | None = None, dtype: str = "f64", shard: bool = True):
    """Pick the GPU (default: LOCAL_RANK or 0), the engine dtype and whether to shard over ranks."""
    _OPTIONS.update(device=device, dtype=dtype, shard=shard)
    _CACHE.clear()


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def _engine_for(dat) -> Engine:
    import os
    dat = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
    rank, world = _rank_world()
    if not _OPTIONS["shard"]:
        rank, world = 0, 1
    key = (dat.shape, zlib.crc32(dat.tobytes()), rank, world, _OPTIONS["dtype"])
    eng = _CACHE.get(key)
    if eng is None:
        if len(_CACHE) >= _MAX_CACHE:
            _CACHE.pop(next(iter(_CACHE))).close()
        n_mut = (dat.shape[1] - 3) // 2
        dev = _OPTIONS["device"]
        if dev is None:
            dev = int(os.environ.get("LOCAL_RANK", "0"))
        eng = Engine(n_mut, device=dev, dtype=_OPTIONS["dtype"])
        rows = dat if world == 1 else dat[_dist.shard_rows(dat, world)[rank]]
        eng.set_cohort(rows)
        # MMHN_FORCE_ALLREDUCE=1: run the collective even with one rank (exercises the RCCL path on a 1-GPU box)
        eng._sharded = world > 1 or os.environ.get("MMHN_FORCE_ALLREDUCE") == "1"
        _CACHE[key] = eng
    return eng


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
    eng = _engine_for(dat)
    sums = _dist.allreduce_sums(eng.cohort_sums(log_theta, log_d_p, log_d_m, with_grad=False)) \
        if eng._sharded else eng.cohort_sums(log_theta, log_d_p, log_d_m, with_grad=False)
    return _dist.combine_sums(sums, eng.N, perc_met)[0]


def score_and_grad(log_theta, log_d_p, log_d_m, dat, perc_met: float):
    """(score, d_theta, d_d_p, d_d_m)  (regularized_optimization.py:163-267)."""
    eng = _engine_for(dat)
    sums = eng.cohort_sums(log_theta, log_d_p, log_d_m, with_grad=True)
    if eng._sharded:
        sums = _dist.allreduce_sums(sums)
    return _dist.combine_sums(sums, eng.N, perc_met)


def _unpack(params, n_total):
    params = np.asarray(params, dtype=np.float64)
    return (params[0:n_total ** 2].reshape((n_total, n_total)), params[n_total ** 2:n_total * (n_total + 1)],
            params[n_total * (n_total + 1):])


def score_reg(params, dat, perc_met: float, penal: Callable, w_penal: float):
    """regularized_optimization.py:133-160."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    sc = score(th, dp, dm, dat, perc_met)
    pen, _ = penal(params, n_total)
    return np.array(-sc + w_penal * pen)


def score_and_grad_reg(params, dat, perc_met: float, penal: Callable, w_penal: float):
    """regularized_optimization.py:270-298."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    sc, d_th, d_d_p, d_d_m = score_and_grad(th, dp, dm, dat, perc_met)
    grad_vec = np.concatenate((d_th.flatten(), d_d_p, d_d_m))
    pen, pen_ = penal(params, n_total)
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
