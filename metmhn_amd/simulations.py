"""Gillespie sampling of the joint PT/MT process on the GPU - the reference's `metmhn/simulations.py` call
surface (`simulate_dat`, `simulate_orders`, :87-147) over `mmhn_simulate` (csrc/sampler.h).

`original_key` takes the place of the `jax.random.PRNGKey`: an int, or anything array-like whose integers are
folded into the 64-bit Philox key.  Streams differ from jax.random's; distributions do not."""
from __future__ import annotations

import numpy as np

from .engine import Engine

_engines: dict = {}


def _engine(n_mut: int) -> Engine:
    from .engine import default_device
    key = (n_mut, default_device())
    if key not in _engines:
        _engines[key] = Engine(n_mut)
    return _engines[key]


def _seed(key) -> int:
    a = np.atleast_1d(np.asarray(key)).astype(np.uint64).ravel()
    s = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        for v in a:
            s = (s ^ v) * np.uint64(0xBF58476D1CE4E5B9)
            s ^= s >> np.uint64(31)
    return int(s)


def simulate_dat(log_theta, pt_d_ef, mt_d_ef, n_sim: int, original_key=0) -> np.ndarray:
    """int8 [n_sim, 2n+2]: genotypes `[PT_0, MT_0, ..., seeding]` + observation order (simulations.py:117-147)."""
    lt = np.asarray(log_theta, dtype=np.float64)
    return _engine(lt.shape[0] - 1).simulate(lt, pt_d_ef, mt_d_ef, n_sim, _seed(original_key))


def simulate_orders(log_theta, pt_d_ef, mt_d_ef, n_sim: int, original_key=0) -> np.ndarray:
    """int8 [n_sim, 2N+2]: event sequences padded with -99, events numbered as simulations.py:100-107."""
    lt = np.asarray(log_theta, dtype=np.float64)
    return _engine(lt.shape[0] - 1).simulate(lt, pt_d_ef, mt_d_ef, n_sim, _seed(original_key), orders=True)[1]
