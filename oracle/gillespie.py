"""Vectorised NumPy Gillespie sampler of the joint PT/MT process - TEST INFRASTRUCTURE ONLY.

Restates metmhn/simulations.py:8-147 (`single_traject`, `simulate_dat`) with a NumPy Generator instead of
jax.random: same state layout, rates, synchronised evolution before seeding, stopping rule and output format,
so the Monte-Carlo property the reference pins in tests/test_likelihood.py (analytic probability ~ simulated
frequency) can be re-tested here.  Sample streams differ from jax.random's, the distribution does not.
"""
from __future__ import annotations

import numpy as np


def simulate_dat(log_theta, pt_d_ef, mt_d_ef, n_sim: int, seed: int = 42) -> np.ndarray:
    """int8 [n_sim, 2*n_mut + 2]: [PT_0, MT_0, ..., PT_{m-1}, MT_{m-1}, seeding, order(0 unpaired / 1 PT first / 2 MT first)]."""
    rng = np.random.default_rng(seed)
    lt = np.asarray(log_theta, dtype=np.float64)
    pt_d = np.asarray(pt_d_ef, dtype=np.float64)
    mt_d = np.asarray(mt_d_ef, dtype=np.float64)
    n = lt.shape[0]                      # events incl. seeding (index n-1)
    n_d = n + 1
    b = np.diag(lt).copy()
    lt_prim = lt.copy()
    lt_prim[:-1, -1] = 0.0               # seeding has no effect on PT mutations (simulations.py:67-68)
    L = 2 * n + 2
    state = np.zeros((n_sim, L), dtype=np.int8)
    t_pt = np.full(n_sim, -1)            # step at which the PT / MT diagnosis happened
    t_mt = np.full(n_sim, -1)
    alive = np.ones(n_sim, dtype=bool)
    step = 0
    while alive.any():
        idx = np.nonzero(alive)[0]
        s = state[idx].astype(np.float64)
        pt, mt = s[:, :n], s[:, n_d:-1]
        r = np.zeros((idx.size, L))
        r[:, :n] = np.exp(pt @ lt_prim.T + b) * (1 - pt)
        r[:, n] = np.exp(pt @ pt_d) * (1 - s[:, n])
        r[:, :n + 1] *= (1 - s[:, n])[:, None]                          # PT frozen once diagnosed
        r[:, n_d:-1] = np.exp(mt @ lt.T + b) * (1 - mt)
        r[:, -1] = np.exp(mt @ mt_d) * (1 - s[:, -1])
        r[:, n_d:] *= (s[:, n - 1] * (1 - s[:, -1]))[:, None]           # MT evolves only after seeding
        cum = np.cumsum(r, axis=1)
        u = rng.random(idx.size) * cum[:, -1]
        ev = (cum < u[:, None]).sum(axis=1)
        seeded = state[idx, n - 1] == 1
        state[idx, ev] = 1
        both = ~seeded                                                   # before seeding both tumours move together
        state[idx[both], ev[both] + n_d] = 1
        hit_pt = (ev == n)
        t_pt[idx[hit_pt]] = step
        t_mt[idx[both & hit_pt]] = step
        hit_mt = (ev == L - 1)
        t_mt[idx[hit_mt]] = step
        st = state[idx]
        done = (st[:, n] * st[:, -1] + st[:, n] * (1 - st[:, n - 1])) >= 1
        alive[idx[done]] = False
        step += 1
    inter = state.reshape((n_sim, 2, n + 1)).transpose(0, 2, 1).reshape(n_sim, -1)   # reshape(-1,2,'F').flatten()
    obs = inter[:, :-3]
    order = np.zeros(n_sim, dtype=np.int8)
    paired = obs[:, -1] == 1
    order[paired] = np.where(t_pt[paired] < t_mt[paired], 1, 2)
    return np.hstack((obs, order[:, None])).astype(np.int8)
