"""Synthetic parameters and cohorts of BASELINE.md section 3 (seeded NumPy generators)."""
from __future__ import annotations

import numpy as np


def random_params(n: int, seed: int | None = None):
    """log_theta (diag ~ N(0,1), off-diagonals non-zero w.p. 0.5 ~ N(0,1)), log_d_p, log_d_m ~ N(0, 0.25)."""
    rng = np.random.default_rng(1000 + n if seed is None else seed)
    N = n + 1
    lt = np.diag(rng.normal(size=N))
    mask = rng.random((N, N)) < 0.5
    np.fill_diagonal(mask, False)
    lt = lt + mask * rng.normal(size=(N, N))
    return lt, rng.normal(size=N) * 0.5, rng.normal(size=N) * 0.5


def full_k_cohort(n: int, n_pat: int, k: int | None = None, seed: int | None = None) -> np.ndarray:
    """Every row type 3, seeding = 1, k-1 ones uniformly without replacement over the 2n PT/MT slots,
    order ~ U{0,1,2}; int8 [n_pat, 2n+3]."""
    rng = np.random.default_rng(2000 + n if seed is None else seed)
    k = n if k is None else k
    dat = np.zeros((n_pat, 2 * n + 3), dtype=np.int8)
    for r in range(n_pat):
        dat[r, rng.choice(2 * n, size=k - 1, replace=False)] = 1
    dat[:, 2 * n] = 1
    dat[:, 2 * n + 1] = rng.integers(0, 3, size=n_pat)
    dat[:, 2 * n + 2] = 3
    return dat


def mixed_cohort(n: int, n_pat: int, seed: int = 0, p_event: float = 0.3) -> np.ndarray:
    """Mixed types with the fractions of examples/recall_study.py:120-125
    (11.5 % never-metastasising; of the rest 10.7 % paired, 38.6 % PT-only, remainder MT-only)."""
    rng = np.random.default_rng(seed)
    dat = np.zeros((n_pat, 2 * n + 3), dtype=np.int8)
    for r in range(n_pat):
        bits = (rng.random(2 * n) < p_event).astype(np.int8)
        u = rng.random()
        if u < 0.115:
            bits[1::2] = 0
            dat[r] = np.concatenate((bits, [0, -99, 0]))
        else:
            v = rng.random()
            if v < 0.107:
                dat[r] = np.concatenate((bits, [1, rng.integers(0, 3), 3]))
            elif v < 0.107 + 0.386:
                bits[1::2] = 0
                dat[r] = np.concatenate((bits, [1, -99, 1]))
            else:
                bits[0::2] = 0
                dat[r] = np.concatenate((bits, [1, -99, 2]))
    return dat
