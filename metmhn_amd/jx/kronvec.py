"""metmhn/jx/kronvec.py entry points (same argument orders)."""
from __future__ import annotations

import numpy as np

from . import engine, n_from_joint


def kronvec(log_theta, p, state, diag: bool = True, transpose: bool = False):
    """kronvec.py:499-539."""
    return engine(n_from_joint(state)).kronvec(log_theta, p, state, diag, transpose)


def kron_diag(log_theta, state, n_state: int):
    """kronvec.py:964-999 (n_state must equal the number of ones in state)."""
    out = engine(n_from_joint(state)).kron_diag(log_theta, state)
    if out.shape[0] != 2 ** n_state:
        raise ValueError("n_state does not match state")
    return out


def diag_scal_p(log_d_p, state, p):
    """kronvec.py:574-602."""
    return engine(n_from_joint(state)).diag_scal(log_d_p, state, p, 0)


def diag_scal_m(log_d_m, state, p):
    """kronvec.py:646-671."""
    return engine(n_from_joint(state)).diag_scal(log_d_m, state, p, 1)


def partial_diag_scal_p(log_d_p, state, p, i: int):
    """kronvec.py:632-644: (dD_p / dlog d_p[i]) * p."""
    return engine(n_from_joint(state)).partial_diag_scal(log_d_p, state, p, i, 0)


def partial_diag_scal_m(log_d_m, state, p, i: int):
    """kronvec.py:704-710: (dD_m / dlog d_m[i]) * p."""
    return engine(n_from_joint(state)).partial_diag_scal(log_d_m, state, p, i, 1)


def obs_states(n_joint: int, state, pt_first: bool = True):
    """kronvec.py:1056-1095: 0/1 mask of the compatible joint states."""
    idx = engine(n_from_joint(state)).obs_indices(state, pt_first)
    mask = np.zeros(2 ** n_joint)
    mask[idx] = 1.0
    return mask


def obs_indices(state, pt_first: bool = True):
    """jnp.where(obs_states(...) == 1, size=...)[0]  (likelihood.py:280,342,375)."""
    return engine(n_from_joint(state)).obs_indices(state, pt_first)
