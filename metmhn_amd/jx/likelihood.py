"""metmhn/jx/likelihood.py entry points (same argument orders, incl. the reference's quirks)."""
from __future__ import annotations

from . import engine, n_from_joint


def R_i_inv_vec(log_theta, log_d_p, log_d_m, x, state, state_size: int, transpose: bool = False):
    """likelihood.py:231-262."""
    return engine(n_from_joint(state)).resolvent(log_theta, log_d_p, log_d_m, x, state, transpose)


def x_partial_Q_y(log_theta, x, y, state):
    """likelihood.py:163-201."""
    return engine(n_from_joint(state)).x_partial_Q_y(log_theta, x, y, state)


def x_partial_D_y(log_d_m, log_d_p, state, x, y):
    """likelihood.py:204-228: takes (log_d_m, log_d_p, ...) and returns (d_dp, d_dm) like the reference."""
    return engine(n_from_joint(state)).x_partial_D_y(log_d_p, log_d_m, state, x, y)
