"""metmhn/jx/vanilla.py entry points (single-tumour MHN, state of length n+1)."""
from __future__ import annotations

import numpy as np

from . import engine


def _n(state):
    return np.asarray(state).shape[0] - 1


def kronvec(log_theta, p, state, diag: bool = True, transpose: bool = False):
    """vanilla.py:78-106."""
    return engine(_n(state)).v_kronvec(log_theta, p, state, diag, transpose)


def R_inv_vec(log_theta, x, state, d_rates=1, transpose: bool = False):
    """vanilla.py:269-305."""
    return engine(_n(state)).v_resolvent(log_theta, x, state, d_rates, transpose)


def x_partial_Q_y(log_theta, x, y, state):
    """vanilla.py:328-393: (val, d_diag)."""
    return engine(_n(state)).v_x_partial_Q_y(log_theta, x, y, state)


def kron_diag(log_theta, state, diag):
    """vanilla.py:247-260: diag(Q) of the single-tumour space times the vector `diag`."""
    return engine(_n(state)).v_kron_diag(log_theta, state, diag)


def scal_d_pt(log_d_p, log_d_m, state, vec):
    """vanilla.py:125-142: (d_p part, d_m part) of the observation rates of an MT-only datapoint, times vec."""
    return engine(_n(state)).v_scal_d_pt(log_d_p, log_d_m, state, vec)


def d_scal_d_pt(log_d_p, log_d_m, state, vec, i: int):
    """vanilla.py:179-187."""
    return engine(_n(state)).v_d_scal_d_pt(log_d_p, log_d_m, state, vec, i)


def x_partial_D_y(log_d_p, log_d_m, state, x, y):
    """vanilla.py:190-203: (d_dp, d_dm); argument order (log_d_p, log_d_m), unlike likelihood.x_partial_D_y."""
    return engine(_n(state)).v_x_partial_D_y(log_d_p, log_d_m, state, x, y)


def gradient(log_theta, state, p_0):
    """vanilla.py:396-418: (d_theta, d_diag, p_theta) with p_theta = R^-1 p_0, x = R^-T e_last / p_theta[-1]."""
    p_theta = R_inv_vec(log_theta, p_0, state)
    x = np.zeros_like(p_theta)
    x[-1] = 1.0 / p_theta[-1]
    x = R_inv_vec(log_theta, x, state, transpose=True)
    d_th, d_diag = x_partial_Q_y(log_theta, x, p_theta, state)
    return d_th, d_diag, p_theta
