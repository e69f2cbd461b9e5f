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


# ---- per-patient entry points (likelihood.py:286-731): one-row cohorts on the engine --------------------
import numpy as _np


def _row(state, order, typ, n):
    r = _np.zeros(2 * n + 3, dtype=_np.int8)
    r[:2 * n + 1] = _np.asarray(state).astype(_np.int8)[:2 * n + 1]
    r[2 * n + 1], r[2 * n + 2] = order, typ
    return r.reshape(1, -1)


def _eval(log_theta, log_d_p, log_d_m, dat, grad):
    n = (dat.shape[1] - 3) // 2
    e = engine(n)
    e.set_cohort(dat)
    if not grad:
        return e.patient_grads(log_theta, log_d_p, log_d_m, with_grad=False)[0]
    lp, g, gp, gm = e.patient_grads(log_theta, log_d_p, log_d_m)
    return lp[0], g[0], gp[0], gm[0]


def _zeros_like_d(log_theta):
    return _np.zeros(_np.asarray(log_theta).shape[0])


def _lp_coupled_0(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:286-316 (one_event.py:141-171 for the one-event space)."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 0, 3, n_from_joint(state_joint)), False)


def _lp_coupled_1(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:320-350."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 1, 3, n_from_joint(state_joint)), False)


def _lp_coupled_2(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:353-384."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 2, 3, n_from_joint(state_joint)), False)


def _g_coupled_0(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:623-662: (log-prob, d_theta, d_dp, d_dm)."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 0, 3, n_from_joint(state_joint)), True)


def _g_coupled_1(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:665-697."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 1, 3, n_from_joint(state_joint)), True)


def _g_coupled_2(log_theta, log_d_p, log_d_m, state_joint, n_prim=None, n_met=None):
    """likelihood.py:700-731."""
    return _eval(log_theta, log_d_p, log_d_m, _row(state_joint, 2, 3, n_from_joint(state_joint)), True)


def _pt_row(state_pt):
    """state_pt = PT slots + seeding slot (length n+1) -> joint-format row of type 0 / 1."""
    st = _np.asarray(state_pt).astype(_np.int8)
    n = st.shape[0] - 1
    joint = _np.zeros(2 * n + 1, dtype=_np.int8)
    joint[0:2 * n:2] = st[:n]
    joint[2 * n] = st[n]
    return _row(joint, -99, 1 if st[n] else 0, n)


def _lp_prim_obs(log_theta, log_d_p, state_pt, n_prim=None):
    """likelihood.py:387-405."""
    return _eval(log_theta, log_d_p, _zeros_like_d(log_theta), _pt_row(state_pt), False)


def _grad_prim_obs(log_theta, log_d_p, state_prim, n_prim=None):
    """likelihood.py:441-461: (log-prob, d_theta, d_dp)."""
    return _eval(log_theta, log_d_p, _zeros_like_d(log_theta), _pt_row(state_prim), True)[:3]


def _az_row(log_theta):
    """the all-zero never-metastasised primary tumour: a type-0 row without any event (regularized_optimization.py:80-81)"""
    n = _np.asarray(log_theta).shape[0] - 1
    return _row(_np.zeros(2 * n + 1, dtype=_np.int8), -99, 0, n)


def _lp_prim_obs_az(log_theta):
    """likelihood.py:408-416: log P(00...0 | theta) = -log(1 + sum_i theta_ii), through the engine's closed-form row kind."""
    return _eval(log_theta, _zeros_like_d(log_theta), _zeros_like_d(log_theta), _az_row(log_theta), False)


def _grad_prim_obs_az(log_theta):
    """likelihood.py:462-476: (log-prob, d_theta, d_dp) of the all-zero primary tumour."""
    return _eval(log_theta, _zeros_like_d(log_theta), _zeros_like_d(log_theta), _az_row(log_theta), True)[:3]


def _mt_row(state_mt):
    """state_mt = MT slots + seeding (= 1), length n+1 -> joint-format row of type 2."""
    st = _np.asarray(state_mt).astype(_np.int8)
    n = st.shape[0] - 1
    joint = _np.zeros(2 * n + 1, dtype=_np.int8)
    joint[1:2 * n:2] = st[:n]
    joint[2 * n] = 1
    return _row(joint, -99, 2, n)


def _lp_met_obs(log_theta, log_d_pt, log_d_mt, state_mt, n_met=None):
    """likelihood.py:419-438."""
    return _eval(log_theta, log_d_pt, log_d_mt, _mt_row(state_mt), False)


def _grad_met_obs(log_theta, log_d_p, log_d_m, state_met, n_met=None):
    """likelihood.py:481-512: (log-prob, d_theta, d_dp, d_dm)."""
    return _eval(log_theta, log_d_p, log_d_m, _mt_row(state_met), True)
