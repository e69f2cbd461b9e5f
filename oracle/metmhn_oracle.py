"""CPU oracle (NumPy, fp64) for the metMHN likelihood/gradient hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module; the product path
(metmhn_amd/) never does and fails loudly when its HIP library is missing.

This is a restatement of the reference algorithm (cbg-ethz/metMHN @ 2024_08_07,
paths below relative to /root/reference) with the SAME pass structure: every
Kronecker summand is applied factor by factor to the lowest one or two index
bits of the restricted vector, followed by a rotation of those bits to the top
(`reshape(-1, w, 'C') @ T` then `flatten('F')`).  It is table driven instead of
one function per factor, but every table entry cites the primitive it restates.

Pinned against the reference's own source executed under tests/tools/jax_standin
(see tests/tools/make_golden.py and tests/golden/*.npz) and against the
first-principles dense oracle in oracle/dense.py.

Layout facts (metmhn/jx/kronvec.py:223-250, metmhn/state.py:248-261):
  state = int8[2n+1] = [PT_0, MT_0, ..., PT_{n-1}, MT_{n-1}, seed]
  restricted vectors have length 2^k, k = #ones in state; index bit b <-> b-th
  active slot (LSB first), for an event active in both tumours the PT bit is
  the lower one; seeding (when active) is the MSB.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# factor application:  p.reshape(-1, w) @ T  ->  flatten('F')
# --------------------------------------------------------------------------


def _apply(p: np.ndarray, t) -> np.ndarray:
    """Apply one Kronecker factor to the lowest bit(s) and rotate them to the top.

    t is None (event inactive: k1d00, kronvec.py:31), a float (k1ns, :209:
    scalar multiple, no bits consumed), a length-w vector (diagonal factor) or
    a (w, w) matrix written for `p_rows @ t` (k2ntt/k4n*, kronvec.py:82-147).
    """
    if t is None:
        return p
    if np.isscalar(t):
        return t * p
    t = np.asarray(t, dtype=np.float64)
    w = t.shape[0]
    m = p.reshape((-1, w), order="C")
    m = m * t if t.ndim == 1 else m @ t
    return m.flatten(order="F")


def _sel(state, j):
    """lax.switch selector of event j: PT + 2*MT (kronvec.py:225)."""
    return int(state[2 * j]) + 2 * int(state[2 * j + 1])


# passive ("diagonal") factor tables, indexed by the selector ---------------
def _f_sync(c, th):      # kronvec.py:226  [k1d00, k2d10, k2d10, k4d100t]
    return (None, [1., 0.], [1., 0.], [1., 0., 0., th])[c]


def _f_prim(c, th):      # kronvec.py:302  [k1d00, k2d1t, k2d11, k4d1t1t]
    return (None, [1., th], [1., 1.], [1., th, 1., th])[c]


def _f_met(c, th):       # kronvec.py:373  [k1d00, k2d11, k2d1t, k4d11tt]
    return (None, [1., 1.], [1., th], [1., 1., th, th])[c]


# acting factors -------------------------------------------------------------
def _k2ntt(th, diag, transpose):      # kronvec.py:82-93
    t = np.zeros((2, 2))
    if transpose:
        t[1, 0] = th
    else:
        t[0, 1] = th
    if diag:
        t[0, 0] = -th
    return t


def _k4(th, diag, transpose, moves):  # kronvec.py:96-147 (k4ns/k4np/k4nm)
    t = np.zeros((4, 4))
    for src, dst in moves:
        if transpose:
            t[dst, src] = th
        else:
            t[src, dst] = th
        if diag:
            t[src, src] = -th
    return t


def _a_sync(c, th, diag, transpose):  # kronvec.py:238 [k1ns, k2dt0, k2dt0, k4ns]
    if c == 0:
        return -th
    if c in (1, 2):
        return [-th, 0.]
    return _k4(th, diag, transpose, [(0, 3)])


def _a_prim(c, th, diag, transpose):  # kronvec.py:315 [k1ns, k2ntt, k2dtt, k4np]
    if c == 0:
        return -th
    if c == 1:
        return _k2ntt(th, diag, transpose)
    if c == 2:
        return [-th, -th]
    return _k4(th, diag, transpose, [(0, 1), (2, 3)])


def _a_met(c, th, diag, transpose):   # kronvec.py:386 [k1ns, k2dtt, k2ntt, k4nm]
    if c == 0:
        return -th
    if c == 1:
        return [-th, -th]
    if c == 2:
        return _k2ntt(th, diag, transpose)
    return _k4(th, diag, transpose, [(0, 2), (1, 3)])


# --------------------------------------------------------------------------
# kronvec: one summand at a time (kronvec.py:214-539)
# --------------------------------------------------------------------------


def kronvec_sync(log_theta, p, i, state, diag=True, transpose=False):
    """kronvec.py:214-287."""
    n = log_theta.shape[0] - 1
    if (not diag) and int(state[2 * i]) + int(state[2 * i + 1]) != 2:
        return p * 0.0
    th = np.exp(log_theta[i, :])
    for j in range(n):
        c = _sel(state, j)
        p = _apply(p, _a_sync(c, th[i], diag, transpose) if j == i else _f_sync(c, th[j]))
    if state[-1] == 1:                      # kronvec.py:246-250  k2d10
        p = _apply(p, [1., 0.])
    return p


def kronvec_prim(log_theta, p, i, state, diag=True, transpose=False):
    """kronvec.py:290-359."""
    n = log_theta.shape[0] - 1
    if ((not diag) and state[2 * i] == 0) or state[-1] == 0:
        return p * 0.0
    th = np.exp(log_theta[i, :])
    for j in range(n):
        c = _sel(state, j)
        p = _apply(p, _a_prim(c, th[i], diag, transpose) if j == i else _f_prim(c, th[j]))
    return _apply(p, [0., 1.])              # kronvec.py:323  k2d01


def kronvec_met(log_theta, p, i, state, diag=True, transpose=False):
    """kronvec.py:362-431."""
    n = log_theta.shape[0] - 1
    if ((not diag) and state[2 * i + 1] == 0) or state[-1] == 0:
        return p * 0.0
    th = np.exp(log_theta[i, :])
    for j in range(n):
        c = _sel(state, j)
        p = _apply(p, _a_met(c, th[i], diag, transpose) if j == i else _f_met(c, th[j]))
    return _apply(p, [0., th[n]])           # kronvec.py:395  k2d0t(theta_in)


def kronvec_seed(log_theta, p, state, diag=True, transpose=False):
    """kronvec.py:434-496."""
    n = log_theta.shape[0] - 1
    if (not diag) and state[-1] == 0:
        return p * 0.0
    th = np.exp(log_theta[n, :])
    for j in range(n):
        p = _apply(p, _f_sync(_sel(state, j), th[j]))
    if state[-1] == 1:
        return _apply(p, _k2ntt(th[n], diag, transpose))
    return -th[n] * p


def kronvec(log_theta, p, state, diag=True, transpose=False):
    """y = Q p restricted to `state` (kronvec.py:499-539)."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    n = log_theta.shape[0] - 1
    y = np.zeros_like(p)
    for i in range(n):
        y = y + kronvec_sync(log_theta, p, i, state, diag, transpose)
        y = y + kronvec_prim(log_theta, p, i, state, diag, transpose)
        y = y + kronvec_met(log_theta, p, i, state, diag, transpose)
    return y + kronvec_seed(log_theta, p, state, diag, transpose)


# --------------------------------------------------------------------------
# diagonal of Q (kronvec.py:713-999)
# --------------------------------------------------------------------------


def kron_diag(log_theta, state, n_state):
    log_theta = np.asarray(log_theta, dtype=np.float64)
    n = log_theta.shape[0] - 1
    y = np.zeros(2 ** n_state)
    for i in range(n):
        th = np.exp(log_theta[i, :])
        # sync part, kronvec.py:713-765  acting: [k1ns, k2dt0, k2dt0, k4dt000]
        d = np.ones(2 ** n_state)
        for j in range(n):
            c = _sel(state, j)
            if j == i:
                t = (-th[i], [-th[i], 0.], [-th[i], 0.], [-th[i], 0., 0., 0.])[c]
            else:
                t = _f_sync(c, th[j])
            d = _apply(d, t)
        if state[-1] == 1:
            d = _apply(d, [1., 0.])
        y = y + d
        if state[-1] == 1:
            # prim part, kronvec.py:768-841  acting: [k1ns, k2dt0, k2dtt, k4dt0t0]
            d = np.ones(2 ** n_state)
            for j in range(n):
                c = _sel(state, j)
                if j == i:
                    t = (-th[i], [-th[i], 0.], [-th[i], -th[i]], [-th[i], 0., -th[i], 0.])[c]
                else:
                    t = _f_prim(c, th[j])
                d = _apply(d, t)
            y = y + _apply(d, [0., 1.])
            # met part, kronvec.py:844-916  acting: [k1ns, k2dtt, k2dt0, k4dtt00]
            d = np.ones(2 ** n_state)
            for j in range(n):
                c = _sel(state, j)
                if j == i:
                    t = (-th[i], [-th[i], -th[i]], [-th[i], 0.], [-th[i], -th[i], 0., 0.])[c]
                else:
                    t = _f_met(c, th[j])
                d = _apply(d, t)
            y = y + _apply(d, [0., th[n]])
    # seeding part, kronvec.py:919-962
    th = np.exp(log_theta[n, :])
    d = np.ones(2 ** n_state)
    for j in range(n):
        d = _apply(d, _f_sync(_sel(state, j), th[j]))
    d = _apply(d, [-th[n], 0.]) if state[-1] == 1 else -th[n] * d
    return y + d


# --------------------------------------------------------------------------
# observation-rate diagonals (kronvec.py:574-710)
# --------------------------------------------------------------------------


def diag_scal_p(log_d_p, state, p):
    """kronvec.py:574-602."""
    d = np.exp(np.asarray(log_d_p, dtype=np.float64))
    n = d.shape[0] - 1
    p = np.asarray(p, dtype=np.float64)
    for j in range(n):
        p = _apply(p, _f_prim(_sel(state, j), d[j]))
    return _apply(p, [1., d[-1]])


def diag_scal_m(log_d_m, state, p):
    """kronvec.py:646-671."""
    d = np.exp(np.asarray(log_d_m, dtype=np.float64))
    n = d.shape[0] - 1
    p = np.asarray(p, dtype=np.float64)
    for j in range(n):
        p = _apply(p, _f_met(_sel(state, j), d[j]))
    return _apply(p, [0., d[-1]])


def partial_diag_scal_p(log_d_p, state, p, i):
    """kronvec.py:605-644."""
    d = np.exp(np.asarray(log_d_p, dtype=np.float64))
    n = d.shape[0] - 1
    p = np.asarray(p, dtype=np.float64)
    which = int(state[2 * i]) + int(i == n)
    if which == 0:
        return p * 0.0
    if which == 2:                                   # partial_le, :635-639
        out = diag_scal_p(log_d_p, state, p).reshape((-1, 2), order="F").copy()
        out[:, 0] = 0.0
        return out.ravel(order="F")
    for j in range(n):
        c = _sel(state, j)
        if j == i:                                   # :618-621  k2d0t | k4d0t0t
            t = [0., d[i]] if c in (1, 2) else [0., d[i], 0., d[i]]
        else:
            t = _f_prim(c, d[j])
        p = _apply(p, t)
    return _apply(p, [1., d[-1]])


def partial_diag_scal_m(log_d_m, state, p, i):
    """kronvec.py:674-710."""
    d = np.exp(np.asarray(log_d_m, dtype=np.float64))
    n = d.shape[0] - 1
    p = np.asarray(p, dtype=np.float64)
    which = int(state[min(2 * n, 2 * i + 1)]) + int(i == n)
    if which == 0:
        return p * 0.0
    if which == 2:
        return diag_scal_m(log_d_m, state, p)
    for j in range(n):
        c = _sel(state, j)
        if j == i:                                   # :696-699  k2d0t | k4d00tt
            t = [0., d[i]] if c in (1, 2) else [0., 0., d[i], d[i]]
        else:
            t = _f_met(c, d[j])
        p = _apply(p, t)
    return _apply(p, [0., d[-1]])


# --------------------------------------------------------------------------
# compatible-state mask (kronvec.py:1031-1095)
# --------------------------------------------------------------------------


def obs_states(n_joint, state, pt_first=True):
    n = (len(state) - 1) // 2
    p = np.ones(2 ** n_joint)
    keep2 = [0., 1.]
    rot2 = [1., 1.]
    tab = {  # selector -> factor, kronvec.py:1075-1084
        True: (None, keep2, rot2, [0., 1., 0., 1.]),     # keep_col2 / shuffle / keep_col1_3
        False: (None, rot2, keep2, [0., 0., 1., 1.]),    # shuffle / keep_col2 / keep_col2_3
    }[bool(pt_first)]
    for j in range(n):
        p = _apply(p, tab[_sel(state, j)])
    if state[-1] != 0:
        p = _apply(p, keep2)
    return p


def obs_indices(n_joint, state, pt_first, n_single):
    """jnp.where(mask == 1, size=2**(n_single-1))[0] (likelihood.py:280,342,375)."""
    idx = np.nonzero(obs_states(n_joint, state, pt_first) == 1.0)[0]
    size = 2 ** (n_single - 1)
    out = np.zeros(size, dtype=np.int64)
    m = min(size, idx.shape[0])
    out[:m] = idx[:m]
    return out


# --------------------------------------------------------------------------
# joint resolvent (likelihood.py:231-262)
# --------------------------------------------------------------------------


def R_i_inv_vec(log_theta, log_d_p, log_d_m, x, state, state_size, transpose=False):
    x = np.asarray(x, dtype=np.float64)
    ones = np.ones_like(x)
    lidg = -1.0 / (kron_diag(log_theta, state, state_size)
                   - (diag_scal_p(log_d_p, state, ones) + diag_scal_m(log_d_m, state, ones)))
    y = lidg * x
    for _ in range(state_size + 1):
        y = lidg * (kronvec(log_theta, y, state, diag=False, transpose=transpose) + x)
    return y


# --------------------------------------------------------------------------
# joint gradient pieces (likelihood.py:25-228)
# --------------------------------------------------------------------------


def _peel(vecs, w):
    """reshape(-1, w, 'C') of several vectors."""
    return [v.reshape((-1, w), order="C") for v in vecs]


def _rot(mats):
    return [m.flatten(order="F") for m in mats]


def _deriv_no_seed(i, d_th_i, x, y, log_theta, state, n):
    """likelihood.py:125-161 (reducers f0-f3, t1, t12, t3 at :25-107)."""
    s = x * kronvec_sync(log_theta, y, i, state)
    p = x * kronvec_prim(log_theta, y, i, state)
    m = x * kronvec_met(log_theta, y, i, state)
    d_th_i = d_th_i.copy()
    d_th_i[-1] = m.sum()
    for j in range(n):
        c = _sel(state, j)
        if j == i:
            if c == 0:                                       # t1
                z = s.sum() + p.sum() + m.sum()
            elif c in (1, 2):                                # t12
                S, P, M = _peel((s, p, m), 2)
                z = S[:, 0].sum() + P.sum() + M.sum()
                s, p, m = _rot((S, P, M))
            else:                                            # t3
                S, P, M = _peel((s, p, m), 4)
                z = S.sum() + P.sum() + M.sum()
                s, p, m = _rot((S, P, M))
        else:
            if c == 0:                                       # f0
                z = 0.0
            elif c == 1:                                     # f1
                S, P, M = _peel((s, p, m), 2)
                z = P[:, 1].sum()
                s, p, m = _rot((S, P, M))
            elif c == 2:                                     # f2
                S, P, M = _peel((s, p, m), 2)
                z = M[:, 1].sum()
                s, p, m = _rot((S, P, M))
            else:                                            # f3
                S, P, M = _peel((s, p, m), 4)
                z = S[:, 3].sum() + P[:, [1, 3]].sum() + M[:, [2, 3]].sum()
                s, p, m = _rot((S, P, M))
        d_th_i[j] = z
    return d_th_i


def x_partial_Q_y(log_theta, x, y, state):
    """likelihood.py:163-201."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = log_theta.shape[0] - 1
    z = np.zeros((n + 1, n + 1))
    for j in range(n):
        z[j, :] = _deriv_no_seed(j, z[:, j], x, y, log_theta, state, n)
    zs = x * kronvec_seed(log_theta, y, state)
    z[-1, -1] = zs.sum()
    for j in range(n):
        c = _sel(state, j)
        if c == 0:                                           # z0
            val = 0.0
        elif c in (1, 2):                                    # z1
            zs = zs.reshape((-1, 2), order="C").flatten(order="F")
            val = 0.0
        else:                                                # z3
            Z = zs.reshape((-1, 4), order="C")
            val = Z[:, 3].sum()
            zs = Z.flatten(order="F")
        z[-1, j] = val
    return z


def x_partial_D_y(log_d_m, log_d_p, state, x, y):
    """likelihood.py:204-228.  NOTE the (log_d_m, log_d_p) argument order and the
    (d_dp, d_dm) return order - reproduced from the reference."""
    n = np.asarray(log_d_m).shape[0]
    d_dp = np.zeros(n)
    d_dm = np.zeros(n)
    for i in range(n):
        d_dp[i] = np.dot(x, partial_diag_scal_p(log_d_p, state, y, i))
        d_dm[i] = np.dot(x, partial_diag_scal_m(log_d_m, state, y, i))
    return d_dp, d_dm


# --------------------------------------------------------------------------
# single-tumour ("vanilla") MHN on n+1 events (vanilla.py)
# --------------------------------------------------------------------------


def diagnosis_theta(log_theta, log_diag_rates):
    """kronvec.py:7-21: subtract log d_j from every off-diagonal entry of column j."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    out = log_theta - np.asarray(log_diag_rates, dtype=np.float64)[None, :]
    np.fill_diagonal(out, np.diagonal(log_theta))
    return out


def v_kronvec_i(log_theta, p, i, state, diag=True, transpose=False):
    """vanilla.py:21-75."""
    n = log_theta.shape[0]
    if (not diag) and state[i] != 1:
        return 0.0 * p
    th = np.exp(log_theta[i, :])
    for j in range(n):
        if j == i:
            t = -th[i] if state[i] == 0 else _k2ntt(th[i], diag, transpose)
        else:
            t = None if state[j] == 0 else [1., th[j]]
        p = _apply(p, t)
    return p


def v_kronvec(log_theta, p, state, diag=True, transpose=False):
    """vanilla.py:78-106."""
    n = log_theta.shape[0]
    return np.sum([v_kronvec_i(log_theta, p, i, state, diag, transpose) for i in range(n)], axis=0)


def v_kron_diag(log_theta, state, ones):
    """vanilla.py:206-260."""
    n = log_theta.shape[0]
    out = np.zeros_like(ones)
    for i in range(n):
        th = np.exp(log_theta[i, :])
        d = ones
        for j in range(n):
            if j == i:
                t = -th[i] if state[i] == 0 else [-th[i], 0.]
            else:
                t = None if state[j] == 0 else [1., th[j]]
            d = _apply(d, t)
        out = out + d
    return out


def v_R_inv_vec(log_theta, x, state, d_rates=1.0, transpose=False):
    """vanilla.py:269-305."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    state_size = int(np.log2(x.shape[0]))
    lidg = -1.0 / (v_kron_diag(log_theta, state, np.ones_like(x)) - d_rates)
    y = lidg * x
    for _ in range(state_size + 1):
        y = lidg * (v_kronvec(log_theta, y, state, False, transpose) + x)
    return y


def v_x_partial_Q_y(log_theta, x, y, state):
    """vanilla.py:328-393."""
    n = log_theta.shape[0]
    val = np.zeros((n, n))
    for i in range(n):
        z = x * v_kronvec_i(log_theta, y, i, state)
        for j in range(n):
            if j == i:
                if state[i] == 0:
                    val[i, j] = z.sum()
                else:
                    Z = z.reshape((-1, 2), order="C")
                    val[i, j] = Z.sum()
                    z = Z.flatten(order="F")
            else:
                if state[j] == 0:
                    val[i, j] = 0.0
                else:
                    Z = z.reshape((-1, 2), order="C")
                    val[i, j] = Z[:, 1].sum()
                    z = Z.flatten(order="F")
    d_diag = -np.sum(val, axis=0) + np.diagonal(val)
    return val, d_diag


def v_gradient(log_theta, state, p_0):
    """vanilla.py:396-418."""
    p_theta = v_R_inv_vec(log_theta, p_0, state)
    x = np.zeros_like(p_theta)
    x[-1] = 1.0 / p_theta[-1]
    x = v_R_inv_vec(log_theta, x, state, transpose=True)
    d_th, d_diag = v_x_partial_Q_y(log_theta, x, p_theta, state)
    return d_th, d_diag, p_theta


def v_scal_d_pt(log_d_p, log_d_m, state, vec):
    """vanilla.py:125-142."""
    d_p = np.exp(np.asarray(log_d_p, dtype=np.float64))
    d_m = np.exp(np.asarray(log_d_m, dtype=np.float64))
    n = d_m.shape[0]
    a, b = vec, vec
    for j in range(n - 1):
        if state[j] == 1:
            a, b = _apply(a, [1., d_p[j]]), _apply(b, [1., d_m[j]])
    return _apply(a, [1., 0.]), _apply(b, [0., d_m[-1]])


def v_d_scal_d_pt(log_d_p, log_d_m, state, vec, i):
    """vanilla.py:145-187."""
    d_p = np.exp(np.asarray(log_d_p, dtype=np.float64))
    d_m = np.exp(np.asarray(log_d_m, dtype=np.float64))
    n = d_p.shape[0] - 1
    which = int(state[i]) + int(i == n)
    if which == 0:
        return 0.0 * vec, 0.0 * vec
    if which == 2:
        return 0.0 * vec, v_scal_d_pt(log_d_p, log_d_m, state, vec)[1]
    a, b = vec, vec
    for j in range(n):
        if j == i:
            a, b = _apply(a, [0., d_p[i]]), _apply(b, [0., d_m[i]])
        elif state[j] == 1:
            a, b = _apply(a, [1., d_p[j]]), _apply(b, [1., d_m[j]])
    return _apply(a, [1., 0.]), _apply(b, [0., d_m[-1]])


def v_x_partial_D_y(log_d_p, log_d_m, state, x, y):
    """vanilla.py:190-203."""
    n = np.asarray(log_d_p).shape[0]
    a = np.zeros(n)
    b = np.zeros(n)
    for i in range(n):
        dp_, dm_ = v_d_scal_d_pt(log_d_p, log_d_m, state, y, i)
        a[i] = np.dot(x, dp_)
        b[i] = np.dot(x, dm_)
    return a, b


# --------------------------------------------------------------------------
# k = 1 closed forms (one_event.py)
# --------------------------------------------------------------------------


def _one_small_Q(log_theta):
    """one_event.py:10-25."""
    base = np.diagonal(log_theta)
    b_r = np.exp(base[:-1])
    e_seed = np.exp(log_theta[:-1, -1]) + 1.0
    return np.array([[-np.exp(base).sum(), 0.0],
                     [np.exp(log_theta[-1, -1]), -np.sum(b_r * e_seed)]])


def _one_R_i_inv_vec(log_theta, x, d_p_le, d_m_le, transpose=False):
    """one_event.py:56-85."""
    R = np.diag([1.0, d_p_le + d_m_le]) - _one_small_Q(log_theta)
    b = np.array(x, dtype=np.float64)
    if not transpose:
        b[0] /= R[0, 0]
        b[1] += -(b[0] * R[1, 0])
        b[1] /= R[1, 1]
    else:
        b[1] /= R[1, 1]
        b[0] += -(b[1] * R[1, 0])
        b[0] /= R[0, 0]
    return b


def _one_x_partial_Q_y(log_theta, x, y):
    """one_event.py:88-113."""
    z = np.zeros_like(log_theta)
    n = log_theta.shape[0]
    for i in range(n):
        th_ii = np.exp(log_theta[i, i])
        th_iM = np.exp(log_theta[i, -1])
        z[i, i] = -th_ii * (x @ np.diag([1.0, 1.0 + th_iM]) @ y)
        z[i, -1] = x @ np.diag([0.0, -th_ii * th_iM]) @ y
    th_MM = np.exp(log_theta[-1, -1])
    z[-1, -1] = x @ np.array([[-th_MM, 0.0], [th_MM, 0.0]]) @ y
    return z


def _one_q_inv_deriv_pth(log_theta, d_p_le, d_m_le, q, p):
    """one_event.py:116-137."""
    n = log_theta.shape[0]
    q = _one_R_i_inv_vec(log_theta, q, d_p_le, d_m_le, True)
    g_2 = _one_x_partial_Q_y(log_theta, q, p)
    d_dm_2 = np.zeros(n)
    d_dm_2[-1] = np.dot(q * np.array([0.0, d_m_le]), p)
    d_dp_2 = np.zeros(n)
    d_dp_2[-1] = np.dot(q * np.array([0.0, d_p_le]), p)
    return g_2, d_dp_2, d_dm_2


def _theta_pt(log_theta, log_d_p):
    """likelihood.py:313-314: theta[:n, n] <- 0 then column scaling by d_p."""
    t = np.array(log_theta, dtype=np.float64, copy=True)
    t[:-1, -1] = 0.0
    return diagnosis_theta(t, log_d_p)


def _one_marginal_pt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, met):
    """one_event.py:229-263."""
    d_p_le = np.exp(log_d_p[-1])
    cond = np.append(np.zeros(1), pTh1[-1]) * d_p_le
    th_dm = diagnosis_theta(log_theta, log_d_m)
    g_1, d_dm_1, pTh2 = v_gradient(th_dm, met, cond)
    score = pTh2[-1]
    q = np.zeros(2)
    q[-1] = 1.0 / score
    q = v_R_inv_vec(th_dm, q, met, transpose=True)
    p = q * np.array([0.0, d_p_le])
    d_dp_1 = np.zeros_like(log_d_p)
    d_dp_1[-1] = np.dot(p, pTh1)
    return score, g_1, d_dp_1, d_dm_1, p


def _one_marginal_mt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, prim):
    """one_event.py:266-303."""
    d_m_le = np.exp(log_d_m[-1])
    cond = np.append(np.zeros(1), pTh1[-1] * d_m_le)
    th_pt = _theta_pt(log_theta, log_d_p)
    g_1, d_dp_1, pTh2 = v_gradient(th_pt, prim, cond)
    g_1[:-1, -1] = 0.0
    score = pTh2[-1]
    q = np.zeros(2)
    q[-1] = 1.0 / score
    q = v_R_inv_vec(th_pt, q, prim, transpose=True)
    p = q * np.array([0.0, d_m_le])
    d_dm_1 = np.zeros_like(log_d_m)
    d_dm_1[-1] = np.dot(p, pTh1)
    return score, g_1, d_dp_1, d_dm_1, p


def _one_lp_coupled(log_theta, log_d_p, log_d_m, state_joint, order):
    """one_event.py:141-226."""
    d_m_le = np.exp(log_d_m[-1])
    d_p_le = np.exp(log_d_p[-1])
    pTh1 = _one_R_i_inv_vec(log_theta, np.array([1.0, 0.0]), d_p_le, d_m_le)
    met = np.append(state_joint[1::2], 1)
    prim = state_joint[0::2]
    tot = 0.0
    if order in (0, 1):
        tot += v_R_inv_vec(diagnosis_theta(log_theta, log_d_m),
                           np.array([0.0, pTh1[-1] * d_p_le]), met)[-1]
    if order != 1:
        tot += v_R_inv_vec(_theta_pt(log_theta, log_d_p),
                           np.array([0.0, pTh1[-1] * d_m_le]), prim)[-1]
    return np.log(tot)


def _one_g_coupled(log_theta, log_d_p, log_d_m, state_joint, order):
    """one_event.py:307-409."""
    prim = state_joint[::2]
    met = np.append(state_joint[1::2], 1)
    d_m_le = np.exp(log_d_m[-1])
    d_p_le = np.exp(log_d_p[-1])
    pTh1 = _one_R_i_inv_vec(log_theta, np.array([1.0, 0.0]), d_p_le, d_m_le)
    if order == 0:
        pf = _one_marginal_pt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, met)
        mf = _one_marginal_mt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, prim)
        full = pf[0] + mf[0]
        g_2, d_dp_2, d_dm_2 = _one_q_inv_deriv_pth(
            log_theta, d_p_le, d_m_le, (pf[4] * pf[0] + mf[4] * mf[0]) / full, pTh1)
        d_dm = (pf[3] * pf[0] + mf[3] * mf[0]) / full - d_dm_2
        d_dp = (pf[2] * pf[0] + mf[2] * mf[0]) / full - d_dp_2
        g = (pf[1] * pf[0] + mf[1] * mf[0]) / full + g_2
        return np.log(full), g, d_dp, d_dm
    if order == 1:
        sc, g_1, d_dp_1, d_dm_1, p = _one_marginal_pt_first(
            log_theta, log_d_p, log_d_m, pTh1, state_joint, met)
    else:
        sc, g_1, d_dp_1, d_dm_1, p = _one_marginal_mt_first(
            log_theta, log_d_p, log_d_m, pTh1, state_joint, prim)
    g_2, d_dp_2, d_dm_2 = _one_q_inv_deriv_pth(log_theta, d_p_le, d_m_le, p, pTh1)
    return np.log(sc), g_1 + g_2, d_dp_1 - d_dp_2, d_dm_1 - d_dm_2


# --------------------------------------------------------------------------
# per-patient log-probabilities (likelihood.py:265-438)
# --------------------------------------------------------------------------


def _cond_p_obs(vec_joint, state_joint, n_joint, n_single, pt_first):
    """likelihood.py:265-283 (also inlined at :341-345, :374-378)."""
    inds = obs_indices(n_joint, state_joint, pt_first, n_single)
    return np.append(np.zeros(2 ** (n_single - 1)), vec_joint[inds]), inds


def lp_coupled(log_theta, log_d_p, log_d_m, state_joint, n_prim, n_met, order):
    """likelihood.py:286-384 (order 0 / 1 / anything else = 2)."""
    n_joint = n_prim + n_met - 1
    if n_joint == 1:
        return _one_lp_coupled(log_theta, log_d_p, log_d_m, state_joint, order)
    p0 = np.zeros(2 ** n_joint)
    p0[0] = 1.0
    pTh1 = R_i_inv_vec(log_theta, log_d_p, log_d_m, p0, state_joint, n_joint)
    tot = 0.0
    if order in (0, 1):
        v, _ = _cond_p_obs(diag_scal_p(log_d_p, state_joint, pTh1), state_joint, n_joint, n_met, True)
        met = np.append(state_joint[1::2], 1)
        tot += v_R_inv_vec(diagnosis_theta(log_theta, log_d_m), v, met)[-1]
    if order != 1:
        v, _ = _cond_p_obs(diag_scal_m(log_d_m, state_joint, pTh1), state_joint, n_joint, n_prim, False)
        prim = state_joint[0::2]
        tot += v_R_inv_vec(_theta_pt(log_theta, log_d_p), v, prim)[-1]
    return np.log(tot)


def lp_prim_obs(log_theta, log_d_p, state_pt, n_prim):
    """likelihood.py:387-405."""
    p0 = np.zeros(2 ** n_prim)
    p0[0] = 1.0
    return np.log(v_R_inv_vec(_theta_pt(log_theta, log_d_p), p0, state_pt, np.ones_like(p0))[-1])


def lp_prim_obs_az(log_theta):
    """likelihood.py:408-416."""
    return np.log(1.0 / (1.0 + np.sum(np.diag(np.exp(log_theta)))))


def lp_met_obs(log_theta, log_d_p, log_d_m, state_mt, n_met):
    """likelihood.py:419-438."""
    p0 = np.zeros(2 ** n_met)
    p0[0] = 1.0
    d_p, d_m = v_scal_d_pt(log_d_p, log_d_m, state_mt, np.ones(2 ** n_met))
    d_rates = d_p + d_m
    pTh = v_R_inv_vec(log_theta, p0, state_mt, d_rates, False)
    return np.log(pTh[-1] * d_rates[-1])


# --------------------------------------------------------------------------
# per-patient gradients (likelihood.py:441-731)
# --------------------------------------------------------------------------


def grad_prim_obs(log_theta, log_d_p, state_prim, n_prim):
    """likelihood.py:441-461."""
    p0 = np.zeros(2 ** n_prim)
    p0[0] = 1.0
    d_th, d_dp, pTh2 = v_gradient(_theta_pt(log_theta, log_d_p), state_prim, p0)
    d_th[:-1, -1] = 0.0
    return np.log(pTh2[-1]), d_th, d_dp


def grad_prim_obs_az(log_theta):
    """likelihood.py:464-478."""
    br = np.exp(np.diag(log_theta))
    lp = np.log(1.0 / (1.0 + np.sum(br)))
    d_th = 1.0 / np.exp(lp) * np.diag(-br / (1.0 + np.sum(br)) ** 2)
    return lp, d_th, np.zeros(log_theta.shape[0])


def grad_met_obs(log_theta, log_d_p, log_d_m, state_met, n_met):
    """likelihood.py:481-512."""
    p0 = np.zeros(2 ** n_met)
    p0[0] = 1.0
    d_p, d_m = v_scal_d_pt(log_d_p, log_d_m, state_met, np.ones(2 ** n_met))
    d_rates = d_p + d_m
    pTh = v_R_inv_vec(log_theta, p0, state_met, d_rates, False)
    score = pTh[-1]
    q = np.zeros(2 ** n_met)
    q[-1] = 1.0 / score
    _, d_dm_1 = v_x_partial_D_y(log_d_p, log_d_m, state_met, q / d_rates[-1], pTh)
    q = v_R_inv_vec(log_theta, q, state_met, d_rates, True)
    d_dp, d_dm_2 = v_x_partial_D_y(log_d_p, log_d_m, state_met, q, pTh)
    d_th, _ = v_x_partial_Q_y(log_theta, q, pTh, state_met)
    return np.log(score * d_rates[-1]), d_th, -d_dp, d_dm_1 - d_dm_2


def _q_inv_deriv_pth(log_theta, log_d_p, log_d_m, q, p, state_joint, n_joint):
    """likelihood.py:516-537."""
    q = R_i_inv_vec(log_theta, log_d_p, log_d_m, q, state_joint, n_joint, transpose=True)
    g_2 = x_partial_Q_y(log_theta, q, p, state_joint)
    d_dp_2, d_dm_2 = x_partial_D_y(log_d_m, log_d_p, state_joint, q, p)
    return g_2, d_dp_2, d_dm_2


def _marginal_pt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, met, n_joint, n_met):
    """likelihood.py:540-578."""
    scal = diag_scal_p(log_d_p, state_joint, pTh1)
    cond, inds = _cond_p_obs(scal, state_joint, n_joint, n_met, True)
    th_dm = diagnosis_theta(log_theta, log_d_m)
    g_1, d_dm_1, pTh2 = v_gradient(th_dm, met, cond)
    score = pTh2[-1]
    q = np.zeros(2 ** n_met)
    q[-1] = 1.0 / score
    q = v_R_inv_vec(th_dm, q, met, transpose=True)
    p = np.zeros(2 ** n_joint)
    p[inds] = q[2 ** (n_met - 1):]
    d_dp_1, _ = x_partial_D_y(log_d_m, log_d_p, state_joint, p, pTh1)
    return score, g_1, d_dp_1, d_dm_1, p


def _marginal_mt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, prim, n_joint, n_prim):
    """likelihood.py:581-620."""
    scal = diag_scal_m(log_d_m, state_joint, pTh1)
    cond, inds = _cond_p_obs(scal, state_joint, n_joint, n_prim, False)
    th_pt = _theta_pt(log_theta, log_d_p)
    g_1, d_dp_1, pTh2 = v_gradient(th_pt, prim, cond)
    g_1[:-1, -1] = 0.0
    score = pTh2[-1]
    q = np.zeros(2 ** n_prim)
    q[-1] = 1.0 / score
    q = v_R_inv_vec(th_pt, q, prim, transpose=True)
    p = np.zeros(2 ** n_joint)
    p[inds] = q[2 ** (n_prim - 1):]
    _, d_dm_1 = x_partial_D_y(log_d_m, log_d_p, state_joint, p, pTh1)
    return score, g_1, d_dp_1, d_dm_1, p


def g_coupled(log_theta, log_d_p, log_d_m, state_joint, n_prim, n_met, order):
    """likelihood.py:623-731 (order 0 / 1 / anything else = 2)."""
    n_joint = n_prim + n_met - 1
    if n_joint == 1:
        return _one_g_coupled(log_theta, log_d_p, log_d_m, state_joint, order)
    prim = state_joint[::2]
    met = np.append(state_joint[1::2], 1)
    p0 = np.zeros(2 ** n_joint)
    p0[0] = 1.0
    pTh1 = R_i_inv_vec(log_theta, log_d_p, log_d_m, p0, state_joint, n_joint)
    if order == 0:
        pf = _marginal_pt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, met, n_joint, n_met)
        mf = _marginal_mt_first(log_theta, log_d_p, log_d_m, pTh1, state_joint, prim, n_joint, n_prim)
        full = pf[0] + mf[0]
        pf_p = diag_scal_p(log_d_p, state_joint, pf[4]) * pf[0] / full
        mf_p = diag_scal_m(log_d_m, state_joint, mf[4]) * mf[0] / full
        g_2, d_dp_2, d_dm_2 = _q_inv_deriv_pth(log_theta, log_d_p, log_d_m, pf_p + mf_p,
                                                pTh1, state_joint, n_joint)
        d_dm = (pf[3] * pf[0] + mf[3] * mf[0]) / full - d_dm_2
        d_dp = (pf[2] * pf[0] + mf[2] * mf[0]) / full - d_dp_2
        g = (pf[1] * pf[0] + mf[1] * mf[0]) / full + g_2
        return np.log(full), g, d_dp, d_dm
    if order == 1:
        sc, g_1, d_dp_1, d_dm_1, p = _marginal_pt_first(
            log_theta, log_d_p, log_d_m, pTh1, state_joint, met, n_joint, n_met)
        p = diag_scal_p(log_d_p, state_joint, p)
    else:
        sc, g_1, d_dp_1, d_dm_1, p = _marginal_mt_first(
            log_theta, log_d_p, log_d_m, pTh1, state_joint, prim, n_joint, n_prim)
        p = diag_scal_m(log_d_m, state_joint, p)
    g_2, d_dp_2, d_dm_2 = _q_inv_deriv_pth(log_theta, log_d_p, log_d_m, p, pTh1, state_joint, n_joint)
    return np.log(sc), g_1 + g_2, d_dp_1 - d_dp_2, d_dm_1 - d_dm_2


# --------------------------------------------------------------------------
# cohort objective (regularized_optimization.py)
# --------------------------------------------------------------------------


def patient_lp(log_theta, log_d_p, log_d_m, row):
    """One row of `dat` -> (log-prob, is_type0); regularized_optimization.py:75-119."""
    row = np.asarray(row)
    n_mut = (row.shape[0] - 3) // 2
    n_total = n_mut + 1
    typ = int(row[-1])
    if typ in (0, 1):
        st = row[0:2 * n_total - 1:2]
        n_prim = int(st.sum())
        if typ == 0 and n_prim == 0:
            return lp_prim_obs_az(log_theta), True
        return lp_prim_obs(log_theta, log_d_p, st, n_prim), typ == 0
    if typ == 2:
        st = np.append(row[1:2 * n_total - 1:2], 1)
        return lp_met_obs(log_theta, log_d_p, log_d_m, st, int(st.sum())), False
    st = row[0:2 * n_mut + 1]
    n_prim = int(st[::2].sum())
    n_met = int(st[1::2].sum() + 1)
    return lp_coupled(log_theta, log_d_p, log_d_m, st, n_prim, n_met, int(row[-2])), False


def patient_grad(log_theta, log_d_p, log_d_m, row):
    """One row of `dat` -> (lp, d_th, d_dp, d_dm, is_type0); regularized_optimization.py:187-254."""
    row = np.asarray(row)
    n_mut = (row.shape[0] - 3) // 2
    n_total = n_mut + 1
    typ = int(row[-1])
    zero = np.zeros(n_total)
    if typ in (0, 1):
        st = row[:-2:2]
        n_prim = int(st.sum())
        if typ == 0 and n_prim == 0:
            lp, g, dp = grad_prim_obs_az(log_theta)
        else:
            lp, g, dp = grad_prim_obs(log_theta, log_d_p, st, n_prim)
        return lp, g, dp, zero, typ == 0
    if typ == 2:
        st = np.append(row[1:-2:2], 1)
        lp, g, dp, dm = grad_met_obs(log_theta, log_d_p, log_d_m, st, int(st.sum()))
        return lp, g, dp, dm, False
    st = row[0:2 * n_mut + 1]
    n_prim = int(st[::2].sum())
    n_met = int(st[1::2].sum() + 1)
    lp, g, dp, dm = g_coupled(log_theta, log_d_p, log_d_m, st, n_prim, n_met, int(row[-2]))
    return lp, g, dp, dm, False


def _weights(dat, perc_met):
    """regularized_optimization.py:121-128."""
    n_em = float(np.sum(dat[:, -3]))
    n_nm = dat.shape[0] - n_em
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em * n_nm != 0 else 1.0
    return w, w * n_em + n_nm


def score(log_theta, log_d_p, log_d_m, dat, perc_met):
    """regularized_optimization.py:55-130."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    log_d_p = np.asarray(log_d_p, dtype=np.float64)
    log_d_m = np.asarray(log_d_m, dtype=np.float64)
    dat = np.asarray(dat)
    s_em, s_pt = 0.0, 0.0
    for i in range(dat.shape[0]):
        lp, is0 = patient_lp(log_theta, log_d_p, log_d_m, dat[i])
        if is0:
            s_pt += lp
        else:
            s_em += lp
    w, n_full = _weights(dat, perc_met)
    return (w * s_em + s_pt) / n_full


def score_and_grad(log_theta, log_d_p, log_d_m, dat, perc_met):
    """regularized_optimization.py:163-267."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    log_d_p = np.asarray(log_d_p, dtype=np.float64)
    log_d_m = np.asarray(log_d_m, dtype=np.float64)
    dat = np.asarray(dat)
    N = log_theta.shape[0]
    s_em, s_pt = 0.0, 0.0
    g_em, g_pt = np.zeros((N, N)), np.zeros((N, N))
    dp_em, dp_pt = np.zeros(N), np.zeros(N)
    dm_em = np.zeros(N)
    for i in range(dat.shape[0]):
        lp, g, dp, dm, is0 = patient_grad(log_theta, log_d_p, log_d_m, dat[i])
        if is0:
            s_pt += lp
            g_pt += g
            dp_pt += dp
        else:
            s_em += lp
            g_em += g
            dp_em += dp
            dm_em += dm
    w, n_full = _weights(dat, perc_met)
    return ((w * s_em + s_pt) / n_full, (w * g_em + g_pt) / n_full,
            (w * dp_em + dp_pt) / n_full, w * dm_em / n_full)


# penalties (regularized_optimization.py:11-52) ------------------------------


def _L1(v, eps=1e-5):
    return np.sum(np.sqrt(v ** 2 + eps))


def _L1_(v, eps=1e-5):
    return v / np.sqrt(v ** 2 + eps)


def symmetric_penal(params, n_total, eps=1e-5):
    params = np.asarray(params, dtype=np.float64)
    th = params[:n_total ** 2].reshape((n_total, n_total)).copy()
    dp = params[n_total ** 2:n_total * (n_total + 1)]
    dm = params[n_total * (n_total + 1):]
    np.fill_diagonal(th, 0.0)
    root = np.sqrt(th ** 2 + th.T ** 2 - th * th.T + eps)
    pen = 0.5 * (np.sum(root) - n_total * np.sqrt(eps)) + _L1(dp, eps) + _L1(dm, eps)
    pen_ = np.concatenate((((2 * th - th.T) / (2 * root)).flatten(), _L1_(dp, eps), _L1_(dm, eps)))
    return pen, pen_


def _unpack(params, n_total):
    params = np.asarray(params, dtype=np.float64)
    return (params[:n_total ** 2].reshape((n_total, n_total)),
            params[n_total ** 2:n_total * (n_total + 1)], params[n_total * (n_total + 1):])


def score_reg(params, dat, perc_met, penal, w_penal):
    """regularized_optimization.py:133-160."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    pen, _ = penal(params, n_total)
    return np.array(-score(th, dp, dm, dat, perc_met) + w_penal * pen)


def score_and_grad_reg(params, dat, perc_met, penal, w_penal):
    """regularized_optimization.py:270-298."""
    n_total = (np.asarray(dat).shape[1] - 3) // 2 + 1
    th, dp, dm = _unpack(params, n_total)
    sc, g, a, b = score_and_grad(th, dp, dm, dat, perc_met)
    pen, pen_ = penal(params, n_total)
    return np.array(-sc + w_penal * pen), -np.concatenate((g.flatten(), a, b)) + w_penal * pen_
