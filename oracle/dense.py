"""First-principles dense oracle (NumPy) for small cases.

TEST INFRASTRUCTURE ONLY (same import rules as oracle/metmhn_oracle.py).

Independent of the Kronecker pass structure: builds the restricted generator Q
of the joint PT/MT chain entry by entry from its transition rules and evaluates
the marginal log-likelihoods with dense linear algebra.  Used to cross-check the
restatement in metmhn_oracle.py, the golden vectors and the HIP kernels.

Transition rules (the model behind metmhn/jx/kronvec.py:214-496; theta_ij =
exp(log_theta[i, j]) is the multiplicative effect of event j on event i):
  before seeding (seed = 0) PT and MT are one lineage: event i fires in BOTH
    tumours at once with rate theta_ii * prod_{j in PT(x)} theta_ij, and seeding
    fires with rate theta_nn * prod_{j in PT(x)} theta_nj;
  after seeding PT_i fires with rate theta_ii * prod_{j in PT(x)} theta_ij and
    MT_i with rate theta_ii * theta_in * prod_{j in MT(x)} theta_ij.
States in which PT != MT while seed = 0 are unreachable and carry zero rates.
The restricted space of an observation `state` keeps the 2^k states that are
subsets of the observed ones; transitions that leave it only feed the diagonal.
"""
from __future__ import annotations

import numpy as np


def _slots(state):
    """Active slots in index-bit order: list of (event, tumour) with tumour 0=PT, 1=MT, 2=seed."""
    n = (len(state) - 1) // 2
    out = []
    for j in range(n):
        if state[2 * j]:
            out.append((j, 0))
        if state[2 * j + 1]:
            out.append((j, 1))
    if state[-1]:
        out.append((n, 2))
    return out


def joint_Q(log_theta, state):
    """Dense restricted generator Q[to, from] (columns sum to <= 0)."""
    th = np.exp(np.asarray(log_theta, dtype=np.float64))
    n = th.shape[0] - 1
    slots = _slots(state)
    k = len(slots)
    bit = {s: b for b, s in enumerate(slots)}
    Q = np.zeros((2 ** k, 2 ** k))
    for x in range(2 ** k):
        pt = {e for b, (e, t) in enumerate(slots) if t == 0 and x >> b & 1}
        mt = {e for b, (e, t) in enumerate(slots) if t == 1 and x >> b & 1}
        seeded = (n, 2) in bit and (x >> bit[(n, 2)] & 1) == 1
        if not seeded:
            if pt != mt:
                continue
            for i in range(n):
                if i in pt:
                    continue
                r = th[i, i] * np.prod([th[i, j] for j in pt])
                Q[x, x] -= r
                if (i, 0) in bit and (i, 1) in bit:
                    Q[x | 1 << bit[(i, 0)] | 1 << bit[(i, 1)], x] += r
            r = th[n, n] * np.prod([th[n, j] for j in pt])
            Q[x, x] -= r
            if (n, 2) in bit:
                Q[x | 1 << bit[(n, 2)], x] += r
        else:
            for i in range(n):
                if i not in pt:
                    r = th[i, i] * np.prod([th[i, j] for j in pt])
                    Q[x, x] -= r
                    if (i, 0) in bit:
                        Q[x | 1 << bit[(i, 0)], x] += r
                if i not in mt:
                    r = th[i, i] * th[i, n] * np.prod([th[i, j] for j in mt])
                    Q[x, x] -= r
                    if (i, 1) in bit:
                        Q[x | 1 << bit[(i, 1)], x] += r
    return Q


def joint_D(log_d_p, log_d_m, state):
    """Observation-rate diagonals D_p, D_m on the restricted joint space."""
    dp = np.exp(np.asarray(log_d_p, dtype=np.float64))
    dm = np.exp(np.asarray(log_d_m, dtype=np.float64))
    slots = _slots(state)
    k = len(slots)
    Dp = np.ones(2 ** k)
    Dm = np.ones(2 ** k)
    for x in range(2 ** k):
        seeded = False
        for b, (e, t) in enumerate(slots):
            if x >> b & 1:
                if t == 0:
                    Dp[x] *= dp[e]
                elif t == 1:
                    Dm[x] *= dm[e]
                else:
                    seeded = True
        if seeded:
            Dp[x] *= dp[-1]
            Dm[x] *= dm[-1]
        else:
            Dm[x] = 0.0
    return Dp, Dm


def single_Q(theta, state):
    """Dense generator of a single-tumour MHN (theta NOT logarithmic) restricted to `state` (len n+1)."""
    ev = [j for j in range(len(state)) if state[j]]
    k = len(ev)
    N = theta.shape[0]
    Q = np.zeros((2 ** k, 2 ** k))
    for x in range(2 ** k):
        S = [ev[b] for b in range(k) if x >> b & 1]
        for i in range(N):
            if i in S:
                continue
            r = theta[i, i] * np.prod([theta[i, j] for j in S])
            Q[x, x] -= r
            if i in ev:
                Q[x | 1 << ev.index(i), x] += r
    return Q


def _single_D(d, state):
    ev = [j for j in range(len(state)) if state[j]]
    k = len(ev)
    D = np.ones(2 ** k)
    for x in range(2 ** k):
        for b in range(k):
            if x >> b & 1:
                D[x] *= d[ev[b]]
    return D


def patient_lp(log_theta, log_d_p, log_d_m, row):
    """log-probability of one `dat` row, dense algebra."""
    log_theta = np.asarray(log_theta, dtype=np.float64)
    row = np.asarray(row)
    n = (row.shape[0] - 3) // 2
    th = np.exp(log_theta)
    dp = np.exp(np.asarray(log_d_p, dtype=np.float64))
    dm = np.exp(np.asarray(log_d_m, dtype=np.float64))
    th_pt = th.copy()
    th_pt[:n, n] = 1.0
    typ, order = int(row[-1]), int(row[-2])
    if typ in (0, 1):
        st = row[:-2:2]
        if typ == 0 and st.sum() == 0:
            return -np.log(1.0 + np.trace(th))
        Q = single_Q(th_pt, st)
        D = _single_D(dp, st)
        e0 = np.zeros(Q.shape[0])
        e0[0] = 1.0
        return np.log(D[-1] * np.linalg.solve(np.diag(D) - Q, e0)[-1])
    if typ == 2:
        st = np.append(row[1:-2:2], 1)
        Q = single_Q(th, st)
        k = int(st.sum())
        D = np.empty(2 ** k)
        Dp_ = _single_D(np.append(dp[:n], 1.0), st)
        Dm_ = _single_D(dm, st)
        half = 2 ** (k - 1)
        D[:half] = Dp_[:half]
        D[half:] = Dm_[half:]
        e0 = np.zeros(2 ** k)
        e0[0] = 1.0
        return np.log(D[-1] * np.linalg.solve(np.diag(D) - Q, e0)[-1])
    st = row[:2 * n + 1]
    slots = _slots(st)
    k = len(slots)
    Q = joint_Q(log_theta, st)
    Dp, Dm = joint_D(log_d_p, log_d_m, st)
    e0 = np.zeros(2 ** k)
    e0[0] = 1.0
    pi = np.linalg.solve(np.diag(Dp + Dm) - Q, e0)
    maskP = sum(1 << b for b, (e, t) in enumerate(slots) if t == 0)
    maskM = sum(1 << b for b, (e, t) in enumerate(slots) if t == 1)
    seedb = 1 << (k - 1)
    tot = 0.0
    if order in (0, 1):
        idx = [x for x in range(2 ** k) if (x & maskP) == maskP and x & seedb]
        met = np.append(st[1::2], 1)
        QM = single_Q(th, met)
        DM = _single_D(dm, met)
        v = np.zeros(QM.shape[0])
        v[len(v) // 2:] = (Dp * pi)[idx]
        tot += DM[-1] * np.linalg.solve(np.diag(DM) - QM, v)[-1]
    if order != 1:
        idx = [x for x in range(2 ** k) if (x & maskM) == maskM and x & seedb]
        prim = st[0::2]
        QP = single_Q(th_pt, prim)
        DP = _single_D(dp, prim)
        v = np.zeros(QP.shape[0])
        v[len(v) // 2:] = (Dm * pi)[idx]
        tot += DP[-1] * np.linalg.solve(np.diag(DP) - QP, v)[-1]
    return np.log(tot)


def score(log_theta, log_d_p, log_d_m, dat, perc_met):
    dat = np.asarray(dat)
    s_em = s_pt = 0.0
    for r in dat:
        lp = patient_lp(log_theta, log_d_p, log_d_m, r)
        if r[-1] == 0:
            s_pt += lp
        else:
            s_em += lp
    n_em = float(dat[:, -3].sum())
    n_nm = dat.shape[0] - n_em
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em * n_nm != 0 else 1.0
    return (w * s_em + s_pt) / (w * n_em + n_nm)
