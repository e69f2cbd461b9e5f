"""Closed-form ("gather") model of the hot path - the arithmetic the HIP kernels implement.

TEST INFRASTRUCTURE ONLY (same import rules as oracle/metmhn_oracle.py).

Where metmhn_oracle.py follows the reference's per-factor pass structure, this
module states the same quantities element-wise, exactly as the device code in
metmhn_amd/csrc does (one lane per state, rates from per-bit product tables,
class-marginal gradient accumulation).  It exists to validate that formulation
against the oracle on the CPU, and as executable documentation of DESIGN.md.

Vocabulary
  slot      one of the 2n+1 entries of `state`; active slots are index bits.
  class     P (primary-tumour bit), M (metastasis bit), S (seeding bit).
  eq(x)     seed(x) = 0 and PT(x) = MT(x): the only seed = 0 states with rates.
  R(i,c,x)  prod_{bits b' of class c set in x} theta[i, ev(b')].
"""
from __future__ import annotations

import numpy as np

P, M, S = 0, 1, 2


class Desc:
    """Bit roles of one restricted space (mirrors `struct Desc` in csrc/desc.h)."""

    def __init__(self, state, mode="joint"):
        state = np.asarray(state).astype(int)
        self.mode = mode
        ev, cls = [], []
        if mode == "joint":
            n = (len(state) - 1) // 2
            for j in range(n):
                if state[2 * j]:
                    ev.append(j); cls.append(P)
                if state[2 * j + 1]:
                    ev.append(j); cls.append(M)
            if state[-1]:
                ev.append(n); cls.append(S)
            self.N = n + 1
        else:                                   # single tumour: every active event is a class-P bit
            for j in range(len(state)):
                if state[j]:
                    ev.append(j); cls.append(P)
            self.N = len(state)
        self.ev, self.cls = np.array(ev, int), np.array(cls, int)
        self.k = len(ev)
        self.maskP = sum(1 << b for b in range(self.k) if cls[b] == P)
        self.maskM = sum(1 << b for b in range(self.k) if cls[b] == M)
        self.seedbit = next((b for b in range(self.k) if cls[b] == S), -1)
        self.pairP = sum(1 << b for b in range(self.k - 1)
                         if cls[b] == P and cls[b + 1] == M and ev[b] == ev[b + 1])
        self.lone = (self.maskP | self.maskM) & ~(self.pairP | self.pairP << 1)
        self.bitP = -np.ones(self.N, int)
        self.bitM = -np.ones(self.N, int)
        for b in range(self.k):
            if cls[b] == P:
                self.bitP[ev[b]] = b
            elif cls[b] == M:
                self.bitM[ev[b]] = b

    def eq(self, x):
        ok = (x & self.lone) == 0
        ok &= ((x & self.pairP) << 1) == (x & (self.pairP << 1))
        if self.seedbit >= 0:
            ok &= (x >> self.seedbit & 1) == 0
        return ok

    def seedset(self, x):
        if self.mode != "joint":
            return np.ones_like(x, bool)
        if self.seedbit < 0:
            return np.zeros_like(x, bool)
        return (x >> self.seedbit & 1) == 1


def _R(th, i, d, c, x):
    """prod over class-c bits set in x of th[i, ev(bit)]."""
    out = np.ones(x.shape)
    for b in range(d.k):
        if d.cls[b] == c:
            out = np.where(x >> b & 1, out * th[i, d.ev[b]], out)
    return out


def _base(th, d, b):
    i = d.ev[b]
    if d.mode == "joint" and d.cls[b] == M:
        return th[i, i] * th[i, d.N - 1]
    return th[i, i]


def _pcls(d, b):
    return P if d.cls[b] == S else d.cls[b]


def offdiag(th, d, p, transpose=False):
    """y = Q_off p (or Q_off^T p), element-wise gather form (kernel k_offdiag)."""
    x = np.arange(2 ** d.k)
    y = np.zeros(2 ** d.k)
    ss = d.seedset(x)
    for b in range(d.k):
        bit = 1 << b
        has = (x >> b & 1) == 1
        src = x & ~bit                                   # rate tables exclude bit b itself
        rate = _base(th, d, b) * _R(th, d.ev[b], d, _pcls(d, b), src)
        if d.cls[b] == S:
            if transpose:
                c = d.eq(x)
                y += np.where(c, rate * p[x | bit], 0.0)
            else:
                c = has & d.eq(x ^ bit)
                y += np.where(c, rate * p[x ^ bit], 0.0)
            continue
        if transpose:
            c = ~has & ss
            y += np.where(c, rate * p[x | bit], 0.0)
        else:
            c = has & ss
            y += np.where(c, rate * p[x ^ bit], 0.0)
        if d.mode == "joint" and (d.pairP >> b & 1):     # synchronised event before seeding
            both = 3 << b
            if transpose:
                c = ((x & both) == 0) & d.eq(x)
                y += np.where(c, rate * p[(x | both) % len(p)], 0.0)
            else:
                c = ((x & both) == both) & d.eq(x ^ both)
                y += np.where(c, rate * p[x ^ both], 0.0)
    return y


def qdiag(th, d):
    """diag(Q) (negative total outflow, incl. transitions leaving the restricted space)."""
    x = np.arange(2 ** d.k)
    out = np.zeros(2 ** d.k)
    n = d.N - 1
    if d.mode != "joint":
        for i in range(d.N):
            free = np.ones_like(x, bool) if d.bitP[i] < 0 else (x >> d.bitP[i] & 1) == 0
            out -= np.where(free, th[i, i] * _R(th, i, d, P, x), 0.0)
        return out
    ss = d.seedset(x)
    e = d.eq(x) if d.seedbit >= 0 else (((x & d.lone) == 0) & (((x & d.pairP) << 1) == (x & (d.pairP << 1))))
    for i in range(n):
        freeP = np.ones_like(x, bool) if d.bitP[i] < 0 else (x >> d.bitP[i] & 1) == 0
        freeM = np.ones_like(x, bool) if d.bitM[i] < 0 else (x >> d.bitM[i] & 1) == 0
        rP = th[i, i] * _R(th, i, d, P, x)
        rM = th[i, i] * th[i, n] * _R(th, i, d, M, x)
        out -= np.where(ss & freeP, rP, 0.0) + np.where(ss & freeM, rM, 0.0)
        out -= np.where(~ss & e & freeP, rP, 0.0)
    out -= np.where(~ss & e, th[n, n] * _R(th, n, d, P, x), 0.0)
    return out


def obs_diag(d, dp, dm):
    """D_p(x), D_m(x) of the joint space (kronvec.py:574-602, 646-671)."""
    x = np.arange(2 ** d.k)
    Dp, Dm = np.ones(2 ** d.k), np.ones(2 ** d.k)
    for b in range(d.k):
        if d.cls[b] == P:
            Dp = np.where(x >> b & 1, Dp * dp[d.ev[b]], Dp)
        elif d.cls[b] == M:
            Dm = np.where(x >> b & 1, Dm * dm[d.ev[b]], Dm)
    ss = d.seedset(x)
    return np.where(ss, Dp * dp[-1], Dp), np.where(ss, Dm * dm[-1], 0.0)


def solve(th, d, dobs, rhs, transpose=False):
    """(diag(dobs) - Q)^-1 rhs by k+1 Jacobi sweeps (likelihood.py:231-262)."""
    lidg = 1.0 / (dobs - qdiag(th, d))
    y = lidg * rhs
    for _ in range(d.k + 1):
        y = lidg * (offdiag(th, d, y, transpose) + rhs)
    return y


def _pdep(vals, mask):
    out = np.zeros_like(vals)
    j = 0
    for b in range(32):
        if mask >> b & 1:
            out |= (vals >> j & 1) << b
            j += 1
    return out


def compat_indices(d, pt_first):
    """Ascending joint indices with all bits of the observed tumour and the seeding bit set."""
    fixed = (d.maskP if pt_first else d.maskM) | (1 << d.seedbit)
    free = d.maskM if pt_first else d.maskP
    cnt = bin(free).count("1")
    return _pdep(np.arange(2 ** cnt), free) | fixed


def grad_accum(th, base, d_cls_bits, ev, N, A_diag, A_bit):
    """Generic small kernel: flows of every event i over one class' subset lattice.

    d_cls_bits: list of the class' bits (local index l <-> subset bit l), ev[l] their events.
    A_diag[S] = -sum p q ; A_bit[l][S] = sum p q(S | l) (defined for l not in S).
    Returns tot[i] = sum of all flows of event i, mar[i, l] = flows from subsets containing l.
    """
    kc = len(ev)
    Sx = np.arange(2 ** kc)
    tot = np.zeros(N)
    mar = np.zeros((N, kc))
    loc = {e: l for l, e in enumerate(ev)}
    for i in range(N):
        if base[i] == 0.0:
            continue
        rate = base[i] * np.ones(2 ** kc)
        for l in range(kc):
            rate = np.where(Sx >> l & 1, rate * th[i, ev[l]], rate)
        if i in loc:
            l = loc[i]
            free = (Sx >> l & 1) == 0
            f = np.where(free, rate * (A_bit[l] + A_diag), 0.0)
        else:
            f = rate * A_diag
        tot[i] = f.sum()
        for l in range(kc):
            mar[i, l] = f[(Sx >> l & 1) == 1].sum()
    return tot, mar


def joint_xQy(th, d, q, p):
    """G[i, j] = q^T (dQ / dlog theta_ij) p on the joint space (likelihood.py:163-201)."""
    N = d.N
    n = N - 1
    G = np.zeros((N, N))
    x = np.arange(2 ** d.k)
    # ---- seed = 1 half: class marginals (kernel k_class_marginals) then small kernel
    if d.seedbit >= 0:
        sb = 1 << d.seedbit
        for c, mask, omask in ((P, d.maskP, d.maskM), (M, d.maskM, d.maskP)):
            bits = [b for b in range(d.k) if d.cls[b] == c]
            kc = len(bits)
            ko = bin(omask).count("1")
            Sx = _pdep(np.arange(2 ** kc), mask)
            Tx = _pdep(np.arange(2 ** ko), omask) | sb
            idx = Sx[:, None] | Tx[None, :]
            W = -(p[idx] * q[idx]).sum(axis=1)
            V = []
            for b in bits:
                free = (Sx >> b & 1) == 0
                V.append(np.where(free, (p[idx] * q[idx | (1 << b)]).sum(axis=1), 0.0))
            base = np.array([th[i, i] * (th[i, n] if c == M else 1.0) if i < n else 0.0 for i in range(N)])
            tot, mar = grad_accum(th, base, bits, [d.ev[b] for b in bits], N, W, V)
            for i in range(n):
                G[i, i] += tot[i]
                if c == M:
                    G[i, n] += tot[i]
                for l, b in enumerate(bits):
                    if d.ev[b] != i:
                        G[i, d.ev[b]] += mar[i, l]
    # ---- seed = 0 region, eq states: synchronised events + seeding (kernel k_eq_flows)
    pairs = [b for b in range(d.k) if d.pairP >> b & 1]
    ke = len(pairs)
    e = np.arange(2 ** ke)
    x0 = np.zeros_like(e)
    for l, b in enumerate(pairs):
        x0 |= (e >> l & 1) * (3 << b)
    A_diag = -(p[x0] * q[x0])
    A_bit = [np.where((e >> l & 1) == 0, p[x0] * q[(x0 | (3 << b)) % len(q)], 0.0) for l, b in enumerate(pairs)]
    base = np.array([th[i, i] for i in range(N)])
    # seeding row behaves like an event whose "bit" is the seeding bit: handle separately
    tot, mar = grad_accum(th, np.append(base[:n], 0.0), pairs, [d.ev[b] for b in pairs], N, A_diag, A_bit)
    for i in range(n):
        G[i, i] += tot[i]
        for l, b in enumerate(pairs):
            if d.ev[b] != i:
                G[i, d.ev[b]] += mar[i, l]
    rate = th[n, n] * np.ones(2 ** ke)
    for l, b in enumerate(pairs):
        rate = np.where(e >> l & 1, rate * th[n, d.ev[b]], rate)
    if d.seedbit >= 0:
        f = rate * (p[x0] * q[x0 | (1 << d.seedbit)] + A_diag)
    else:
        f = rate * A_diag
    G[n, n] += f.sum()
    for l, b in enumerate(pairs):
        G[n, d.ev[b]] += f[(e >> l & 1) == 1].sum()
    return G


def joint_xDy(d, dp, dm, q, p):
    """(d_dp, d_dm) = q^T (dD/dlog d) p : weighted bit marginals (likelihood.py:204-228)."""
    Dp, Dm = obs_diag(d, dp, dm)
    x = np.arange(2 ** d.k)
    wp, wm = q * p * Dp, q * p * Dm
    d_dp, d_dm = np.zeros(d.N), np.zeros(d.N)
    for b in range(d.k):
        has = (x >> b & 1) == 1
        if d.cls[b] == P:
            d_dp[d.ev[b]] = wp[has].sum()
        elif d.cls[b] == M:
            d_dm[d.ev[b]] = wm[has].sum()
        else:
            d_dp[-1] = wp[has].sum()
            d_dm[-1] = wm[has].sum()
    return d_dp, d_dm


def single_xQy(th, d, q, p):
    """vanilla.py:328-393 in flow form: (val[i, j], d_diag[j])."""
    N = d.N
    bits = list(range(d.k))
    A_diag = -(p * q)
    x = np.arange(2 ** d.k)
    A_bit = [np.where((x >> b & 1) == 0, p * q[x | (1 << b)], 0.0) for b in bits]
    base = np.array([th[i, i] for i in range(N)])
    tot, mar = grad_accum(th, base, bits, list(d.ev), N, A_diag, A_bit)
    val = np.zeros((N, N))
    for i in range(N):
        val[i, i] = tot[i]
        for b in bits:
            if d.ev[b] != i:
                val[i, d.ev[b]] = mar[i, b]
    return val, -val.sum(axis=0) + np.diagonal(val)


# --------------------------------------------------------------------------
# per-patient pipelines (what Engine::score_and_grad runs per bucket)
# --------------------------------------------------------------------------


def _theta_sets(log_theta, log_d_p, log_d_m):
    th = np.exp(log_theta)
    dp, dm = np.exp(log_d_p), np.exp(log_d_m)
    n = th.shape[0] - 1
    thM = th / dm[None, :]
    np.fill_diagonal(thM, np.diagonal(th))
    thP = th.copy()
    thP[:n, n] = 1.0
    thP = thP / dp[None, :]
    np.fill_diagonal(thP, np.diagonal(th))
    return th, thM, thP, dp, dm


def _single_grad(thx, d, rhs, seed_scale=None):
    """fwd solve, adjoint solve and flow gradient of a single-tumour space with D = I."""
    ones = np.ones(2 ** d.k)
    pth = solve(thx, d, ones, rhs)
    return pth


def patient_grad(log_theta, log_d_p, log_d_m, row):
    """(lp, d_th, d_dp, d_dm) of one `dat` row in the closed-form formulation."""
    row = np.asarray(row).astype(int)
    n = (row.shape[0] - 3) // 2
    N = n + 1
    th, thM, thP, dp, dm = _theta_sets(np.asarray(log_theta, float), np.asarray(log_d_p, float),
                                       np.asarray(log_d_m, float))
    typ, order = row[-1], row[-2]
    zero = np.zeros(N)
    if typ in (0, 1):
        st = row[:-2:2]
        if typ == 0 and st.sum() == 0:
            br = np.diagonal(th)
            return -np.log1p(br.sum()), np.diag(-br / (1.0 + br.sum())), zero, zero
        d = Desc(st, "single")
        ones = np.ones(2 ** d.k)
        e0 = np.zeros(2 ** d.k); e0[0] = 1.0
        pth = solve(thP, d, ones, e0)
        el = np.zeros(2 ** d.k); el[-1] = 1.0 / pth[-1]
        q = solve(thP, d, ones, el, transpose=True)
        g, dd = single_xQy(thP, d, q, pth)
        g[:n, n] = 0.0
        return np.log(pth[-1]), g, dd, zero
    if typ == 2:
        st = np.append(row[1:-2:2], 1)
        d = Desc(st, "single")
        x = np.arange(2 ** d.k)
        sb = d.k - 1                                        # seeding is the MSB
        a, b = np.ones(2 ** d.k), np.ones(2 ** d.k)
        for bb in range(d.k - 1):
            a = np.where(x >> bb & 1, a * dp[d.ev[bb]], a)
            b = np.where(x >> bb & 1, b * dm[d.ev[bb]], b)
        seeded = (x >> sb & 1) == 1
        drp, drm = np.where(seeded, 0.0, a), np.where(seeded, b * dm[-1], 0.0)
        dr = drp + drm
        e0 = np.zeros(2 ** d.k); e0[0] = 1.0
        pth = solve(th, d, dr, e0)
        el = np.zeros(2 ** d.k); el[-1] = 1.0 / pth[-1]
        q = solve(th, d, dr, el, transpose=True)
        g, _ = single_xQy(th, d, q, pth)
        d_dp, d_dm2, d_dm1 = np.zeros(N), np.zeros(N), np.zeros(N)
        for bb in range(d.k):
            has = (x >> bb & 1) == 1
            d_dp[d.ev[bb]] = (q * pth * drp)[has].sum() if bb != sb else 0.0
            d_dm2[d.ev[bb]] = (q * pth * drm)[has].sum()
            d_dm1[d.ev[bb]] = 1.0
        return np.log(pth[-1] * dr[-1]), g, -d_dp, d_dm1 - d_dm2
    # ---- paired
    st = row[:2 * n + 1]
    d = Desc(st, "joint")
    Dp, Dm = obs_diag(d, dp, dm)
    e0 = np.zeros(2 ** d.k); e0[0] = 1.0
    pi = solve(th, d, Dp + Dm, e0)
    met = np.append(st[1::2], 1)
    prim = st[0::2]
    parts = []
    if order in (0, 1):
        parts.append((True, Desc(met, "single"), thM, Dp))
    if order != 1:
        parts.append((False, Desc(prim, "single"), thP, Dm))
    fwd = []
    for pt_first, ds, thx, Dobs in parts:
        idx = compat_indices(d, pt_first)
        v = np.zeros(2 ** ds.k)
        v[2 ** (ds.k - 1):] = (Dobs * pi)[idx]
        pth2 = solve(thx, ds, np.ones(2 ** ds.k), v)
        fwd.append((idx, v, pth2))
    full = sum(f[2][-1] for f in fwd)
    G, d_dp, d_dm = np.zeros((N, N)), np.zeros(N), np.zeros(N)
    rhsJ = np.zeros(2 ** d.k)
    for (pt_first, ds, thx, Dobs), (idx, v, pth2) in zip(parts, fwd):
        el = np.zeros(2 ** ds.k); el[-1] = 1.0 / full      # adjoint seeded with 1/full: linear mixing
        qm = solve(thx, ds, np.ones(2 ** ds.k), el, transpose=True)
        g1, dd = single_xQy(thx, ds, qm, pth2)
        half = qm[2 ** (ds.k - 1):]
        dot = float(half @ v[2 ** (ds.k - 1):])
        act = np.zeros(N)
        act[-1] = 1.0
        if pt_first:
            G += g1
            d_dm += dd
            act[d.ev[d.cls == P]] = 1.0
            d_dp += dot * act
        else:
            g1[:n, n] = 0.0
            G += g1
            d_dp += dd
            act[d.ev[d.cls == M]] = 1.0
            d_dm += dot * act
        rhsJ[idx] += Dobs[idx] * half
    qJ = solve(th, d, Dp + Dm, rhsJ, transpose=True)
    G += joint_xQy(th, d, qJ, pi)
    a, b = joint_xDy(d, dp, dm, qJ, pi)
    return np.log(full), G, d_dp - a, d_dm - b
