"""Test infrastructure: a scalar model of the schedule of `k_wsolve` (metmhn_amd/csrc/wsolve.h).

NOT part of the product path.  It executes, thread by thread and step by step, exactly the data movement the HIP
kernel performs for one joint problem in the class-sorted "window" layout, on small configurable sizes, and checks
every value a thread picks up against the identity (external index, window block) it expects there.  What is
modelled (reference semantics: the triangular solve of R_i_inv_vec, metmhn/jx/likelihood.py:231-262, on the seeded
half, where D - Q = A_R (+) A_C is a Kronecker sum over the two tumour classes):

  state (S, T): S = row-class subset, T = column-class subset
      forward     y[S,T] = (rhs + sum_{i in S} rR[i][S^i] y[S^i,T] + sum_{b in T} rC[b][T^b] y[S,T^b]) / (dR[S] + dC[T])
      transposed  y[S,T] = (rhs + sum_{i !in S} rR[i][S] y[S|i,T] + sum_{b !in T} rC[b][T] y[S,T|b]) / (dR[S] + dC[T])

  thread = row (w, l): LB lane bits l (exchanged by lane permutes straight out of the neighbour's registers) and WVB
  wave bits w (exchanged through a two-slot ring in LDS); a thread keeps a WINDOW of 2^h blocks of NC = 2^RB columns in
  registers; everything above (external bits: remaining column bits, then remaining row bits) is the thread's own
  earlier output in global memory.

  skew: a wave at wave-level lam (popcount of w, transposed: of ~w) runs lam BLOCKS behind, a lane at lane-level m runs
  m whole WINDOWS behind.  Hence (i) every wave works on one static window slot per step, (ii) a lane's lower
  neighbours hold the block it needs in THEIR window slot of the same number (written one window pass earlier, not
  yet overwritten), (iii) a wave's lower neighbour waves published that block in the ring one step earlier.
"""
import numpy as np


def popc(v):
    return bin(v).count("1")


def direct(kR, kC, rR, rC, dR, dC, rhs, tr):
    NR, NCc = 1 << kR, 1 << kC
    Y = np.zeros((NR, NCc))
    rows = range(NR) if not tr else range(NR - 1, -1, -1)
    for S in rows:
        cols = range(NCc) if not tr else range(NCc - 1, -1, -1)
        for T in cols:
            z = rhs[S, T]
            for i in range(kR):
                if not tr and (S >> i) & 1:
                    z += rR[i][S ^ (1 << i)] * Y[S ^ (1 << i), T]
                if tr and not (S >> i) & 1:
                    z += rR[i][S] * Y[S | (1 << i), T]
            for b in range(kC):
                if not tr and (T >> b) & 1:
                    z += rC[b][T ^ (1 << b)] * Y[S, T ^ (1 << b)]
                if tr and not (T >> b) & 1:
                    z += rC[b][T] * Y[S, T | (1 << b)]
            Y[S, T] = z / (dR[S] + dC[T])
    return Y


def emulate(LB, WVB, RB, h, nXc, nXr, tr, seed=0):
    """Run the schedule; returns (max abs error vs the direct solve, number of global steps)."""
    rng = np.random.default_rng(seed)
    TBITS = LB + WVB
    kR, kC = TBITS + nXr, RB + h + nXc
    NC, H, nX = 1 << RB, 1 << h, nXc + nXr
    NXS = 1 << nX
    NR, NCc = 1 << kR, 1 << kC
    rR = rng.uniform(0.5, 1.5, (kR, NR))
    rC = rng.uniform(0.5, 1.5, (kC, NCc))
    dR = rng.uniform(2.0, 3.0, NR)
    dC = rng.uniform(2.0, 3.0, NCc)
    rhs = rng.normal(size=(NR, NCc))
    ref = direct(kR, kC, rR, rC, dR, dC, rhs, tr)

    nthr = 1 << TBITS
    NAN = float("nan")
    W = np.full((nthr, H, NC), 0.0)                 # register windows (start as zeros, as in the kernel)
    Wtag = np.full((nthr, H), -1)                   # external index of the block a slot holds
    ring = np.zeros((2, nthr, NC))
    rtag = np.full((2, nthr, 2), -1)                # (Sigma, beta) of the published block
    G = np.full((NXS, H, nthr, NC), NAN)            # global memory, block-major
    lmask, wmask = (1 << LB) - 1, (1 << WVB) - 1
    NSIG = NXS + LB
    nsteps = NSIG * H + WVB

    def lvl(bits, nb):
        return popc(bits) if not tr else nb - popc(bits)

    for s in range(nsteps):
        newW = {}
        for thr in range(nthr):
            l, w = thr & lmask, thr >> LB
            lam, m = lvl(w, WVB), lvl(l, LB)
            t = s - lam
            if t < 0 or t >= NSIG * H:
                continue                                                  # (the wave only meets the barrier)
            sig, bi = divmod(t, H)
            beta = bi if not tr else H - 1 - bi
            Sg = sig - m
            if Sg < 0 or Sg >= NXS:
                continue                                                  # inactive lane: results discarded
            Sigma = Sg if not tr else NXS - 1 - Sg
            Tx, Sx = Sigma & ((1 << nXc) - 1), Sigma >> nXc
            S = l | (w << LB) | (Sx << TBITS)
            acc = np.zeros(NC)
            # lane bits: the neighbour lane's register window, slot beta
            for i in range(LB):
                has = (l >> i) & 1
                if (has and not tr) or (not has and tr):
                    nb = thr ^ (1 << i)
                    assert Wtag[nb, beta] == Sigma, ("lane nbr holds", Wtag[nb, beta], "want", Sigma, s, thr)
                    coef = rR[i][S ^ (1 << i)] if not tr else rR[i][S]
                    acc += coef * W[nb, beta]
            # wave bits: the ring slot written one global step ago
            for j in range(WVB):
                i = LB + j
                has = (w >> j) & 1
                if (has and not tr) or (not has and tr):
                    nb = thr ^ (1 << i)
                    slot = (s - 1) & 1
                    assert tuple(rtag[slot, nb]) == (Sigma, beta), (tuple(rtag[slot, nb]), Sigma, beta, s, thr)
                    coef = rR[i][S ^ (1 << i)] if not tr else rR[i][S]
                    acc += coef * ring[slot, nb]
            # external bits: own earlier blocks from global memory
            for j in range(nX):
                has = (Sigma >> j) & 1
                if (has and not tr) or (not has and tr):
                    Sn = Sigma ^ (1 << j)
                    v = G[Sn, beta, thr]
                    assert not np.isnan(v).any(), "external block not written yet"
                    for c in range(NC):
                        T = c | (beta << RB) | (Tx << (RB + h))
                        if j < nXc:
                            b = RB + h + j
                            coef = rC[b][T ^ (1 << b)] if not tr else rC[b][T]
                        else:
                            i = TBITS + (j - nXc)
                            coef = rR[i][S ^ (1 << i)] if not tr else rR[i][S]
                        acc[c] += coef * v[c]
            # window bits: own register slots of this window pass
            for j in range(h):
                has = (beta >> j) & 1
                if (has and not tr) or (not has and tr):
                    sb = beta ^ (1 << j)
                    src = newW.get((thr, sb))
                    if src is None:
                        assert Wtag[thr, sb] == Sigma, ("window slot", Wtag[thr, sb], Sigma)
                        src = W[thr, sb]
                    b = RB + j
                    for c in range(NC):
                        T = c | (beta << RB) | (Tx << (RB + h))
                        coef = rC[b][T ^ (1 << b)] if not tr else rC[b][T]
                        acc[c] += coef * src[c]
            # the block itself
            y = np.zeros(NC)
            order = range(NC) if not tr else range(NC - 1, -1, -1)
            for c in order:
                T = c | (beta << RB) | (Tx << (RB + h))
                z = acc[c] + rhs[S, T]
                for r in range(RB):
                    has = (c >> r) & 1
                    if (has and not tr) or (not has and tr):
                        z += (rC[r][T ^ (1 << r)] if not tr else rC[r][T]) * y[c ^ (1 << r)]
                y[c] = z / (dR[S] + dC[T])
            newW[(thr, beta)] = y
            G[Sigma, beta, thr] = y
        # all lanes of a wave execute a step together: window / ring writes land after every read of the step
        for (thr, beta), y in newW.items():
            l, w = thr & lmask, thr >> LB
            lam, m = lvl(w, WVB), lvl(l, LB)
            sig = (s - lam) // H
            Sg = sig - m
            Sigma = Sg if not tr else NXS - 1 - Sg
            W[thr, beta] = y
            Wtag[thr, beta] = Sigma
            ring[s & 1, thr] = y
            rtag[s & 1, thr] = (Sigma, beta)
    out = np.zeros((NR, NCc))
    for Sigma in range(NXS):
        Tx, Sx = Sigma & ((1 << nXc) - 1), Sigma >> nXc
        for beta in range(H):
            for thr in range(nthr):
                S = thr | (Sx << TBITS)
                for c in range(NC):
                    out[S, c | (beta << RB) | (Tx << (RB + h))] = G[Sigma, beta, thr, c]
    return float(np.max(np.abs(out - ref))), nsteps


if __name__ == "__main__":
    for cfg in [(2, 1, 1, 1, 1, 0), (2, 2, 1, 2, 1, 1), (3, 2, 2, 1, 2, 0), (3, 1, 1, 2, 0, 2), (2, 2, 2, 2, 2, 1)]:
        for tr in (False, True):
            err, ns = emulate(*cfg, tr)
            print(cfg, "tr" if tr else "fwd", "steps", ns, "max err", err)
