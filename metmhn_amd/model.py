"""Order likelihoods and likeliest orders: the host-side mirror of metmhn/model.py (class MetMHN).

SURVEY.md §8 row f-4.  Same constructor, `likelihood(order, met_status, first_obs)` and
`likeliest_order(state, met_status, first_obs)` as the reference (model.py:184-376), same event
codes (2i = event i in the primary tumour, 2i+1 = event i in the metastasis, 2n = seeding; before
the seeding an event is written as the pair 2i, 2i+1), same error behaviour.

What runs where.  The diagonal of the restricted joint rate matrix, `_get_diag_paired`
(model.py:434-446 -> jx/kronvec.py kron_diag), is the device part and goes through the C ABI
(`mmhn_kron_diag`); without the HIP library it raises, there is no host substitute.  Everything
else in the reference's model.py is host code (NumPy/BLAS loops, dicts of candidate orders) and is
host code here.

How it differs from the reference inside.  A path's probability is a product of
"rate of this event / (observation rate + exit rate of the state it leads to)" factors.  With two
observations the first one may fall anywhere in the tail of the order, so a prefix carries a vector
    a   = probability of the prefix with no observation made yet,
    b_P = sum over admissible earlier "primary tumour observed here" points of the probability
          with the metastasis continuing on its own (model.py:1428-1538, 1540-1553),
    b_M = the same with the metastasis observed first (model.py:1555-1680),
and every later factor multiplies these by non-negative numbers.  `likelihood` walks one order with
that recurrence; `likeliest_order` runs it over the lattice of sub-states in index order and keeps,
per sub-state, only the candidates no other candidate dominates component-wise.  The reference keeps
EVERY order for seeded sub-states that do not yet hold all first-observed events
(model.py:574-619) and encodes orders as factorial-base int32 (k <= 12,
int_order_conversion.pyx:9); here the candidate lists stay short and orders are linked tuples, so
k is bounded by the 2^k lattice only.  Same maximum, same arg max when it is unique.
"""
from __future__ import annotations

import warnings
from types import SimpleNamespace

import numpy as np

from .jx import kronvec as _kronvec
from .state import MetState, State

_FIRST_OBS = ("PT", "Met", "unknown", "sync")
_STATUS_ERR = "met_status must be one of 'isMetastasis', 'absent', 'present', 'isPaired'"
_FIRST_ERR = "first_obs must be one of 'PT', 'Met', 'unknown', 'sync'"


def _subset_sums(weights) -> np.ndarray:
    """out[x] = sum of weights[b] over the bits b of x."""
    out = np.zeros(1)
    for w in weights:
        out = np.concatenate((out, out + w))
    return out


def _subset_index(flags) -> np.ndarray:
    """out[x] = the bits of x whose flag is set, packed together (a software pext)."""
    out, nxt = np.zeros(1, dtype=np.int64), 0
    for f in flags:
        out = np.concatenate((out, out + (1 << nxt))) if f else np.concatenate((out, out))
        nxt += bool(f)
    return out


def _single_diag(theta: np.ndarray, n_events: int, events: list) -> np.ndarray:
    """Diagonal of a one-tumour rate matrix over `n_events` events, restricted to `events`:
    minus the summed rates of every event not yet present (model.py:378-432)."""
    diag = np.zeros(1 << len(events))
    for i in range(n_events):
        rate = np.exp(theta[i, i] + _subset_sums(theta[i, events]))
        if i in events:                       # already present in the upper half of its own bit
            rate[(np.arange(rate.size) >> events.index(i) & 1) == 1] = 0.0
        diag -= rate
    return diag


def _pareto(vecs: list) -> list:
    """Indices of the vectors no other vector dominates (>= everywhere, > somewhere); one of equals."""
    keep = []
    for i, v in enumerate(vecs):
        for j, w in enumerate(vecs):
            if j != i and all(wc >= vc for wc, vc in zip(w, v)) and (any(wc > vc for wc, vc in zip(w, v)) or j < i):
                break
        else:
            keep.append(i)
    return keep


class MetMHN:
    """The metastasis MHN with its two observation-rate vectors (model.py:175-211)."""

    def __init__(self, log_theta, obs1, obs2, events: list = None, meta: dict = None):
        self.log_theta = np.array(log_theta, dtype=np.float64)
        self.obs1 = np.array(obs1, dtype=np.float64)
        self.obs2 = np.array(obs2, dtype=np.float64)
        self.events = events
        self.meta = meta
        self.n = self.log_theta.shape[1] - 1
        # the primary tumour does not feel the seeding (model.py:207-208)
        self._pt_log_theta = self.log_theta.copy()
        self._pt_log_theta[:-1, -1] = 0.0

    # ------------------------------------------------------------------ diagonals
    def _get_diag_unpaired(self, state: State, seeding: bool = True) -> np.ndarray:
        """model.py:378-432."""
        nn = self.n + 1 if seeding else self.n
        return _single_diag(self.log_theta, nn, [j for j in range(nn) if j in state])

    def _get_diag_paired(self, state: MetState) -> np.ndarray:
        """model.py:434-446: on the device, through mmhn_kron_diag."""
        return np.asarray(_kronvec.kron_diag(self.log_theta, state.to_seq().astype(np.int32), len(state)),
                          dtype=np.float64)

    # ------------------------------------------------------------------ public entry points
    def likeliest_order(self, state, met_status: str, first_obs: str = None):
        """model.py:213-293: (order, probability)."""
        if isinstance(state, np.ndarray):
            state = MetState.from_seq(state)
        if met_status == "isMetastasis":
            if len(state.PT) > 0:
                raise ValueError("PT part of the state was not empty, but met_status is 'isMetastasis'.")
            if not state.Seeding:
                raise ValueError("Seeding was not observed, but met_status is 'isMetastasis'.")
            return self._likeliest_order_unpaired_mt(state.MT)
        if met_status == "absent":
            if len(state.MT) > 0 or state.MT_events:
                raise ValueError("Met part of the state was not empty, but met_status is 'absent'.")
            if state.Seeding:
                raise ValueError("Seeding was observed, but met_status is 'absent'.")
            return self._likeliest_order_unpaired_pt(state.PT_S)
        if met_status == "present":
            if tuple(state.MT) != (self.n,):
                raise ValueError("Met part of the state was not empty, but met_status is 'present', not 'isPaired'.")
            return self._likeliest_order_unpaired_pt(state.PT_S)
        if met_status == "isPaired":
            if first_obs not in _FIRST_OBS:
                raise ValueError(_FIRST_ERR)
            if first_obs == "sync":
                warnings.warn("Synchronous development is deprecated.", DeprecationWarning)
            return self._likeliest_order_paired(state, first_obs)
        raise ValueError(_STATUS_ERR)

    def likelihood(self, order, met_status: str, first_obs: str = None) -> float:
        """model.py:295-376: probability of exactly this order of events being what is observed."""
        order = tuple(int(e) for e in order)
        seeding = 2 * self.n
        if met_status == "isMetastasis":
            if any(e % 2 == 0 and e != seeding for e in order):
                raise ValueError("PT event in order, but met_status is 'isMetastasis'.")
            if seeding not in order:
                raise ValueError("Seeding event not in order, but met_status is 'isMetastasis'.")
            return self._likelihood_unpaired_mt(order)
        if met_status in ("absent", "present"):
            if any(e % 2 == 1 for e in order):
                raise ValueError(f"Met event in order, but met_status is '{met_status}'.")
            if met_status == "absent" and seeding in order:
                raise ValueError("Seeding event in order, but met_status is 'absent'.")
            if met_status == "present" and seeding not in order:
                raise ValueError("Seeding event not in order, but met_status is 'present'.")
            return self._likelihood_unpaired_pt(order)
        if met_status == "isPaired":
            if first_obs not in _FIRST_OBS:
                raise ValueError(_FIRST_ERR)
            if first_obs == "sync":
                warnings.warn("Synchronous development is deprecated.", DeprecationWarning)
            return self._likelihood_paired(order, first_obs)
        raise ValueError(_STATUS_ERR)

    # ------------------------------------------------------------------ one tumour
    def _single_tables(self, theta, state: State, obs_after):
        """Lattice tables of a one-tumour chain over the n+1 events, restricted to `state`.
        The observation rate follows obs1 until the seeding is in and `obs_after` from then on."""
        ev = list(state)
        k = len(ev)
        t1, t2 = _subset_sums(self.obs1[ev]), _subset_sums(obs_after[ev])
        x = np.arange(1 << k)
        seeded = (x >> (k - 1) & 1).astype(bool) if self.n in state else np.zeros(1 << k, dtype=bool)
        diag = _single_diag(theta, self.n + 1, ev)
        den = np.where(seeded, np.exp(t2), np.exp(t1)) - diag
        num = [np.exp(_subset_sums(theta[e, ev])) for e in ev]
        final = np.exp(t2[-1] if seeded[-1] else t1[-1])
        return SimpleNamespace(ev=ev, k=k, den=den, num=num, final=final)

    @staticmethod
    def _single_walk(T, events) -> float:
        x, p = 0, 1.0 / T.den[0]
        for e in events:
            b = T.ev.index(e)
            if x >> b & 1:
                raise ValueError("an event occurs twice in the order")
            x |= 1 << b
            p *= T.num[b][x] / T.den[x]
        return float(p * T.final)

    @staticmethod
    def _single_viterbi(T):
        """Best path to every sub-state in index order (every predecessor has a smaller index)."""
        best = np.zeros(1 << T.k)
        last = np.zeros(1 << T.k, dtype=np.int64)
        best[0] = 1.0 / T.den[0]
        for x in range(1, 1 << T.k):
            top, arg = -1.0, -1
            for b in range(T.k):
                if x >> b & 1:
                    cand = best[x ^ (1 << b)] * T.num[b][x]
                    if cand > top:
                        top, arg = cand, b
            best[x], last[x] = top / T.den[x], arg
        x, rev = (1 << T.k) - 1, []
        while x:
            rev.append(T.ev[last[x]])
            x ^= 1 << last[x]
        return np.array(rev[::-1], dtype=np.int64), float(best[-1] * T.final)

    def _likelihood_unpaired_mt(self, order) -> float:
        """model.py:1391-1426: a metastasis seen once (obs2), the chain feeling the seeding."""
        events = [e // 2 for e in order]
        T = self._single_tables(self.log_theta, State(events, size=self.n + 1), self.obs2)
        return self._single_walk(T, events)

    def _likeliest_order_unpaired_mt(self, state: State):
        """model.py:448-501."""
        T = self._single_tables(self.log_theta, state, self.obs2)
        events, p = self._single_viterbi(T)
        codes = 2 * events + 1
        codes[events == self.n] = 2 * self.n
        return codes, p

    def _likelihood_unpaired_pt(self, order) -> float:
        """model.py:343,356: the reference hands these to mhn.oMHN(vstack(theta with the seeding's
        column zeroed, obs1)).order_likelihood -- a one-tumour chain over the n+1 events whose
        observation rate is exp(obs1 . state); expansion in the reference's tests/test_orders.py:76-100."""
        events = [e // 2 for e in order]
        T = self._single_tables(self._pt_log_theta, State(events, size=self.n + 1), self.obs1)
        return self._single_walk(T, events)

    def _likeliest_order_unpaired_pt(self, state: State):
        """model.py:260-274 (mhn.oMHN.likeliest_order on the same chain)."""
        T = self._single_tables(self._pt_log_theta, state, self.obs1)
        events, p = self._single_viterbi(T)
        return 2 * events, p

    # ------------------------------------------------------------------ both tumours
    def _paired_tables(self, state: MetState, first_obs: str):
        """Lattice tables over the 2^k sub-states of `state` (bit b = its b-th occupied slot)."""
        n, th = self.n, self.log_theta
        slots = list(state)
        k = len(slots)
        if not state.Seeding:
            raise ValueError("a paired sample needs the seeding event")
        kind = [2 if s == 2 * n else s & 1 for s in slots]            # 0 PT, 1 MT, 2 seeding
        ev = [n if s == 2 * n else s // 2 for s in slots]
        in_pt = [kd != 1 for kd in kind]                                # slots obs1 / PT rates look at
        in_mt = [kd != 0 for kd in kind]
        s1 = _subset_sums([self.obs1[e] if f else 0.0 for e, f in zip(ev, in_pt)])
        s2 = _subset_sums([self.obs2[e] if f else 0.0 for e, f in zip(ev, in_mt)])
        seeded = (np.arange(1 << k) >> (k - 1) & 1).astype(bool)
        o1, o2 = np.exp(s1), np.exp(s2)
        den = o1 + np.where(seeded, o2, 0.0) - self._get_diag_paired(state)
        # numerator of the event in slot b: its row of theta summed over the PT slots (a PT event;
        # before the seeding both tumours agree) or over the MT slots and the seeding (a metastasis
        # event, the seeding itself)  (model.py:1466-1503)
        def row(b):
            flags = [kd == 0 for kd in kind] if kind[b] == 0 else in_mt
            return np.exp(_subset_sums([th[ev[b], e] if f else 0.0 for e, f in zip(ev, flags)]))
        num = [row(b) for b in range(k)]
        T = SimpleNamespace(k=k, slots=slots, kind=kind, seeded=seeded, den=den, num=num, o1=o1, o2=o2,
                            pt_first=first_obs in ("PT", "unknown"), mt_first=first_obs in ("Met", "unknown"),
                            sync=first_obs == "sync",
                            pt_mask=sum(1 << b for b in range(k) if kind[b] == 0),
                            mt_mask=sum(1 << b for b in range(k) if kind[b] == 1),
                            joint=sum(1 << b for b in range(k - 1)
                                      if kind[b] == 0 and kind[b + 1] == 1 and ev[b] == ev[b + 1]))
        # the tumour that is left after the first observation runs on alone (model.py:1515-1536,
        # 1643-1663): the metastasis with the seeding's effects under obs2, or the primary tumour
        # without them under obs1 (which still counts the seeding)
        if T.pt_first:
            du = self._get_diag_unpaired(state.MT)
            T.den_mt = o2 - du[_subset_index(in_mt)]
        if T.mt_first:
            du = self._get_diag_unpaired(state.PT, seeding=False)
            T.den_pt = o1 - du[_subset_index([kd == 0 for kd in kind])]
        return T

    @staticmethod
    def _settle(T, x: int, v):
        """(a, b_P, b_M) with "the first observation happens now" folded into the b's, which is
        admissible once the seeding and every event of the first-observed tumour are in."""
        a, bp, bm = v
        if T.seeded[x]:
            if T.pt_first and x & T.pt_mask == T.pt_mask:
                bp = bp + a * T.o1[x] / T.den_mt[x]
            if T.mt_first and x & T.mt_mask == T.mt_mask:
                bm = bm + a * T.o2[x] / T.den_pt[x]
        return a, bp, bm

    @staticmethod
    def _advance(T, x: int, v, b: int):
        """Add the event in slot b (a joint event, slots b and b+1, before the seeding)."""
        if not T.seeded[x] and b != T.k - 1:
            y = x | 3 << b
            return y, (v[0] * T.num[b][y] / T.den[y], 0.0, 0.0)
        y = x | 1 << b
        a, bp, bm = MetMHN._settle(T, x, v)
        num = T.num[b][y]
        bp = bp * num / T.den_mt[y] if T.pt_first and T.kind[b] == 1 else 0.0
        bm = bm * num / T.den_pt[y] if T.mt_first and T.kind[b] == 0 else 0.0
        return y, (a * num / T.den[y], bp, bm)

    @staticmethod
    def _total(T, v) -> float:
        full = (1 << T.k) - 1
        a, bp, bm = MetMHN._settle(T, full, v)
        if T.sync:
            return float(a * (T.o1[full] + T.o2[full]))           # model.py:1746-1751
        return float(bp * T.o2[full] + bm * T.o1[full])

    def _likelihood_paired(self, order, first_obs: str) -> float:
        """model.py:1540-1553, 1667-1750: all admissible first-observation points summed."""
        if len(set(order)) != len(order):
            raise ValueError("an event occurs twice in the order")
        if 2 * self.n not in order:
            raise ValueError("Seeding event not in order, but met_status is 'isPaired'.")
        T = self._paired_tables(MetState(order, size=2 * self.n + 1), first_obs)
        x, v, i = 0, (1.0 / T.den[0], 0.0, 0.0), 0
        while i < len(order):
            b = T.slots.index(order[i])
            if not T.seeded[x] and b != T.k - 1:
                if not (T.joint >> b & 1 and i + 1 < len(order) and order[i + 1] == order[i] + 1):
                    raise ValueError("before the seeding an event must occur in both tumours (2i, 2i+1)")
                i += 1
            x, v = self._advance(T, x, v, b)
            i += 1
        return self._total(T, v)

    def _likeliest_order_paired(self, state: MetState, first_obs: str):
        """model.py:503-1389 (_likeliest_order_pt_mt / _mt_pt / _unknown / _sync)."""
        if not state.reachable:
            raise ValueError("This state is not reachable by mhn.")
        T = self._paired_tables(state, first_obs)
        k, top = T.k, 1 << (T.k - 1)
        # front[x]: candidates (vector, predecessor candidate, slot, joint event?) nobody dominates
        front = {0: [((1.0 / T.den[0], 0.0, 0.0), None, -1, False)]}
        for y in range(1, 1 << k):
            cands = []
            if not y & top:
                lo = y & T.joint
                if y != lo | lo << 1:
                    continue                                    # tumours differ before the seeding
                for b in range(k - 1):
                    if lo >> b & 1:
                        x = y ^ 3 << b
                        cands.extend((self._advance(T, x, c[0], b)[1], c, b, True) for c in front[x])
                front[y] = [max(cands, key=lambda c: c[0][0])]
                continue
            for b in range(k):
                x = y ^ 1 << b
                if y >> b & 1 and x in front:
                    cands.extend((self._advance(T, x, c[0], b)[1], c, b, False) for c in front[x])
            keep = _pareto([self._settle(T, y, c[0]) for c in cands])
            front[y] = [cands[i] for i in keep]
        best = max(front[(1 << k) - 1], key=lambda c: self._total(T, c[0]))
        rev, c = [], best
        while c[1] is not None:
            rev.extend([T.slots[c[2] + 1], T.slots[c[2]]] if c[3] else [T.slots[c[2]]])
            c = c[1]
        return tuple(int(s) for s in rev[::-1]), self._total(T, best[0])
