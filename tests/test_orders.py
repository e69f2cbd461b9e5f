"""Order likelihoods / likeliest orders (metmhn_amd.model.MetMHN).

Checked against tests/golden/orders.npz (the reference's metmhn/model.py run by
tests/tools/make_golden_orders.py), against the closed-form expansions the reference's own
tests/test_orders.py spells out, and against brute-force enumeration of every admissible order.

CPU tests replace the ONE device call of the class (the restricted joint diagonal, mmhn_kron_diag)
with the oracle's restatement of jx/kronvec.py kron_diag; the -m gpu tests run the class as shipped.
"""
import itertools
import warnings

import numpy as np
import pytest

import metmhn_amd.model as model_mod
from metmhn_amd.model import MetMHN
from metmhn_amd.state import MetState, State
from oracle import metmhn_oracle as orc

KINDS = ["isMetastasis", "PT", "Met", "unknown", "sync"]
REL = 1e-10      # fp64 products of <= 2k factors; the golden values agree to ~1e-15


class _OracleDiag:
    @staticmethod
    def kron_diag(log_theta, state, n_state):
        return orc.kron_diag(np.asarray(log_theta), np.asarray(state), n_state)


@pytest.fixture
def host_only(monkeypatch):
    monkeypatch.setattr(model_mod, "_kronvec", _OracleDiag)
    warnings.simplefilter("ignore", DeprecationWarning)


def _call(kind):
    return ("isMetastasis", None) if kind == 0 else ("isPaired", KINDS[kind])


def _check_golden(golden):
    d = golden("orders")
    for m in (0, 1):
        pre = f"m{m}_"
        mod = MetMHN(d[pre + "theta"], d[pre + "obs1"], d[pre + "obs2"])
        n = mod.n
        off = d[pre + "du_off"]
        for c in range(len(off) - 1):
            seed = bool(d[pre + "du_seed"][c])
            st = State.from_seq(d[pre + "du_state"][c][:n + 1 if seed else n])
            np.testing.assert_allclose(mod._get_diag_unpaired(st, seeding=seed),
                                       d[pre + "du_val"][off[c]:off[c + 1]], rtol=1e-12)
        for kind, order, p in zip(d[pre + "lk_kind"], d[pre + "lk_order"], d[pre + "lk_p"]):
            status, first = _call(kind)
            got = mod.likelihood(tuple(int(e) for e in order if e >= 0), status, first)
            assert abs(got - p) <= REL * p, (m, kind, order)
        for kind, st, order, p in zip(d[pre + "lo_kind"], d[pre + "lo_state"], d[pre + "lo_order"], d[pre + "lo_p"]):
            status, first = _call(kind)
            got_o, got_p = mod.likeliest_order(MetState.from_seq(st), status, first)
            assert abs(got_p - p) <= REL * p, (m, kind, st)
            assert tuple(int(e) for e in got_o) == tuple(int(e) for e in order if e >= 0)   # bit-exact indices


def test_golden_orders_host(golden, host_only):
    _check_golden(golden)


@pytest.mark.gpu
def test_golden_orders_device_diag(golden):
    """As shipped: the joint diagonal comes from the HIP library through the C ABI."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        _check_golden(golden)


@pytest.mark.gpu
def test_paired_diag_device_matches_oracle():
    rng = np.random.default_rng(5)
    n = 7
    mod = MetMHN(rng.normal(0, 0.5, (n + 1, n + 1)), rng.normal(0, 0.3, n + 1), rng.normal(0, 0.3, n + 1))
    st = np.zeros(2 * n + 1, dtype=bool)
    st[[0, 1, 3, 4, 6, 9, 10, 11, 13, 14]] = True
    got = mod._get_diag_paired(MetState.from_seq(st))
    ref = orc.kron_diag(mod.log_theta, st.astype(np.int32), int(st.sum()))
    np.testing.assert_allclose(got, ref, rtol=1e-12)


def _model(n=5, seed=0):
    rng = np.random.default_rng(seed)
    th = rng.normal(0.0, 0.5, (n + 1, n + 1))
    th[np.diag_indices(n + 1)] = rng.normal(-1.0, 0.5, n + 1)
    return MetMHN(th, 2 * rng.random(n + 1) + 1, 2 * rng.random(n + 1) + 1)


def test_unpaired_closed_forms(host_only):
    """The three expansions of the reference's tests/test_orders.py:66-128, n = 5: a primary tumour
    without / with a later metastasis (the observation-MHN the reference takes from PyPI `mhn`),
    and a lone metastasis."""
    mod = _model()
    n, th, o1, o2, e = mod.n, mod.log_theta, mod.obs1, mod.obs2, np.exp
    seeding = 2 * n
    every = State.from_seq([1] * (n + 1))
    ds = mod._get_diag_unpaired(every)                     # the seeding can still come
    dn = mod._get_diag_unpaired(every, seeding=False)      # the seeding is in and not felt
    absent = (e(th[0, 0]) / (1 - ds[0])
              * e(th[3, [0, 3]].sum()) / (e(o1[0]) - ds[1])
              * e(th[2, [0, 3, 2]].sum()) / (e(o1[[0, 3]].sum()) - ds[1 + 8])
              * e(o1[[0, 2, 3]].sum()) / (e(o1[[0, 2, 3]].sum()) - ds[1 + 4 + 8]))
    present = (e(th[0, 0]) / (1 - ds[0])
               * e(th[3, [0, 3]].sum()) / (e(o1[0]) - ds[1])
               * e(th[n, [n, 0, 3]].sum()) / (e(o1[[0, 3]].sum()) - ds[1 + 8])
               * e(th[2, [0, 3, 2]].sum()) / (e(o1[[0, 3, n]].sum()) - dn[1 + 8])
               * e(o1[[0, 2, 3, n]].sum()) / (e(o1[[0, 2, 3, n]].sum()) - dn[1 + 4 + 8]))
    lone = (e(th[0, 0]) / (1 - ds[0])
            * e(th[3, [0, 3]].sum()) / (e(o1[0]) - ds[1])
            * e(th[n, [n, 0, 3]].sum()) / (e(o1[[0, 3]].sum()) - ds[1 + 8])
            * e(th[2, [0, 3, 2, n]].sum()) / (e(o2[[0, 3, n]].sum()) - ds[1 + 8 + 2 ** n])
            * e(o2[[0, 2, 3, n]].sum()) / (e(o2[[0, 2, 3, n]].sum()) - ds[1 + 4 + 8 + 2 ** n]))
    assert mod.likelihood((0, 6, 4), "absent") == pytest.approx(absent, rel=1e-12)
    assert mod.likelihood((0, 6, seeding, 4), "present") == pytest.approx(present, rel=1e-12)
    assert mod.likelihood((1, 7, seeding, 5), "isMetastasis") == pytest.approx(lone, rel=1e-12)


def test_paired_timed_closed_form(host_only):
    """In the manner of the reference's tests/test_orders.py:130-221: the order (0,1,S,4,3,5) with
    the primary tumour seen first, every factor written out.  The first observation can fall after
    the PT event 4, after the MT event 3 or after the MT event 5 (model.py:139-144)."""
    mod = _model(seed=3)
    n, th, o1, o2, e = mod.n, mod.log_theta, mod.obs1, mod.obs2, np.exp
    S = 2 * n
    st = MetState([0, 1, 3, 4, 5, S], size=2 * n + 1)
    dj = orc.kron_diag(th, st.to_seq().astype(np.int32), len(st))      # slots 0,1,3,4,5,S -> bits 0..5
    du = mod._get_diag_unpaired(State([0, 1, 2, n], size=n + 1))         # MT events 0,1,2 + seeding -> bits 0..3
    head = (e(th[0, 0]) / (1 - dj[0])
            * e(th[n, [0, n]].sum()) / (e(o1[0]) - dj[1 + 2])
            * e(th[2, [0, 2]].sum()) / (e(o1[[0, n]].sum()) + e(o2[[0, n]].sum()) - dj[1 + 2 + 32])
            / (e(o1[[0, 2, n]].sum()) + e(o2[[0, n]].sum()) - dj[1 + 2 + 8 + 32]))
    # first observation right after the PT event 4, then MT events 1 (code 3) and 2 (code 5) alone
    split_early = (head * e(o1[[0, 2, n]].sum()) / (e(o2[[0, n]].sum()) - du[1 + 8])
                   * e(th[1, [0, 1, n]].sum()) / (e(o2[[0, 1, n]].sum()) - du[1 + 2 + 8])
                   * e(th[2, [0, 1, 2, n]].sum()) / (e(o2[[0, 1, 2, n]].sum()) - du[15])
                   * e(o2[[0, 1, 2, n]].sum()))
    # ... after MT event 1 as well (it happened with both tumours still unobserved)
    j1 = e(th[1, [0, 1, n]].sum()) / (e(o1[[0, 2, n]].sum()) + e(o2[[0, 1, n]].sum()) - dj[1 + 2 + 4 + 8 + 32])
    split_mid = (head * j1 * e(o1[[0, 2, n]].sum()) / (e(o2[[0, 1, n]].sum()) - du[1 + 2 + 8])
                 * e(th[2, [0, 1, 2, n]].sum()) / (e(o2[[0, 1, 2, n]].sum()) - du[15])
                 * e(o2[[0, 1, 2, n]].sum()))
    j2 = e(th[2, [0, 1, 2, n]].sum()) / (e(o1[[0, 2, n]].sum()) + e(o2[[0, 1, 2, n]].sum()) - dj[63])
    split_late = head * j1 * j2 * e(o1[[0, 2, n]].sum()) / (e(o2[[0, 1, 2, n]].sum()) - du[15]) * e(o2[[0, 1, 2, n]].sum())
    got = mod.likelihood((0, 1, S, 4, 3, 5), "isPaired", "PT")
    assert got == pytest.approx(split_early + split_mid + split_late, rel=1e-12)


def _all_orders(state: MetState):
    """Every order the chain can take to a seeded `state`."""
    n = state.n
    both = [i for i in state.PT_events if i in state.MT_events]
    for r in range(len(both) + 1):
        for pre in itertools.permutations(both, r):
            head = [c for i in pre for c in (2 * i, 2 * i + 1)] + [2 * n]
            rest = [2 * i for i in state.PT_events if i not in pre] + [2 * i + 1 for i in state.MT_events if i not in pre]
            for tail in itertools.permutations(rest):
                yield tuple(head) + tail


@pytest.mark.parametrize("first_obs", ["PT", "Met", "unknown", "sync"])
def test_likeliest_is_the_maximum_over_all_orders(host_only, first_obs):
    mod = _model(n=4, seed=11)
    for slots in ([0, 1, 2, 5, 6, 7, 8], [0, 1, 4, 5, 3, 8], [2, 3, 4, 7, 8], [1, 8], [0, 8], [8]):
        st = MetState(slots, size=9)
        order, p = mod.likeliest_order(st, "isPaired", first_obs)
        table = {o: mod.likelihood(o, "isPaired", first_obs) for o in _all_orders(st)}
        best = max(table, key=table.get)
        assert p == pytest.approx(table[best], rel=1e-12)
        assert table[tuple(order)] == pytest.approx(p, rel=1e-12)
        assert set(order) == set(slots)


def test_unpaired_likeliest_is_the_maximum(host_only):
    mod = _model(n=5, seed=2)
    S = 10
    for slots, status in (([1, 5, 9, S], "isMetastasis"), ([2, 4, 8, S], "present"), ([0, 4, 6], "absent")):
        st = MetState(slots, size=11)
        order, p = mod.likeliest_order(st, status)
        assert p == pytest.approx(mod.likelihood(order, status), rel=1e-12)      # test_orders.py:259-277
        assert p >= max(mod.likelihood(o, status) for o in itertools.permutations(slots)) * (1 - 1e-12)
        assert sorted(int(e) for e in order) == slots


def test_unknown_is_the_sum_of_both_first_observations(host_only):
    mod = _model(n=4, seed=5)
    order = (2, 3, 8, 0, 5, 7, 1)
    assert mod.likelihood(order, "isPaired", "unknown") == pytest.approx(
        mod.likelihood(order, "isPaired", "PT") + mod.likelihood(order, "isPaired", "Met"), rel=1e-13)


def _random_order(rng, state: MetState):
    n = state.n
    pre = [i for i in state.PT_events if i in state.MT_events and rng.random() < 0.5]
    rng.shuffle(pre)
    rest = [2 * i for i in state.PT_events if i not in pre] + [2 * i + 1 for i in state.MT_events if i not in pre]
    rng.shuffle(rest)
    return tuple([c for i in pre for c in (2 * i, 2 * i + 1)] + [2 * n] + rest)


def test_beyond_the_reference_size_limit(host_only):
    """k = 14 occupied slots: past the reference's factorial-base int32 order code (k <= 12)."""
    mod = _model(n=7, seed=8)
    st = MetState([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14], size=15)
    rng = np.random.default_rng(0)
    orders = [_random_order(rng, st) for _ in range(300)]
    for first_obs in ("PT", "Met", "unknown"):
        order, p = mod.likeliest_order(st, "isPaired", first_obs)
        assert p == pytest.approx(mod.likelihood(order, "isPaired", first_obs), rel=1e-12)
        assert p >= max(mod.likelihood(o, "isPaired", first_obs) for o in orders)
        # no single swap of neighbours improves on it
        for i in range(len(order) - 1):
            o = list(order)
            o[i], o[i + 1] = o[i + 1], o[i]
            try:
                assert mod.likelihood(o, "isPaired", first_obs) <= p * (1 + 1e-12)
            except ValueError:
                pass                                            # the swap left the chain's paths


def test_unreachable_and_invalid_states(host_only):
    """tests/test_orders.py:18-64 (reference)."""
    mod = _model(n=4)
    for first_obs in ("PT", "Met", "unknown", "sync"):
        with pytest.raises(ValueError):
            mod.likeliest_order(np.array([0, 1, 1, 1, 0, 0, 0, 1, 0]), "isPaired", first_obs)
    bad = [("isMetastasis", [0, 1, 1, 1, 0, 0, 0, 1, 1]), ("isMetastasis", [0, 1, 0, 1, 0, 0, 0, 0, 0]),
           ("present", [1, 1, 0, 1, 1, 0, 0, 0, 1]), ("present", [1, 0, 0, 0, 1, 0, 0, 0, 0]),
           ("absent", [1, 1, 0, 1, 1, 0, 0, 0, 1]), ("absent", [1, 0, 0, 0, 1, 0, 0, 0, 1])]
    for status, st in bad:
        with pytest.raises(ValueError):
            mod.likeliest_order(np.array(st), status)
    with pytest.raises(ValueError):
        mod.likeliest_order(np.array([1, 1, 0, 0, 0, 0, 0, 0, 1]), "isPaired", "first")
    with pytest.raises(ValueError):
        mod.likeliest_order(np.array([1, 1, 0, 0, 0, 0, 0, 0, 1]), "paired")
    with pytest.raises(ValueError):
        mod.likelihood((0, 8, 1), "isPaired", "PT")            # event 0 alone before the seeding
    with pytest.raises(ValueError):
        mod.likelihood((0, 2, 8), "isMetastasis")
    with pytest.raises(ValueError):
        mod.likelihood((0, 2), "present")
    with pytest.raises(ValueError):
        mod.likelihood((0, 8), "absent")
    with pytest.warns(DeprecationWarning):
        warnings.simplefilter("default", DeprecationWarning)
        mod.likelihood((0, 1, 8), "isPaired", "sync")


def test_state_mirror():
    """metmhn/state.py:201-300."""
    st = MetState.from_seq(np.array([1, 1, 0, 1, 1, 0, 0, 0, 1]))
    assert (st.n, st.size, len(st), st.data) == (4, 9, 5, 0b100011011)
    assert st.events == (0, 1, 3, 4, 8)
    assert (st.PT_events, st.MT_events, st.Seeding) == ((0, 2), (0, 1), (4,))
    assert list(st.PT) == [0, 2] and st.PT.size == 4
    assert list(st.PT_S) == [0, 2, 4] and list(st.MT) == [0, 1, 4]
    assert st.reachable and 3 in st and 2 not in st
    np.testing.assert_array_equal(st.to_seq(), [1, 1, 0, 1, 1, 0, 0, 0, 1])
    un = MetState([0, 3], size=9)
    assert not un.reachable and list(un.MT) == []            # no seeding: the MT view is empty
    assert MetState([0, 1], size=9).reachable
    assert MetState(0b11, size=9) == MetState([0, 1], size=9)
    un.add(8), un.discard(3)
    assert un.events == (0, 8)
    with pytest.raises(ValueError):
        State(-1, size=3)
    with pytest.raises(TypeError):
        State(1.5, size=3)
