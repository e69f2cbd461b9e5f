"""The property the reference pins in tests/test_likelihood.py:10-132: the analytic probability of a datapoint
equals its frequency among Gillespie-simulated patients (there: 1e5 jax.random samples, 2 significant digits;
here: own NumPy sampler oracle/gillespie.py following simulations.py, 4e5 samples, 4.5 binomial sigmas)."""
import numpy as np
import pytest

N_SIM = 400_000


@pytest.fixture(scope="module")
def sim():
    from oracle import gillespie
    rng = np.random.default_rng(42)
    n_mut = 3
    lt = np.diag(rng.normal(size=n_mut + 1))
    off = rng.random((n_mut + 1, n_mut + 1)) < 0.3
    np.fill_diagonal(off, False)
    lt = lt + off * rng.normal(size=(n_mut + 1, n_mut + 1))
    dp = rng.normal(0, 1, size=n_mut + 1)
    dm = rng.normal(0, 1, size=n_mut + 1)
    dat = gillespie.simulate_dat(lt, dp, dm, N_SIM, seed=7)
    return lt, dp, dm, dat


def _cases(dat):
    geno, order = dat[:, :-1], dat[:, -1]
    ones = np.ones(6, dtype=np.int8)
    out = []

    def exact(g, sel=None):
        m = (geno == np.asarray(g, dtype=np.int8)).all(axis=1)
        return int((m if sel is None else m & sel).sum())
    # (dat row, simulated count) - rows as in test_likelihood.py
    out.append(("prim", [1] * 6 + [0, -99, 0], exact([1] * 6 + [0])))
    out.append(("prim_az", [0] * 6 + [0, -99, 0], exact([0] * 6 + [0])))
    pt_all = (geno[:, 0:6:2] == 1).all(axis=1) & (geno[:, 6] == 1)
    mt_all = (geno[:, 1:6:2] == 1).all(axis=1) & (geno[:, 6] == 1)
    out.append(("prim_met", [1] * 6 + [1, -99, 1], int(pt_all.sum())))
    out.append(("met", [1] * 6 + [1, -99, 2], int(mt_all.sum())))
    out.append(("coupled_0", [1] * 6 + [1, 0, 3], exact([1] * 7)))
    out.append(("coupled_1", [1] * 6 + [1, 1, 3], exact([1] * 7, order == 1)))
    out.append(("coupled_2", [1] * 6 + [1, 2, 3], exact([1] * 7, order == 2)))
    out.append(("empty", [0] * 6 + [1, 0, 3], exact([0] * 6 + [1])))
    out.append(("mixed_1", [1, 0, 1, 1, 0, 1, 1, 1, 3], exact([1, 0, 1, 1, 0, 1, 1], order == 1)))
    return out


def _check(score_fn, sim):
    lt, dp, dm, dat = sim
    for name, row, count in _cases(dat):
        p = float(np.exp(score_fn(lt, dp, dm, np.array([row], dtype=np.int8), 0)))
        sigma = np.sqrt(p * (1 - p) / N_SIM)
        assert count >= 25, f"{name}: too few samples ({count})"
        assert abs(count / N_SIM - p) < 4.5 * sigma, f"{name}: simulated {count / N_SIM:.5f} vs analytic {p:.5f}"


def test_oracle_matches_simulated_frequencies(sim):
    from oracle import metmhn_oracle as O
    _check(O.score, sim)


@pytest.mark.gpu
def test_engine_matches_simulated_frequencies(sim):
    import metmhn_amd.regularized_optimization as ro
    _check(ro.score, sim)


@pytest.fixture(scope="module")
def gpu_sim(sim):
    from metmhn_amd import simulations
    lt, dp, dm, _ = sim
    return lt, dp, dm, simulations.simulate_dat(lt, dp, dm, N_SIM, original_key=123)


@pytest.mark.gpu
def test_gpu_sampler_matches_analytic_probabilities(gpu_sim):
    """Samples drawn by the HIP sampler (csrc/sampler.h) against the engine's analytic probabilities and
    against the oracle's: the reference's test_likelihood.py property with both sides on the device."""
    import metmhn_amd.regularized_optimization as ro
    from oracle import metmhn_oracle as O
    _check(ro.score, gpu_sim)
    _check(O.score, gpu_sim)


@pytest.mark.gpu
def test_gpu_sampler_matches_numpy_sampler(sim, gpu_sim):
    """Two-sample check, every distinct (genotype, order) row: counts from the HIP sampler and from the NumPy
    restatement of simulations.py agree within 5 sigma of the pooled binomial; formats are identical."""
    a, b = sim[3], gpu_sim[3]
    assert a.shape == b.shape and b.dtype == np.int8
    assert set(np.unique(b[:, -1])) <= {0, 1, 2} and set(np.unique(b[:, :-1])) <= {0, 1}
    assert (b[b[:, -2] == 0][:, -1] == 0).all() and (b[b[:, -2] == 1][:, -1] > 0).all()     # order only for paired rows
    assert (b[b[:, -2] == 0][:, 0:-2:2] == b[b[:, -2] == 0][:, 1:-2:2]).all()               # unseeded: PT == MT
    keys = lambda d: (d.astype(np.int64) * (3 ** np.arange(d.shape[1]))).sum(axis=1)
    ka, ca = np.unique(keys(a), return_counts=True)
    kb, cb = np.unique(keys(b), return_counts=True)
    da, db = dict(zip(ka, ca)), dict(zip(kb, cb))
    worst = 0.0
    for k in set(da) | set(db):
        x, y = da.get(k, 0), db.get(k, 0)
        if x + y < 200:
            continue
        p = (x + y) / (2 * N_SIM)
        z = abs(x - y) / np.sqrt(2 * N_SIM * p * (1 - p))
        worst = max(worst, z)
    assert worst < 5.0, worst


@pytest.mark.gpu
def test_gpu_sampler_orders_and_determinism(sim):
    from metmhn_amd import simulations
    lt, dp, dm, _ = sim
    n = lt.shape[0]
    d1 = simulations.simulate_dat(lt, dp, dm, 5000, original_key=5)
    d2 = simulations.simulate_dat(lt, dp, dm, 5000, original_key=5)
    d3 = simulations.simulate_dat(lt, dp, dm, 5000, original_key=6)
    assert (d1 == d2).all() and (d1 != d3).any()
    assert (simulations.simulate_dat(lt, dp, dm, 100, original_key=5) == d1[:100]).all()     # prefix-stable
    od = simulations.simulate_orders(lt, dp, dm, 5000, original_key=5)
    assert od.shape == (5000, 2 * n + 2)
    for row, o in zip(d1[:500], od[:500]):
        ev = o[o != -99]
        assert len(set(ev.tolist())) == len(ev)                          # every event at most once
        seeded = (n - 1) in ev
        pt = np.zeros(n + 1, dtype=np.int8); mt = np.zeros(n + 1, dtype=np.int8)
        s = False
        for e in ev:
            if e <= n:
                pt[e] = 1
                if not s:
                    mt[e] = 1
                if e == n - 1:
                    s = True
            else:
                assert s                                                 # MT events only after seeding
                mt[e - n - 1] = 1
        assert (row[0:2 * (n - 1):2] == pt[:n - 1]).all() and (row[1:2 * (n - 1):2] == mt[:n - 1]).all()
        assert row[2 * (n - 1)] == int(seeded)
        if seeded:
            assert row[-1] == (1 if list(ev).index(n) < list(ev).index(2 * n + 1) else 2)
        else:
            assert row[-1] == 0 and ev[-1] == n
    assert simulations.simulate_dat(lt, dp, dm, 0).shape == (0, 2 * (n - 1) + 2)


@pytest.mark.gpu
def test_recall_study_recovers_ground_truth():
    """Sampler -> cohort composition of the real data -> independence start -> learn_mhn on the engine
    (examples/recall_study.py:110-170 with a synthetic ground truth): the fit explains the data at least as well
    as the truth and recovers base rates, interactions and the signs of the strong effects."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "recall_study", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "recall_study.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run(n_mut=8, n_dat=5000, lam=1e-3, seed=42, n_sim=200_000, verbose=False)
    assert r["score_fit"] >= r["score_truth"] - 1e-3
    assert r["r_diag"] > 0.85 and r["r_off"] > 0.7 and r["sign_strong"] > 0.8, r
