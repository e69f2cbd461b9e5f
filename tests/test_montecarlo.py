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
        assert count > 50, f"{name}: too few samples ({count})"
        assert abs(count / N_SIM - p) < 4.5 * sigma, f"{name}: simulated {count / N_SIM:.5f} vs analytic {p:.5f}"


def test_oracle_matches_simulated_frequencies(sim):
    from oracle import metmhn_oracle as O
    _check(O.score, sim)


@pytest.mark.gpu
def test_engine_matches_simulated_frequencies(sim):
    import metmhn_amd.regularized_optimization as ro
    _check(ro.score, sim)
