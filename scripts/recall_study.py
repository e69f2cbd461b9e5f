"""End-to-end recall study on the GPU, the procedure of examples/recall_study.py:110-170 with a synthetic ground
truth: simulate patients from known parameters (HIP Gillespie sampler), subsample them to the composition of the
real cohort (11.5 % never metastasised; of the rest 10.7 % paired, 38.6 % PT-only, remainder MT-only), fit
`learn_mhn` from the independence start on the engine, and compare the fit with the ground truth.

    python scripts/recall_study.py [n_mut] [n_dat] [lambda]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metmhn_amd import simulations, synthetic
from metmhn_amd.Utilityfunctions import create_dat, indep
import metmhn_amd.regularized_optimization as ro


def run(n_mut=8, n_dat=5000, lam=1e-3, seed=42, n_sim=200_000, ftol=1e-6, verbose=True):
    rng = np.random.default_rng(seed)
    lt, dp, dm = synthetic.random_params(n_mut, seed=seed)
    lt = lt * 0.7
    np.fill_diagonal(lt, np.diag(lt) - 1.0)             # moderate base rates: genotypes neither empty nor full
    t0 = time.perf_counter()
    sim = simulations.simulate_dat(lt, dp, dm, n_sim, original_key=seed)
    t_sim = time.perf_counter() - t0
    n_nm = int(round(0.115 * n_dat)); n_em = n_dat - n_nm
    n_c = int(round(0.107 * n_em)); n_pm = int(round(0.386 * n_em)); n_mo = n_em - n_c - n_pm
    dat = create_dat(sim, n_em, n_nm, n_c, n_pm, n_mo, rng)
    th0, dp0, dm0 = indep(dat)
    t0 = time.perf_counter()
    th, a, b = ro.learn_mhn(th0, dp0, dm0, dat, 0.65, ro.symmetric_penal, lam, opt_ftol=ftol, opt_v=False)
    t_fit = time.perf_counter() - t0
    off = ~np.eye(n_mut + 1, dtype=bool)
    strong = off & (np.abs(lt) > 0.5)
    res = dict(r_diag=float(np.corrcoef(np.diag(lt), np.diag(th))[0, 1]),
               r_off=float(np.corrcoef(lt[off], th[off])[0, 1]),
               sign_strong=float((np.sign(lt[strong]) == np.sign(th[strong])).mean()) if strong.any() else 1.0,
               score_fit=float(ro.score(th, a, b, dat, 0.65)), score_truth=float(ro.score(lt, dp, dm, dat, 0.65)),
               t_sim=t_sim, t_fit=t_fit, n_dat=int(dat.shape[0]))
    if verbose:
        print(f"n_mut={n_mut}: {n_sim} simulated in {t_sim * 1e3:.0f} ms, {dat.shape[0]} datapoints "
              f"(types {np.bincount(dat[:, -1], minlength=4)}), fit in {t_fit:.1f} s")
        print({k: round(v, 4) if isinstance(v, float) else v for k, v in res.items()})
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 8, int(a[1]) if len(a) > 1 else 5000, float(a[2]) if len(a) > 2 else 1e-3)
