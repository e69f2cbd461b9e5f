#!/usr/bin/env python3
"""tests/golden/luad_cpu_fit.npz: the fit of the LUAD-reduced cohort (BASELINE configs[0]) by SciPy's L-BFGS-B on the CPU oracle
(oracle/metmhn_ref.c through oracle/cref.py), with the reference's own settings (perc_met 0.2, lambda 1e-3, start indep(dat),
examples/data_analysis.ipynb cell 14; regularized_optimization.py:301-334) but a tight stopping rule (ftol 1e-10), so that the
GPU engine's fit can be compared to it parameter by parameter (tests/test_gpu_parity.py).

    python tests/tools/make_golden_luad_cpu_fit.py          (reads tests/golden/luad_indep.npz; ~20 min on 8 cores)
"""
import os
import sys
import time

import numpy as np
import scipy.optimize as opt

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from oracle import cref, metmhn_oracle as O      # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "luad_indep.npz"))
dat, pm, lam = g["dat"], float(g["perc_met"]), float(g["lam"])
N = g["indep_theta"].shape[0]
em = dat[:, -1] != 0
n_em = float(dat[:, -3].sum()); n_nm = dat.shape[0] - n_em
w = pm * n_nm / ((1 - pm) * n_em)
nf = w * n_em + n_nm
calls = [0]


def fun(params):
    lt, dp, dm = params[:N * N].reshape(N, N), params[N * N:N * N + N], params[N * N + N:]
    lp, G, a, b = cref.patients(lt, dp, dm, dat)
    s = (w * lp[em].sum() + lp[~em].sum()) / nf
    grad = np.concatenate((((w * G[em].sum(0) + G[~em].sum(0)) / nf).flatten(), (w * a[em].sum(0) + a[~em].sum(0)) / nf,
                           w * b[em].sum(0) / nf))
    pen, pen_ = O.symmetric_penal(params, N)
    calls[0] += 1
    if calls[0] % 25 == 0:
        print(calls[0], float(-s + lam * pen), flush=True)
    return float(-s + lam * pen), -grad + lam * np.asarray(pen_)


x0 = np.concatenate((g["indep_theta"].flatten(), g["indep_dp"], g["indep_dm"]))
t0 = time.time()
res = opt.minimize(fun=fun, jac=True, x0=x0, method="L-BFGS-B", options={"maxiter": 100000, "ftol": 1e-10})
print(res.message, res.nit, res.nfev, res.fun, f"{time.time() - t0:.0f} s")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "luad_cpu_fit.npz"), theta=res.x[:N * N].reshape(N, N),
                    dp=res.x[N * N:N * N + N], dm=res.x[N * N + N:], objective=np.float64(res.fun), nit=np.int64(res.nit),
                    nfev=np.int64(res.nfev), ftol=np.float64(1e-10), perc_met=np.float64(pm), lam=np.float64(lam),
                    grad_norm=np.float64(np.linalg.norm(res.jac)))
