#!/usr/bin/env python3
"""examples/analysis.py:104-113 of the reference on the engine: learn_mhn of the 28-event LUAD cohort from indep(dat)
(tests/golden/luad28.npz; perc_met 0.2, lambda 1e-3 as in the LUAD fixtures, SciPy L-BFGS-B with the reference's default
ftol 1e-4 and with 1e-8) - wall time, evaluations, objective, and the objective at the reference's published parameters.
    python scripts/luad28_fit.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metmhn_amd.regularized_optimization as ro

g = np.load(os.path.join(ROOT, "tests", "golden", "luad28.npz"))
dat, pm, lam = g["dat"], float(g["perc_met"]), float(g["lam"])
pub = np.concatenate((g["fit_theta"].flatten(), g["fit_dp"], g["fit_dm"]))
v_pub = float(ro.score_reg(pub, dat, pm, ro.symmetric_penal, lam))
print(f"objective at the published parameters: {v_pub:.10f} (fixture {float(g['fit_reg_value']):.10f})")
calls = [0]
orig = ro.score_and_grad_reg


def counted(*a, **k):
    calls[0] += 1
    return orig(*a, **k)


ro.score_and_grad_reg = counted
for ftol in (1e-4, 1e-8):
    calls[0] = 0
    t0 = time.perf_counter()
    th, dp, dm = ro.learn_mhn(g["indep_theta"], g["indep_dp"], g["indep_dm"], dat, pm, ro.symmetric_penal, lam, opt_ftol=ftol, opt_v=False)
    dt = time.perf_counter() - t0
    v = float(ro.score_reg(np.concatenate((th.flatten(), dp, dm)), dat, pm, ro.symmetric_penal, lam))
    print(f"learn_mhn ftol {ftol:g}: {dt:.2f} s, {calls[0]} evaluations ({dt / max(calls[0], 1) * 1e3:.2f} ms each incl. SciPy), objective {v:.10f}")
