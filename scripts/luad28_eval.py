#!/usr/bin/env python3
"""The 28-event LUAD cohort (what examples/analysis.py of the reference really fits; tests/golden/luad28.npz) through the
engine: parity of score / gradient against the fixture at both parameter points, per-patient check of the paired rows, and
the time of an evaluation.
    python scripts/luad28_eval.py [reps=20] [point=indep|fit]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metmhn_amd import Engine

g = np.load(os.path.join(ROOT, "tests", "golden", "luad28.npz"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
points = [sys.argv[2]] if len(sys.argv) > 2 else ["indep", "fit"]
dat = g["dat"]
n = (dat.shape[1] - 3) // 2
e = Engine(n)
t0 = time.perf_counter()
e.set_cohort(dat)
print(f"set_cohort {time.perf_counter() - t0:.3f} s", flush=True)
pm = float(g["perc_met"])
for pt in points:
    lt, dp, dm = g[pt + "_theta"], g[pt + "_dp"], g[pt + "_dm"]
    s, G, a, b = e.score_and_grad(lt, dp, dm, pm)
    rel = lambda x, y: float(np.max(np.abs(np.asarray(x) - y)) / max(np.max(np.abs(y)), 1e-300))
    print(pt, "score", float(s), "golden", float(g[pt + "_score"]), "rel err: score %.2e d_th %.2e d_dp %.2e d_dm %.2e" % (
        abs(float(s) - float(g[pt + "_score"])) / abs(float(g[pt + "_score"])), rel(G, g[pt + "_d_th"]), rel(a, g[pt + "_d_dp"]),
        rel(b, g[pt + "_d_dm"])), flush=True)
    lp = e.patient_grads(lt, dp, dm)[0]
    print(pt, "per-patient lp: max abs err", float(np.max(np.abs(lp - g[pt + "_lp"]))), flush=True)
    ts, tsc = [], []
    for _ in range(reps):
        t1 = time.perf_counter()
        e.score_and_grad(lt, dp, dm, pm)
        ts.append(time.perf_counter() - t1)
    for _ in range(reps):
        t1 = time.perf_counter()
        e.score(lt, dp, dm, pm)
        tsc.append(time.perf_counter() - t1)
    print(pt, "ms per evaluation: with gradient %.3f (min %.3f), score only %.3f" % (np.median(ts) * 1e3, min(ts) * 1e3, np.median(tsc) * 1e3), flush=True)
