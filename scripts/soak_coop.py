#!/usr/bin/env python3
"""Soak of the cooperative tile launch (csrc/tsolve.h): N evaluations of the 28-event LUAD cohort, every result compared with the
fixture - a stale read of another workgroup's tile (a hand-off that is only wrong now and then, under uneven load) would show as a
deviation.  The side streams of the evaluation (small-space launches, the staged own-problem chain with its own cooperative
launches) run next to the joint solves: the load is uneven by construction.
    python scripts/soak_coop.py [evaluations=2000]      (MMHN_COOP_WGS=<n> to vary the number of workgroups)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metmhn_amd import Engine, distributed as D

g = np.load(os.path.join(ROOT, "tests", "golden", "luad28.npz"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dat = g["dat"]
n = (dat.shape[1] - 3) // 2
e = Engine(n)
e.set_cohort(dat)
pm = float(g["perc_met"])
worst = {"indep": [0.0, 0.0], "fit": [0.0, 0.0]}
t0 = time.perf_counter()
for it in range(N):
    pt = "fit" if it & 1 else "indep"
    s, G, a, b = D.combine_sums(e.cohort_sums(g[pt + "_theta"], g[pt + "_dp"], g[pt + "_dm"]), n + 1, pm)
    ds = abs(float(s) - float(g[pt + "_score"])) / abs(float(g[pt + "_score"]))
    dg = float(np.max(np.abs(G - g[pt + "_d_th"]))) / float(np.max(np.abs(g[pt + "_d_th"])))
    worst[pt][0] = max(worst[pt][0], ds)
    worst[pt][1] = max(worst[pt][1], dg)
    if not (ds < 1e-9 and dg < 1e-7):
        print(f"evaluation {it} ({pt}): score off by {ds:.3e}, gradient by {dg:.3e}", flush=True)
        sys.exit(1)
dt = time.perf_counter() - t0
print(f"{N} evaluations in {dt:.1f} s ({dt / N * 1e3:.3f} ms each), workgroups {os.environ.get('MMHN_COOP_WGS', 'default')}: "
      f"worst relative deviation score {max(worst['indep'][0], worst['fit'][0]):.2e}, d_theta {max(worst['indep'][1], worst['fit'][1]):.2e}")
