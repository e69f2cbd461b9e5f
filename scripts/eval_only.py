#!/usr/bin/env python3
"""Two evaluations of the bench cohort and nothing else: the command the PMC passes of scripts/pmc_eval.sh profile.
    python scripts/eval_only.py [patients=5000] [n=20] [dtype=f64]      python scripts/eval_only.py luad"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic

import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "luad":            # the LUAD-reduced cohort at indep(dat) (tests/golden/luad_indep.npz)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "luad_indep.npz"))
    dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
    n = 20
    sys.argv[1:2] = []
else:
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
    if os.environ.get("AB_SHAPES"):                           # e.g. "12,12;13,11": only rows with these (larger, smaller) class sizes
        want = {tuple(int(v) for v in s.split(",")) for s in os.environ["AB_SHAPES"].split(";")}
        big = np.asarray(synthetic.full_k_cohort(n, 8 * P, seed=2000 + n))
        kp, km = big[:, 0:2 * n:2].sum(1), big[:, 1:2 * n:2].sum(1)
        dat = big[[i for i in range(len(big)) if (max(kp[i], km[i]), min(kp[i], km[i])) in want][:P]]
        os.environ.setdefault("MMHN_PSOLVE_MIN", "1")
e = Engine(n, dtype=sys.argv[3] if len(sys.argv) > 3 else "f64")
e.set_cohort(dat)
for _ in range(2):
    s = e.cohort_sums(lt, dp, dm)
print(s[0])
