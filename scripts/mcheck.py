#!/usr/bin/env python3
"""Matrix-path solves (MMHN_MSOLVE) against oracle/metmhn_fast.c on n = 20 full-k patients, then the bench cohort timing.
    python scripts/mcheck.py [patients=5000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metmhn_amd import Engine, synthetic
from metmhn_amd import _lib

n = 20
lt, dp, dm = synthetic.random_params(n)
if os.environ.get("MCHECK_PARITY", "1") == "1":
    from oracle import cref
    dat = synthetic.full_k_cohort(n, 12, seed=2000 + n)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    os.environ["MMHN_PSOLVE_MIN"] = "1"
    for ms in ("1", "0"):
        os.environ["MMHN_MSOLVE"] = ms
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        print(f"MSOLVE={ms}: lp err {np.max(np.abs(r[0] - lp) / np.abs(lp)):.3e}  g err {np.max(np.abs(r[1] - g)):.3e}"
              f"  dp err {np.max(np.abs(r[2] - a)):.3e}  dm err {np.max(np.abs(r[3] - b)):.3e}", flush=True)
    del os.environ["MMHN_PSOLVE_MIN"]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
for ms in os.environ.get("MCHECK_MODES", "1,0").split(","):
    os.environ["MMHN_MSOLVE"] = ms
    e = Engine(n)
    e.set_cohort(dat)
    e.cohort_sums(lt, dp, dm)
    e.reset_counters() if hasattr(e, "reset_counters") else None
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        s = e.cohort_sums(lt, dp, dm)
    dt = (time.perf_counter() - t0) / reps
    c = e.counters() if hasattr(e, "counters") else None
    print(f"MSOLVE={ms}: {dt * 1e3:.2f} ms per evaluation, score sum {s[0]:.12g}", flush=True)
    if c is not None:
        print("   ", c, flush=True)
    e.close()
