#!/usr/bin/env python3
"""Per-patient gradients of the small-space path (csrc/small.h) against the staged kernels it replaces (MMHN_SMALL=0) on
300 paired rows at n = 16 with 10 - 16 active slots: which rows differ, their types and sizes (debugging aid).
    python scripts/small_vs_staged.py"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from metmhn_amd import Engine, synthetic
n = 16
lt, dp, dm = synthetic.random_params(n, seed=77)
dat = np.vstack([synthetic.full_k_cohort(n, 50, k=kk, seed=900 + kk) for kk in (10, 12, 13, 14, 15, 16)])
res = {}
for small in ("1", "0"):
    os.environ["MMHN_SMALL"] = small
    e = Engine(n); e.set_cohort(dat); res[small] = e.patient_grads(lt, dp, dm); e.close()
a, b = res["1"], res["0"]
bad = [i for i in range(dat.shape[0]) if not np.allclose(a[1][i], b[1][i], rtol=1e-7, atol=1e-10)]
print("bad patients", len(bad))
for i in bad[:12]:
    row = dat[i]
    pt = int(row[0:2*n:2].sum()); mt = int(row[1:2*n:2].sum())
    d = np.abs(a[1][i] - b[1][i]); ii = np.unravel_index(d.argmax(), d.shape)
    print(i, "type", row[-1], "order", row[-2], "pt", pt, "mt", mt, "seed", row[2*n], "maxdiff", d.max(), "at", ii, "nbad", int((d > 1e-9).sum()))
