#!/usr/bin/env python3
"""Interleaved A/B timing of environment switches on the bench cohort, one fresh process per variant and round:
    python scripts/ab_env.py [--patients 5000] [--rounds 3] [--n 20] [--dtype f64] "MMHN_WSOLVE=1" "MMHN_WSOLVE=0" "default" ...
A variant is a space-separated list of NAME=value settings (or "default"); MMHN_LIB=<path> selects another build.
Prints ms per evaluation and the per-kernel-class milliseconds; the first variant's objective is the parity reference."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %r)
from metmhn_amd import Engine, synthetic
P, n, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lt, dp, dm = synthetic.random_params(n)
shapes = os.environ.get("AB_SHAPES")            # e.g. "14,10;15,9": only paired rows with these (larger, smaller) class sizes
if shapes:
    want = {tuple(int(v) for v in s.split(",")) for s in shapes.split(";")}
    big = np.asarray(synthetic.full_k_cohort(n, 8 * P, seed=2000 + n))
    kp, km = big[:, 0:2 * n:2].sum(1), big[:, 1:2 * n:2].sum(1)
    keep = [i for i in range(len(big)) if (max(kp[i], km[i]), min(kp[i], km[i])) in want][:P]
    dat = big[keep]
else:
    dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
e = Engine(n, dtype=dtype); e.set_cohort(dat)
for _ in range(2): s = e.cohort_sums(lt, dp, dm)
e.reset_counters()
t0 = time.perf_counter()
for _ in range(5): s = e.cohort_sums(lt, dp, dm)
dt = (time.perf_counter() - t0) / 5
c = e.counters()
print(json.dumps(dict(rows=len(dat), ms=dt * 1e3, fwd=c["psolve_fwd"]["ms"] / 5, adj=c["psolve_adj"]["ms"] / 5, marg=c["pclass"]["ms"] / 5,
                      other=c["other_solve"]["ms"] / 5, s0=float(s[0]), g=float(np.abs(s[4:]).sum()))))
''' % ROOT

args = sys.argv[1:]
P, rounds, n, dtype = 5000, 3, 20, "f64"
variants = []
i = 0
while i < len(args):
    if args[i] == "--patients": P = int(args[i + 1]); i += 2
    elif args[i] == "--rounds": rounds = int(args[i + 1]); i += 2
    elif args[i] == "--n": n = int(args[i + 1]); i += 2
    elif args[i] == "--dtype": dtype = args[i + 1]; i += 2
    else: variants.append(args[i]); i += 1
ref = None
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        if v != "default":
            for kv in v.split():
                name, val = kv.split("=", 1)
                env[name] = val
        out = subprocess.run([sys.executable, "-c", CHILD, str(P), str(n), dtype], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(v, "FAILED", out.stderr[-800:]); continue
        if ref is None:
            ref = d
        print(f"round {r} {v:40s} {d['rows']} rows {d['ms']:8.2f} ms  fwd {d['fwd']:6.2f} adj {d['adj']:6.2f} marg {d['marg']:6.2f} other {d['other']:5.2f}"
              f"  d_lp {abs(d['s0'] - ref['s0']) / abs(ref['s0']):.1e} d_g {abs(d['g'] - ref['g']) / abs(ref['g']):.1e}", flush=True)
