#!/usr/bin/env python3
"""Interleaved A/B timing of library builds on the bench cohort (one process per build, rounds interleaved):
    python scripts/ab_eval.py [--patients 5000] [--rounds 3] build_ab/liba.so build_ab/libb.so ...
Each round runs every build once (fresh process: 2 warm-up + 5 timed evaluations) and prints ms per evaluation and
the per-kernel-class milliseconds; the first build's objective value is the parity reference for the others."""
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
n, P = 20, int(sys.argv[1])
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
e = Engine(n); e.set_cohort(dat)
for _ in range(2): s = e.cohort_sums(lt, dp, dm)
e.reset_counters()
t0 = time.perf_counter()
for _ in range(5): s = e.cohort_sums(lt, dp, dm)
dt = (time.perf_counter() - t0) / 5
c = e.counters()
print(json.dumps(dict(ms=dt * 1e3, fwd=c["psolve_fwd"]["ms"] / 5, adj=c["psolve_adj"]["ms"] / 5, marg=c["pclass"]["ms"] / 5,
                      other=c["other_solve"]["ms"] / 5, s0=float(s[0]), g=float(np.abs(s[4:]).sum()))))
''' % ROOT

args = sys.argv[1:]
P, rounds = 5000, 3
libs = []
i = 0
while i < len(args):
    if args[i] == "--patients": P = int(args[i + 1]); i += 2
    elif args[i] == "--rounds": rounds = int(args[i + 1]); i += 2
    else: libs.append(args[i]); i += 1
ref = None
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["MMHN_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, "-c", CHILD, str(P)], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(lib, "FAILED", out.stderr[-800:]); continue
        if ref is None:
            ref = d
        print(f"round {r} {lib:32s} {d['ms']:8.2f} ms  fwd {d['fwd']:6.2f} adj {d['adj']:6.2f} marg {d['marg']:6.2f} other {d['other']:5.2f}"
              f"  d_lp {abs(d['s0'] - ref['s0']) / abs(ref['s0']):.1e} d_g {abs(d['g'] - ref['g']) / abs(ref['g']):.1e}", flush=True)
