#!/usr/bin/env python3
"""Batched kronvec launches only (the command behind the kronvec PMC passes and A/B timings):
    python scripts/kv_only.py [n=20] [k=20] [batch=64] [iters=20]
prints ms per launch and algorithmic TB/s for Q_off p, Q_off^T p and the fused Jacobi step, and checks Q_off p of one
vector against the CPU oracle's C port at k <= 16."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
k = int(sys.argv[2]) if len(sys.argv) > 2 else n
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lt, dp, dm = synthetic.random_params(n)
st = synthetic.full_k_cohort(n, 1, k=k, seed=2000 + n)[0, :2 * n + 1]
e = Engine(n)
V = 2 ** int(st.sum()) * 8
for name, tr, jac, mult in (("kronvec", 0, 0, 2), ("kronvec_T", 1, 0, 2), ("jacobi_step", 0, 1, 4)):
    ms, live, tot = min(e.bench_kronvec(lt, st, batch, iters, transpose=tr, jacobi=jac, tiles=True) for _ in range(3))
    tile_b = V * batch / tot
    live_b = (live + tot) * tile_b if not jac else (live + 3 * tot) * tile_b      # bytes the launch has to move
    print(f"{name:12s} {ms:8.4f} ms/launch  {mult * V * batch / ms / 1e9:7.3f} TB/s on SURVEY 8(d) bytes ({mult * V * batch / ms / 1e9 / 8:.3f} of 8 TB/s)"
          f"  {live_b / ms / 1e9:7.3f} TB/s on the bytes it must move ({live_b / ms / 1e9 / 8:.3f});  {live} of {tot} tiles carry values")
