"""Only the batched kronvec leg: python scripts/kv_only.py <n> <k> <batch> <iters> [jacobi] [transpose]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metmhn_amd import Engine, synthetic
n, k, batch, iters = (int(a) for a in sys.argv[1:5])
jac = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tr = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lt, dp, dm = synthetic.random_params(n)
st = synthetic.full_k_cohort(n, 1, k=k)[0, :2 * n + 1]
e = Engine(n)
ms = e.bench_kronvec(lt, st, batch, iters, transpose=tr, jacobi=jac)
V = 2 ** k * 8
print(f"n={n} k={k} batch={batch} iters={iters} jac={jac} tr={tr}: {ms:.4f} ms/launch, alg {(4 if jac else 2) * V * batch / ms / 1e6:.1f} GB/s; "
      f"per launch: vector bytes r/w {V * batch} / {V * batch}, table bytes {(k * k + 2 * k * 64) * 8 * batch}")
