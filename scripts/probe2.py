import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic
n, P = int(sys.argv[1]), int(sys.argv[2])
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P)
e = Engine(n); e.set_cohort(dat)
e.cohort_sums(lt, dp, dm)
e.reset_counters(); t0 = time.time()
for _ in range(3): s = e.cohort_sums(lt, dp, dm)
dt = (time.time() - t0) / 3; c = e.counters()
print(f"eval {dt*1e3:.1f} ms ({dt/P*1e6:.1f} us/patient), solves {c['sweep_ms']/3:.1f} ms, rest {dt*1e3 - c['sweep_ms']/3:.1f} ms, lp {s[0]:.6f}")
