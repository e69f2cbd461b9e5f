"""Gillespie sampler throughput: HIP sampler (mmhn_simulate, incl. the download) vs the NumPy restatement of
simulations.py on the host.  python scripts/bench_sampler.py [n_mut] [gpu samples] [cpu samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metmhn_amd import Engine, synthetic
from oracle import gillespie                      # baseline only

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
lt, dp, dm = synthetic.random_params(n)
e = Engine(n)
e.simulate(lt, dp, dm, 1000, 1)
t0 = time.perf_counter(); d = e.simulate(lt, dp, dm, ng, 2); tg = time.perf_counter() - t0
t0 = time.perf_counter(); c = gillespie.simulate_dat(lt, dp, dm, nc, seed=3); tc = time.perf_counter() - t0
print(f"n_mut={n}: GPU {ng / tg / 1e6:.2f} M samples/s ({ng} in {tg * 1e3:.0f} ms incl. download), "
      f"NumPy {nc / tc / 1e3:.1f} k samples/s ({nc} in {tc:.2f} s); mean events set GPU {d[:, :-1].sum(1).mean():.2f} "
      f"NumPy {c[:, :-1].sum(1).mean():.2f}; paired fraction GPU {(d[:, -2] == 1).mean():.4f} NumPy {(c[:, -2] == 1).mean():.4f}")
