"""Two engines (own streams) on one GPU, each with half of the cohort, evaluated concurrently from two threads
vs one engine with the whole cohort: does overlapping the latency-bound solves of one half with the
bandwidth-bound marginal pass of the other raise the throughput?  python scripts/probe_dual.py [n] [patients]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metmhn_amd import Engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
e = Engine(n); e.set_cohort(dat); e.cohort_sums(lt, dp, dm)
t0 = time.perf_counter()
for _ in range(4): s = e.cohort_sums(lt, dp, dm)
t1 = (time.perf_counter() - t0) / 4
e.close()
halves = [dat[:P // 2], dat[P // 2:]]
es = [Engine(n) for _ in halves]
for x, d in zip(es, halves):
    x.set_cohort(d); x.cohort_sums(lt, dp, dm)
res = [None, None]
def work(i, offset):
    if offset: time.sleep(offset)
    for _ in range(4): res[i] = es[i].cohort_sums(lt, dp, dm)
for off in (0.0, 0.02):
    ths = [threading.Thread(target=work, args=(i, off * i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    t2 = (time.perf_counter() - t0 - off) / 4
    print(f"one engine {P}: {t1 * 1e3:.1f} ms/eval; two engines {P // 2} each, concurrent (start offset {off * 1e3:.0f} ms): {t2 * 1e3:.1f} ms per pair; "
          f"sum check {s[0]:.6f} vs {res[0][0] + res[1][0]:.6f}")

# per-call version: both halves start inside one call, the second one delayed, and are joined before returning
def one(i, delay):
    if delay: time.sleep(delay)
    res[i] = es[i].cohort_sums(lt, dp, dm)
for off in (0.0, 0.005, 0.010, 0.015, 0.020, 0.030):
    t0 = time.perf_counter()
    for _ in range(6):
        ths = [threading.Thread(target=one, args=(i, off * i)) for i in range(2)]
        for t in ths: t.start()
        for t in ths: t.join()
    print(f"per call, second half delayed {off * 1e3:.0f} ms: {(time.perf_counter() - t0) / 6 * 1e3:.1f} ms per evaluation")
