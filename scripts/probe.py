import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P)
e = Engine(n)
t0 = time.time(); e.set_cohort(dat); print("set_cohort", time.time()-t0)
for it in range(3):
    e.reset_counters()
    t0 = time.time(); s = e.cohort_sums(lt, dp, dm); dt = time.time()-t0
    c = e.counters()
    print(f"eval {dt*1e3:.1f} ms  ({dt/P*1e6:.1f} us/patient)  sweep {c['sweep_ms']:.1f} ms in {c['sweep_launches']} launches, {c['sweep_alg_bytes']/c['sweep_ms']/1e6:.1f} GB/s alg;  lp_sum {s[0]:.10f}")
t0 = time.time(); s2 = e.cohort_sums(lt, dp, dm, with_grad=False); print("score only", (time.time()-t0)*1e3, "ms", s2[0])
st = dat[0, :2*n+1]
for jac in (0, 1):
    for tr in (0, 1):
        ms = e.bench_kronvec(lt, st, min(P, 64), 10, transpose=tr, jacobi=jac)
        V = 2**int(st.sum()) * 8
        by = (4 if jac else 2) * V * min(P, 64)
        print(f"kronvec jac={jac} tr={tr}: {ms:.3f} ms/launch  {by/ms/1e6:.1f} GB/s algorithmic")
