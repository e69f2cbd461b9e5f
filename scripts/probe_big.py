import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic
n, P, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P)
e = Engine(n, dtype=dt); t0 = time.time(); e.set_cohort(dat); print("set_cohort", round(time.time() - t0, 2), "s")
t0 = time.time(); lp, g, a, b = e.patient_grads(lt, dp, dm); print("first eval", round(time.time() - t0, 3), "s")
t0 = time.time(); s = e.cohort_sums(lt, dp, dm); dtm = time.time() - t0
print(f"n={n} k={n} P={P} {dt}: eval {dtm*1e3:.1f} ms ({dtm/P*1e3:.2f} ms/patient); lp[:3]={lp[:3]}, finite={np.isfinite(g).all()}, |G|={np.linalg.norm(g[0]):.6f}")
np.save(f"gpurun_out/big_{n}_{dt}.npy", np.concatenate((lp, g.reshape(P, -1).sum(1))))
