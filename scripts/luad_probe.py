import sys, os, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from metmhn_amd import Engine
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/luad_indep.npz"))
dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
e = Engine(20); e.set_cohort(dat)
for wg in (True, False):
    for _ in range(3): e.cohort_sums(lt, dp, dm, with_grad=wg)
    t = time.perf_counter()
    for _ in range(20): e.cohort_sums(lt, dp, dm, with_grad=wg)
    print("with_grad", wg, (time.perf_counter() - t) / 20 * 1e3, "ms")
