import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, _lib
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/luad_indep.npz"))
dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
sel = None
if len(sys.argv) > 1:      # only patients whose largest single space has k in [lo, hi]
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    n = 20; typ = dat[:, -1]; pt = dat[:, 0:2*n:2].sum(1); mt = dat[:, 1:2*n:2].sum(1); seed = dat[:, 2*n]
    ks = np.where(typ <= 1, pt + seed, np.where(typ == 2, mt + 1, np.maximum(mt + 1, pt + 1)))
    dat = dat[(ks >= lo) & (ks <= hi) & (typ != 3)]
print(dat.shape)
e = Engine(20); e.set_cohort(dat)
e.cohort_sums(lt, dp, dm)
lib = _lib.load()
lib.mmhn_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
out = (C.c_double * 16)()
lib.mmhn_debug_stamps(e.h, out, 1)
e.cohort_sums(lt, dp, dm)
lib.mmhn_debug_stamps(e.h, out, 1)
v = np.array(out[:8]); names = ["setup", "fwd solve", "seed/lp", "adjoint", "dots+bitmarg", "grad rows"]
for i, nm in enumerate(names): print(f"{nm:14s} {v[i] / max(v.sum(), 1) * 100:5.1f} %  {v[i]:.3e} cycles")
