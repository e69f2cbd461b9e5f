#!/usr/bin/env python3
"""Where the small-space kernels (csrc/small.h) spend their cycles on the paired rows of the LUAD-reduced cohort.
Diagnostic build: csrc copied, the flushes of the tile solvers removed and STAMP_FLUSH(SPB == 64 ? 0 : 8) in small.h, -DMMHN_STAMPS:
    MMHN_LIB=build_ab/libsstamps.so python scripts/luad_stamps.py [all]
Shares only (the stamped build's fences forbid overlaps the product kernel has)."""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from metmhn_amd import Engine, _lib

g = np.load(os.path.join(R, "tests", "golden", "luad_indep.npz"))
dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
if len(sys.argv) < 2:
    dat = dat[dat[:, -1] == 3]
print("rows", dat.shape[0])
e = Engine(20)
e.set_cohort(dat)
e.cohort_sums(lt, dp, dm)
lib = _lib.load()
lib.mmhn_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
out = (C.c_double * 16)()
lib.mmhn_debug_stamps(e.h, out, 1)
e.cohort_sums(lt, dp, dm)
lib.mmhn_debug_stamps(e.h, out, 1)
v = np.array(out[:])
names = ["0 setup: descriptor", "1 setup: gathers from theta", "2 setup: Lc / Uc / observation products", "3 setup: 1 / diagonal",
         "4 forward solve, score", "5 adjoint solve, q out, dots", "6 gradient rows", "7"]
for base, nm in ((0, "one-wave class (wave 0 of each workgroup)"), (8, "256 / 1024-thread classes")):
    tot = v[base:base + 8].sum()
    print(f"{nm}: total {tot:.0f}")
    for i in range(8):
        if v[base + i]:
            print(f"   {names[i]:35s} {100 * v[base + i] / tot:5.1f} %  {v[base + i]:10.0f}")
