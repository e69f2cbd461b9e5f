#!/usr/bin/env python3
"""Where k_psolve spends its cycles on the paired rows of the LUAD-reduced cohort (diagnostic build: csrc copied, the
flushes of small.h removed, -DMMHN_STAMPS):  MMHN_LIB=build_ab/libpstamps.so python scripts/luad_psolve_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from metmhn_amd import Engine, _lib

g = np.load(os.path.join(R, "tests", "golden", "luad_indep.npz"))
dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
dat = dat[dat[:, -1] == 3]
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
P = dat.shape[0]
names = ["0 tile setup (dl fetch, hx) + barrier", "1 Utab, rhs, barrier", "2 step A (neighbour tiles)", "3 popcount order + 1/diag + barrier",
         "4 step B (levels)", "5 step C stores + end barrier", "6 -", "7 per-patient setup / eq block"]
for half, nm in ((0, "forward"), (8, "adjoint")):
    tot = v[half:half + 8].sum()
    print(f"{nm}: {tot / P:.0f} cycles per patient (wave 0), shares:")
    for i in range(8):
        if v[half + i]:
            print(f"   {names[i]:45s} {100 * v[half + i] / tot:5.1f} %   {v[half + i] / P:8.0f} cyc/patient")
