#!/usr/bin/env python3
"""Where a k_wsolve step spends its cycles (diagnostic build -DMMHN_STAMPS; wave 0 of every workgroup):
    MMHN_LIB=build_ab/lib_stamps.so python scripts/wstamps.py [patients]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metmhn_amd import Engine, synthetic, _lib

n = 20
P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
e = Engine(n)
e.set_cohort(dat)
e.cohort_sums(lt, dp, dm)
lib = _lib.load()
lib.mmhn_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
out = (C.c_double * 16)()
lib.mmhn_debug_stamps(e.h, out, 1)
e.reset_counters()
e.cohort_sums(lt, dp, dm)
lib.mmhn_debug_stamps(e.h, out, 1)
v = np.array(out[:])
names = ["0 request ext 0,1 + lane moves (DPP)", "1 take ext 0,1 (wait), request ext 2,3", "2 wave moves (ring)",
         "3 window moves, take ext 2,3", "4 rhs, block solve, stores", "5 barrier", "6 begin of a pass", "7 patient enters / leaves"]
for half, nm in ((0, "forward"), (8, "adjoint")):
    tot = v[half:half + 8].sum()
    print(f"{nm}: {tot / P:.0f} cycles per patient (wave 0), shares:")
    for i in range(8):
        if v[half + i]:
            print(f"   {names[i]:45s} {100 * v[half + i] / tot:5.1f} %   {v[half + i] / P:10.0f} cyc/patient")
print(e.counters())
