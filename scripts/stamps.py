#!/usr/bin/env python3
"""Where a k_psolve tile spends its cycles (diagnostic build with -DMMHN_STAMPS, see scripts/build_variants.sh):
    MMHN_LIB=build_ab/libmetmhn_stamps.so python scripts/stamps.py [patients]
Shares only - the stamped build's fences forbid overlaps the product kernel has; never quote its run time."""
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
names = ["0 tile setup (dl fetch, hx) + barrier", "1 Utab, rhs, barrier", "2 step A (neighbour tiles)", "3 popcount order + 1/diag + barrier",
         "4 step B (13 levels)", "5 step C stores + end barrier", "6 -", "7 eq block"]
for half, nm in ((0, "forward"), (8, "adjoint")):
    tot = v[half:half + 8].sum()
    print(f"{nm}: {tot / P / 128:.0f} cycles per tile (wave 0), shares:")
    for i in range(8):
        if v[half + i]:
            print(f"   {names[i]:45s} {100 * v[half + i] / tot:5.1f} %   {v[half + i] / P / 128:8.0f} cyc/tile")
print(e.counters())
