#!/usr/bin/env python3
"""Where a tile of the tile solver (csrc/tsolve.h) spends its cycles on the 28-event LUAD cohort (diagnostic build -DMMHN_STAMPS):
    MMHN_LIB=build_ab/libstamps.so python scripts/tile_stamps.py
Shares only - the stamped build's fences forbid overlaps the product kernel has; never quote its run time."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metmhn_amd import Engine, _lib

g = np.load(os.path.join(ROOT, "tests", "golden", "luad28.npz"))
dat = g["dat"]
e = Engine(28)
e.set_cohort(dat)
lt, dp, dm = g["fit_theta"], g["fit_dp"], g["fit_dm"]
e.cohort_sums(lt, dp, dm)
lib = _lib.load()
lib.mmhn_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
out = (C.c_double * 16)()
lib.mmhn_debug_stamps(e.h, out, 1)
e.cohort_sums(lt, dp, dm)
lib.mmhn_debug_stamps(e.h, out, 1)
v = np.array(out[:])
names = ["0 descriptor, pext tables, 1/diag fetch", "1 rate tables (tile_tables)", "2 right-hand side", "3 wait for the tiles read",
         "4 step A (neighbour tiles)", "5 step B (in-tile solve)", "6 step C (stores, drained)"]
for half, nm in ((0, "forward"), (8, "adjoint")):
    tiles = v[half + 7]
    tot = v[half:half + 7].sum()
    if not tiles:
        continue
    print(f"{nm}: {tiles:.0f} tiles, {tot / tiles:.0f} cycles per tile (wave 0), shares:")
    for i in range(7):
        print(f"   {names[i]:45s} {100 * v[half + i] / tot:5.1f} %   {v[half + i] / tiles:8.0f} cyc/tile")
