// gfx950 kernels of the metMHN hot path (closed-form "gather" formulation) - every kernel family of the index-order
// formulation (round 5: one header per family; this file is the umbrella the other headers and engine.hip include):
//   common.h     tile constants, per-problem tables (k_prep, tile_tables), device helpers, in-kernel stamps
//   kv.h         Kronecker products: k_sweep, k_hx, k_kv                      (kronvec.py:499-539)
//   psolve.h     joint solves, one workgroup per patient: k_psolve2           (likelihood.py:231-262)
//   diag.h       diagonal quantities: k_diag                                  (kronvec.py:574-710, 964-999)
//   marg.h       joint <-> marginal transfers, seeds, staged init             (likelihood.py:540-620)
//   classmarg.h  class marginals: k_class_marg, k_eq_flows, k_pclass          (likelihood.py:25-201)
//   gradrows.h   gradient rows, bit marginals: k_grad_rows, k_bit_marg        (likelihood.py:163-228, vanilla.py:328-393)
//   assemble.h   per-patient assembly, cohort reduction                       (likelihood.py:441-731)
// Tile solves (k_tsolve, k_csolve) live in tsolve.h, the window-layout kernels in wsolve.h / wclass.h, the small-space
// kernels in small.h, the Gillespie sampler in sampler.h.
#pragma once
#include "common.h"
#include "kv.h"
#include "psolve.h"
#include "diag.h"
#include "marg.h"
#include "classmarg.h"
#include "gradrows.h"
#include "assemble.h"
