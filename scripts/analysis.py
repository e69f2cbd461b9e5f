#!/usr/bin/env python3
"""Equivalent of the reference's examples/analysis.py on the GPU engine:
events CSV + annotation CSV -> dat -> indep start -> (optional cross-validation) -> learn_mhn -> params CSV."""
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metmhn_amd.regularized_optimization as reg_opt   # noqa: E402
import metmhn_amd.Utilityfunctions as utils             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("source_data"); ap.add_argument("source_annot"); ap.add_argument("target")
ap.add_argument("--lam", type=float, default=1e-3)
ap.add_argument("--pm_ratio", type=float, default=None)
ap.add_argument("--cv", action="store_true")
ap.add_argument("--cv_start", type=float, default=1e-4); ap.add_argument("--cv_end", type=float, default=1e-2)
ap.add_argument("--cv_splits", type=int, default=5); ap.add_argument("--cv_folds", type=int, default=5)
ap.add_argument("--seed", type=int, default=42); ap.add_argument("--logs", default="analysis.log")
a = ap.parse_args()
logging.basicConfig(filename=a.logs, filemode="w", level=logging.INFO, force=True,
                    format="%(asctime)s %(levelname)-8s %(message)s")
dat, events = utils.load_cohort(a.source_data, a.source_annot)
perc_met = a.pm_ratio if a.pm_ratio is not None else dat[:, -3].sum() / (dat.shape[0] - dat[:, -3].sum())
lam = a.lam
if a.cv:
    lams = 10 ** np.linspace(np.log10(a.cv_start), np.log10(a.cv_end), a.cv_splits)
    res = utils.cross_val(dat, reg_opt.symmetric_penal, lams, a.cv_folds, perc_met, key=a.seed)
    lam = lams[np.argmax(np.mean(res.to_numpy(), axis=0))]
th0, dp0, dm0 = utils.indep(dat)
theta, d_p, d_m = reg_opt.learn_mhn(th0, dp0, dm0, dat, perc_met, reg_opt.symmetric_penal, lam)
utils.save_params(a.target, theta, d_p, d_m, events)
