"""Workflow callers of the hot path: the parts of metmhn/Utilityfunctions.py and
examples/analysis.py that sit directly around score / learn_mhn.  Host-side NumPy / pandas, same
names and argument meaning as the reference; the likelihood work goes to the GPU engine through
metmhn_amd.regularized_optimization.

Not mirrored (plots, Gillespie sampling, state-space helpers): out of scope, see DESIGN.md.
"""
from __future__ import annotations

import logging
from typing import Callable

import numpy as np

from .regularized_optimization import learn_mhn, score


def indep(dat) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Initial estimate of theta, d_p, d_m from marginal event counts (Utilityfunctions.py:157-183)."""
    dat = np.asarray(dat)
    n_coupled = int(np.sum(dat[:, -1] == 3))
    n = (dat.shape[1] - 3) // 2
    n_single = dat.shape[0] - n_coupled
    theta = np.zeros((n + 1, n + 1))
    for i in range(n):
        mut_count = np.sum(dat[:, 2 * i].astype(np.int64) + dat[:, 2 * i + 1].astype(np.int64))
        if mut_count == 0:
            theta[i, i] = -1e10
        else:
            theta[i, i] = np.log(mut_count / (2 * n_coupled + n_single - mut_count + 1e-10))
    seed_count = np.sum(dat[:, -3].astype(np.int64))
    theta[n, n] = np.log(seed_count / (n_coupled + n_single - seed_count + 1e-10))
    return theta, np.zeros(n + 1), np.zeros(n + 1)


def cross_val(dat, penal_fun: Callable, splits, n_folds: int, m_p_corr: float, key: int = 42,
              parallel_folds: bool = False, _learn=None, _score=None):
    """n_folds cross-validation over the penalty weights `splits` (Utilityfunctions.py:186-231).

    `key` seeds a NumPy Generator for the row permutation (the reference uses jax.random.permutation;
    the fold assignment therefore differs from the reference's for the same key, the procedure does not).
    Returns a DataFrame [n_folds x len(splits)] of held-out scores.

    `parallel_folds`: the len(splits) x n_folds fits are independent, so under torch.distributed (one process
    per GPU) rank r takes the jobs `distributed.fold_jobs` assigns to it, fits them on its own GPU with the whole
    training fold (no patient sharding inside a fit) and ONE all-reduce of the score matrix at the end gives every
    rank the full table.  Without an initialised process group it is the plain loop.
    """
    import pandas as pd
    from . import distributed as D
    from . import regularized_optimization as ro
    learn = learn_mhn if _learn is None else _learn
    score_fn = score if _score is None else _score
    dat = np.asarray(dat)
    splits = np.asarray(splits, dtype=np.float64)
    rng = np.random.default_rng(key)
    shuffled = dat[rng.permutation(dat.shape[0])]
    runs = np.zeros((n_folds, splits.shape[0]))
    batch_size = int(np.ceil(dat.shape[0] / n_folds))
    rank, world = ro._rank_world() if parallel_folds else (0, 1)
    prev_shard = ro._OPTIONS["shard"]
    if world > 1:
        ro.configure(device=ro._OPTIONS["device"], dtype=ro._OPTIONS["dtype"], shard=False)
    logging.info("Crossvalidation started")
    try:
        for i, fold in D.fold_jobs(splits.size, n_folds, rank, world):
            n_dat = shuffled.shape[0]
            start = batch_size * fold
            stop = min(batch_size * (fold + 1), n_dat)
            train = np.concatenate((shuffled[:start], shuffled[stop:]))
            th0, dp0, dm0 = indep(train)
            th, dp, dm = learn(th0, dp0, dm0, train, m_p_corr, penal_fun, splits[i], opt_v=False)
            runs[fold, i] = score_fn(th, dp, dm, shuffled[start:stop], m_p_corr)
            logging.info(f"Lambda: {splits[i]} Fold: {fold} Test Score: {runs[fold, i]}")
    finally:
        if world > 1:
            ro.configure(device=ro._OPTIONS["device"], dtype=ro._OPTIONS["dtype"], shard=prev_shard)
    if world > 1:
        runs = D.allreduce_sums(runs.ravel()).reshape(runs.shape)
    return pd.DataFrame(runs, columns=splits, index=np.arange(n_folds))


def create_dat(dat_sim_full, n_em: int, n_nm: int, n_c: int, n_pm: int, n_mo: int, rng) -> np.ndarray:
    """Subsample simulated patients (rows of `simulate_dat`) to a cohort with the given composition and append
    the type column (examples/recall_study.py:32-47): `n_nm` never-metastasised PT-only rows (type 0, all-zero
    rows dropped first), and of the metastasised ones `n_c` paired (3), `n_mo` MT-only (2), `n_pm` PT-only (1).
    `rng`: NumPy Generator (the reference draws the indices with jax.random.choice)."""
    sim = np.asarray(dat_sim_full)
    po = sim[sim[:, -2] == 0]
    po = po[po.sum(axis=1) != 0]
    po = po[rng.choice(po.shape[0], size=n_nm, replace=False)]
    em = sim[sim[:, -2] != 0]
    idx = rng.choice(em.shape[0], size=n_em, replace=False)
    col = lambda rows, t: np.hstack((rows, np.full((rows.shape[0], 1), t, dtype=np.int8)))
    out = np.vstack((col(po, 0), col(em[idx[:n_c]], 3), col(em[idx[n_c:n_c + n_mo]], 2), col(em[idx[n_c + n_mo:]], 1)))
    return out.astype(np.int8)


def categorize(x) -> int:
    """Type of a datapoint from its annotation (Utilityfunctions.py:98-113)."""
    import pandas as pd
    if x["paired"] == 0:
        return {"absent": 0, "present": 1, "isMetastasis": 2}.get(x["metaStatus"], pd.NA if x["metaStatus"] == "unknown" else -1)
    if x["paired"] == 1:
        return 3
    return pd.NA


def load_cohort(events_csv: str, annot_csv: str, muts: list[str] | None = None):
    """Events CSV + annotation CSV -> (dat int8 [n_pat, 2n+3], event names); examples/analysis.py:49-83.

    `muts`: the P./M. event columns to keep (pairs, PT column first); default: every event column.
    """
    import pandas as pd
    annot = pd.read_csv(annot_csv)
    mut = pd.read_csv(events_csv)
    mut.rename(columns={"Unnamed: 0": "patientID"}, inplace=True)
    d = pd.merge(mut, annot.loc[:, ["patientID", "metaStatus"]], on=["patientID", "patientID"])
    if muts is None:
        muts = list(d.columns[1:-4])
    d["type"] = d.apply(categorize, axis=1)
    d["Seeding"] = d["type"].apply(lambda x: pd.NA if pd.isna(x) else 0 if x == 0 else 1)
    d["M.AgeAtSeqRep"] = pd.to_numeric(d["M.AgeAtSeqRep"], errors="coerce")
    d["P.AgeAtSeqRep"] = pd.to_numeric(d["P.AgeAtSeqRep"], errors="coerce")
    d["diag_order"] = d["M.AgeAtSeqRep"] - d["P.AgeAtSeqRep"]
    d["diag_order"] = d["diag_order"].apply(lambda x: pd.NA if pd.isna(x) else 2 if x < 0 else 1 if x > 0 else 0)
    d["diag_order"] = d["diag_order"].astype(pd.Int64Dtype())
    cleaned = d.loc[~pd.isna(d["type"]), muts + ["Seeding", "diag_order", "type"]]
    dat = cleaned.to_numpy(dtype=np.int8, na_value=-99)
    events = [c.split(".")[1] for c in muts[::2]] + ["Seeding"]
    return dat, events


def save_params(path: str, theta, d_p, d_m, events: list[str]):
    """Parameter CSV with rows d_p, d_m, theta (examples/analysis.py:115-119)."""
    import pandas as pd
    tab = np.vstack((np.asarray(d_p).reshape(1, -1), np.asarray(d_m).reshape(1, -1), np.asarray(theta)))
    pd.DataFrame(tab, columns=events).to_csv(path)
