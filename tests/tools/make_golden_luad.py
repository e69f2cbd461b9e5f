#!/usr/bin/env python3
"""Golden vectors of the LUAD-reduced cohort (BASELINE.json configs[0]).

Runs ONLY in the build container (needs /root/reference, which never travels); takes ~10-20
minutes per leg because the reference's source runs eagerly under the NumPy stand-in:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/tools/jax_standin:/root/reference \
    python tests/tools/make_golden_luad.py [indep|fit|luad28]

Writes tests/golden/luad_<leg>.npz (data only: inputs + the reference's outputs):
  indep: the 4 852 x 43 int8 `dat` built as examples/analysis.py:49-72 does (20 '(M)' mutations of
         examples/recall_study.py:58-64), `indep(dat)` and score_and_grad / score_and_grad_reg there
         (perc_met 0.2, lambda 1e-3: the settings of the reference's own LUAD fit,
         examples/data_analysis.ipynb cell 14).
  fit:   the parameters the reference published for that fit (results/luad/luad_g14_20muts.csv: rows d_p,
         d_m, theta; examples/analysis.py:115-119) and the reference's score_and_grad_reg at them.
  luad28: the cohort examples/analysis.py really fits - ALL 28 events (20 mutations + 8 copy-number events,
         `muts = list(dat.columns[1:-4])`, examples/analysis.py:55): 4 852 x 59 int8 `dat`, the reference's `indep(dat)`
         and the parameters it published for that fit (results/luad/luad_g14_cv_20muts_8cnvs.csv).  Paired rows reach
         k = 21 there: the reference's source under the stand-in would take days, so the VALUES of this leg come from the
         C restatements (oracle/metmhn_ref.c for every row, oracle/metmhn_fast.c for the paired rows), which are pinned to
         the reference on the other fixtures (tests/test_oracle_golden.py) - this file pins the cohort, the start point and
         the published parameters, and keeps the GPU tests off a 2-minute CPU evaluation.
"""
import os
import sys
import time
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "golden")

import jax.numpy as jnp  # noqa: E402  (the stand-in)
import metmhn.regularized_optimization as ro  # noqa: E402
import metmhn.Utilityfunctions as ru  # noqa: E402

J = jnp.array
A = np.asarray
PERC_MET, LAM = 0.2, 1e-3


def luad_dat(all_events=False):
    import pandas as pd
    base = "/root/reference/data/luad/"
    annot = pd.read_csv(base + "G14_LUAD_sampleSelection.csv")
    mut = pd.read_csv(base + "G14_LUAD_Events.csv")
    mut.rename(columns={"Unnamed: 0": "patientID"}, inplace=True)
    d = pd.merge(mut, annot.loc[:, ["patientID", "metaStatus"]], on=["patientID", "patientID"])
    genes = ["TP53", "KRAS", "EGFR", "STK11", "KEAP1", "RBM10", "SMARCA4", "ATM", "NF1", "PTPRD", "PTPRT",
             "ARID1A", "BRAF", "PIK3CA", "EPHA3", "FAT1", "SETD2", "RB1", "MET", "KMT2C"]
    muts = [f"{t}.{g} (M)" for g in genes for t in ("P", "M")]
    if all_events:                                            # examples/analysis.py:55
        muts = list(d.columns[1:-4])
        genes = [c.split(".", 1)[1] for c in muts[::2]]
    d["type"] = d.apply(ru.categorize, axis=1)
    d["Seeding"] = d["type"].apply(lambda x: pd.NA if pd.isna(x) else 0 if x == 0 else 1)
    d["M.AgeAtSeqRep"] = pd.to_numeric(d["M.AgeAtSeqRep"], errors="coerce")
    d["P.AgeAtSeqRep"] = pd.to_numeric(d["P.AgeAtSeqRep"], errors="coerce")
    d["diag_order"] = d["M.AgeAtSeqRep"] - d["P.AgeAtSeqRep"]
    d["diag_order"] = d["diag_order"].apply(lambda x: pd.NA if pd.isna(x) else 2 if x < 0 else 1 if x > 0 else 0)
    d["diag_order"] = d["diag_order"].astype(pd.Int64Dtype())
    cleaned = d.loc[~pd.isna(d["type"]), muts + ["Seeding", "diag_order", "type"]]
    return cleaned.to_numpy(dtype=np.int8, na_value=-99), genes


def evaluate(out, pre, lt, dp, dm, dat):
    params = np.concatenate((A(lt).flatten(), A(dp), A(dm)))
    t0 = time.time()
    s, g, a, b = ro.score_and_grad(J(lt), J(dp), J(dm), J(dat), PERC_MET)
    out[pre + "score"] = np.float64(np.asarray(s).reshape(-1)[0])
    out[pre + "d_th"], out[pre + "d_dp"], out[pre + "d_dm"] = A(g), A(a), A(b)
    pen, pen_ = ro.symmetric_penal(params, A(lt).shape[0])
    out[pre + "reg_value"] = np.float64(-out[pre + "score"] + LAM * float(pen))
    out[pre + "reg_grad"] = -np.concatenate((A(g).flatten(), A(a), A(b))) + LAM * A(pen_)
    print(pre, "score", out[pre + "score"], "reg", out[pre + "reg_value"], f"{time.time() - t0:.0f} s", flush=True)


def cohort_sums(lp, g, a, b, dat, perc_met):
    """regularized_optimization.py:256-266 on per-patient rows"""
    em = dat[:, -1] != 0
    n_em, n_nm = float(dat[:, -3].sum()), float(dat.shape[0] - dat[:, -3].sum())
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em > 0 and n_nm > 0 else 1.0
    den = w * n_em + n_nm
    f = lambda x: (w * x[em].sum(0) + x[~em].sum(0)) / den
    return f(lp), f(g), f(a), w * b[em].sum(0) / den


def luad28():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import cref
    import pandas as pd
    dat, events = luad_dat(all_events=True)
    print("LUAD, 28 events:", dat.shape, np.bincount(dat[:, -1]), flush=True)
    out = {"perc_met": np.float64(PERC_MET), "lam": np.float64(LAM), "dat": dat}
    th, dp, dm = ru.indep(J(dat))
    out["indep_theta"], out["indep_dp"], out["indep_dm"] = A(th), A(dp), A(dm)
    df = pd.read_csv("/root/reference/results/luad/luad_g14_cv_20muts_8cnvs.csv", index_col=0)
    arr = df.to_numpy(dtype=np.float64)                      # rows: d_p, d_m, theta (examples/analysis.py:115-119)
    assert arr.shape == (31, 29) and list(df.columns[:-1]) == events
    out["fit_dp"], out["fit_dm"], out["fit_theta"] = arr[0], arr[1], arr[2:]
    paired = np.flatnonzero(dat[:, -1] == 3)
    out["paired_rows"] = paired
    for pre in ("indep_", "fit_"):
        lt, a_, b_ = out[pre + "theta"], out[pre + "dp"], out[pre + "dm"]
        t0 = time.time()
        lp, g, a, b = cref.patients(lt, a_, b_, dat)                          # metmhn_ref.c: every row
        print(pre, f"metmhn_ref.c {time.time() - t0:.0f} s", flush=True)
        t0 = time.time()
        lpf, gf, af, bf = cref.fast_patients(lt, a_, b_, dat[paired])         # metmhn_fast.c: the paired rows
        print(pre, f"metmhn_fast.c {time.time() - t0:.0f} s; the two ports differ by",
              np.abs(lpf - lp[paired]).max(), np.abs(gf - g[paired]).max(), flush=True)
        s, G, ga, gb = cohort_sums(lp, g, a, b, dat, PERC_MET)
        out[pre + "score"], out[pre + "d_th"], out[pre + "d_dp"], out[pre + "d_dm"] = np.float64(s), G, ga, gb
        out[pre + "lp"] = lp
        out[pre + "lp_fast"] = lpf
        out[pre + "g_fast_norm"] = np.sqrt((gf ** 2).sum((1, 2)) + (af ** 2).sum(1) + (bf ** 2).sum(1))
        params = np.concatenate((lt.flatten(), a_, b_))
        pen, pen_ = ro.symmetric_penal(params, lt.shape[0])
        out[pre + "reg_value"] = np.float64(-s + LAM * float(pen))
        out[pre + "reg_grad"] = -np.concatenate((G.flatten(), ga, gb)) + LAM * A(pen_)
        print(pre, "score", s, "reg", out[pre + "reg_value"], flush=True)
    np.savez_compressed(os.path.join(OUT, "luad28.npz"), **out)


def main():
    leg = sys.argv[1] if len(sys.argv) > 1 else "indep"
    if leg == "luad28":
        return luad28()
    dat, genes = luad_dat()
    print("LUAD-reduced:", dat.shape, np.bincount(dat[:, -1]), flush=True)
    out = {"perc_met": np.float64(PERC_MET), "lam": np.float64(LAM)}
    if leg == "indep":
        out["dat"] = dat
        th, dp, dm = ru.indep(J(dat))
        out["indep_theta"], out["indep_dp"], out["indep_dm"] = A(th), A(dp), A(dm)
        evaluate(out, "indep_", A(th), A(dp), A(dm), dat)
    else:
        import pandas as pd
        df = pd.read_csv("/root/reference/results/luad/luad_g14_20muts.csv", index_col=0)
        arr = df.to_numpy(dtype=np.float64)                  # rows: d_p, d_m, theta (examples/analysis.py:115-119)
        assert arr.shape == (23, 21) and [c.split(" ")[0] for c in df.columns[:-1]] == genes
        out["fit_dp"], out["fit_dm"], out["fit_theta"] = arr[0], arr[1], arr[2:]
        evaluate(out, "fit_", arr[2:], arr[0], arr[1], dat)
    np.savez_compressed(os.path.join(OUT, f"luad_{leg}.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir("/root/reference/metmhn"):
        sys.exit("needs /root/reference (build container only)")
    main()
