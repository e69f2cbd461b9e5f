#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own source.

Runs ONLY in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 \
    PYTHONPATH=tests/tools/jax_standin:/root/reference \
    python tests/tools/make_golden.py            # primitives, vanilla, patients, cohorts
    python tests/tools/make_golden.py large      # large.npz: three paired rows with k = 16 - 18 (minutes per row)

The reference (cbg-ethz/metMHN @ 2024_08_07) is pure Python on JAX; jax is not
installed here, so its source is executed eagerly under the NumPy stand-in in
tests/tools/jax_standin (identical elementwise arithmetic, fp64; reductions may
differ from XLA in summation order by ~1e-16 relative).  Inputs are seeded;
outputs are whatever the reference functions return.  The files written are
data only (inputs + expected outputs).
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "golden")

import jax.numpy as jnp  # noqa: E402  (the stand-in)
import metmhn.regularized_optimization as ro  # noqa: E402
import metmhn.jx.kronvec as rk  # noqa: E402
import metmhn.jx.likelihood as rl  # noqa: E402
import metmhn.jx.vanilla as rv  # noqa: E402
import metmhn.Utilityfunctions as ru  # noqa: E402

J = jnp.array
A = np.asarray


def rand_params(rng, n, scale=1.0):
    lt = np.diag(rng.normal(size=n + 1))
    off = rng.random((n + 1, n + 1)) < 0.5
    np.fill_diagonal(off, False)
    lt = lt + off * rng.normal(size=(n + 1, n + 1)) * scale
    return lt, rng.normal(size=n + 1) * 0.5, rng.normal(size=n + 1) * 0.5


def primitives():
    rng = np.random.default_rng(20240807)
    out = {}
    c = 0
    specs = [(2, None), (2, None), (3, None), (3, None), (3, None), (4, None), (4, None), (4, None),
             (5, None), (5, None), (3, "seed_only"), (3, "noseed"), (4, "noseed"), (3, "all"),
             (4, "pt_only"), (4, "mt_only"), (5, "all"), (6, None)]
    for n, kind in specs:
        lt, dp, dm = rand_params(rng, n)
        st = rng.integers(0, 2, size=2 * n + 1).astype(np.int8)
        st[-1] = 1
        if kind == "seed_only":
            st[:-1] = 0
        elif kind == "noseed":
            st[-1] = 0
            if st.sum() == 0:
                st[0] = 1
        elif kind == "all":
            st[:] = 1
        elif kind == "pt_only":
            st[1:-1:2] = 0
        elif kind == "mt_only":
            st[0:-1:2] = 0
        k = int(st.sum())
        p = rng.normal(size=2 ** k)
        x = rng.normal(size=2 ** k)
        pre = f"c{c}_"
        out[pre + "log_theta"], out[pre + "log_d_p"], out[pre + "log_d_m"] = lt, dp, dm
        out[pre + "state"], out[pre + "p"], out[pre + "x"] = st, p, x
        for dg in (0, 1):
            for tr in (0, 1):
                out[pre + f"kv_d{dg}_t{tr}"] = A(rk.kronvec(J(lt), J(p), J(st), bool(dg), bool(tr)))
        out[pre + "kron_diag"] = A(rk.kron_diag(J(lt), J(st), k))
        if st[-1] == 1:
            out[pre + "dsp"] = A(rk.diag_scal_p(J(dp), J(st), J(p)))
            out[pre + "dsm"] = A(rk.diag_scal_m(J(dm), J(st), J(p)))
            # (dD/dlog d[i]) p for every event index (kronvec.py:632-644, :704-710)
            out[pre + "pdsp"] = np.array([A(rk.partial_diag_scal_p(J(dp), J(st), J(p), i)) for i in range(n + 1)])
            out[pre + "pdsm"] = np.array([A(rk.partial_diag_scal_m(J(dm), J(st), J(p), i)) for i in range(n + 1)])
            n_prim = int(st[::2].sum())
            n_met = int(st[1::2].sum() + 1)
            for name, pf, ns in (("pf", True, n_met), ("mf", False, n_prim)):
                mask = A(rk.obs_states(k, J(st), pf))
                out[pre + "obs_" + name] = mask
                out[pre + "idx_" + name] = A(jnp.where(J(mask) == 1., size=2 ** (ns - 1))[0]).astype(np.int64)
            if k >= 2:
                for tr in (0, 1):
                    out[pre + f"R_t{tr}"] = A(rl.R_i_inv_vec(J(lt), J(dp), J(dm), J(x), J(st), k, bool(tr)))
                out[pre + "xQy"] = A(rl.x_partial_Q_y(J(lt), J(x), J(p), J(st)))
                a, b = rl.x_partial_D_y(J(dm), J(dp), J(st), J(x), J(p))
                out[pre + "xDy_dp"], out[pre + "xDy_dm"] = A(a), A(b)
        c += 1
    out["n_cases"] = np.int64(c)
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), **out)
    print("primitives:", c, "cases")


def vanilla():
    rng = np.random.default_rng(77)
    out = {}
    c = 0
    for n in (2, 3, 3, 4, 4, 5, 6):
        lt, dp, dm = rand_params(rng, n)
        st = rng.integers(0, 2, size=n + 1).astype(np.int8)
        if c % 2 == 0:
            st[-1] = 1
        if st.sum() == 0:
            st[0] = 1
        k = int(st.sum())
        p = rng.normal(size=2 ** k)
        x = rng.normal(size=2 ** k)
        dr = np.exp(rng.normal(size=2 ** k) * 0.3)
        pre = f"c{c}_"
        out[pre + "log_theta"], out[pre + "log_d_p"], out[pre + "log_d_m"] = lt, dp, dm
        out[pre + "state"], out[pre + "p"], out[pre + "x"], out[pre + "d_rates"] = st, p, x, dr
        for dg in (0, 1):
            for tr in (0, 1):
                out[pre + f"kv_d{dg}_t{tr}"] = A(rv.kronvec(J(lt), J(p), J(st), bool(dg), bool(tr)))
        out[pre + "kron_diag"] = A(rv.kron_diag(J(lt), J(st), J(np.ones(2 ** k))))
        for tr in (0, 1):
            out[pre + f"R1_t{tr}"] = A(rv.R_inv_vec(J(lt), J(x), J(st), 1, bool(tr)))
            out[pre + f"Rd_t{tr}"] = A(rv.R_inv_vec(J(lt), J(x), J(st), J(dr), bool(tr)))
        g, dd = rv.x_partial_Q_y(J(lt), J(x), J(p), J(st))
        out[pre + "xQy"], out[pre + "xQy_ddiag"] = A(g), A(dd)
        p0 = np.abs(p)
        g, dd, pth = rv.gradient(J(lt), J(st), J(p0))
        out[pre + "p0"], out[pre + "grad_th"], out[pre + "grad_ddiag"], out[pre + "grad_pth"] = p0, A(g), A(dd), A(pth)
        if st[-1] == 1:
            a, b = rv.scal_d_pt(J(dp), J(dm), J(st), J(p))
            out[pre + "scal_dp"], out[pre + "scal_dm"] = A(a), A(b)
            a, b = rv.x_partial_D_y(J(dp), J(dm), J(st), J(x), J(p))
            out[pre + "xDy_dp"], out[pre + "xDy_dm"] = A(a), A(b)
            ds = [rv.d_scal_d_pt(J(dp), J(dm), J(st), J(p), i) for i in range(n + 1)]      # vanilla.py:179-187
            out[pre + "dscal_dp"] = np.array([A(t[0]) for t in ds])
            out[pre + "dscal_dm"] = np.array([A(t[1]) for t in ds])
        c += 1
    out["n_cases"] = np.int64(c)
    np.savez_compressed(os.path.join(OUT, "vanilla.npz"), **out)
    print("vanilla:", c, "cases")


def patients():
    """One-row cohorts of every type / order, incl. the k = 1 paired states (one_event.py)."""
    rng = np.random.default_rng(42)
    out = {}
    c = 0
    for n in (3, 4, 5):
        lt, dp, dm = rand_params(rng, n, 0.8)
        rows = []
        for typ in (0, 0, 1, 1, 2, 2):
            bits = rng.binomial(1, 0.6, 2 * n)
            if typ in (0, 1):
                bits[1::2] = 0
            else:
                bits[0::2] = 0
            rows.append(list(bits) + [0 if typ == 0 else 1, -99, typ])
        rows.append([0] * (2 * n) + [0, -99, 0])                 # all-zero never-metastasised PT
        rows.append([0] * (2 * n) + [1, -99, 1])                 # type 1 with no mutation
        rows.append([0] * (2 * n) + [1, -99, 2])                 # type 2 with no mutation
        for order in (0, 1, 2, -99):
            for _ in range(3):
                rows.append(list(rng.binomial(1, 0.6, 2 * n)) + [1, order, 3])
            rows.append([0] * (2 * n) + [1, order, 3])           # k = 1
            lone = [0] * (2 * n)
            lone[int(rng.integers(0, 2 * n))] = 1
            rows.append(lone + [1, order, 3])                    # k = 2
        dat = np.array(rows, dtype=np.int8)
        lps, gths, gdps, gdms, sc = [], [], [], [], []
        for r in dat:
            one = J(r.reshape(1, -1))
            s, g, a, b = ro.score_and_grad(J(lt), J(dp), J(dm), one, 0.5)
            lps.append(float(np.asarray(s).reshape(-1)[0]))
            gths.append(A(g))
            gdps.append(A(a))
            gdms.append(A(b))
            sc.append(float(np.asarray(ro.score(J(lt), J(dp), J(dm), one, 0.5)).reshape(-1)[0]))
        pre = f"c{c}_"
        out[pre + "log_theta"], out[pre + "log_d_p"], out[pre + "log_d_m"] = lt, dp, dm
        out[pre + "dat"] = dat
        out[pre + "lp_grad"], out[pre + "lp_score"] = np.array(lps), np.array(sc)
        out[pre + "d_th"], out[pre + "d_dp"], out[pre + "d_dm"] = np.array(gths), np.array(gdps), np.array(gdms)
        c += 1
    out["n_cases"] = np.int64(c)
    np.savez_compressed(os.path.join(OUT, "patients.npz"), **out)
    print("patients:", c, "parameter sets")


def _cohort_entry(out, pre, lt, dp, dm, dat, perc_met, lam):
    params = np.concatenate((lt.flatten(), dp, dm))
    out[pre + "log_theta"], out[pre + "log_d_p"], out[pre + "log_d_m"] = lt, dp, dm
    out[pre + "dat"], out[pre + "perc_met"], out[pre + "lam"] = dat, np.float64(perc_met), np.float64(lam)
    s, g, a, b = ro.score_and_grad(J(lt), J(dp), J(dm), J(dat), perc_met)
    out[pre + "score"] = np.float64(np.asarray(s).reshape(-1)[0])
    out[pre + "score_only"] = np.float64(np.asarray(ro.score(J(lt), J(dp), J(dm), J(dat), perc_met)).reshape(-1)[0])
    out[pre + "d_th"], out[pre + "d_dp"], out[pre + "d_dm"] = A(g), A(a), A(b)
    v, gr = ro.score_and_grad_reg(params, J(dat), perc_met, ro.symmetric_penal, lam)
    out[pre + "reg_value"], out[pre + "reg_grad"] = np.float64(v), A(gr)
    out[pre + "reg_value_only"] = np.float64(ro.score_reg(params, J(dat), perc_met, ro.symmetric_penal, lam))
    pen, pen_ = ro.symmetric_penal(params, lt.shape[0])
    out[pre + "pen"], out[pre + "pen_grad"] = np.float64(pen), A(pen_)
    th_i, dp_i, dm_i = ru.indep(J(dat))                      # Utilityfunctions.py:157-183
    out[pre + "indep_theta"], out[pre + "indep_dp"], out[pre + "indep_dm"] = A(th_i), A(dp_i), A(dm_i)


def cohorts():
    out = {}
    # C.1 of SURVEY.md: RNG-free anchor
    lt = np.arange(16).reshape(4, 4) * 0.1 - 0.8
    dp = np.log(np.array([1., 2, 3, 4]))
    dm = np.log(np.array([.5, 1.5, 2.5, 3.5]))
    dat = np.array([[1, 1, 0, 1, 1, 0, 1, 1, 3], [1, 0, 1, 1, 0, 1, 1, 2, 3], [0, 1, 1, 1, 1, 1, 1, 0, 3],
                    [0, 0, 0, 0, 0, 0, 1, 0, 3], [1, 0, 1, 0, 0, 0, 0, -99, 0], [0, 0, 0, 0, 0, 0, 0, -99, 0],
                    [1, 0, 0, 0, 1, 0, 1, -99, 1], [0, 1, 0, 1, 0, 0, 1, -99, 2]], dtype=np.int8)
    _cohort_entry(out, "c0_", lt, dp, dm, dat, 0.8, 0.4)

    # random mixed cohort, n = 5
    rng = np.random.default_rng(5)
    n = 5
    lt, dp, dm = rand_params(rng, n, 0.7)
    rows = []
    for r in range(40):
        typ = int(rng.choice([0, 1, 2, 3], p=[0.15, 0.2, 0.25, 0.4]))
        bits = rng.binomial(1, 0.45, 2 * n)
        if typ in (0, 1):
            bits[1::2] = 0
            rows.append(list(bits) + [typ, -99, typ])
        elif typ == 2:
            bits[0::2] = 0
            rows.append(list(bits) + [1, -99, 2])
        else:
            rows.append(list(bits) + [1, int(rng.choice([0, 1, 2, -99], p=[.3, .3, .3, .1])), 3])
    _cohort_entry(out, "c1_", lt, dp, dm, np.array(rows, dtype=np.int8), 0.35, 1e-2)

    # cohort without never-metastasised patients (w = 1 branch, regularized_optimization.py:126-127)
    only_em = np.array([r for r in rows if r[-1] != 0], dtype=np.int8)[:12]
    _cohort_entry(out, "c2_", lt, dp, dm, only_em, 0.5, 1e-3)

    # LUAD-reduced sub-sample (data/luad/*.csv, labelled as examples/analysis.py:49-72,
    # 20 '(M)' mutations as examples/recall_study.py:58-64); small-k rows in file order
    import pandas as pd
    base = "/root/reference/data/luad/"
    annot = pd.read_csv(base + "G14_LUAD_sampleSelection.csv")
    mut = pd.read_csv(base + "G14_LUAD_Events.csv")
    mut.rename(columns={"Unnamed: 0": "patientID"}, inplace=True)
    d = pd.merge(mut, annot.loc[:, ["patientID", "metaStatus"]], on=["patientID", "patientID"])
    genes = ["TP53", "KRAS", "EGFR", "STK11", "KEAP1", "RBM10", "SMARCA4", "ATM", "NF1", "PTPRD", "PTPRT",
             "ARID1A", "BRAF", "PIK3CA", "EPHA3", "FAT1", "SETD2", "RB1", "MET", "KMT2C"]
    muts = [f"{t}.{g} (M)" for g in genes for t in ("P", "M")]
    d["type"] = d.apply(ru.categorize, axis=1)
    d["Seeding"] = d["type"].apply(lambda x: pd.NA if pd.isna(x) else 0 if x == 0 else 1)
    d["M.AgeAtSeqRep"] = pd.to_numeric(d["M.AgeAtSeqRep"], errors="coerce")
    d["P.AgeAtSeqRep"] = pd.to_numeric(d["P.AgeAtSeqRep"], errors="coerce")
    d["diag_order"] = d["M.AgeAtSeqRep"] - d["P.AgeAtSeqRep"]
    d["diag_order"] = d["diag_order"].apply(lambda x: pd.NA if pd.isna(x) else 2 if x < 0 else 1 if x > 0 else 0)
    d["diag_order"] = d["diag_order"].astype(pd.Int64Dtype())
    cleaned = d.loc[~pd.isna(d["type"]), muts + ["Seeding", "diag_order", "type"]]
    full = cleaned.to_numpy(dtype=np.int8, na_value=-99)
    print("LUAD-reduced:", full.shape, np.bincount(full[:, -1]))
    pick = []
    for typ, cnt in ((0, 8), (1, 8), (2, 8), (3, 24)):
        rows_t = [r for r in full if r[-1] == typ and r[:-2].sum() <= 9]
        pick += rows_t[:cnt]
    sub = np.array(pick, dtype=np.int8)
    th0, dp0, dm0 = ru.indep(J(sub))
    rng = np.random.default_rng(14)
    th0 = np.maximum(np.asarray(th0), -8.0)          # indep() puts -1e10 on never-seen events
    off = rng.normal(size=th0.shape) * 0.3 * (rng.random(th0.shape) < 0.3)
    np.fill_diagonal(off, 0.0)
    _cohort_entry(out, "c3_", th0 + off, np.asarray(dp0) + rng.normal(size=21) * 0.2,
                  np.asarray(dm0) + rng.normal(size=21) * 0.2, sub, 0.2, 1e-3)
    out["n_cases"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "cohorts.npz"), **out)
    print("cohorts: 4")


def large():
    """Three paired rows in the regime the window kernels run in (>= 10 bits in one class, >= 4 in the other):
    n = 12, (kP, kM) = (10, 5), (11, 6), (6, 11), orders 0 / 1 / 2 -> k = 16, 18, 18, through the reference's
    _g_coupled_0/1/2 (likelihood.py:623-731) as regularized_optimization.py:227-254 calls them."""
    import time
    rng = np.random.default_rng(1618)
    n = 12
    lt, dp, dm = rand_params(rng, n, 0.6)
    rows = []
    for (kP, kM), order in (((10, 5), 0), ((11, 6), 1), ((6, 11), 2)):
        r = np.zeros(2 * n + 3, dtype=np.int8)
        r[2 * rng.choice(n, kP, replace=False)] = 1
        r[2 * rng.choice(n, kM, replace=False) + 1] = 1
        r[2 * n], r[2 * n + 1], r[2 * n + 2] = 1, order, 3
        rows.append(r)
    dat = np.array(rows, dtype=np.int8)
    out = {"log_theta": lt, "log_d_p": dp, "log_d_m": dm, "dat": dat}
    lps, gths, gdps, gdms = [], [], [], []
    for r in dat:
        t0 = time.time()
        s, g, a, b = ro.score_and_grad(J(lt), J(dp), J(dm), J(r.reshape(1, -1)), 0.5)
        lps.append(float(np.asarray(s).reshape(-1)[0]))
        gths.append(A(g)); gdps.append(A(a)); gdms.append(A(b))
        print("large: k =", int(r[:2 * n + 1].sum()), "lp", lps[-1], f"{time.time() - t0:.0f} s", flush=True)
    out["lp"], out["d_th"], out["d_dp"], out["d_dm"] = np.array(lps), np.array(gths), np.array(gdps), np.array(gdms)
    np.savez_compressed(os.path.join(OUT, "large.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir("/root/reference/metmhn"):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        large()
        sys.exit(0)
    primitives()
    vanilla()
    patients()
    cohorts()
