"""The CPU oracle against the golden vectors emitted by the reference's own source
(tests/tools/make_golden.py) and against the first-principles dense oracle."""
import numpy as np
import pytest

from oracle import metmhn_oracle as O
from oracle import dense as D

TOL = dict(rtol=1e-10, atol=1e-12)


def _cases(g):
    return range(int(g["n_cases"]))


def test_primitives(golden):
    g = golden("primitives")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"]
        st, p, x = g[pre + "state"], g[pre + "p"], g[pre + "x"]
        k = int(st.sum())
        for dg in (0, 1):
            for tr in (0, 1):
                np.testing.assert_allclose(O.kronvec(lt, p, st, bool(dg), bool(tr)), g[pre + f"kv_d{dg}_t{tr}"], **TOL)
        np.testing.assert_allclose(O.kron_diag(lt, st, k), g[pre + "kron_diag"], **TOL)
        if st[-1] == 1:
            np.testing.assert_allclose(O.diag_scal_p(dp, st, p), g[pre + "dsp"], **TOL)
            np.testing.assert_allclose(O.diag_scal_m(dm, st, p), g[pre + "dsm"], **TOL)
            for i in range(dp.shape[0]):                       # kronvec.py:605-710
                np.testing.assert_allclose(O.partial_diag_scal_p(dp, st, p, i), g[pre + "pdsp"][i], **TOL)
                np.testing.assert_allclose(O.partial_diag_scal_m(dm, st, p, i), g[pre + "pdsm"][i], **TOL)
            n_prim, n_met = int(st[::2].sum()), int(st[1::2].sum() + 1)
            for name, pf, ns in (("pf", True, n_met), ("mf", False, n_prim)):
                assert np.array_equal(O.obs_states(k, st, pf), g[pre + "obs_" + name])
                assert np.array_equal(O.obs_indices(k, st, pf, ns), g[pre + "idx_" + name])   # bit-exact
            if k >= 2:
                for tr in (0, 1):
                    np.testing.assert_allclose(O.R_i_inv_vec(lt, dp, dm, x, st, k, bool(tr)), g[pre + f"R_t{tr}"], **TOL)
                np.testing.assert_allclose(O.x_partial_Q_y(lt, x, p, st), g[pre + "xQy"], **TOL)
                a, b = O.x_partial_D_y(dm, dp, st, x, p)
                np.testing.assert_allclose(a, g[pre + "xDy_dp"], **TOL)
                np.testing.assert_allclose(b, g[pre + "xDy_dm"], **TOL)


def test_vanilla(golden):
    g = golden("vanilla")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"]
        st, p, x, dr = g[pre + "state"], g[pre + "p"], g[pre + "x"], g[pre + "d_rates"]
        for dg in (0, 1):
            for tr in (0, 1):
                np.testing.assert_allclose(O.v_kronvec(lt, p, st, bool(dg), bool(tr)), g[pre + f"kv_d{dg}_t{tr}"], **TOL)
        np.testing.assert_allclose(O.v_kron_diag(lt, st, np.ones_like(p)), g[pre + "kron_diag"], **TOL)
        for tr in (0, 1):
            np.testing.assert_allclose(O.v_R_inv_vec(lt, x, st, 1.0, bool(tr)), g[pre + f"R1_t{tr}"], **TOL)
            np.testing.assert_allclose(O.v_R_inv_vec(lt, x, st, dr, bool(tr)), g[pre + f"Rd_t{tr}"], **TOL)
        a, b = O.v_x_partial_Q_y(lt, x, p, st)
        np.testing.assert_allclose(a, g[pre + "xQy"], **TOL)
        np.testing.assert_allclose(b, g[pre + "xQy_ddiag"], **TOL)
        a, b, c_ = O.v_gradient(lt, st, g[pre + "p0"])
        np.testing.assert_allclose(a, g[pre + "grad_th"], **TOL)
        np.testing.assert_allclose(b, g[pre + "grad_ddiag"], **TOL)
        np.testing.assert_allclose(c_, g[pre + "grad_pth"], **TOL)
        if st[-1] == 1:
            a, b = O.v_scal_d_pt(dp, dm, st, p)
            np.testing.assert_allclose(a, g[pre + "scal_dp"], **TOL)
            np.testing.assert_allclose(b, g[pre + "scal_dm"], **TOL)
            a, b = O.v_x_partial_D_y(dp, dm, st, x, p)
            np.testing.assert_allclose(a, g[pre + "xDy_dp"], **TOL)
            np.testing.assert_allclose(b, g[pre + "xDy_dm"], **TOL)
            for i in range(dp.shape[0]):                       # vanilla.py:179-187
                a, b = O.v_d_scal_d_pt(dp, dm, st, p, i)
                np.testing.assert_allclose(a, g[pre + "dscal_dp"][i], **TOL)
                np.testing.assert_allclose(b, g[pre + "dscal_dm"][i], **TOL)


def test_patients(golden):
    g = golden("patients")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        for r, row in enumerate(dat):
            lp, is0 = O.patient_lp(lt, dp, dm, row)
            np.testing.assert_allclose(lp, g[pre + "lp_score"][r], **TOL)
            lp2, gth, gdp, gdm, _ = O.patient_grad(lt, dp, dm, row)
            np.testing.assert_allclose(lp2, g[pre + "lp_grad"][r], **TOL)
            np.testing.assert_allclose(gth, g[pre + "d_th"][r], **TOL)
            np.testing.assert_allclose(gdp, g[pre + "d_dp"][r], **TOL)
            np.testing.assert_allclose(gdm, g[pre + "d_dm"][r], **TOL)
            # independent dense oracle on the log-probability
            np.testing.assert_allclose(D.patient_lp(lt, dp, dm, row), lp, rtol=1e-9)


def test_cohorts(golden):
    g = golden("cohorts")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        pm, lam = float(g[pre + "perc_met"]), float(g[pre + "lam"])
        s, gth, gdp, gdm = O.score_and_grad(lt, dp, dm, dat, pm)
        np.testing.assert_allclose(s, g[pre + "score"], **TOL)
        np.testing.assert_allclose(O.score(lt, dp, dm, dat, pm), g[pre + "score_only"], **TOL)
        np.testing.assert_allclose(gth, g[pre + "d_th"], **TOL)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"], **TOL)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"], **TOL)
        params = np.concatenate((lt.flatten(), dp, dm))
        v, gr = O.score_and_grad_reg(params, dat, pm, O.symmetric_penal, lam)
        np.testing.assert_allclose(v, g[pre + "reg_value"], **TOL)
        np.testing.assert_allclose(gr, g[pre + "reg_grad"], **TOL)
        np.testing.assert_allclose(O.score_reg(params, dat, pm, O.symmetric_penal, lam), g[pre + "reg_value_only"], **TOL)
        pen, pen_ = O.symmetric_penal(params, lt.shape[0])
        np.testing.assert_allclose(pen, g[pre + "pen"], **TOL)
        np.testing.assert_allclose(pen_, g[pre + "pen_grad"], **TOL)


def test_survey_anchor_c1(golden):
    """SURVEY.md Appendix C.1 known-answer values."""
    g = golden("cohorts")
    assert abs(float(g["c0_score"]) - (-5.3401612761381285)) < 1e-13
    assert abs(float(g["c0_reg_value"]) - 9.25186325475162) < 1e-12
    assert abs(np.linalg.norm(g["c0_reg_grad"]) - 2.1596327054018287) < 1e-12


def test_gradient_is_fd_of_score():
    """Property the reference's tests/test_gradient.py pins: analytic gradient == forward FD of score."""
    rng = np.random.default_rng(3)
    n = 3
    lt = rng.normal(size=(4, 4)) * 0.6
    dp, dm = np.log([1., 2, 3, 4]), np.log([.5, 1.5, 2.5, 3.5])
    rows = [list(rng.binomial(1, .6, 6)) + [1, o, 3] for o in (0, 1, 2)]
    rows += [[1, 0, 0, 0, 1, 0, 0, -99, 0], [1, 0, 1, 0, 0, 0, 1, -99, 1], [0, 1, 0, 0, 0, 1, 1, -99, 2],
             [0] * 6 + [1, 0, 3]]
    dat = np.array(rows, dtype=np.int8)
    s, gth, gdp, gdm = O.score_and_grad(lt, dp, dm, dat, 0.4)
    h = 1e-6
    for (i, j) in [(0, 0), (1, 2), (3, 1), (2, 3), (3, 3)]:
        e = np.zeros((4, 4)); e[i, j] = h
        fd = (O.score(lt + e, dp, dm, dat, 0.4) - O.score(lt - e, dp, dm, dat, 0.4)) / (2 * h)
        assert abs(fd - gth[i, j]) < 1e-7
    for i in range(4):
        e = np.zeros(4); e[i] = h
        assert abs((O.score(lt, dp + e, dm, dat, .4) - O.score(lt, dp - e, dm, dat, .4)) / (2 * h) - gdp[i]) < 1e-7
        assert abs((O.score(lt, dp, dm + e, dat, .4) - O.score(lt, dp, dm - e, dat, .4)) / (2 * h) - gdm[i]) < 1e-7


def test_optimised_cpu_variant_matches_reference_structure_port():
    """oracle/metmhn_fast.c (gather formulation, substitution solves; second CPU baseline of bench.py) against
    oracle/metmhn_ref.c (reference pass structure) on random paired rows: every order code, sparse / dense /
    empty / full genotypes, n = 3, 5, 8."""
    from oracle import cref
    from metmhn_amd import synthetic
    rng = np.random.default_rng(5)
    for n in (3, 5, 8):
        lt, dp, dm = synthetic.random_params(n, seed=50 + n)
        rows = []
        for _ in range(30):
            bits = (rng.random(2 * n) < rng.choice([0.1, 0.5, 0.9])).astype(np.int8)
            rows.append(np.concatenate((bits, [1, int(rng.choice([0, 1, 2, -99])), 3])))
        rows.append(np.concatenate((np.zeros(2 * n, np.int8), [1, 0, 3])))
        rows.append(np.concatenate((np.ones(2 * n, np.int8), [1, 2, 3])))
        dat = np.array(rows, dtype=np.int8)
        a = cref.patients(lt, dp, dm, dat)
        b = cref.fast_patients(lt, dp, dm, dat)
        for x, y in zip(a, b):
            np.testing.assert_allclose(y, x, rtol=1e-11, atol=1e-13)
    with pytest.raises(ValueError):
        cref.fast_patients(lt, dp, dm, np.array([[1, 0] * n + [1, -99, 1]], dtype=np.int8))


def test_c_ports_pinned_to_reference_golden(golden):
    """oracle/metmhn_ref.c (reference pass structure) and oracle/metmhn_fast.c (gather formulation) - the checkers of
    the random sweeps and of the full-size GPU tests - directly against the reference's own per-patient outputs
    (tests/golden/patients.npz: every datapoint type / order, k = 1 and k = 2 spaces included)."""
    from oracle import cref
    g = golden("patients")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        lp, gth, gdp, gdm = cref.patients(lt, dp, dm, dat)
        np.testing.assert_allclose(lp, g[pre + "lp_grad"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gth, g[pre + "d_th"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"], rtol=1e-10, atol=1e-12)
        paired = dat[:, -1] == 3
        lp, gth, gdp, gdm = cref.fast_patients(lt, dp, dm, dat[paired])
        np.testing.assert_allclose(lp, g[pre + "lp_grad"][paired], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(gth, g[pre + "d_th"][paired], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"][paired], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"][paired], rtol=1e-10, atol=1e-12)


def _cohort_from_patients(lp, gt, gp, gm, dat, perc_met):
    """regularized_optimization.py:121-130, :256-266 on per-patient results."""
    em = dat[:, -1] != 0
    n_em = float(dat[:, -3].sum())
    n_nm = dat.shape[0] - n_em
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em * n_nm != 0 else 1.0
    nf = w * n_em + n_nm
    return ((w * lp[em].sum() + lp[~em].sum()) / nf, (w * gt[em].sum(0) + gt[~em].sum(0)) / nf,
            (w * gp[em].sum(0) + gp[~em].sum(0)) / nf, w * gm[em].sum(0) / nf)


def test_luad_reduced_anchor_cpu(golden):
    """BASELINE configs[0] / SURVEY Appendix C.2 on the host: the C port on the full 4 852-row LUAD-reduced cohort against
    the reference's score and gradient at indep(dat) and at the published fit (tests/tools/make_golden_luad.py)."""
    from oracle import cref
    gi, gf = golden("luad_indep"), golden("luad_fit")
    dat = gi["dat"]
    assert dat.shape == (4852, 43) and list(np.bincount(dat[:, -1])) == [595, 1677, 2127, 453]
    for g, pre in ((gi, "indep_"), (gf, "fit_")):
        lt, dp, dm = g[pre + "theta"], g[pre + "dp"], g[pre + "dm"]
        s, G, a, b = _cohort_from_patients(*cref.patients(lt, dp, dm, dat), dat, float(g["perc_met"]))
        np.testing.assert_allclose(s, g[pre + "score"], rtol=1e-12)
        np.testing.assert_allclose(G, g[pre + "d_th"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(a, g[pre + "d_dp"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(b, g[pre + "d_dm"], rtol=1e-9, atol=1e-13)
        params = np.concatenate((lt.flatten(), dp, dm))
        pen, pen_ = O.symmetric_penal(params, lt.shape[0])
        np.testing.assert_allclose(-s + float(g["lam"]) * pen, g[pre + "reg_value"], rtol=1e-12)
    np.testing.assert_allclose(float(gi["indep_score"]), -8.43382859658627, rtol=1e-13)
    np.testing.assert_allclose(np.linalg.norm(gi["indep_d_th"]), 1.66178987553837, rtol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(gi["indep_d_dp"]), 0.4435122531351038, rtol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(gi["indep_d_dm"]), 0.03620795819315152, rtol=1e-12)


def test_c_ports_pinned_to_reference_at_window_shapes(golden):
    """reference-generated vectors in the regime the headline kernels run in.  tests/golden/large.npz
    (tests/tools/make_golden.py large) holds three paired rows, n = 12, (kP, kM) = (10, 5), (11, 6), (6, 11), orders 0 / 1 / 2,
    k = 16 / 18 / 18, evaluated by the reference's own _g_coupled_0/1/2 (likelihood.py:623-731, through
    regularized_optimization.score_and_grad on one-row cohorts; 15 - 70 minutes per row under the NumPy stand-in).  Both C ports -
    the checkers of every full-size GPU test - must reproduce them to 1e-10: the chain window kernels -> metmhn_fast.c -> reference
    is direct at these shapes."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "large.npz")):
        pytest.skip("large.npz not generated (tests/tools/make_golden.py large)")
    from oracle import cref
    g = golden("large")
    lt, dp, dm, dat = g["log_theta"], g["log_d_p"], g["log_d_m"], g["dat"]
    n = (dat.shape[1] - 3) // 2
    kP, kM = dat[:, 0:2 * n:2].sum(1), dat[:, 1:2 * n:2].sum(1)
    assert [(int(a), int(b)) for a, b in zip(kP, kM)] == [(10, 5), (11, 6), (6, 11)] and list(dat[:, -2]) == [0, 1, 2]
    for name, fn in (("metmhn_ref.c", cref.patients), ("metmhn_fast.c", cref.fast_patients)):
        lp, gth, gdp, gdm = fn(lt, dp, dm, dat)
        np.testing.assert_allclose(lp, g["lp"], rtol=1e-10, err_msg=name)
        np.testing.assert_allclose(gth, g["d_th"], rtol=1e-10, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(gdp, g["d_dp"], rtol=1e-10, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(gdm, g["d_dm"], rtol=1e-10, atol=1e-12, err_msg=name)


def test_luad28_fixture_cpu(golden):
    """The 28-event LUAD cohort (examples/analysis.py:55; tests/tools/make_golden_luad.py luad28): the fixture's values come
    from oracle/metmhn_ref.c - here the second C port (metmhn_fast.c, gather formulation) reproduces the stored per-patient
    log-probabilities of all 453 paired rows (k up to 21) at both parameter points, the NumPy restatement (reference pass
    structure, oracle/metmhn_oracle.py) those of the unpaired rows and of the small paired ones it finishes in seconds, and the
    cohort formula the stored score."""
    from oracle import cref
    g = golden("luad28")
    dat = g["dat"]
    assert dat.shape == (4852, 59) and list(np.bincount(dat[:, -1])) == [595, 1677, 2127, 453]
    paired = np.flatnonzero(dat[:, -1] == 3)
    assert np.array_equal(paired, g["paired_rows"])
    for pre in ("indep_", "fit_"):
        lt, dp, dm = g[pre + "theta"], g[pre + "dp"], g[pre + "dm"]
        lpf, gf, af, bf = cref.fast_patients(lt, dp, dm, dat[paired])
        np.testing.assert_allclose(lpf, g[pre + "lp"][paired], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(np.sqrt((gf ** 2).sum((1, 2)) + (af ** 2).sum(1) + (bf ** 2).sum(1)), g[pre + "g_fast_norm"], rtol=1e-10)
    # NumPy restatement on a sample: every 40th row with at most 9 active slots
    k = dat[:, :-2].sum(1)
    rows = [r for r in range(0, dat.shape[0], 40) if k[r] <= 9]
    lt, dp, dm = g["fit_theta"], g["fit_dp"], g["fit_dm"]
    for r in rows:
        one = dat[r:r + 1]
        s = float(O.score(lt, dp, dm, one, 0.5))
        np.testing.assert_allclose(s, g["fit_lp"][r], rtol=1e-10, atol=1e-12, err_msg=f"row {r}")
    em = dat[:, -1] != 0
    pm = float(g["perc_met"])
    n_em = float(dat[:, -3].sum()); n_nm = dat.shape[0] - n_em
    w = pm * n_nm / ((1 - pm) * n_em)
    for pre in ("indep_", "fit_"):
        lp = g[pre + "lp"]
        np.testing.assert_allclose((w * lp[em].sum() + lp[~em].sum()) / (w * n_em + n_nm), g[pre + "score"], rtol=1e-13)


def test_window_schedule_model():
    """oracle/wschedule.py: the scalar model of k_wsolve's schedule (lanes skewed by whole register windows, waves by
    blocks, external bits from the thread's own earlier output) solves the Kronecker-sum system exactly, forward and
    transposed, and every value a thread picks up is the block it expects (the model asserts the identities)."""
    from oracle import wschedule
    for cfg in ((2, 1, 1, 1, 1, 0), (2, 2, 1, 2, 1, 1), (3, 2, 2, 1, 2, 0), (3, 1, 1, 2, 0, 2)):
        for tr in (False, True):
            err, _ = wschedule.emulate(*cfg, tr, seed=3)
            assert err < 1e-13, (cfg, tr, err)
