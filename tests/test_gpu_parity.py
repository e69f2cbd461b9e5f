"""Parity of the HIP path (through the C ABI) with the golden vectors of the reference and
with the CPU oracle on seeded inputs.  fp64 tolerance: 1e-6 relative (BASELINE.json
north_star); indices bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-6          # north_star tolerance for fp64 likelihoods / gradients
TIGHT = dict(rtol=1e-9, atol=1e-11)   # what the fp64 kernels actually deliver on small cases


def _cases(g):
    return range(int(g["n_cases"]))


@pytest.fixture(scope="module")
def engines():
    from metmhn_amd import Engine
    cache = {}

    def get(n):
        if n not in cache:
            cache[n] = Engine(n)
        return cache[n]
    yield get
    for e in cache.values():
        e.close()


def test_joint_primitives_golden(golden, engines):
    g = golden("primitives")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"]
        st, p, x = g[pre + "state"], g[pre + "p"], g[pre + "x"]
        n = (st.shape[0] - 1) // 2
        k = int(st.sum())
        e = engines(n)
        for dg in (0, 1):
            for tr in (0, 1):
                np.testing.assert_allclose(e.kronvec(lt, p, st, bool(dg), bool(tr)), g[pre + f"kv_d{dg}_t{tr}"],
                                           err_msg=f"case {c} kronvec d{dg} t{tr}", **TIGHT)
        np.testing.assert_allclose(e.kron_diag(lt, st), g[pre + "kron_diag"], err_msg=f"case {c} kron_diag", **TIGHT)
        if st[-1] == 1:
            np.testing.assert_allclose(e.diag_scal(dp, st, p, 0), g[pre + "dsp"], **TIGHT)
            np.testing.assert_allclose(e.diag_scal(dm, st, p, 1), g[pre + "dsm"], **TIGHT)
            for name, pf in (("pf", True), ("mf", False)):
                assert np.array_equal(e.obs_indices(st, pf), g[pre + "idx_" + name]), f"case {c} obs {name}"
            if k >= 2:
                for tr in (0, 1):
                    np.testing.assert_allclose(e.resolvent(lt, dp, dm, x, st, bool(tr)), g[pre + f"R_t{tr}"],
                                               err_msg=f"case {c} resolvent t{tr}", **TIGHT)
                np.testing.assert_allclose(e.x_partial_Q_y(lt, x, p, st), g[pre + "xQy"],
                                           err_msg=f"case {c} xQy", **TIGHT)
                a, b = e.x_partial_D_y(dp, dm, st, x, p)
                np.testing.assert_allclose(a, g[pre + "xDy_dp"], err_msg=f"case {c} xDy dp", **TIGHT)
                np.testing.assert_allclose(b, g[pre + "xDy_dm"], err_msg=f"case {c} xDy dm", **TIGHT)


def test_single_primitives_golden(golden, engines):
    g = golden("vanilla")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, st, p, x, dr = g[pre + "log_theta"], g[pre + "state"], g[pre + "p"], g[pre + "x"], g[pre + "d_rates"]
        e = engines(st.shape[0] - 1)
        for dg in (0, 1):
            for tr in (0, 1):
                np.testing.assert_allclose(e.v_kronvec(lt, p, st, bool(dg), bool(tr)), g[pre + f"kv_d{dg}_t{tr}"],
                                           err_msg=f"case {c} v_kronvec d{dg} t{tr}", **TIGHT)
        for tr in (0, 1):
            np.testing.assert_allclose(e.v_resolvent(lt, x, st, None, bool(tr)), g[pre + f"R1_t{tr}"], **TIGHT)
            np.testing.assert_allclose(e.v_resolvent(lt, x, st, dr, bool(tr)), g[pre + f"Rd_t{tr}"], **TIGHT)
        a, b = e.v_x_partial_Q_y(lt, x, p, st)
        np.testing.assert_allclose(a, g[pre + "xQy"], err_msg=f"case {c} v_xQy", **TIGHT)
        np.testing.assert_allclose(b, g[pre + "xQy_ddiag"], **TIGHT)


def test_reference_module_surface(golden):
    """The metmhn.jx-style mirrors take the reference's argument orders."""
    from metmhn_amd.jx import kronvec as K, likelihood as L, vanilla as V
    g = golden("primitives")
    pre = "c5_"
    lt, dp, dm, st, p, x = (g[pre + s] for s in ("log_theta", "log_d_p", "log_d_m", "state", "p", "x"))
    k = int(st.sum())
    np.testing.assert_allclose(K.kronvec(lt, p, st, diag=False, transpose=True), g[pre + "kv_d0_t1"], **TIGHT)
    np.testing.assert_allclose(K.kron_diag(lt, st, k), g[pre + "kron_diag"], **TIGHT)
    np.testing.assert_allclose(K.diag_scal_p(dp, st, p), g[pre + "dsp"], **TIGHT)
    assert np.array_equal(K.obs_states(k, st, True), g[pre + "obs_pf"])
    np.testing.assert_allclose(L.R_i_inv_vec(lt, dp, dm, x, st, k, transpose=True), g[pre + "R_t1"], **TIGHT)
    a, b = L.x_partial_D_y(dm, dp, st, x, p)           # reference order: (log_d_m, log_d_p, ...)
    np.testing.assert_allclose(a, g[pre + "xDy_dp"], **TIGHT)
    np.testing.assert_allclose(b, g[pre + "xDy_dm"], **TIGHT)
    gv = golden("vanilla")
    np.testing.assert_allclose(V.R_inv_vec(gv["c3_log_theta"], gv["c3_x"], gv["c3_state"], 1, True), gv["c3_R1_t1"],
                               **TIGHT)


def test_patients_golden(golden, engines):
    g = golden("patients")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        e = engines((dat.shape[1] - 3) // 2)
        e.set_cohort(dat)
        lp, gth, gdp, gdm = e.patient_grads(lt, dp, dm)
        for r in range(dat.shape[0]):
            msg = f"set {c} row {r} {dat[r]}"
            np.testing.assert_allclose(lp[r], g[pre + "lp_grad"][r], err_msg=msg, **TIGHT)
            np.testing.assert_allclose(gth[r], g[pre + "d_th"][r], err_msg=msg, **TIGHT)
            np.testing.assert_allclose(gdp[r], g[pre + "d_dp"][r], err_msg=msg, **TIGHT)
            np.testing.assert_allclose(gdm[r], g[pre + "d_dm"][r], err_msg=msg, **TIGHT)
        lp_only = e.patient_grads(lt, dp, dm, with_grad=False)
        np.testing.assert_allclose(lp_only, g[pre + "lp_score"], **TIGHT)


def test_cohorts_golden(golden):
    import metmhn_amd.regularized_optimization as ro
    g = golden("cohorts")
    for c in _cases(g):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        pm, lam = float(g[pre + "perc_met"]), float(g[pre + "lam"])
        s, gth, gdp, gdm = ro.score_and_grad(lt, dp, dm, dat, pm)
        np.testing.assert_allclose(s, g[pre + "score"], rtol=RTOL, err_msg=f"cohort {c}")
        np.testing.assert_allclose(s, g[pre + "score"], **TIGHT)
        np.testing.assert_allclose(ro.score(lt, dp, dm, dat, pm), g[pre + "score_only"], **TIGHT)
        np.testing.assert_allclose(gth, g[pre + "d_th"], **TIGHT)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"], **TIGHT)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"], **TIGHT)
        params = np.concatenate((lt.flatten(), dp, dm))
        v, gr = ro.score_and_grad_reg(params, dat, pm, ro.symmetric_penal, lam)
        np.testing.assert_allclose(v, g[pre + "reg_value"], **TIGHT)
        np.testing.assert_allclose(gr, g[pre + "reg_grad"], **TIGHT)
        np.testing.assert_allclose(ro.score_reg(params, dat, pm, ro.symmetric_penal, lam), g[pre + "reg_value_only"],
                                   **TIGHT)


def test_multi_tile_against_oracle(engines):
    """k = 14 > tile bits: neighbour tiles, several workgroups per vector."""
    from oracle import metmhn_oracle as O
    from metmhn_amd import synthetic
    n = 7
    lt, dp, dm = synthetic.random_params(n, seed=11)
    st = np.ones(2 * n + 1, dtype=np.int8)
    st[3] = 0
    k = int(st.sum())
    rng = np.random.default_rng(0)
    p, x = rng.normal(size=2 ** k), rng.normal(size=2 ** k)
    e = engines(n)
    for tr in (False, True):
        np.testing.assert_allclose(e.kronvec(lt, p, st, True, tr), O.kronvec(lt, p, st, True, tr), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(e.resolvent(lt, dp, dm, x, st, False), O.R_i_inv_vec(lt, dp, dm, x, st, k), rtol=1e-8,
                               atol=1e-10)
    np.testing.assert_allclose(e.x_partial_Q_y(lt, x, p, st), O.x_partial_Q_y(lt, x, p, st), rtol=1e-8, atol=1e-8)
    a, b = e.x_partial_D_y(dp, dm, st, x, p)
    a2, b2 = O.x_partial_D_y(dm, dp, st, x, p)
    np.testing.assert_allclose(a, a2, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(b, b2, rtol=1e-8, atol=1e-8)


def test_synthetic_cohort_against_oracle():
    """BASELINE config-2 shaped rows (n = k = 12 ... scaled to n = 8 so the oracle finishes in seconds)."""
    from oracle import metmhn_oracle as O
    from metmhn_amd import synthetic
    import metmhn_amd.regularized_optimization as ro
    n = 8
    lt, dp, dm = synthetic.random_params(n)
    dat = np.vstack((synthetic.full_k_cohort(n, 6), synthetic.mixed_cohort(n, 30, seed=3)))
    s, gth, gdp, gdm = ro.score_and_grad(lt, dp, dm, dat, 0.5)
    s2, g2, a2, b2 = O.score_and_grad(lt, dp, dm, dat, 0.5)
    np.testing.assert_allclose(s, s2, rtol=RTOL)
    np.testing.assert_allclose(gth, g2, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(gdp, a2, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(gdm, b2, rtol=RTOL, atol=1e-9)


def test_cohort_edited_in_place_is_evaluated_afresh():
    """The reference's objective is a pure function of `dat`; ours keeps the cohort's layout in HBM keyed by the array's
    identity.  The guard against in-place edits runs next to the GPU (regularized_optimization._result): an edited
    array gives the edited cohort's value, at once and on the following call."""
    from metmhn_amd import synthetic
    import metmhn_amd.regularized_optimization as ro
    n = 6
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.mixed_cohort(n, 200, seed=11)
    edited = dat.copy()
    edited[:, -3] = 0                                       # every row: no metastasis observed ...
    edited[:, 1:2 * n:2] = 0
    edited[:, -1] = 0                                       # ... type 0
    want_before, want_after = ro.score(lt, dp, dm, dat.copy(), 0.3), ro.score(lt, dp, dm, edited, 0.3)
    assert abs(want_before - want_after) > 1e-3
    work = dat.copy()
    assert ro.score(lt, dp, dm, work, 0.3) == want_before
    work[:] = edited                                        # same array object, new contents
    assert ro.score(lt, dp, dm, work, 0.3) == want_after
    s, *_ = ro.score_and_grad(lt, dp, dm, work, 0.3)
    assert abs(s - want_after) <= 1e-12 * abs(want_after)


def test_random_patients_against_c_oracle(monkeypatch):
    """Randomised sweep: n = 1..9 events, every type / order code (incl. -99 and invalid order values on paired
    rows), dense and sparse genotypes, empty and full rows; per-patient log-prob and gradients from both kernel
    schedules (per tile, per patient) against the C restatement of the reference pass structure."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    cref.load()
    rng = np.random.default_rng(20240807)
    for n in (1, 2, 3, 5, 7, 9):
        lt, dp, dm = synthetic.random_params(n, seed=100 + n)
        rows = []
        for _ in range(60):
            dens = rng.choice([0.0, 0.15, 0.5, 0.85, 1.0])
            bits = (rng.random(2 * n) < dens).astype(np.int8)
            typ = int(rng.integers(0, 4))
            if typ == 0:
                bits[1::2] = 0
                rows.append(np.concatenate((bits, [0, -99, 0])))
            elif typ == 1:
                bits[1::2] = 0
                rows.append(np.concatenate((bits, [1, -99, 1])))
            elif typ == 2:
                bits[0::2] = 0
                rows.append(np.concatenate((bits, [1, -99, 2])))
            else:
                rows.append(np.concatenate((bits, [1, int(rng.choice([0, 1, 2, -99, 3])), 3])))
        dat = np.array(rows, dtype=np.int8)
        lp, g, a, b = cref.patients(lt, dp, dm, dat, with_grad=True)
        for pmin in ("1", "1000000"):
            monkeypatch.setenv("MMHN_PSOLVE_MIN", pmin)
            e = Engine(n)
            e.set_cohort(dat)
            r = e.patient_grads(lt, dp, dm)
            e.close()
            np.testing.assert_allclose(r[0], lp, rtol=1e-9, atol=1e-12, err_msg=f"n={n} lp pmin={pmin}")
            np.testing.assert_allclose(r[1], g, rtol=1e-8, atol=1e-10, err_msg=f"n={n} d_theta pmin={pmin}")
            np.testing.assert_allclose(r[2], a, rtol=1e-8, atol=1e-10, err_msg=f"n={n} d_dp pmin={pmin}")
            np.testing.assert_allclose(r[3], b, rtol=1e-8, atol=1e-10, err_msg=f"n={n} d_dm pmin={pmin}")


def test_full_size_patients_against_optimised_cpu_variant(monkeypatch):
    """BASELINE configs[2] at full size: n = 20, k = 20 (2^20-state vectors) paired patients, per-patient
    log-prob and gradients from both kernel schedules against oracle/metmhn_fast.c (an independent CPU
    implementation of the same mathematics, itself pinned to the reference-structure port on small cases)."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 20
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 64, seed=2000 + n)                      # (metmhn_fast.c: ~21 ms per patient and core)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    for pmin in ("1", "1000000"):
        monkeypatch.setenv("MMHN_PSOLVE_MIN", pmin)
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], lp, rtol=1e-10)
        np.testing.assert_allclose(r[1], g, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10)
    # 300 patients at n = 16 with k = 10..16 active slots: PT / MT splits from 1 : 15 to 15 : 1 (class bits above the
    # class-aligned tile, partial tiles, lone-heavy and pair-heavy layouts)
    n = 16
    lt, dp, dm = synthetic.random_params(n, seed=77)
    dat = np.vstack([synthetic.full_k_cohort(n, 50, k=kk, seed=900 + kk) for kk in (10, 12, 13, 14, 15, 16)])
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    for pmin in ("1", "1000000"):
        monkeypatch.setenv("MMHN_PSOLVE_MIN", pmin)
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], lp, rtol=1e-10)
        np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10)


def test_small_batches_match_one_batch(engines):
    """A tiny workspace limit forces many batches: same result."""
    from metmhn_amd import Engine, synthetic
    n = 8
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.mixed_cohort(n, 60, seed=5)
    e1 = engines(n)
    e1.set_cohort(dat)
    a = e1.cohort_sums(lt, dp, dm)
    e2 = Engine(n, workspace_bytes=1 << 20)
    e2.set_cohort(dat)
    b = e2.cohort_sums(lt, dp, dm)
    e2.close()
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def test_empty_and_error_paths(engines):
    from metmhn_amd import Engine
    e = engines(3)
    with pytest.raises(ValueError):
        e.set_cohort(np.zeros((2, 5), dtype=np.int8))
    with pytest.raises(RuntimeError):
        e.set_cohort(np.array([[0, 0, 0, 0, 0, 0, 0, 0, 7]], dtype=np.int8))       # bad type
    with pytest.raises(RuntimeError):
        e.set_cohort(np.array([[1, 1, 0, 0, 0, 0, 0, 1, 3]], dtype=np.int8))       # paired row without seeding
    with pytest.raises(RuntimeError):
        Engine(3, device=99)
    # round 3 entry points: shapes are checked before anything reaches the device; an evaluation begun in two halves must be
    # collected with the matching _end, and nothing else may start in between
    st = np.array([1, 1, 0, 1, 1, 0, 1], dtype=np.int8)                            # k = 5
    lt = np.zeros((4, 4))
    with pytest.raises(ValueError):
        e.kronvec_batched(lt, np.ones((2, 16)), st)
    with pytest.raises(ValueError):
        e.jacobi_step_batched(lt, np.zeros(4), np.zeros(4), np.ones((2, 32)), np.ones((3, 32)), st)
    y = e.kronvec_batched(lt, np.ones((2, 32)), st, diag=False)
    np.testing.assert_allclose(y[0], e.kronvec(lt, np.ones(32), st, diag=False), rtol=1e-13)
    e.set_cohort(np.array([[1, 1, 0, 1, 1, 0, 1, 1, 3]], dtype=np.int8))
    e.cohort_wsums_begin(lt, np.zeros(4), np.zeros(4), 1.0)
    with pytest.raises(RuntimeError):
        e.cohort_sums_end()                                                        # begun as wsums
    ws = e.cohort_wsums_end()
    assert ws.shape == (1 + 16 + 8,) and np.isfinite(ws).all()


def test_substitution_solver_matches_jacobi_iteration(monkeypatch):
    """Default tile-level substitution vs the reference's k+1 Jacobi sweeps (MMHN_SOLVER=jacobi)."""
    from metmhn_amd import Engine, synthetic
    n = 9
    lt, dp, dm = synthetic.random_params(n)
    dat = np.vstack((synthetic.full_k_cohort(n, 8, k=9), synthetic.full_k_cohort(n, 4, k=15, seed=5),
                     synthetic.mixed_cohort(n, 30, seed=9)))
    e1 = Engine(n)
    e1.set_cohort(dat)
    a = e1.cohort_sums(lt, dp, dm)
    e1.close()
    monkeypatch.setenv("MMHN_SOLVER", "jacobi")
    e2 = Engine(n)
    e2.set_cohort(dat)
    b = e2.cohort_sums(lt, dp, dm)
    e2.close()
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)


def test_learn_mhn_matches_oracle_optimizer():
    """Drop-in under SciPy's L-BFGS-B (regularized_optimization.py:301-334): same optimum as the
    same optimizer run on the CPU oracle's objective."""
    import scipy.optimize as opt
    from oracle import metmhn_oracle as O
    from metmhn_amd import synthetic
    import metmhn_amd.regularized_optimization as ro
    n = 4
    dat = synthetic.mixed_cohort(n, 60, seed=21, p_event=0.35)
    N = n + 1
    th0 = np.diag(np.full(N, -1.0))
    dp0, dm0 = np.zeros(N), np.zeros(N)
    th, dp, dm = ro.learn_mhn(th0, dp0, dm0, dat, 0.3, ro.symmetric_penal, 0.05, opt_iter=200, opt_ftol=1e-10,
                              opt_v=False)
    x0 = np.concatenate((th0.flatten(), dp0, dm0))
    ref = opt.minimize(fun=O.score_and_grad_reg, jac=True, x0=x0, method="L-BFGS-B",
                       args=(dat, 0.3, O.symmetric_penal, 0.05), options={"maxiter": 200, "ftol": 1e-10})
    got = np.concatenate((th.flatten(), dp, dm))
    f_got = float(O.score_reg(got, dat, 0.3, O.symmetric_penal, 0.05))
    assert abs(f_got - float(ref.fun)) < 1e-7 * max(1.0, abs(float(ref.fun)))
    np.testing.assert_allclose(got, ref.x, atol=5e-4)


def test_fp32_engine_close_to_fp64():
    """dtype="f32" engine (BASELINE config 5 dtype) on a small cohort: looser tolerance by construction."""
    from oracle import metmhn_oracle as O
    from metmhn_amd import Engine, synthetic, distributed as D
    n = 6
    lt, dp, dm = synthetic.random_params(n)
    dat = np.vstack((synthetic.full_k_cohort(n, 6), synthetic.mixed_cohort(n, 40, seed=4)))
    e = Engine(n, dtype="f32")
    e.set_cohort(dat)
    s, g, a, b = D.combine_sums(e.cohort_sums(lt, dp, dm), n + 1, 0.5)
    e.close()
    s2, g2, a2, b2 = O.score_and_grad(lt, dp, dm, dat, 0.5)
    np.testing.assert_allclose(s, s2, rtol=2e-5)
    np.testing.assert_allclose(g, g2, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(a, a2, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(b, b2, rtol=2e-3, atol=2e-5)


def test_fp32_engine_tracks_fp64_engine_on_large_spaces():
    """No CPU oracle finishes k = 18 in seconds: compare the fp32 engine (e_0 pre-scaled by 2^60) with the
    fp64 engine on the same paired patients; log-probs to 1e-4, every gradient component within 2e-3 relative +
    2e-5 absolute."""
    from metmhn_amd import Engine, synthetic
    n = 18
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 6)
    res = []
    for dt in ("f64", "f32"):
        e = Engine(n, dtype=dt)
        e.set_cohort(dat)
        res.append(e.patient_grads(lt, dp, dm))
        e.close()
    (lp64, g64, a64, b64), (lp32, g32, a32, b32) = res
    assert np.all(np.isfinite(lp32)) and np.all(np.isfinite(g32))
    np.testing.assert_allclose(lp32, lp64, rtol=1e-4)
    for nm, x32, x64 in (("d_theta", g32, g64), ("d_dp", a32, a64), ("d_dm", b32, b64)):
        err = np.abs(x32 - x64)
        assert (err <= 2e-3 * np.abs(x64) + 2e-5).all(), f"{nm}: max abs err {err.max():.3e}"


def test_cross_val_workflow():
    """Utilityfunctions.cross_val (:186-231) on the engine: shape, finiteness, and each cell equals the
    held-out oracle score of the parameters learn_mhn returns for that fold."""
    from oracle import metmhn_oracle as O
    from metmhn_amd import synthetic
    import metmhn_amd.regularized_optimization as ro
    import metmhn_amd.Utilityfunctions as U
    n = 3
    dat = synthetic.mixed_cohort(n, 40, seed=11, p_event=0.4)
    lams = np.array([1e-2, 1e-1])
    res = U.cross_val(dat, ro.symmetric_penal, lams, 2, 0.3, key=5)
    assert res.shape == (2, 2) and np.isfinite(res.to_numpy()).all()
    shuffled = dat[np.random.default_rng(5).permutation(dat.shape[0])]
    train, test = shuffled[20:], shuffled[:20]
    th0, dp0, dm0 = U.indep(train)
    th, dp, dm = ro.learn_mhn(th0, dp0, dm0, train, 0.3, ro.symmetric_penal, lams[0], opt_v=False)
    np.testing.assert_allclose(res.iloc[0, 0], O.score(th, dp, dm, test, 0.3), rtol=1e-8)


def _row(n, pt, mt, order):
    r = np.zeros(2 * n + 3, dtype=np.int8)
    for j in pt:
        r[2 * j] = 1
    for j in mt:
        r[2 * j + 1] = 1
    r[2 * n], r[2 * n + 1], r[2 * n + 2] = 1, order, 3
    return r


def test_awkward_layouts_substitution_vs_jacobi(monkeypatch):
    """Multi-tile joint spaces (k = 13..19) with skewed layouts: only PT bits, only MT bits, all events
    paired (many PT == MT states, pair straddling the 2^12 tile boundary), lone-heavy, mixed.  The default
    engine (pruned tile lists, class-table diagonal, on-the-fly adjoint right-hand side) must agree with the
    Jacobi path (every tile, vector diagonal, dense right-hand side) and, for the smallest, with the oracle."""
    from oracle import metmhn_oracle as O
    from metmhn_amd import Engine, synthetic
    n = 12
    lt, dp, dm = synthetic.random_params(n, seed=5)
    rows = [
        _row(n, range(12), [], 1), _row(n, range(12), [], 0),                  # k = 13, PT only
        _row(n, [], range(12), 2), _row(n, [], range(12), 0),                  # k = 13, MT only
        _row(n, range(8), range(8), 0), _row(n, range(9), range(9), 1),        # k = 17 / 19, all paired
        _row(n, range(6), range(6, 12), 2), _row(n, range(6), range(6, 12), 0),  # k = 13, all lone
        _row(n, [0, 1, 2, 3, 4, 5, 6], [5, 6, 7, 8, 9, 10, 11], -99),          # k = 15, pair at bits 10..13
        _row(n, [0, 2, 4, 5, 6, 7, 8, 9, 11], [1, 3, 5, 6, 10], 0),            # k = 15 mixed
        _row(n, range(1), range(12), 1), _row(n, range(11), [11], 2),          # very skewed class sizes
    ]
    dat = np.array(rows, dtype=np.int8)
    e1 = Engine(n)
    e1.set_cohort(dat)
    a = e1.patient_grads(lt, dp, dm)
    e1.close()
    monkeypatch.setenv("MMHN_SOLVER", "jacobi")
    e2 = Engine(n)
    e2.set_cohort(dat)
    b = e2.patient_grads(lt, dp, dm)
    e2.close()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11)
    lp, g, gp, gm, _ = O.patient_grad(lt, dp, dm, dat[6])                      # k = 13 oracle cross-check
    np.testing.assert_allclose(a[0][6], lp, rtol=1e-10)
    np.testing.assert_allclose(a[1][6], g, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[2][6], gp, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[3][6], gm, rtol=1e-8, atol=1e-10)


def test_per_patient_kernels_match_tile_kernels(monkeypatch):
    """One-workgroup-per-patient kernels (k_psolve, k_pclass: class-aligned tiles, register accumulators, class
    bits above the tile, > 15 class bits left to k_class_marg) against the per-tile kernels on k = 14..20
    spaces with skewed PT / MT splits, and against the C oracle on a k = 15 space with 12 PT bits."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 20
    lt, dp, dm = synthetic.random_params(n, seed=11)
    rows = [
        _row(n, range(3), range(4, 20), 0),                     # kP = 3, kM = 16: class marginals fall back
        _row(n, range(16), [17, 18, 19], 1),                    # kP = 16
        _row(n, range(13), range(10, 16), 2),                   # kP = 13, kM = 6, three pairs
        _row(n, list(range(0, 20, 2)) + [1], range(1, 17, 2), 0),   # kP = 11, kM = 8
        _row(n, range(10), range(5, 14), 1),                    # kP = 10, kM = 9
        _row(n, [0, 19], range(6, 19), 2),                      # kP = 2, kM = 13
        _row(n, range(12), [12, 13], 0),                        # k = 15, kP = 12 (two class bits above the tile)
        _row(n, [], range(14), -99),                            # k = 15, MT only
        _row(n, range(7), range(7), 1),                         # k = 15, all paired
    ]
    dat = np.array(rows, dtype=np.int8)
    res = []
    for pmin in ("1", "1000000"):
        monkeypatch.setenv("MMHN_PSOLVE_MIN", pmin)
        # the per-patient kernels do not clear the solution buffers: NaN-fill them to show that no state of a
        # skipped (dead) tile ever reaches arithmetic
        monkeypatch.setenv("MMHN_POISON", "1" if pmin == "1" else "0")
        e = Engine(n)
        e.set_cohort(dat)
        res.append(e.patient_grads(lt, dp, dm))
        e.close()
    monkeypatch.setenv("MMHN_POISON", "0")
    for x, y in zip(*res):
        assert np.isfinite(x).all()
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11)
    cref.load()
    lp, g, gp, gm = cref.patients(lt, dp, dm, dat[6:7], with_grad=True)
    a = res[0]
    np.testing.assert_allclose(a[0][6], lp[0], rtol=1e-10)
    np.testing.assert_allclose(a[1][6], g[0], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[2][6], gp[0], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[3][6], gm[0], rtol=1e-8, atol=1e-10)


def test_per_patient_reference_entry_points(golden):
    """ssr._g_coupled_* / _lp_* / _grad_*_obs and mhn.gradient mirrors (reference argument orders)."""
    from metmhn_amd.jx import likelihood as L, vanilla as V
    g = golden("patients")
    lt, dp, dm, dat = g["c0_log_theta"], g["c0_log_d_p"], g["c0_log_d_m"], g["c0_dat"]
    n = (dat.shape[1] - 3) // 2
    seen = set()
    for r, row in enumerate(dat):
        typ, order = int(row[-1]), int(row[-2])
        key = (typ, order if typ == 3 else 0)
        if key in seen or (typ == 0 and row[:-2].sum() == 0):
            continue
        seen.add(key)
        st = row[:2 * n + 1]
        if typ == 3:
            fn_g = {0: L._g_coupled_0, 1: L._g_coupled_1}.get(order, L._g_coupled_2)
            fn_l = {0: L._lp_coupled_0, 1: L._lp_coupled_1}.get(order, L._lp_coupled_2)
            n_prim, n_met = int(st[::2].sum()), int(st[1::2].sum() + 1)
            lp, gth, gdp, gdm = fn_g(lt, dp, dm, st, n_prim, n_met)
            np.testing.assert_allclose(fn_l(lt, dp, dm, st, n_prim, n_met), g["c0_lp_score"][r], **TIGHT)
            np.testing.assert_allclose(gdm, g["c0_d_dm"][r], **TIGHT)
        elif typ == 2:
            sm = np.append(st[1::2], 1)
            lp, gth, gdp, gdm = L._grad_met_obs(lt, dp, dm, sm, int(sm.sum()))
            np.testing.assert_allclose(L._lp_met_obs(lt, dp, dm, sm, int(sm.sum())), g["c0_lp_score"][r], **TIGHT)
            np.testing.assert_allclose(gdm, g["c0_d_dm"][r], **TIGHT)
        else:
            sp = st[0::2]
            lp, gth, gdp = L._grad_prim_obs(lt, dp, sp, int(sp.sum()))
            np.testing.assert_allclose(L._lp_prim_obs(lt, dp, sp, int(sp.sum())), g["c0_lp_score"][r], **TIGHT)
        np.testing.assert_allclose(lp, g["c0_lp_grad"][r], **TIGHT)
        np.testing.assert_allclose(gth, g["c0_d_th"][r], **TIGHT)
        np.testing.assert_allclose(gdp, g["c0_d_dp"][r], **TIGHT)
    assert len(seen) >= 7
    gv = golden("vanilla")
    d_th, d_diag, pth = V.gradient(gv["c4_log_theta"], gv["c4_state"], gv["c4_p0"])
    np.testing.assert_allclose(d_th, gv["c4_grad_th"], **TIGHT)
    np.testing.assert_allclose(d_diag, gv["c4_grad_ddiag"], **TIGHT)
    np.testing.assert_allclose(pth, gv["c4_grad_pth"], **TIGHT)


# ---- round 2: BASELINE configurations that had no parity test, new entry points, sharded engine ---------------------
import os as _os

GOLDEN = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p

def test_config1_n12_1000_patients_both_schedules(monkeypatch):
    """BASELINE configs[1] at full size: n = 12, 1 000 paired patients with k = 12 (one 2^12 tile each), fp64; every
    patient's log-prob and gradient from both kernel schedules against oracle/metmhn_fast.c."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 12
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 1000, seed=2012)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    for pmin in ("1", "1000000"):
        monkeypatch.setenv("MMHN_PSOLVE_MIN", pmin)
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], lp, rtol=1e-9, err_msg=f"lp pmin={pmin}")
        np.testing.assert_allclose(r[1], g, rtol=1e-9, atol=1e-11, err_msg=f"d_theta pmin={pmin}")
        np.testing.assert_allclose(r[2], a, rtol=1e-9, atol=1e-11, err_msg=f"d_dp pmin={pmin}")
        np.testing.assert_allclose(r[3], b, rtol=1e-9, atol=1e-11, err_msg=f"d_dm pmin={pmin}")


def test_config4_n25_fp32_against_fp64_cpu():
    """BASELINE configs[4] dtype and size: n = k = 25 (2^25-state vectors, 128 MiB fp32 each), engine dtype f32, against
    oracle/metmhn_fast.c in fp64.  fp32 bar (necessarily looser than 1e-6): log-prob 1e-4 relative, EVERY
    gradient component within 2e-3 relative + 2e-5 absolute (the worst component is printed).  Also one k = 22 patient:
    fp32 engine against the fp64 engine at the same bar."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 25
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 3, seed=2025)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    e = Engine(n, dtype="f32")
    e.set_cohort(dat)
    r = e.patient_grads(lt, dp, dm)
    e.close()
    assert np.all(np.isfinite(r[0])) and np.all(np.isfinite(r[1]))
    np.testing.assert_allclose(r[0], lp, rtol=1e-4)
    for x32, x64, nm in ((r[1], g, "d_theta"), (r[2], a, "d_dp"), (r[3], b, "d_dm")):
        err, tol = _fp32_report(f"n=25 {nm}", x32, x64)
        assert (err <= tol).all(), nm
    n = 22
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 1, seed=2022)
    res = []
    for dt in ("f64", "f32"):
        e = Engine(n, dtype=dt)
        e.set_cohort(dat)
        res.append(e.patient_grads(lt, dp, dm))
        e.close()
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=1e-4)
    for nm, x32, x64 in zip(("d_theta", "d_dp", "d_dm"), res[1][1:], res[0][1:]):
        err, tol = _fp32_report(f"k=22 {nm}", x32, x64)
        assert (err <= tol).all(), nm


def test_partial_diag_scal_golden(golden):
    """kronvec.partial_diag_scal_p / _m (kronvec.py:605-710) for every event index."""
    from metmhn_amd.jx import kronvec as K
    g = golden("primitives")
    seen = 0
    for c in _cases(g):
        pre = f"c{c}_"
        if pre + "pdsp" not in g:
            continue
        st, p, dp, dm = g[pre + "state"], g[pre + "p"], g[pre + "log_d_p"], g[pre + "log_d_m"]
        for i in range(dp.shape[0]):
            np.testing.assert_allclose(K.partial_diag_scal_p(dp, st, p, i), g[pre + "pdsp"][i], **TIGHT)
            np.testing.assert_allclose(K.partial_diag_scal_m(dm, st, p, i), g[pre + "pdsm"][i], **TIGHT)
        seen += 1
    assert seen >= 10


def test_vanilla_observation_rate_primitives_golden(golden):
    """vanilla.kron_diag, scal_d_pt, d_scal_d_pt, x_partial_D_y (vanilla.py:115-260) through their own entry points."""
    from metmhn_amd.jx import vanilla as V
    g = golden("vanilla")
    seen = 0
    for c in _cases(g):
        pre = f"c{c}_"
        lt, st, p, x = g[pre + "log_theta"], g[pre + "state"], g[pre + "p"], g[pre + "x"]
        np.testing.assert_allclose(V.kron_diag(lt, st, np.ones_like(p)), g[pre + "kron_diag"], **TIGHT)
        np.testing.assert_allclose(V.kron_diag(lt, st, p), g[pre + "kron_diag"] * p, **TIGHT)
        if pre + "scal_dp" not in g:
            continue
        dp, dm = g[pre + "log_d_p"], g[pre + "log_d_m"]
        a, b = V.scal_d_pt(dp, dm, st, p)
        np.testing.assert_allclose(a, g[pre + "scal_dp"], **TIGHT)
        np.testing.assert_allclose(b, g[pre + "scal_dm"], **TIGHT)
        a, b = V.x_partial_D_y(dp, dm, st, x, p)
        np.testing.assert_allclose(a, g[pre + "xDy_dp"], **TIGHT)
        np.testing.assert_allclose(b, g[pre + "xDy_dm"], **TIGHT)
        if pre + "dscal_dp" in g:
            for i in range(dp.shape[0]):
                a, b = V.d_scal_d_pt(dp, dm, st, p, i)
                np.testing.assert_allclose(a, g[pre + "dscal_dp"][i], **TIGHT)
                np.testing.assert_allclose(b, g[pre + "dscal_dm"][i], **TIGHT)
        seen += 1
    assert seen >= 3


def test_in_library_rccl_allreduce_single_rank(monkeypatch, golden):
    """MMHN_FORCE_ALLREDUCE=1 with a 1-rank nccl group: Engine.cohort_sums -> k_pack_sums -> ncclAllReduce on the
    engine's stream (mmhn_comm_init) -> combine_sums, against the reference's cohort golden."""
    import torch
    import torch.distributed as dist
    import metmhn_amd.regularized_optimization as ro
    monkeypatch.setenv("MMHN_FORCE_ALLREDUCE", "1")
    monkeypatch.setenv("MMHN_STRICT_COMM", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ro.configure(device=0)
        g = golden("cohorts")
        for c in (0, 1):
            pre = f"c{c}_"
            dat = g[pre + "dat"]
            s, gt, gp, gm = ro.score_and_grad(g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], dat, float(g[pre + "perc_met"]))
            eng = ro._engine_for(dat)
            assert eng._sharded and eng._device_comm
            np.testing.assert_allclose(s, g[pre + "score"], rtol=1e-9)
            np.testing.assert_allclose(gt, g[pre + "d_th"], **TIGHT)
            np.testing.assert_allclose(gp, g[pre + "d_dp"], **TIGHT)
            np.testing.assert_allclose(gm, g[pre + "d_dm"], **TIGHT)
    finally:
        ro.configure()
        dist.destroy_process_group()


def _shard_worker(rank, world, port, q):
    """One rank of the sharded-engine test: real Engine on cuda:0 over its LPT shard; gloo carries the all-reduce
    (two ranks cannot share one GPU under RCCL)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import metmhn_amd.regularized_optimization as ro
    ro.configure(device=0)
    g = np.load(os.path.join(root, "tests", "golden", "cohorts.npz"))
    res = ro.score_and_grad(g["c1_log_theta"], g["c1_log_d_p"], g["c1_log_d_m"], g["c1_dat"], float(g["c1_perc_met"]))
    eng = ro._engine_for(g["c1_dat"])
    q.put((rank, eng.n_pat, [np.asarray(r) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_engine_matches_golden(golden):
    """world_size 2, each rank a real engine over its own patient shard, one all-reduce: every rank gets the
    unsharded (reference) result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = golden("cohorts")
    assert sorted(r[0] for r in got) == [0, 1]
    assert sum(r[1] for r in got) == g["c1_dat"].shape[0] and all(r[1] > 0 for r in got)
    for _, _, res in got:
        np.testing.assert_allclose(res[0], g["c1_score"], rtol=1e-9)
        np.testing.assert_allclose(res[1], g["c1_d_th"], **TIGHT)
        np.testing.assert_allclose(res[2], g["c1_d_dp"], **TIGHT)
        np.testing.assert_allclose(res[3], g["c1_d_dm"], **TIGHT)


def test_engines_on_explicit_device_and_guard():
    """Every ABI entry selects the engine's GPU itself (DevGuard) and restores the caller's device."""
    import torch
    from metmhn_amd import Engine, synthetic
    n = 5
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.mixed_cohort(n, 20, seed=9)
    e = Engine(n, device=0)
    e.set_cohort(dat)
    r0 = e.patient_grads(lt, dp, dm, with_grad=False)
    assert torch.cuda.current_device() == 0
    e2 = Engine(n)                      # default device: LOCAL_RANK or 0
    e2.set_cohort(dat)
    np.testing.assert_allclose(e2.patient_grads(lt, dp, dm, with_grad=False), r0, rtol=1e-13)
    e.close(); e2.close()
    with pytest.raises(RuntimeError):
        Engine(n, device=10_000)


def test_luad_reduced_anchor(golden):
    """BASELINE configs[0] / SURVEY Appendix C.2: the 4 852 x 43 LUAD-reduced cohort at indep(dat), perc_met 0.2:
    score = -8.43382859658627, and the reference's full gradient (fixture made by tests/tools/make_golden_luad.py)."""
    import os
    if not os.path.exists(os.path.join(GOLDEN, "luad_indep.npz")):
        pytest.skip("luad_indep.npz not generated")
    import metmhn_amd.regularized_optimization as ro
    from metmhn_amd import Utilityfunctions as U
    g = golden("luad_indep")
    dat = g["dat"]
    assert dat.shape == (4852, 43) and list(np.bincount(dat[:, -1])) == [595, 1677, 2127, 453]
    th, dp, dm = U.indep(dat)
    np.testing.assert_allclose(th, g["indep_theta"], rtol=1e-12)
    s, gt, gp, gm = ro.score_and_grad(g["indep_theta"], g["indep_dp"], g["indep_dm"], dat, 0.2)
    np.testing.assert_allclose(s, -8.43382859658627, rtol=1e-9)
    np.testing.assert_allclose(s, g["indep_score"], rtol=1e-10)
    np.testing.assert_allclose(np.linalg.norm(gt), 1.66178987553837, rtol=1e-8)
    np.testing.assert_allclose(gt, g["indep_d_th"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gp, g["indep_d_dp"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gm, g["indep_d_dm"], rtol=1e-7, atol=1e-10)
    params = np.concatenate((g["indep_theta"].flatten(), g["indep_dp"], g["indep_dm"]))
    v, gr = ro.score_and_grad_reg(params, dat, 0.2, ro.symmetric_penal, 1e-3)
    np.testing.assert_allclose(v, g["indep_reg_value"], rtol=1e-9)
    np.testing.assert_allclose(gr, g["indep_reg_grad"], rtol=1e-7, atol=1e-10)


def test_every_route_in_one_batch(monkeypatch):
    """Per-problem dispatch (round 5): ONE batch in which all three joint routes and both single-tumour paths are active at
    once - window-shaped paired rows (k = 15 .. 17; route RT_W once the thresholds are lowered), multi-tile rows outside the window
    shapes (RT_P), rows with the seeding bit inside the tile and a few deep ones left over (RT_T, the cooperative launch), paired
    rows whose marginal space exceeds a tile and unpaired rows of 13 - 15 bits (staged kernels, both groups) next to small ones
    (small-space path).  Random slots, every order, every type; per-patient log-probabilities and gradients against
    oracle/metmhn_ref.c (reference pass structure) - with the default thresholds (everything on the tile route), with the
    thresholds at 8 (all routes), and with level-by-level launches."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 16
    rng = np.random.default_rng(516)
    lt, dp, dm = synthetic.random_params(n)
    rows = []

    def paired(kp, km, order):
        r = np.zeros(2 * n + 3, dtype=np.int8)
        r[2 * rng.choice(n, kp, replace=False)] = 1
        r[2 * rng.choice(n, km, replace=False) + 1] = 1
        r[2 * n], r[2 * n + 1], r[2 * n + 2] = 1, order, 3
        return r

    for i in range(12):                                          # window shapes
        kp, km = [(10, 5), (11, 4), (4, 11), (12, 4), (5, 10), (10, 6)][i % 6]
        rows.append(paired(kp, km, i % 3))
    for i in range(12):                                          # multi-tile, outside the window shapes
        kp, km = [(8, 7), (7, 7), (9, 5), (6, 8), (9, 7), (13, 2)][i % 6]
        rows.append(paired(kp, km, [0, 1, 2, -99][i % 4]))
    for i in range(24):                                          # small joint spaces
        rows.append(paired(int(rng.integers(0, 6)), int(rng.integers(0, 6)), i % 3))
    for i in range(30):                                          # unpaired rows of every type, some beyond a tile
        typ = i % 3
        kk = int(rng.integers(12, 15)) if i % 5 == 0 else int(rng.integers(0, 9))
        r = np.zeros(2 * n + 3, dtype=np.int8)
        r[2 * rng.choice(n, kk, replace=False) + (1 if typ == 2 else 0)] = 1
        r[2 * n], r[2 * n + 1], r[2 * n + 2] = (0 if typ == 0 else 1), -99, typ
        rows.append(r)
    dat = np.array(rows, dtype=np.int8)
    rng.shuffle(dat, axis=0)
    lp, g, a, b = cref.patients(lt, dp, dm, dat)
    for env in ({}, {"MMHN_PSOLVE_MIN": "8", "MMHN_WSOLVE_MIN": "8", "MMHN_POISON": "1"}, {"MMHN_COOP": "0", "MMHN_PSOLVE_MIN": "8"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], lp, rtol=1e-9, atol=1e-12, err_msg=str(env))
        np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10, err_msg=str(env))
        np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10, err_msg=str(env))
        np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10, err_msg=str(env))
        for k_ in env:
            monkeypatch.delenv(k_)


def test_reference_name_mirrors_one_event_and_all_zero(golden):
    """VERDICT r4 missing 5: `metmhn.jx.one_event` (k = 1 paired rows) and `_lp_prim_obs_az` / `_grad_prim_obs_az`
    (likelihood.py:408-416, 462-476) by name, against the reference's values in patients.npz (the all-zero type-0 row and the
    k = 1 paired rows of every order) and the closed form."""
    from metmhn_amd.jx import likelihood as L, one_event as OE
    g = golden("patients")
    pre = "c0_"
    lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
    n = (dat.shape[1] - 3) // 2
    np.testing.assert_allclose(L._lp_prim_obs_az(lt), -np.log(1.0 + np.exp(np.diag(lt)).sum()), rtol=1e-12)
    az = [i for i in range(dat.shape[0]) if dat[i, -1] == 0 and dat[i, :2 * n + 1].sum() == 0][0]
    lp, gth, gdp = L._grad_prim_obs_az(lt)
    np.testing.assert_allclose(lp, g[pre + "lp_grad"][az], rtol=1e-12)
    np.testing.assert_allclose(gth, g[pre + "d_th"][az], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(gdp, g[pre + "d_dp"][az], atol=1e-13)
    for i in range(dat.shape[0]):
        if dat[i, -1] != 3 or dat[i, :2 * n + 1].sum() != 1:
            continue
        order = int(dat[i, -2])
        fn = {0: OE._g_coupled_0, 1: OE._g_coupled_1}.get(order, OE._g_coupled_2)
        lp, gth, gdp, gdm = fn(lt, dp, dm, dat[i, :2 * n + 1])
        np.testing.assert_allclose(lp, g[pre + "lp_grad"][i], rtol=1e-12)
        np.testing.assert_allclose(gth, g[pre + "d_th"][i], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"][i], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"][i], rtol=1e-10, atol=1e-13)


def test_cooperative_launch_soak(monkeypatch, golden):
    """A hand-off that is only wrong now and then (a stale read of another workgroup's tile under uneven load) would show as a
    deviating evaluation: 400 evaluations of the 28-event LUAD cohort each with 37, the default (one per CU) and 1 024 workgroups -
    fewer workgroups than tiles in flight, and more workgroups than can be resident (the queue keeps either deadlock-free) -, every
    result against the fixture.  (`scripts/soak_coop.py` is the long form: 9 000 evaluations = 36 000 cooperative launches, all exact
    to 2e-16 / 3e-15.)"""
    import os
    if not os.path.exists(os.path.join(GOLDEN, "luad28.npz")):
        pytest.skip("luad28.npz not generated")
    from metmhn_amd import Engine, distributed as D
    g = golden("luad28")
    dat, pm = g["dat"], float(g["perc_met"])
    for wgs in ("37", None, "1024"):
        if wgs is None:
            monkeypatch.delenv("MMHN_COOP_WGS", raising=False)
        else:
            monkeypatch.setenv("MMHN_COOP_WGS", wgs)
        e = Engine(28)
        e.set_cohort(dat)
        for it in range(400):
            pt = "fit" if it & 1 else "indep"
            s, G, a, b = D.combine_sums(e.cohort_sums(g[pt + "_theta"], g[pt + "_dp"], g[pt + "_dm"]), 29, pm)
            assert abs(float(s) - float(g[pt + "_score"])) <= 1e-9 * abs(float(g[pt + "_score"])), (wgs, it)
            np.testing.assert_allclose(G, g[pt + "_d_th"], rtol=1e-7, atol=1e-10, err_msg=f"workgroups {wgs} evaluation {it}")
        e.close()
    monkeypatch.delenv("MMHN_COOP_WGS", raising=False)


def test_cooperative_solve_spins_are_bounded(monkeypatch):
    """csrc/tsolve.h: every wait of the one-launch tile solve is bounded.  With MMHN_COOP_FAULT=1 the first tile of every
    cooperative launch never raises its flag: its dependants must run into the bound of their spin, set the abort word (which
    ends every other spin at once), the launch must drain, and the host call must fail with a clear message - within seconds,
    leaving the GPU usable: a fresh engine evaluates the same cohort correctly afterwards."""
    import time
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 16
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 6, k=15, seed=77)                 # multi-tile joint problems on the tile route
    lp = cref.fast_patients(lt, dp, dm, dat)[0]
    monkeypatch.setenv("MMHN_COOP_FAULT", "1")
    e = Engine(n)
    e.set_cohort(dat)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="timed out"):
        e.patient_grads(lt, dp, dm)
    assert time.perf_counter() - t0 < 60.0
    e.close()
    monkeypatch.delenv("MMHN_COOP_FAULT")
    e = Engine(n)
    e.set_cohort(dat)
    r = e.patient_grads(lt, dp, dm)
    e.close()
    np.testing.assert_allclose(r[0], lp, rtol=1e-10)


def test_luad28_real_workload(monkeypatch, golden):
    """The cohort examples/analysis.py of the reference really fits (`muts = list(dat.columns[1:-4])`,
    examples/analysis.py:55): ALL 28 events, 4 852 x 59 int8, 453 paired rows with k = 1 .. 21 (110 rows with k >= 13 carry 99 % of
    the work, class splits like (8, 9), (9, 10), (11, 8)), single-tumour spaces of up to 17 bits - a heterogeneous cohort: every
    joint problem on its own route (per-problem dispatch), the tiles of the large ones in ONE cooperative launch with several
    workgroups per patient (csrc/tsolve.h: k_csolve), patients with a single-tumour space of more than a tile on the staged
    kernels next to the small-space path.  Fixture: tests/tools/make_golden_luad.py luad28 (values by oracle/metmhn_ref.c /
    metmhn_fast.c, which are pinned to the reference elsewhere) at indep(dat) and at the reference's published parameters
    (results/luad/luad_g14_cv_20muts_8cnvs.csv).  1e-9 on the score, 1e-7 on the gradients; three dispatches: the default,
    level-by-level launches instead of the cooperative one (MMHN_COOP=0), and every window / multi-tile problem forced onto its
    per-patient kernel (MMHN_PSOLVE_MIN=1) with the solution buffers NaN-poisoned."""
    import os
    if not os.path.exists(os.path.join(GOLDEN, "luad28.npz")):
        pytest.skip("luad28.npz not generated")
    from metmhn_amd import Engine, Utilityfunctions as U
    g = golden("luad28")
    dat = g["dat"]
    assert dat.shape == (4852, 59) and list(np.bincount(dat[:, -1])) == [595, 1677, 2127, 453]
    th, dp, dm = U.indep(dat)
    np.testing.assert_allclose(th, g["indep_theta"], rtol=1e-12)
    pm = float(g["perc_met"])
    for env in ({}, {"MMHN_COOP": "0"}, {"MMHN_PSOLVE_MIN": "1", "MMHN_POISON": "1"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        e = Engine(28)
        e.set_cohort(dat)
        for pt in ("indep", "fit"):
            lt, a, b = g[pt + "_theta"], g[pt + "_dp"], g[pt + "_dm"]
            s, G, ga, gb = e.score_and_grad(lt, a, b, pm)
            np.testing.assert_allclose(s, g[pt + "_score"], rtol=1e-9, err_msg=f"{pt} {env}")
            np.testing.assert_allclose(G, g[pt + "_d_th"], rtol=1e-7, atol=1e-10, err_msg=f"{pt} {env}")
            np.testing.assert_allclose(ga, g[pt + "_d_dp"], rtol=1e-7, atol=1e-10, err_msg=f"{pt} {env}")
            np.testing.assert_allclose(gb, g[pt + "_d_dm"], rtol=1e-7, atol=1e-10, err_msg=f"{pt} {env}")
            np.testing.assert_allclose(float(e.score(lt, a, b, pm)), g[pt + "_score"], rtol=1e-9)
        lp = e.patient_grads(g["fit_theta"], g["fit_dp"], g["fit_dm"])[0]
        np.testing.assert_allclose(lp, g["fit_lp"], rtol=1e-9, atol=1e-11, err_msg=f"per-patient log-probs {env}")
        e.close()
        for k_ in env:
            monkeypatch.delenv(k_)


def test_luad_fit_reaches_published_objective(golden):
    """learn_mhn from indep(dat) with the reference's own LUAD settings (perc_met 0.2, lambda 1e-3;
    examples/data_analysis.ipynb cell 14).  (a) The engine's objective equals the reference's at the published parameters
    (results/luad/luad_g14_20muts.csv, evaluated by the reference in luad_fit.npz).  Those parameters are NOT the optimum of
    that objective (5.94956 there), so the fit is compared with what the SAME optimiser reaches on the CPU oracle: (b)
    tests/golden/luad_cpu_fit.npz holds SciPy's L-BFGS-B on oracle/metmhn_ref.c from the same start with ftol 1e-10
    (tests/tools/make_golden_luad_cpu_fit.py: 151 iterations, objective 5.4774638); the engine's fit with the same stopping rule
    must reach that objective to 1e-7 relative and those parameters to 1e-3."""
    import os
    for f in ("luad_indep.npz", "luad_fit.npz", "luad_cpu_fit.npz"):
        if not os.path.exists(os.path.join(GOLDEN, f)):
            pytest.skip("LUAD fixtures not generated")
    import metmhn_amd.regularized_optimization as ro
    gi, gf, gc = golden("luad_indep"), golden("luad_fit"), golden("luad_cpu_fit")
    dat = gi["dat"]
    pf = np.concatenate((gf["fit_theta"].flatten(), gf["fit_dp"], gf["fit_dm"]))
    v_pub, g_pub = ro.score_and_grad_reg(pf, dat, 0.2, ro.symmetric_penal, 1e-3)
    np.testing.assert_allclose(v_pub, gf["fit_reg_value"], rtol=1e-9)
    np.testing.assert_allclose(g_pub, gf["fit_reg_grad"], rtol=1e-6, atol=1e-9)
    # the engine at the CPU optimiser's end point: same objective, and a gradient as small as the CPU's
    pc = np.concatenate((gc["theta"].flatten(), gc["dp"], gc["dm"]))
    v_c, g_c = ro.score_and_grad_reg(pc, dat, 0.2, ro.symmetric_penal, 1e-3)
    np.testing.assert_allclose(v_c, float(gc["objective"]), rtol=1e-10)
    assert np.linalg.norm(g_c) <= 2.0 * float(gc["grad_norm"]) + 1e-9
    th, dp, dm = ro.learn_mhn(gi["indep_theta"], gi["indep_dp"], gi["indep_dm"], dat, 0.2, ro.symmetric_penal, 1e-3,
                              opt_ftol=float(gc["ftol"]), opt_v=False)
    v_fit = float(ro.score_reg(np.concatenate((th.flatten(), dp, dm)), dat, 0.2, ro.symmetric_penal, 1e-3))
    assert v_fit < float(gf["fit_reg_value"])
    assert abs(v_fit - float(gc["objective"])) <= 1e-7 * float(gc["objective"]), (v_fit, float(gc["objective"]))
    # parameters: events the cohort never shows sit at indep()'s -1e10-like floor on both sides; compare through exp
    # for the diagonal (base rates) and directly elsewhere
    np.testing.assert_allclose(np.exp(np.diag(th)), np.exp(np.diag(gc["theta"])), rtol=1e-3, atol=1e-6)
    off = ~np.eye(th.shape[0], dtype=bool)
    np.testing.assert_allclose(th[off], gc["theta"][off], atol=1e-3)
    np.testing.assert_allclose(dp, gc["dp"], atol=1e-3)
    np.testing.assert_allclose(dm, gc["dm"], atol=1e-3)


def test_staged_and_fused_small_paths_agree(monkeypatch, golden):
    """The small-space path (csrc/small.h: one workgroup per patient) and the staged kernels it replaces, on the mixed
    golden cohorts: both against the reference's values."""
    from metmhn_amd import Engine, distributed as D
    g = golden("cohorts")
    for small in ("1", "0"):
        monkeypatch.setenv("MMHN_SMALL", small)
        for c in (0, 1, 3):
            pre = f"c{c}_"
            dat = g[pre + "dat"]
            e = Engine((dat.shape[1] - 3) // 2)
            e.set_cohort(dat)
            s, gt, gp, gm = D.combine_sums(e.cohort_sums(g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"]),
                                           g[pre + "log_theta"].shape[0], float(g[pre + "perc_met"]))
            e.close()
            np.testing.assert_allclose(s, g[pre + "score"], rtol=1e-9, err_msg=f"small={small} cohort {c}")
            np.testing.assert_allclose(gt, g[pre + "d_th"], **TIGHT)
            np.testing.assert_allclose(gp, g[pre + "d_dp"], **TIGHT)
            np.testing.assert_allclose(gm, g[pre + "d_dm"], **TIGHT)


@pytest.mark.gpu
def test_split_evaluation_and_error_order(golden):
    """mmhn_cohort_sums_begin / _end: same sums as the one-call form, host work in between, misuse is an error."""
    from metmhn_amd import Engine
    g = golden("cohorts")
    lt, dp, dm, dat = g["c1_log_theta"], g["c1_log_d_p"], g["c1_log_d_m"], g["c1_dat"]
    e = Engine((dat.shape[1] - 3) // 2)
    e.set_cohort(dat)
    ref = e.cohort_sums(lt, dp, dm)
    e.cohort_sums_begin(lt, dp, dm)
    lt2 = lt.copy()                                           # the parameter arrays may be reused at once
    lt2[:] = 0.0
    with pytest.raises(RuntimeError):
        e.cohort_sums_begin(lt, dp, dm)                       # one evaluation in flight per handle
    with pytest.raises(RuntimeError):
        e.set_cohort(dat)                                     # ... and the cohort stays while it is
    with pytest.raises(RuntimeError):
        e.patient_grads(lt, dp, dm)
    np.testing.assert_array_equal(e.cohort_sums_end(), ref)
    with pytest.raises(RuntimeError):
        e.cohort_sums_end()
    np.testing.assert_array_equal(e.cohort_sums(lt, dp, dm), ref)
    e.close()


@pytest.mark.gpu
def test_batched_kronvec_at_benchmarked_shape():
    """mmhn_kronvec_batched = the launch mmhn_bench_kronvec times (k_kv: 256 tiles per vector, 8 tile bits, neighbour
    pipeline, scalar-unit U rows), at the benchmarked shape n = k = 20 with B = 4 random vectors.  The device output
    starts as NaNs, so every state of y - including the tiles where Q_off has no entries - must be written by that one
    launch.  Against ref_kronvec (oracle/metmhn_ref.c: the reference's factor-by-factor passes) to 1e-9, both
    transposes, diag 0 / 1; the same at k = 16 and k = 14."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    cref.load()
    rng = np.random.default_rng(77)
    for n, kk, B in ((20, 20, 4), (16, 16, 3), (20, 14, 2)):
        lt, _, _ = synthetic.random_params(n, seed=300 + kk)
        st = synthetic.full_k_cohort(n, 1, k=kk, seed=500 + kk)[0, :2 * n + 1]
        assert int(st.sum()) == kk
        p = rng.random((B, 2 ** kk)) + 0.01
        with Engine(n) as e:
            for tr in (False, True):
                for diag in (False, True):
                    y = e.kronvec_batched(lt, p, st, diag=diag, transpose=tr)
                    assert np.isfinite(y).all(), f"k={kk} tr={tr} diag={diag}: unwritten states"
                    for b in range(B if kk < 20 else 2):            # (the CPU passes take seconds at k = 20)
                        ref = cref.kronvec(lt, p[b], st, diag=diag, transpose=tr)
                        np.testing.assert_allclose(y[b], ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max(),
                                                   err_msg=f"k={kk} tr={tr} diag={diag} b={b}")
            # single-vector entry point and the batched one agree
            np.testing.assert_allclose(e.kronvec(lt, p[0], st, diag=False), e.kronvec_batched(lt, p[:1], st, diag=False)[0],
                                       rtol=1e-13)
            ms, live, tot = e.bench_kronvec(lt, st, B, 2, tiles=True)
            assert ms > 0 and 0 < live <= tot == B * max(1, 2 ** (kk - 12))


@pytest.mark.gpu
def test_fused_jacobi_step_at_benchmarked_shape():
    """The fused Jacobi step mmhn_bench_kronvec times (k_kv with the 1/diag and right-hand-side epilogue, every tile
    launched, y starting as NaNs) against lidg * (ref_kronvec(p, diag=False) + rhs) with the oracle's diagonal,
    n = k = 20 and k = 14, both transposes (likelihood.py:249-255)."""
    from oracle import cref, metmhn_oracle as O
    from metmhn_amd import Engine, synthetic
    cref.load()
    rng = np.random.default_rng(78)
    for n, kk, B in ((20, 20, 2), (20, 14, 2)):
        lt, dp, dm = synthetic.random_params(n, seed=310 + kk)
        st = synthetic.full_k_cohort(n, 1, k=kk, seed=510 + kk)[0, :2 * n + 1]
        p = rng.random((B, 2 ** kk)) + 0.01
        rhs = rng.random((B, 2 ** kk))
        ones = np.ones(2 ** kk)
        lidg = 1.0 / (O.diag_scal_p(dp, st, ones) + O.diag_scal_m(dm, st, ones) - O.kron_diag(lt, st, kk))
        with Engine(n) as e:
            for tr in (False, True):
                y = e.jacobi_step_batched(lt, dp, dm, p, rhs, st, transpose=tr)
                assert np.isfinite(y).all()
                for b in range(B):
                    ref = lidg * (cref.kronvec(lt, p[b], st, diag=False, transpose=tr) + rhs[b])
                    np.testing.assert_allclose(y[b], ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max(), err_msg=f"k={kk} tr={tr}")


def _one_rank_comm(e):
    """In-library RCCL all-reduce with a one-rank communicator (what MMHN_FORCE_ALLREDUCE=1 attaches): the collective
    code path of the sharded engine on a single GPU."""
    from metmhn_amd.engine import unique_id
    e.comm_init(unique_id(), 0, 1)


@pytest.mark.gpu
def test_long_launches_at_the_largest_event_count():
    """The 1 024-thread instantiations the long launches take (k_prep, k_diag, k_scatter_marg; one-row k_grad_rows) at n = 30 events
    - the dynamic LDS of k_prep / k_diag grows with the event count (N = 31: 49 / 65 KB, above the default window) - on 320 unpaired
    rows with 13 - 14 active events (single-tumour spaces of four tiles: the staged kernels, more than 256 problems and tiles in every
    launch) and 300 paired rows with small joint spaces; log-probabilities and gradients against oracle/metmhn_ref.c."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 30
    lt, dp, dm = synthetic.random_params(n, seed=31)
    rng = np.random.default_rng(77)
    rows = []
    for r in range(320):
        bits = np.zeros(2 * n, dtype=np.int8)
        ev = rng.choice(n, size=13 + (r & 1), replace=False)
        typ = r % 3
        if typ == 2:
            bits[2 * ev + 1] = 1
            rows.append(np.concatenate((bits, [1, -99, 2])))
        else:
            bits[2 * ev] = 1
            rows.append(np.concatenate((bits, [typ, -99, typ])))
    for r in range(300):
        bits = np.zeros(2 * n, dtype=np.int8)
        bits[rng.choice(2 * n, size=int(rng.integers(2, 9)), replace=False)] = 1
        rows.append(np.concatenate((bits, [1, int(rng.integers(0, 3)), 3])))
    dat = np.array(rows, dtype=np.int8)
    lp, g, a, b = cref.patients(lt, dp, dm, dat, with_grad=True)
    e = Engine(n)
    e.set_cohort(dat)
    r = e.patient_grads(lt, dp, dm)
    e.close()
    np.testing.assert_allclose(r[0], lp, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_full_size_gradient_is_directional_difference_of_score():
    """The reference's own test strategy (tests/test_gradient.py:9-31,68-69: analytic gradient of score_and_grad against finite
    differences of score) at the headline size, where no oracle finishes in seconds: 320 paired patients with n = k = 20 of every
    order - the window route of the engine (forward solve, adjoint solve, class marginals, gradient rows) - and 64 rows of mixed types
    (PT-only, MT-only, never-metastasising, paired with fewer events: the tile and small-space routes, EM and NM rows weighted by
    perc_met), the regularised objective and its gradient through the reference's entry points, central differences of score_reg
    along four random directions of the 440-dimensional parameter vector (h = 1e-5: truncation ~1e-10, rounding ~1e-11 relative)."""
    from metmhn_amd import regularized_optimization as ro, synthetic
    n = 20
    lt, dp, dm = synthetic.random_params(n)
    dat = np.vstack((np.asarray(synthetic.full_k_cohort(n, 320, seed=77)), np.asarray(synthetic.mixed_cohort(n, 64, seed=3, p_event=0.3))))
    assert {0, 1, 2, 3} <= set(int(t) for t in dat[:, -1])
    params = np.concatenate((np.asarray(lt).flatten(), dp, dm))
    v, g = ro.score_and_grad_reg(params, dat, 0.3, ro.symmetric_penal, 1e-2)
    assert np.isfinite(v) and np.isfinite(g).all()
    rng = np.random.default_rng(5)
    h = 1e-5
    for _ in range(4):
        u = rng.standard_normal(params.size)
        u /= np.linalg.norm(u)
        fd = (float(ro.score_reg(params + h * u, dat, 0.3, ro.symmetric_penal, 1e-2))
              - float(ro.score_reg(params - h * u, dat, 0.3, ro.symmetric_penal, 1e-2))) / (2 * h)
        np.testing.assert_allclose(float(g @ u), fd, rtol=2e-6, atol=1e-9)
    # one coordinate of each block, the reference's forward difference (h = 1e-8, rtol 1e-4)
    for idx in (3 * (n + 1) + 7, (n + 1) ** 2 + 4, (n + 1) ** 2 + (n + 1) + 9):
        e = np.zeros(params.size)
        e[idx] = 1e-8
        fd = (float(ro.score_reg(params + e, dat, 0.3, ro.symmetric_penal, 1e-2)) - float(v)) / 1e-8
        np.testing.assert_allclose(g[idx], fd, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_config3_rank_shard_n20_50000_patients():
    """BASELINE configs[3] (n = 20, 50 000 patients over 8 GPUs) as ONE rank sees it: the LPT shard 0 of the 50 000-row
    cohort (6 250 rows, regularized_optimization.py:256-266 is what shards), evaluated through the pre-combined
    all-reduce payload with the in-library RCCL all-reduce attached; 8 random rows of the shard against
    oracle/metmhn_fast.c, and the shard's weighted sums against the sum of its per-patient rows."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic, distributed as D
    n, P, W = 20, 50000, 8
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
    parts = D.shard_rows(dat, W)
    assert sorted(np.concatenate(parts).tolist()) == list(range(P)) and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    rows = dat[parts[0]]
    assert rows.shape[0] == P // W
    w, n_full = D.em_weight(float(dat[:, -3].sum()), float(P), 0.5)
    with Engine(n) as e:
        e.set_cohort(rows)
        _one_rank_comm(e)
        e.cohort_wsums_begin(lt, dp, dm, w)
        ws = e.cohort_wsums_end()
        lp, g, gp, gm = e.patient_grads(lt, dp, dm)
    N = n + 1
    assert ws.shape == (1 + N * N + 2 * N,) and np.isfinite(ws).all()
    # every row is a paired (EM) patient: the payload is w times the plain sums of the shard's rows
    np.testing.assert_allclose(ws[0], w * lp.sum(), rtol=1e-11)
    np.testing.assert_allclose(ws[1:1 + N * N].reshape(N, N), w * g.sum(axis=0), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ws[1 + N * N:1 + N * N + N], w * gp.sum(axis=0), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ws[1 + N * N + N:], w * gm.sum(axis=0), rtol=1e-9, atol=1e-9)
    pick = np.random.default_rng(5).choice(rows.shape[0], size=8, replace=False)
    rlp, rg, ra, rb = cref.fast_patients(lt, dp, dm, rows[pick])
    np.testing.assert_allclose(lp[pick], rlp, rtol=1e-10)
    np.testing.assert_allclose(g[pick], rg, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gp[pick], ra, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gm[pick], rb, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_shard_partial_sums_add_up():
    """8 LPT shards of a 64-row mixed cohort (all dat types), each evaluated by an engine of its own with the
    collective attached: the 8 partial buffers add up to the unsharded buffer to 1e-12 - both layouts (raw sums and
    the pre-combined payload)."""
    from metmhn_amd import Engine, synthetic, distributed as D
    n = 7
    lt, dp, dm = synthetic.random_params(n, seed=21)
    dat = np.vstack((synthetic.mixed_cohort(n, 48, seed=4, p_event=0.45), synthetic.full_k_cohort(n, 16, seed=9)))
    w, n_full = D.em_weight(float(dat[:, -3].sum()), float(dat.shape[0]), 0.3)
    with Engine(n) as e:
        e.set_cohort(dat)
        full = e.cohort_sums(lt, dp, dm)
        e.cohort_wsums_begin(lt, dp, dm, w)
        wfull = e.cohort_wsums_end()
    acc, wacc = np.zeros_like(full), np.zeros_like(wfull)
    for part in D.shard_rows(dat, 8):
        with Engine(n) as e:
            e.set_cohort(dat[part])
            _one_rank_comm(e)
            acc += e.cohort_sums(lt, dp, dm)
            e.cohort_wsums_begin(lt, dp, dm, w)
            wacc += e.cohort_wsums_end()
    np.testing.assert_allclose(acc, full, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(wacc, wfull, rtol=1e-12, atol=1e-12)
    ref = D.combine_sums(full, n + 1, 0.3)
    got = D.split_wsums(wacc, n + 1, n_full)
    for a, b in zip(ref, got):
        np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)


def _fp32_report(tag, x32, x64):
    """max per-component error of an fp32 gradient against fp64: absolute, and relative to 2e-3 |x| + 2e-5"""
    err = np.abs(x32 - x64)
    tol = 2e-3 * np.abs(x64) + 2e-5
    worst = np.unravel_index(np.argmax(err / tol), err.shape)
    print(f"[fp32] {tag}: max abs err {err.max():.3e}, worst component {worst}: {x32[worst]:.6e} vs {x64[worst]:.6e} "
          f"({(err / tol).max():.3f} of the bar)")
    return err, tol


@pytest.mark.gpu
def test_config4_rank_shard_n25_fp32_multi_batch():
    """BASELINE configs[4] (n = 25, 10 000 patients, fp32, 8 GPUs) as ONE rank sees it: the 1 250-row LPT shard 0 -
    2 x 1 250 x 128 MiB of solution vectors do not fit the workspace, so the engine runs it in several batches - with
    the collective attached; 8 random rows against oracle/metmhn_fast.c (fp64) at the fp32 bar: log-prob 1e-4,
    every gradient component within 2e-3 relative + 2e-5 absolute; the payload equals the sum of the per-patient rows."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic, distributed as D
    n, P, W = 25, 10000, 8
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, P, seed=2000 + n)
    part = D.shard_rows(dat, W)[0]
    rows = dat[part]
    assert rows.shape[0] == P // W
    w, n_full = D.em_weight(float(dat[:, -3].sum()), float(P), 0.5)
    with Engine(n, dtype="f32") as e:
        e.set_cohort(rows)
        _one_rank_comm(e)
        e.cohort_wsums_begin(lt, dp, dm, w)
        ws = e.cohort_wsums_end()
        lp, g, gp, gm = e.patient_grads(lt, dp, dm)
    N = n + 1
    assert np.isfinite(ws).all() and np.isfinite(lp).all()
    np.testing.assert_allclose(ws[0], w * lp.sum(), rtol=1e-9)
    # (two fp32 evaluations, 1 250 terms per sum: agreement to fp32 rounding of the partial sums)
    np.testing.assert_allclose(ws[1:1 + N * N].reshape(N, N), w * g.sum(axis=0), rtol=2e-5, atol=2e-5)
    pick = np.random.default_rng(6).choice(rows.shape[0], size=8, replace=False)
    rlp, rg, ra, rb = cref.fast_patients(lt, dp, dm, rows[pick])
    np.testing.assert_allclose(lp[pick], rlp, rtol=1e-4)
    for tag, x32, x64 in (("d_theta", g[pick], rg), ("d_dp", gp[pick], ra), ("d_dm", gm[pick], rb)):
        err, tol = _fp32_report(tag, x32, x64)
        assert (err <= tol).all(), tag


@pytest.mark.gpu
def test_batched_kronvec_and_jacobi_step_fp32():
    """The fp32 instantiations of the batched product and of the fused Jacobi step (BASELINE configs[4] runs the engine in
    fp32) against the fp64 CPU port at k = 16: 2e-5 of the vector's largest entry."""
    from oracle import cref, metmhn_oracle as O
    from metmhn_amd import Engine, synthetic
    cref.load()
    n = kk = 16
    lt, dp, dm = synthetic.random_params(n, seed=41)
    st = synthetic.full_k_cohort(n, 1, k=kk, seed=541)[0, :2 * n + 1]
    rng = np.random.default_rng(79)
    p = rng.random((3, 2 ** kk)) + 0.01
    rhs = rng.random((3, 2 ** kk))
    ones = np.ones(2 ** kk)
    lidg = 1.0 / (O.diag_scal_p(dp, st, ones) + O.diag_scal_m(dm, st, ones) - O.kron_diag(lt, st, kk))
    with Engine(n, dtype="f32") as e:
        for tr in (False, True):
            y = e.kronvec_batched(lt, p, st, diag=False, transpose=tr)
            z = e.jacobi_step_batched(lt, dp, dm, p, rhs, st, transpose=tr)
            assert np.isfinite(y).all() and np.isfinite(z).all()
            for b in range(3):
                ref = cref.kronvec(lt, p[b], st, diag=False, transpose=tr)
                np.testing.assert_allclose(y[b], ref, rtol=0, atol=2e-5 * np.abs(ref).max())
                refz = lidg * (ref + rhs[b])
                np.testing.assert_allclose(z[b], refz, rtol=0, atol=2e-5 * np.abs(refz).max())


@pytest.mark.gpu
def test_small_space_path_fp32_on_luad_cohort(golden):
    """The small-space kernels (csrc/small.h: side-by-side marginal problems, in-kernel marginal right-hand sides) in
    fp32 on the LUAD-reduced cohort, per patient against the fp64 C port (oracle/metmhn_ref.c, cref.patients - itself
    pinned to the reference on this cohort) at the fp32 bar: log-prob 1e-4 relative, every gradient component within
    2e-3 relative + 2e-5 absolute; the fp64 engine on the same rows against the same port at 1e-9."""
    from oracle import cref
    from metmhn_amd import Engine
    g = golden("luad_indep")
    dat, lt, dp, dm = g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"]
    paired = np.flatnonzero(dat[:, -1] == 3)
    rows = np.concatenate((paired, np.arange(0, dat.shape[0], 23)))          # every paired row + a stride of the others
    sub = dat[rows]
    ref = cref.patients(lt, dp, dm, sub)
    res = {}
    for dt in ("f64", "f32"):
        with Engine((dat.shape[1] - 3) // 2, dtype=dt) as e:
            e.set_cohort(sub)
            res[dt] = e.patient_grads(lt, dp, dm)
    for x, y in zip(res["f64"], ref):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res["f32"][0], ref[0], rtol=1e-4, atol=1e-5)
    for nm, x32, x64 in zip(("d_theta", "d_dp", "d_dm"), res["f32"][1:], ref[1:]):
        err, tol = _fp32_report(f"LUAD {nm}", x32, x64)
        assert (err <= tol).all()


def _paired_rows(n, rng, kp, km, count):
    """type-3 rows with kp PT-only and km MT-only events (disjoint), orders cycling 0 / 1 / 2"""
    rows = []
    for r in range(count):
        ev = rng.permutation(n)
        bits = np.zeros(2 * n, dtype=np.int8)
        bits[2 * ev[:kp]] = 1
        bits[2 * ev[kp:kp + km] + 1] = 1
        rows.append(np.concatenate((bits, [1, r % 3, 3])).astype(np.int8))
    return rows


@pytest.mark.gpu
def test_paired_rows_across_the_small_space_classes():
    """Paired rows whose marginal problems fall in different size classes of csrc/small.h, in the three launch layouts
    the engine chooses between: few rows with a 10-bit marginal (they ride in the merged 256-thread launch, two-part
    rows side by side at 2 x 128 threads), more than 16 of them (own 1024-thread launch on a side stream), an 11-bit
    marginal (never merged); orders 0 / 1 / 2 (both parts, PT-first only, MT-first only).  Per patient against the
    optimised C oracle."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 11
    lt, dp, dm = synthetic.random_params(n, seed=31)
    rng = np.random.default_rng(5)
    small = _paired_rows(n, rng, 2, 3, 12) + _paired_rows(n, rng, 5, 1, 9) + _paired_rows(n, rng, 3, 6, 9)
    layouts = {
        "merged (7 rows with a 10-bit part)": _paired_rows(n, rng, 2, 9, 10) + small,
        "own launch (20 rows with a 10-bit part)": _paired_rows(n, rng, 2, 9, 15) + _paired_rows(n, rng, 9, 2, 15) + small,
        "11 bits": _paired_rows(n, rng, 1, 10, 6) + _paired_rows(n, rng, 2, 9, 6) + small,
    }
    for name, rows in layouts.items():
        dat = np.stack(rows)
        lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
        with Engine(n) as e:
            e.set_cohort(dat)
            r = e.patient_grads(lt, dp, dm)
        np.testing.assert_allclose(r[0], lp, rtol=1e-10, err_msg=name)
        np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10, err_msg=name)
        np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10, err_msg=name)
        np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10, err_msg=name)


@pytest.mark.gpu
def test_window_solve_lane_moves():
    """The six lane exchanges of the window solve (csrc/wsolve.h: DPP quad permutes, row shifts / rotate, swizzle,
    bpermute): a lane that has the move along bit i receives from lane ^ (1 << i)."""
    from metmhn_amd import Engine
    with Engine(4) as e:
        for tr in (False, True):
            src = e.debug_lane_moves(transposed=tr)
            for i in range(6):
                for lane in range(64):
                    has = ((lane >> i) & 1) == (0 if tr else 1)
                    if has:
                        assert src[i, lane] == lane ^ (1 << i), (tr, i, lane, src[i, lane])


@pytest.mark.gpu
def test_window_kernels_against_reference_generated_vectors(monkeypatch, golden):
    """tests/golden/large.npz - three paired rows at window shapes ((kP, kM) = (10, 5), (11, 6), (6, 11),
    k = 16 - 18) evaluated by the REFERENCE's own source (likelihood.py:623-731 under the NumPy stand-in, tests/tools/
    make_golden.py large).  The HIP window path (forced: MMHN_PSOLVE_MIN=1, MMHN_WSOLVE=1, buffers NaN-poisoned) must
    reproduce them to 1e-9, and so must the tile route (MMHN_WSOLVE=0: the cooperative launch) and the level-by-level launches."""
    import os
    if not os.path.exists(os.path.join(GOLDEN, "large.npz")):
        pytest.skip("large.npz not generated (tests/tools/make_golden.py large)")
    from metmhn_amd import Engine
    g = golden("large")
    lt, dp, dm, dat = g["log_theta"], g["log_d_p"], g["log_d_m"], g["dat"]
    n = (dat.shape[1] - 3) // 2
    for env in ({"MMHN_PSOLVE_MIN": "1", "MMHN_WSOLVE": "1", "MMHN_POISON": "1"}, {"MMHN_WSOLVE": "0"},
                {"MMHN_WSOLVE": "0", "MMHN_COOP": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        e = Engine(n)
        e.set_cohort(dat)
        r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], g["lp"], rtol=1e-9, err_msg=str(env))
        np.testing.assert_allclose(r[1], g["d_th"], rtol=1e-9, atol=1e-12, err_msg=str(env))
        np.testing.assert_allclose(r[2], g["d_dp"], rtol=1e-9, atol=1e-12, err_msg=str(env))
        np.testing.assert_allclose(r[3], g["d_dm"], rtol=1e-9, atol=1e-12, err_msg=str(env))
        for k_ in env:
            monkeypatch.delenv(k_)


@pytest.mark.gpu
def test_window_layout_solves_match_cpu_port(monkeypatch):
    """MMHN_WSOLVE=1: the joint solves in the window layout (csrc/wsolve.h - 15 index bits on the chip: lane bits
    exchanged out of the neighbour lane's register window, wave bits through an LDS ring, the rest the thread's own
    history) against oracle/metmhn_fast.c on n = k = 20 patients of every order, and against the tile kernels on a
    cohort that mixes shapes the window path takes (10 - 15 row-class bits, 5 - 9 column-class bits, either class as
    rows, no / many pairs) with shapes it leaves to the tile kernels (per-problem dispatch)."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 20
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 16, seed=2000 + n)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    monkeypatch.setenv("MMHN_PSOLVE_MIN", "1")
    monkeypatch.setenv("MMHN_WSOLVE", "1")
    monkeypatch.setenv("MMHN_POISON", "1")
    e = Engine(n)
    e.set_cohort(dat)
    r = e.patient_grads(lt, dp, dm)
    e.close()
    np.testing.assert_allclose(r[0], lp, rtol=1e-10)
    np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10)
    rows = [
        _row(n, range(16), [17, 18, 19], 1),                    # 16 row bits: beyond the window path
        _row(n, range(15), range(14, 19), 2),                   # kP = 15, kM = 5, one pair
        _row(n, range(13), range(10, 16), 2),                   # kP = 13, kM = 6, three pairs
        _row(n, list(range(0, 20, 2)) + [1], range(1, 17, 2), 0),   # kP = 11, kM = 8
        _row(n, range(10), range(5, 14), 1),                    # kP = 10, kM = 9
        _row(n, range(9), range(6, 19), 0),                     # kM = 13 rows, pairs
        _row(n, range(5), range(5, 15), 0),                     # kM = 10 rows, no pair, k = 16: no external bits
        _row(n, range(10), range(10, 15), 1),                   # k = 16, kP = 10 rows
        _row(n, range(12), [12, 13], 0),                        # k = 15: too few column bits for the window path
        _row(n, range(2, 15), range(12, 18), 0),                # kP = 13, kM = 6 again (three pairs) ...
        _row(n, range(1, 14), [0, 3, 14, 15, 16, 19], 1),       # ... and a third one: a same-shape chain of three
    ]
    # (a) mixed shapes: the single-tumour spaces of the large rows exceed a tile -> staged marginal kernels (k_gather_marg);
    # (b) rows whose marginal spaces fit a tile -> the small-space kernels read the window layout (small.h);
    # each with the layout read in place ("1"), converted to index order after every solve ("2"), and without it ("0")
    small = [
        _row(n, range(5), range(5, 15), 0),                     # kM = 10 rows, kP = 5
        _row(n, range(10), range(10, 15), 1),                   # kP = 10 rows
        _row(n, range(10), range(8, 14), 2),                    # kP = 10, kM = 6, two pairs
        _row(n, range(3, 9), range(0, 11), 0),                  # kM = 11 rows, six pairs
    ]
    for cohort in (rows, small):
        dat = np.array(cohort, dtype=np.int8)
        res = []
        for ms in ("1", "2", "0"):
            monkeypatch.setenv("MMHN_WSOLVE", ms)
            e = Engine(n)
            e.set_cohort(dat)
            res.append(e.patient_grads(lt, dp, dm))
            e.close()
        for r1 in res[:2]:
            for x, y in zip(r1, res[2]):
                assert np.isfinite(x).all()
                np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11)
        #  the same cohorts with ONE and TWO workgroups: every same-shape run becomes one chain (patients at chain
        # position > 0: both table buffers re-used, patients entering / leaving the pipeline mid-stream, pinfo of the second
        # buffer), and there are more chains than workgroups (the longest-first deal with filler entries)
        monkeypatch.setenv("MMHN_WSOLVE", "1")
        for wgs in ("1", "2"):
            monkeypatch.setenv("MMHN_WSOLVE_WGS", wgs)
            e = Engine(n)
            e.set_cohort(dat)
            r1 = e.patient_grads(lt, dp, dm)
            e.close()
            for x, y in zip(r1, res[2]):
                assert np.isfinite(x).all()
                np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11, err_msg=f"MMHN_WSOLVE_WGS={wgs}")
        monkeypatch.delenv("MMHN_WSOLVE_WGS")



@pytest.mark.gpu
def test_window_route_two_workgroups_per_patient(monkeypatch):
    """MMHN_WSPLIT=1 (csrc/wsolve.h, SPLIT; measured slower and off by default - DESIGN.md section 6): a patient's passes split on the top
    bit of the external index between two workgroups, the second one reading the first one's half behind a progress word.  n = k = 20
    patients of every order against oracle/metmhn_fast.c, with eight pairs (chains of two patients, both table buffers of both
    workgroups in use) and with one pair per patient; NaN-poisoned vectors."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 20
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.full_k_cohort(n, 16, seed=2000 + n)
    lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
    monkeypatch.setenv("MMHN_WSPLIT", "1")
    monkeypatch.setenv("MMHN_PSOLVE_MIN", "1")
    monkeypatch.setenv("MMHN_POISON", "1")
    for wgs in ("16", None):
        if wgs is None:
            monkeypatch.delenv("MMHN_WSOLVE_WGS", raising=False)
        else:
            monkeypatch.setenv("MMHN_WSOLVE_WGS", wgs)
        e = Engine(n)
        e.set_cohort(dat)
        for _ in range(2):                                      # (the second evaluation meets the progress words of the first)
            r = e.patient_grads(lt, dp, dm)
        e.close()
        np.testing.assert_allclose(r[0], lp, rtol=1e-10)
        np.testing.assert_allclose(r[1], g, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[2], a, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(r[3], b, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_window_path_fp32_k25_shapes(monkeypatch):
    """The fp32 window path (BASELINE configs[4]: k = 25, nine external bits): column-class rates as two factors in LDS
    (csrc/wsolve.h, WCfg<float>::FACT), tables packed per shape, up to 12 column bits, 18 row bits, 12 paired events.
    Designed shapes - two patients of one shape in a row (a chain), either class as rows, the extremes of every bound -
    against oracle/metmhn_fast.c (fp64) at the fp32 bar, (a) all k = 25: the instantiation with the number of external
    bits known at compile time, (b) mixed k: the generic one; and every row again on the tile kernels (MMHN_WSOLVE=0)."""
    from oracle import cref
    from metmhn_amd import Engine, synthetic
    n = 25
    lt, dp, dm = synthetic.random_params(n)
    r = list(range(n))
    k25 = [
        _row(n, r[0:12], r[1:13], 0),                           # (12, 12), eleven pairs
        _row(n, r[5:17], r[4:16], 2),                           # (12, 12) again: chained behind the first
        _row(n, r[0:18], r[12:18], 1),                          # (18, 6): eight external row bits, every MT event paired
        _row(n, r[0:11], r[8:21], 0),                           # MT rows (13, 11), three pairs
        _row(n, r[0:15], r[15:24], 1),                          # (15, 9), no pair
        _row(n, r[3:15], r[5:17], 2),                           # (12, 12), ten pairs
        _row(n, r[0:16], r[10:18], 0),                          # (16, 8)
    ]
    mixed = [
        _row(n, r[0:10], r[10:20], 0),                          # k = 21: (10, 10), no external row bit
        _row(n, r[0:16], r[16:22], 1),                          # k = 23: (16, 6)
        _row(n, r[2:14], r[0:12], 2),                           # k = 25: (12, 12)
        _row(n, r[0:7], r[5:17], 0),                            # k = 20: MT rows (12, 7)
        _row(n, r[0:19], r[19:24], 0),                          # k = 25, 19 row bits: beyond the window path
    ]
    monkeypatch.setenv("MMHN_PSOLVE_MIN", "1")
    monkeypatch.setenv("MMHN_POISON", "1")
    for tag, cohort in (("k25", k25), ("mixed", mixed)):
        dat = np.array(cohort, dtype=np.int8)
        lp, g, a, b = cref.fast_patients(lt, dp, dm, dat)
        # (ms, workgroups): "1" with ONE / TWO workgroups chains every same-shape run - the three (12, 12) rows become one
        # chain of three, the extreme shapes run at chain positions > 0 of a workgroup that changes shape
        for ms, wgs in (("1", None), ("0", None), ("1", "1"), ("1", "2")):
            monkeypatch.setenv("MMHN_WSOLVE", ms)
            if wgs is None:
                monkeypatch.delenv("MMHN_WSOLVE_WGS", raising=False)
            else:
                monkeypatch.setenv("MMHN_WSOLVE_WGS", wgs)
            e = Engine(n, dtype="f32")
            e.set_cohort(dat)
            res = e.patient_grads(lt, dp, dm)
            e.close()
            assert all(np.isfinite(x).all() for x in res)
            np.testing.assert_allclose(res[0], lp, rtol=1e-4)
            for x32, x64, nm in ((res[1], g, "d_theta"), (res[2], a, "d_dp"), (res[3], b, "d_dm")):
                err, tol = _fp32_report(f"{tag} WSOLVE={ms} WGS={wgs} {nm}", x32, x64)
                assert (err <= tol).all(), (tag, ms, wgs, nm)
        monkeypatch.delenv("MMHN_WSOLVE_WGS", raising=False)


def _rccl_worker(rank, world, port, q):
    """One rank of the in-library RCCL test: its own GPU, backend nccl, MMHN_STRICT_COMM=1 (no fallback to torch's
    collective), the golden cohort c0 sharded over the ranks."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MMHN_STRICT_COMM"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import metmhn_amd.regularized_optimization as ro
    ro.configure(device=rank)
    g = np.load(os.path.join(root, "tests", "golden", "cohorts.npz"))
    res = ro.score_and_grad(g["c0_log_theta"], g["c0_log_d_p"], g["c0_log_d_m"], g["c0_dat"], float(g["c0_perc_met"]))
    eng = ro._engine_for(g["c0_dat"])
    q.put((rank, eng.n_pat, bool(eng._device_comm), [np.asarray(r) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_in_library_rccl_two_ranks(golden):
    """The first multi-rank run of the in-library all-reduce (ncclCommInitRank with two ranks, one ncclAllReduce of the
    pre-combined 1 + N^2 + 2N doubles per evaluation on each engine's stream): two processes, one GPU each, started before
    anything touches a GPU here.  Needs two GPUs: skipped on a one-GPU box, runs by itself on any multi-GPU box.
    Result of every rank == the unsharded reference result to 1e-12."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = golden("cohorts")
    assert sorted(r[0] for r in got) == [0, 1]
    assert all(r[2] for r in got), "the device communicator was not attached"
    assert sum(r[1] for r in got) == g["c0_dat"].shape[0]
    for _, _, _, res in got:
        np.testing.assert_allclose(res[0], g["c0_score"], rtol=1e-12)
        np.testing.assert_allclose(res[1], g["c0_d_th"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(res[2], g["c0_d_dp"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(res[3], g["c0_d_dm"], rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
def test_cache_guard_full_crc_and_in_place_refresh(golden):
    """The cohort cache is keyed on the identity of `dat`; its guard hashes the whole array up to 256 KB (and always with
    configure(strict_guard=True)): an in-place edit of ANY row is noticed, the SAME engine gets the new rows (a caller
    holding it keeps a live handle), and the result equals a fresh evaluation of the edited cohort."""
    import metmhn_amd.regularized_optimization as ro
    from metmhn_amd import synthetic
    n = 6
    lt, dp, dm = synthetic.random_params(n)
    dat = synthetic.mixed_cohort(n, 300, seed=11)                            # 300 rows: beyond the 64 sampled rows
    ro.configure()
    s0 = ro.score(lt, dp, dm, dat, 0.3)
    eng = ro._engine_for(dat)
    sampled = set(np.linspace(0, dat.shape[0] - 1, 64).astype(int).tolist())
    row = next(r for r in range(150, 300) if r not in sampled)               # not one of the 64 evenly spaced samples
    old = dat[row].copy()
    dat[row] = dat[(row + 7) % dat.shape[0]]
    assert not np.array_equal(old, dat[row])
    s1 = ro.score(lt, dp, dm, dat, 0.3)
    assert ro._engine_for(dat) is eng and eng.h                              # same engine, still alive
    ref = ro.score(lt, dp, dm, dat.copy(), 0.3)
    np.testing.assert_allclose(s1, ref, rtol=1e-13)
    assert abs(float(s1) - float(s0)) > 0
    big = synthetic.mixed_cohort(n, 40000, seed=12)                          # 600 KB: sampled guard unless strict
    ro.configure(strict_guard=True)
    t0 = ro.score(lt, dp, dm, big, 0.3)
    sampled = set(np.linspace(0, big.shape[0] - 1, 64).astype(int).tolist())
    r2 = next(r for r in range(12345, 13000) if r not in sampled and not np.array_equal(big[r], big[54]))
    big[r2] = big[54]
    t1 = ro.score(lt, dp, dm, big, 0.3)
    np.testing.assert_allclose(t1, ro.score(lt, dp, dm, big.copy(), 0.3), rtol=1e-13)
    ro.configure()
