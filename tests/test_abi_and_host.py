"""CPU-side checks: the C-ABI library loads and exports every symbol include/metmhn_amd.h
declares; host logic (sharding, all-reduce combination, penalties, synthetic generators)."""
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from metmhn_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "metmhn_amd.h")).read()
    declared = set(re.findall(r"\b(mmhn_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    bound = set(_lib.SIGNATURES) | set(_lib.OTHER_SYMBOLS)
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from metmhn_amd import Engine
    with pytest.raises(RuntimeError, match="no .*device|no HIP device"):
        Engine(3)


def test_penalties_match_oracle(golden):
    import metmhn_amd.regularized_optimization as ro
    g = golden("cohorts")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        lt, dp, dm = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"]
        params = np.concatenate((lt.flatten(), dp, dm))
        pen, pen_ = ro.symmetric_penal(params, lt.shape[0])
        np.testing.assert_allclose(pen, g[pre + "pen"], rtol=1e-13)
        np.testing.assert_allclose(pen_, g[pre + "pen_grad"], rtol=1e-13, atol=1e-15)


def _oracle_sums(lt, dp, dm, dat):
    """What mmhn_cohort_sums returns, computed by the CPU oracle (test infrastructure)."""
    from oracle import metmhn_oracle as O
    N = lt.shape[0]
    s = np.zeros(4 + 2 * N * N + 3 * N)
    g_em, g_nm = np.zeros((N, N)), np.zeros((N, N))
    p_em, p_nm, m_em = np.zeros(N), np.zeros(N), np.zeros(N)
    for row in dat:
        lp, g, a, b, is0 = O.patient_grad(lt, dp, dm, row)
        if is0:
            s[1] += lp; g_nm += g; p_nm += a
        else:
            s[0] += lp; g_em += g; p_em += a; m_em += b
    s[2] = dat[:, -3].sum()
    s[3] = dat.shape[0]
    s[4:] = np.concatenate((g_em.ravel(), g_nm.ravel(), p_em, p_nm, m_em))
    return s


def test_combine_sums_matches_reference_weighting(golden):
    from metmhn_amd import distributed as D
    g = golden("cohorts")
    for c in (0, 1, 2):
        pre = f"c{c}_"
        lt, dp, dm, dat = g[pre + "log_theta"], g[pre + "log_d_p"], g[pre + "log_d_m"], g[pre + "dat"]
        s, gth, gdp, gdm = D.combine_sums(_oracle_sums(lt, dp, dm, dat), lt.shape[0], float(g[pre + "perc_met"]))
        np.testing.assert_allclose(s, g[pre + "score"], rtol=1e-12)
        np.testing.assert_allclose(gth, g[pre + "d_th"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gdp, g[pre + "d_dp"], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(gdm, g[pre + "d_dm"], rtol=1e-10, atol=1e-13)


def test_shard_rows_partition():
    from metmhn_amd import distributed as D, synthetic
    dat = synthetic.mixed_cohort(6, 101, seed=1)
    for w in (1, 2, 3, 8):
        parts = D.shard_rows(dat, w)
        allr = np.sort(np.concatenate(parts))
        assert np.array_equal(allr, np.arange(101))
        cost = D.patient_cost(dat)
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() <= loads.mean() + cost.max() + 1e-9          # LPT bound


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metmhn_amd import distributed as D
    g = np.load(os.path.join(ROOT, "tests", "golden", "cohorts.npz"))
    lt, dp, dm, dat = g["c1_log_theta"], g["c1_log_d_p"], g["c1_log_d_m"], g["c1_dat"]
    rows = D.shard_rows(dat, world)[rank]
    local = _oracle_sums(lt, dp, dm, dat[rows])           # stands in for Engine.cohort_sums of this rank's shard
    tot = D.allreduce_sums(local)
    res = D.combine_sums(tot, lt.shape[0], float(g["c1_perc_met"]))
    if rank == 0:
        q.put([np.asarray(r) for r in res])
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_reproduces_full_cohort(golden):
    """world_size-2 patient sharding + one all-reduce == the single-process result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = golden("cohorts")
    np.testing.assert_allclose(res[0], g["c1_score"], rtol=1e-12)
    np.testing.assert_allclose(res[1], g["c1_d_th"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[2], g["c1_d_dp"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[3], g["c1_d_dm"], rtol=1e-10, atol=1e-13)


def _worker_wsums(rank, world, port, q):
    """Pre-combined payload (1 + N^2 + 2N doubles, weight from the GLOBAL counts) through the fixed-order host reduce."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metmhn_amd import distributed as D
    g = np.load(os.path.join(ROOT, "tests", "golden", "cohorts.npz"))
    lt, dp, dm, dat = g["c1_log_theta"], g["c1_log_d_p"], g["c1_log_d_m"], g["c1_dat"]
    N = lt.shape[0]
    rows = D.shard_rows(dat, world)[rank]
    s = _oracle_sums(lt, dp, dm, dat[rows])               # this rank's shard, layout of mmhn_cohort_sums
    w, n_full = D.em_weight(float(dat[:, -3].sum()), float(dat.shape[0]), float(g["c1_perc_met"]))
    em_s, nm_s = s[0], s[1]
    o = 4
    g_em = s[o:o + N * N]; o += N * N
    g_nm = s[o:o + N * N]; o += N * N
    p_em = s[o:o + N]; o += N
    p_nm = s[o:o + N]; o += N
    m_em = s[o:o + N]
    ws = np.concatenate(([w * em_s + nm_s], w * g_em + g_nm, w * p_em + p_nm, w * m_em))    # what k_pack_wsums writes
    a = D.allreduce_sums_fixed_order(ws)
    b = D.allreduce_sums_fixed_order(ws)
    c = D.allreduce_sums(ws)
    assert np.array_equal(a, b)                            # same association every time
    np.testing.assert_allclose(a, c, rtol=1e-14)
    res = D.split_wsums(a, N, n_full)
    if rank == 0:
        q.put([np.asarray(r) for r in res] + [np.asarray(len(ws))])
    dist.destroy_process_group()


def test_two_rank_precombined_payload_fixed_order(golden):
    """the all-reduce payload pre-combined with the global EM / NM weight (1 + N^2 + 2N doubles) and the
    MMHN_REDUCE=host_fixed_order reduction (every partial to every rank, summed in rank order) reproduce the
    single-process result; two reductions of the same partials are bit-identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_wsums, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = golden("cohorts")
    N = g["c1_log_theta"].shape[0]
    assert int(res[4]) == 1 + N * N + 2 * N
    np.testing.assert_allclose(res[0], g["c1_score"], rtol=1e-12)
    np.testing.assert_allclose(res[1], g["c1_d_th"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[2], g["c1_d_dp"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(res[3], g["c1_d_dm"], rtol=1e-10, atol=1e-13)


def _worker_comm_setup(rank, world, port, q, case):
    """collective_init over the store of a 2-rank gloo job (the RCCL set-up itself is stood in for by `init_fn`)."""
    sys.path.insert(0, ROOT)
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.distributed.distributed_c10d import _get_default_store
    from metmhn_amd import distributed as D

    def init_fn():
        if case == "one_fails_early" and rank == 0:
            raise RuntimeError("cannot load RCCL")            # fails BEFORE entering the collective set-up
        if case == "one_fails_early":
            time.sleep(120)                                   # ... while its peer sits in ncclCommInitRank
        if case == "all_fail":
            raise RuntimeError(f"no device on rank {rank}")

    t0 = time.monotonic()
    try:
        out = D.collective_init(init_fn, rank, world, _get_default_store(), timeout=60.0)
        q.put((rank, "returned", out, time.monotonic() - t0))
    except D.CommAbandoned as exc:
        q.put((rank, "abandoned", str(exc), time.monotonic() - t0))
    dist.barrier() if case != "one_fails_early" else None     # (the abandoned job is expected to end; no collective after it)
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["all_ok", "all_fail", "one_fails_early"])
def test_communicator_setup_is_decided_through_the_store(case):
    """a rank that fails before ncclCommInitRank must not leave its peers in it for MMHN_COMM_TIMEOUT.
    The set-up is decided through the torch.distributed store (metmhn_amd/distributed.py: collective_init): all ok -> True on
    every rank; every rank failing quickly -> (False, messages) on every rank (they fall back together); one rank failing
    early while the other is still blocked -> BOTH give up within seconds (CommAbandoned), not after the 60 s timeout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_comm_setup, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    if case == "all_ok":
        assert all(r[1] == "returned" and r[2] == (True, "") for r in res), res
    elif case == "all_fail":
        assert all(r[1] == "returned" and r[2][0] is False and "no device on rank 0" in r[2][1] and "rank 1" in r[2][1] for r in res), res
    else:
        assert all(r[1] == "abandoned" for r in res), res
        assert "cannot load RCCL" in res[1][2]                # the blocked rank names the failure it saw
        assert max(r[3] for r in res) < 15.0, res             # seconds, not the timeout


def test_cohort_cache_guard_notices_in_place_edits():
    from metmhn_amd import regularized_optimization as ro, synthetic
    dat = synthetic.mixed_cohort(5, 300, seed=3)
    c0 = ro._sample_crc(dat)
    dat2 = dat.copy()
    dat2[:, -2] = 0                                        # relabel every row in place
    assert ro._sample_crc(dat2) != c0 and ro._sample_crc(dat.copy()) == c0


def _cv_stub_learn(th0, dp0, dm0, train, m_p_corr, penal, w, opt_v=False):
    return th0 + w, dp0, dm0                    # stands in for the optimizer (GPU-only); depends on lambda


def _cv_stub_score(th, dp, dm, test, m_p_corr):
    from oracle import metmhn_oracle as O
    return O.score(th, dp, dm, test, m_p_corr)


def _cv_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metmhn_amd.Utilityfunctions import cross_val
    import metmhn_amd.regularized_optimization as ro
    g = np.load(os.path.join(ROOT, "tests", "golden", "cohorts.npz"))
    df = cross_val(g["c1_dat"], ro.symmetric_penal, [1e-3, 1e-2, 1e-1], 3, 0.4, key=3, parallel_folds=True,
                   _learn=_cv_stub_learn, _score=_cv_stub_score)
    q.put((rank, df.to_numpy()))
    dist.destroy_process_group()


def test_fold_parallel_cross_val_two_ranks(golden):
    """cross_val with the 3 x 3 independent fits dealt over 2 ranks (gloo) == the sequential loop; every rank
    ends with the full table.  The optimizer / scorer are CPU stand-ins (the engine needs a GPU)."""
    import torch.multiprocessing as mp
    from metmhn_amd import distributed as D
    from metmhn_amd.Utilityfunctions import cross_val
    import metmhn_amd.regularized_optimization as ro
    jobs = [D.fold_jobs(3, 3, r, 2) for r in range(2)]
    assert sorted(jobs[0] + jobs[1]) == [(i, f) for i in range(3) for f in range(3)] and abs(len(jobs[0]) - len(jobs[1])) <= 1
    g = golden("cohorts")
    ref = cross_val(g["c1_dat"], ro.symmetric_penal, [1e-3, 1e-2, 1e-1], 3, 0.4, key=3,
                    _learn=_cv_stub_learn, _score=_cv_stub_score).to_numpy()
    assert np.isfinite(ref).all() and len(np.unique(ref)) == ref.size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(got[r], ref, rtol=1e-13)


def test_synthetic_generators():
    from metmhn_amd import synthetic
    d = synthetic.full_k_cohort(12, 50)
    assert d.shape == (50, 27) and d.dtype == np.int8
    assert (d[:, :-2].sum(axis=1) == 12).all() and (d[:, -1] == 3).all() and set(np.unique(d[:, -2])) <= {0, 1, 2}
    lt, dp, dm = synthetic.random_params(12)
    assert lt.shape == (13, 13) and dp.shape == (13,)
    m = synthetic.mixed_cohort(5, 200, seed=2)
    assert set(np.unique(m[:, -1])) == {0, 1, 2, 3}
    assert (m[m[:, -1] == 0, -3] == 0).all() and (m[m[:, -1] != 0, -3] == 1).all()


def test_indep_matches_reference(golden):
    from metmhn_amd.Utilityfunctions import indep
    g = golden("cohorts")
    for c in range(int(g["n_cases"])):
        th, dp, dm = indep(g[f"c{c}_dat"])
        np.testing.assert_allclose(th, g[f"c{c}_indep_theta"], rtol=1e-13, atol=0)
        assert not dp.any() and not dm.any()


def test_load_cohort_roundtrip(tmp_path):
    """CSV -> dat labelling of examples/analysis.py:49-83 on a hand-made 5-row table."""
    import pandas as pd
    from metmhn_amd.Utilityfunctions import load_cohort, save_params
    ev = pd.DataFrame({"Unnamed: 0": ["a", "b", "c", "d", "e"],
                       "P.X (M)": [1, 0, 0, 1, 1], "M.X (M)": [0, 0, 1, 1, 0],
                       "P.Y (M)": [0, 1, 0, 1, 0], "M.Y (M)": [0, 0, 0, 0, 1],
                       "P.AgeAtSeqRep": [50, 60, "No primary included", 40, 55],
                       "M.AgeAtSeqRep": ["No metastasis included", "No metastasis included", 70, 45, 50],
                       "paired": [0, 0, 0, 1, 1]})
    an = pd.DataFrame({"patientID": ["a", "b", "c", "d", "e"],
                       "metaStatus": ["absent", "present", "isMetastasis", "present", "unknown"]})
    ev.to_csv(tmp_path / "ev.csv", index=False)
    an.to_csv(tmp_path / "an.csv", index=False)
    dat, events = load_cohort(str(tmp_path / "ev.csv"), str(tmp_path / "an.csv"))
    assert events == ["X (M)", "Y (M)", "Seeding"]
    exp = np.array([[1, 0, 0, 0, 0, -99, 0], [0, 0, 1, 0, 1, -99, 1], [0, 1, 0, 0, 1, -99, 2],
                    [1, 1, 1, 0, 1, 1, 3], [1, 0, 0, 1, 1, 2, 3]], dtype=np.int8)
    assert np.array_equal(dat, exp)
    save_params(str(tmp_path / "p.csv"), np.eye(3), np.zeros(3), np.ones(3), events)
    back = pd.read_csv(tmp_path / "p.csv", index_col=0)
    assert back.shape == (5, 3)


def test_bench_line_contract():
    """The committed bench line of the final tree (profiles/) carries every field the driver's contract names,
    with the metric of BASELINE.json, and bench.py parses."""
    import ast
    import glob
    import json
    ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]_bench_line.json")))      # the latest round's line
    d = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == base["metric"] and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and abs(d["value"] - d["n_gpus"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    # round 3: metric 2 with its three byte counts, the Jacobi pricing of SURVEY 8(d), per-rank times, the PMC source stamp
    for name in ("kronvec", "kronvec_T", "jacobi_step"):
        kv = d["kronvec"][name]
        for k in ("ms_per_launch", "frac_of_peak", "frac_of_peak_live", "tiles_per_launch", "tiles_with_entries"):
            assert k in kv, (name, k)
        assert 0 < kv["frac_of_peak_live"] <= kv["frac_of_peak"] < 1 and kv["tiles_with_entries"] <= kv["tiles_per_launch"]
    assert d["eval_floor"]["B_pat_equivalent"]["bytes_per_patient"] == 151 * 2 ** 20 * 8
    assert d["rank_ms_per_step"]["max"] >= d["rank_ms_per_step"]["min"] > 0
    assert "csrc" in r.get("traffic_source", "") and r["traffic"] > r["alg_bytes_per_launch"]
