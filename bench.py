#!/usr/bin/env python3
"""Benchmark of the metMHN hot path on MI355X.

One "step" = one full-cohort evaluation of regularized_optimization.score_and_grad_reg
(value + full gradient, perc_met = 0.5, symmetric_penal, lambda = 1e-3): parameter upload,
every kernel of the likelihood/gradient pipeline, the RCCL all-reduce of the partial sums
when N > 1 (inside the library, on the engine's stream), the download and the host-side
penalty.  Workload at every N: BASELINE.json configs[2] per GPU - synthetic "full-k" cohort,
n = 20 events, 5 000 paired patients per GPU (k = 20 -> 2^20-state vectors, 8 MiB fp64
each), weak scaling; the cohort is resident in HBM before the timed region.

    python bench.py                      # 1 GPU
    python bench.py --gpus N             # starts N ranks itself (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # the driver's form

Rank 0 prints ONE JSON line.  `value` counts 5 000-patient cohort evaluations per second
over all ranks (N GPUs evaluate an N x 5 000 patient cohort per step).  `roofline` is the
dominant kernel OF THE TIMED REGION (k_wsolve, forward instantiation), timed with HIP events
on the engine's stream inside that region; every figure in the line can be recomputed from
`alg_bytes_per_launch`, `avg_launch_ms` and the CSVs under profiles/.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the engine's four streams on hardware queues of their own: metmhn_amd/_lib.py sets the same default when it loads the library;
# here it is in place before anything can have initialised the HIP runtime)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r5_traffic.json")
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r5_counters.json")
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector: half of the 157.3 TFLOP/s fp32 vector rate of MI355X_MICROARCH.md


def csrc_sha16():
    """Fingerprint of the kernel sources: the offline PMC figures are only quoted for the tree they were taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "metmhn_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=20, help="events (k = n active bits per patient)")
    ap.add_argument("--patients", type=int, default=5000, help="patients per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--workload", default="full-k", choices=["full-k", "luad", "luad28"],
                    help="full-k: BASELINE configs[2]; luad: the LUAD-reduced cohort of configs[0] (small-k regime); luad28: the "
                         "28-event LUAD cohort the reference's examples/analysis.py fits (heterogeneous, k up to 21)")
    ap.add_argument("--min-seconds", type=float, default=5.0,
                    help="the timed region is repeated in whole multiples of --steps until it has lasted this long (steps_run in "
                         "the line; value and ms_per_step stay per step)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="patients timed on the reference-structure CPU baseline (-1: two per core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the kronvec / stream / small-cohort legs")
    ap.add_argument("--kronvec-batch", type=int, default=64)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4],
                    help="BASELINE.json configs[]: 2 = n 20, 5 000 patients per GPU (default, weak scaling); 3 = n 20, 50 000 "
                         "patients over the GPUs (6 250 per GPU at --gpus 8); 4 = n 25, 10 000 patients over the GPUs, fp32")
    a = ap.parse_args()
    if a.config == 3:
        a.n, a.patients, a.dtype = 20, max(1, 50000 // a.gpus), "f64"
    elif a.config == 4:
        a.n, a.patients, a.dtype = 25, max(1, 10000 // a.gpus), "f32"
    return a


def host_cores():
    """CPU share of this process: affinity, capped by the cgroup quota and by 16 (one GPU's share of the box)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(cores, 16))


_CPU_SCRIPT = """
import sys, time, json, numpy as np
sys.path.insert(0, {root!r})
from oracle import cref                      # checker / baseline only
from metmhn_amd import synthetic
n, sample, cores = {n}, {sample}, {cores}
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, {patients}, seed=2000 + n)
cref.load()
t0 = time.perf_counter()
cref.patients(lt, dp, dm, dat[:sample], with_grad=True, threads=cores, patient_parallel=True)
print(json.dumps(dict(dt=time.perf_counter() - t0)))
"""


_CPU_FAST_SCRIPT = """
import sys, time, json, numpy as np
sys.path.insert(0, {root!r})
from oracle import cref                      # checker / baseline only
from metmhn_amd import synthetic
n, sample, cores = {n}, {sample}, {cores}
lt, dp, dm = synthetic.random_params(n)
dat = synthetic.full_k_cohort(n, {patients}, seed=2000 + n)
cref.load_fast()
cref.fast_patients(lt, dp, dm, dat[:cores], threads=cores)
t0 = time.perf_counter()
res = cref.fast_patients(lt, dp, dm, dat[:sample], threads=cores)
dt = time.perf_counter() - t0
np.savez({out!r}, lp=res[0], g=res[1], gp=res[2], gm=res[3])
print(json.dumps(dict(dt=dt)))
"""


def _cpu_env(cores):
    return dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_WAIT_POLICY="PASSIVE", GOMP_SPINCOUNT="0", OMP_PROC_BIND="false")


def cpu_baseline_optimised(n, patients, budget_s=300):
    """oracle/metmhn_fast.c: the same mathematics as the GPU engine (closed-form rates, substitution solves,
    class-marginal gradients) on the host cores ("so the speed-up is not flattered by a deliberately slow
    baseline").  Sample: 32 patients per core of the same cohort; its per-patient results are handed back so that
    the engine's evaluation of the bench cohort is CHECKED against them."""
    cores = host_cores()
    sample = min(patients, 32 * cores)
    out = os.path.join(tempfile.gettempdir(), f"mmhn_bench_cpu_{os.getpid()}.npz")
    code = _CPU_FAST_SCRIPT.format(root=ROOT, n=n, sample=sample, cores=cores, patients=patients, out=out)
    try:
        res = subprocess.run([sys.executable, "-c", code], env=_cpu_env(cores), capture_output=True, text=True, timeout=budget_s)
        dt = json.loads(res.stdout.strip().splitlines()[-1])["dt"]
        ref = dict(np.load(out))
        os.unlink(out)
    except Exception as exc:
        return dict(value=None, unit="evals/s", cores=cores, kind="port",
                    sample=f"optimised CPU variant did not finish within {budget_s} s ({type(exc).__name__})"), None
    return dict(value=(sample / dt) / 5000.0, unit="evals/s", cores=cores, kind="port",
                sample=f"{sample} of the {patients} patients of the same n={n} cohort (log-lik + gradient, gather "
                       f"formulation with substitution solves = the engine's own algorithm in C, OpenMP over patients "
                       f"on {cores} threads), {dt:.2f} s wall, extrapolated linearly to 5000 patients"), ref


def cpu_baseline(n, patients, sample, budget_s=600):
    """C restatement of the reference pass structure (oracle/metmhn_ref.c) on the host cores, in a fresh process
    (own OpenMP runtime, passive waits, thread count = CPU share), on a bounded sample of the same cohort: two
    patients per core, OpenMP over patients."""
    cores = host_cores()
    if sample < 0:
        sample = 2 * cores
    code = _CPU_SCRIPT.format(root=ROOT, n=n, sample=sample, cores=cores, patients=patients)
    try:
        res = subprocess.run([sys.executable, "-c", code], env=_cpu_env(cores), capture_output=True, text=True, timeout=budget_s)
        dt = json.loads(res.stdout.strip().splitlines()[-1])["dt"]
    except Exception as exc:                      # report the failure instead of hanging the bench
        return dict(value=None, unit="evals/s", cores=cores, kind="port",
                    sample=f"CPU baseline did not finish within {budget_s} s ({type(exc).__name__})")
    return dict(value=(sample / dt) / 5000.0, unit="evals/s", cores=cores, kind="port",
                sample=f"{sample} of the {patients} patients of the same n={n} cohort ({sample // cores} per core; "
                       f"log-lik + gradient, reference pass structure, OpenMP over patients on {cores} threads), "
                       f"{dt:.1f} s wall, extrapolated linearly to 5000 patients")


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: this parent never touches the GPU; it starts the N ranks as a
    child `torch.distributed.run` (127.0.0.1 rendezvous, free port) and relays rank 0's JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def luad28_cohort():
    """The 28-event LUAD cohort (4 852 x 59 int8, tests/golden/luad28.npz: derived from the reference's data/luad CSVs as
    examples/analysis.py:49-72 does, by tests/tools/make_golden_luad.py luad28) at the reference's published parameters
    (results/luad/luad_g14_cv_20muts_8cnvs.csv)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "luad28.npz"))
    return (g["dat"], g["fit_theta"], g["fit_dp"], g["fit_dm"], float(g["perc_met"]),
            "28-event LUAD cohort of examples/analysis.py (tests/golden/luad28.npz), published parameters", g)


def luad_cohort(n_default=20):
    """The LUAD-reduced cohort (4 852 x 43 int8, tests/golden/luad_indep.npz: data derived from the reference's
    data/luad CSVs by tests/tools/make_golden_luad.py) and the `indep` start; a synthetic cohort of the same
    composition if the fixture is absent."""
    path = os.path.join(ROOT, "tests", "golden", "luad_indep.npz")
    if os.path.exists(path):
        g = np.load(path)
        return g["dat"], g["indep_theta"], g["indep_dp"], g["indep_dm"], float(g["perc_met"]), "LUAD-reduced (tests/golden/luad_indep.npz)"
    from metmhn_amd import synthetic
    lt, dp, dm = synthetic.random_params(n_default)
    return synthetic.mixed_cohort(n_default, 4852, seed=3), lt, dp, dm, 0.2, "synthetic mixed cohort, 4 852 rows (fixture absent)"


def real_cohort_leg(ro, note, reps=20):
    """One evaluation of the 28-event LUAD cohort: ms per evaluation (value + gradient, score only), the result against the
    fixture, and a per-kernel breakdown of ONE evaluation timed with HIP events around the solve / class-marginal launches (a
    second engine with MMHN_TIME_KERNELS=1: the events cost host time, so the breakdown's evaluation is slower than the
    untimed one quoted as ms_per_eval)."""
    from metmhn_amd import Engine
    dat, lt, dp, dm, pm, name, g = luad28_cohort()
    n = (dat.shape[1] - 3) // 2
    sp = np.concatenate((np.asarray(lt).flatten(), dp, dm))
    v, gr = ro.score_and_grad_reg(sp, dat, pm, ro.symmetric_penal, 1e-3)
    err_v = abs(float(v) - float(g["fit_reg_value"])) / abs(float(g["fit_reg_value"]))
    err_g = float(np.max(np.abs(gr - g["fit_reg_grad"])) / np.max(np.abs(g["fit_reg_grad"])))
    ts, tsc = [], []
    for _ in range(reps):
        t1 = time.perf_counter()
        ro.score_and_grad_reg(sp, dat, pm, ro.symmetric_penal, 1e-3)
        ts.append(time.perf_counter() - t1)
    for _ in range(reps):
        t1 = time.perf_counter()
        ro.score_reg(sp, dat, pm, ro.symmetric_penal, 1e-3)
        tsc.append(time.perf_counter() - t1)
    k = dat[dat[:, -1] == 3][:, :2 * n + 1].sum(1)
    out = {"workload": name, "rows": int(dat.shape[0]), "paired_rows": int((dat[:, -1] == 3).sum()), "k_max": int(k.max()),
           "paired_rows_k_ge_13": int((k >= 13).sum()), "seeded_states": float((2.0 ** (k - 1)).sum()),
           "ms_per_eval_with_grad": float(np.median(ts) * 1e3), "ms_per_eval_score_only": float(np.median(tsc) * 1e3),
           "evals_per_s": float(1.0 / np.median(ts)),
           "check": {"against": "tests/golden/luad28.npz (oracle/metmhn_ref.c on every row)", "rel_err_value": err_v,
                     "rel_err_grad_max": err_g, "tolerance": 1e-7}}
    os.environ["MMHN_TIME_KERNELS"] = "1"
    try:
        e = Engine(n)
        e.set_cohort(dat)
        e.cohort_sums(lt, dp, dm)
        e.reset_counters()
        nrep = 5
        for _ in range(nrep):
            e.cohort_sums(lt, dp, dm)
        c = e.counters()
        e.close()
    finally:
        del os.environ["MMHN_TIME_KERNELS"]
    names = {"csolve_fwd": "k_csolve<double,false> (joint forward solve: tiles of every paired patient in one cooperative launch)",
             "csolve_adj": "k_csolve<double,true> (joint adjoint solve)",
             "pclass": "k_pclass<double> (class marginals as work items: a large patient is several workgroups)",
             "other_solve": "k_csolve / k_tsolve (single-tumour spaces of more than a tile: forward + adjoint)",
             "psolve_fwd": "k_wsolve / k_psolve2 forward (not used by this cohort)", "psolve_adj": "k_wsolve / k_psolve2 adjoint"}
    bd = {}
    for key, label in names.items():
        if c[key]["launches"]:
            bd[key] = {"kernel": label, "ms_per_eval": c[key]["ms"] / nrep, "launches_per_eval": c[key]["launches"] / nrep,
                       "alg_bytes_per_eval": c[key]["alg_bytes"] / nrep}
    out["kernel_breakdown"] = bd
    out["kernel_breakdown_eval_ms"] = c["eval_ms"] / nrep
    return out


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a))
    T0 = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    import torch
    import torch.distributed as dist
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from metmhn_amd import synthetic
    import metmhn_amd.regularized_optimization as ro
    ro.configure(device=local, dtype=a.dtype, shard=True)

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)

    if a.workload in ("luad", "luad28"):
        dat, lt, dp, dm, perc_met, wl_name = luad_cohort() if a.workload == "luad" else luad28_cohort()[:6]
        n = (dat.shape[1] - 3) // 2
        dat = np.vstack([dat] * world)
        unit_rows = dat.shape[0] // world
    else:
        n = a.n
        lt, dp, dm = synthetic.random_params(n)
        # global cohort = `world` blocks of `patients` rows; the engine of rank r keeps the LPT shard r
        dat = np.vstack([synthetic.full_k_cohort(n, a.patients, seed=2000 + n + 7919 * r) for r in range(world)])
        perc_met, unit_rows = 0.5, a.patients
        wl_name = (f"synthetic full-k cohort, n={n} events, {a.patients} paired patients per GPU, 2^{n}-state vectors, "
                   f"{a.dtype} (BASELINE.json configs[{a.config}]{' per GPU' if a.config == 2 else f' over {world} GPUs'}; "
                   f"value counts evaluations of 5 000 patients)")
    N = n + 1
    params = np.concatenate((np.asarray(lt).flatten(), dp, dm))
    esz = 8 if a.dtype == "f64" else 4

    note("building cohort layout")
    eng = ro._engine_for(dat)                    # uploads and lays out this rank's shard (outside the timed region)
    note(f"cohort resident (in-library RCCL communicator: {bool(getattr(eng, '_device_comm', False))})")

    def step():
        return ro.score_and_grad_reg(params, dat, perc_met, ro.symmetric_penal, 1e-3)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
        note("warmup step done")
    eng.reset_counters()
    # The timed region: EXACTLY a.steps steps between two fences - repeated (whole multiples of a.steps, every rank the same
    # count) until it has lasted --min-seconds, so that a driver sampling the GPU from outside sees it busy for seconds, not
    # for the 0.25 s of five steps.  Every figure of the line stays per step.
    reps = 1
    if a.min_seconds > 0:
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps = max(1, int(np.ceil(a.min_seconds / max(float(t.item()) * a.steps, 1e-9))))
        eng.reset_counters()
    steps_run = reps * a.steps
    fence()
    t0 = time.perf_counter()
    for _ in range(steps_run):
        val, grad = step()
    fence()
    dt = time.perf_counter() - t0
    steps_asked, a.steps = a.steps, steps_run                  # (every per-step figure below divides by the steps that ran)
    rank_ms = [dt / a.steps * 1e3]
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                       # per-rank times: a slow rank shows up in the line
        rank_ms = [float(x.item()) / a.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cnt = eng.counters()
    note(f"timed region done: {dt:.3f} s for {a.steps} steps")

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        unit = 5000.0 if a.workload == "full-k" else float(unit_rows)
        value = world * (unit_rows / unit) * a.steps / dt
        try:
            traffic = json.load(open(TRAFFIC_FILE))
        except Exception:
            traffic = {}
        tsrc = traffic.get("source", {})
        if tsrc.get("csrc_sha16") != csrc_sha16():      # PMC passes of another tree: not quoted
            traffic = {}
        traffic_source = (f"recorded offline: profiles/r5_traffic.json (git {tsrc.get('git_head', '?')}, csrc {tsrc.get('csrc_sha16', '?')}; "
                          "rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes over this same command; "
                          "2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)")
        traffic_ok = a.workload == "full-k" and n == 20 and a.dtype == "f64" and a.patients == 5000 and bool(traffic)
        ek = traffic.get("eval_kernels", {}) if traffic_ok else {}
        try:
            counters = json.load(open(COUNTERS_FILE))
        except Exception:
            counters = {}
        if counters.get("source", {}).get("csrc_sha16") != csrc_sha16() or not traffic_ok:
            counters = {}

        # ---- measured stream bandwidth of this box (next to the nominal 8 TB/s)
        stream = {}
        if not a.no_extras:
            stream = {"copy_GBps": eng.bench_stream(1 << 30, 10, "copy"), "triad_GBps": eng.bench_stream(1 << 30, 10, "triad"),
                      "what": "16 B/lane copy b=a / triad a=b+s*c over 1 GiB arrays, HIP events, bytes moved / time"}
            note("stream leg done")

        # ---- per-kernel roofline of the timed region (HIP events on the engine's stream around every launch)
        def kern(name, label, alg_note):
            c = cnt[name]
            if c["launches"] == 0:
                return None
            avg_ms = c["ms"] / c["launches"]
            per_launch = c["alg_bytes"] / c["launches"]
            ach = per_launch / avg_ms / 1e6
            tr = ek.get(name, {})
            o = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                 "traffic": tr.get("bytes_per_launch"), "kernel": label, "launches": int(c["launches"]),
                 "avg_launch_ms": avg_ms, "alg_bytes_per_launch": per_launch, "alg_bytes": alg_note,
                 "ms_per_step": c["ms"] / a.steps}
            if tr:
                o["traffic_source"] = traffic_source
                o["traffic_over_alg"] = tr["bytes_per_launch"] / per_launch
                o["traffic_GBps"] = tr["bytes_per_launch"] / avg_ms / 1e6
            if stream:
                o["frac_of_measured_copy"] = ach / stream["copy_GBps"]
            # what bounds the kernel besides HBM (solves only): useful fp64 FMAs of the substitution = one per existing move
            # of every seeded state + one for the diagonal = 2^K (K / 2 + 1) per patient (K = k - 1 index bits of the seeded
            # half), against the fp64 vector peak; instruction counters from a separate PMC pass (profiles/r5_counters.json)
            if name in ("psolve_fwd", "psolve_adj") and a.workload == "full-k":
                K = n - 1
                fmas = per_launch / esz * (K / 2.0 + 1.0)          # (states written by the launch) x (K / 2 + 1)
                pk = FP64_VECTOR_PEAK_TFLOPS * (1.0 if a.dtype == "f64" else 2.0)
                tag = "fp64" if a.dtype == "f64" else "fp32"
                o[f"useful_{tag}_fma_per_launch"] = fmas
                o[f"{tag}_fma_TFLOPs"] = 2.0 * fmas / avg_ms / 1e9
                o[f"{tag}_fma_frac"] = o[f"{tag}_fma_TFLOPs"] / pk
                ck = counters.get("kernels", {}).get(name)
                if ck:
                    states64 = per_launch / esz / 64.0
                    o["valu_wave_insts_per_64_states"] = ck["SQ_INSTS_VALU"] / states64
                    o["wave_insts_per_64_states"] = sum(ck.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                                                                                 "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) / states64
                    o["waves_per_simd"] = ck.get("waves_per_simd")
                    o["counters_source"] = f"recorded offline: profiles/r5_counters.json (git {counters['source'].get('git_head', '?')}), rocprofv3 --pmc SQ_INSTS_*"
            return o

        T = "double" if a.dtype == "f64" else "float"
        wsolve = os.environ.get("MMHN_WSOLVE", "1") != "0"
        ksolve = "k_wsolve" if wsolve else "k_psolve2"
        rf_fwd = kern("psolve_fwd", f"{ksolve}<{T},false> (forward substitution solve of the joint problems"
                      + (", window layout: a chain of patients per workgroup)" if wsolve else ", one workgroup per patient)"),
                      "solution written once: the seeded half, 2^(k-1) x sizeof(dtype) per patient")
        rf_adj = kern("psolve_adj", f"{ksolve}<{T},true> (adjoint substitution solve)", "solution written once")
        rf_marg = kern("pclass", f"{'k_wclass' if wsolve else 'k_pclass'}<{T}> (class marginals of pi (x) q)", "pi and q_J read once (seeded halves)")
        rf_other = kern("other_solve", "k_csolve / k_tsolve (tile solves of the single-tumour problems of more than a tile: one cooperative launch each)",
                        "per tile: solution written once (+ dense rhs / lidg vector reads)")
        # the patient shards: rows and LPT cost (2^k (k + 1), distributed.patient_cost) per rank
        from metmhn_amd import distributed as D
        shard = None
        try:
            parts = D.shard_rows(dat, world) if world > 1 else [np.arange(dat.shape[0])]
            cost = D.patient_cost(dat)
            loads = [float(cost[p_].sum()) for p_ in parts]
            shard = {"rows_per_rank": [int(len(p_)) for p_ in parts], "cost_per_rank": loads,
                     "cost_imbalance_max_over_mean": max(loads) / (sum(loads) / len(loads)) if sum(loads) > 0 else 1.0}
        except Exception as exc:
            shard = {"error": repr(exc)}
        live_bytes = (cnt["psolve_fwd"]["alg_bytes"] / max(cnt["psolve_fwd"]["launches"], 1)) if rf_fwd else None
        dominant = rf_fwd or rf_other or {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
        out = {
            "metric": "full-cohort log-lik+grad evals/sec at n=20 events; kronvec HBM GB/s",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": steps_asked, "steps_run": steps_run, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if a.config == 2 else "strong", "vs_baseline": None,
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "dtype": a.dtype, "data": "synthetic" if a.workload == "full-k" else "LUAD genotypes (derived fixture), indep() / published parameters",
            "config": {"workload": wl_name, "patients_total": int(dat.shape[0]), "perc_met": perc_met, "penalty": "symmetric_penal 1e-3",
                       "parallelism": f"patient-shard x{world}, one RCCL all-reduce of {1 + N * N + 2 * N} f64 per eval (EM / NM pre-combined on the device) "
                                      f"({'inside the library on the engine stream' if getattr(eng, '_device_comm', False) else 'none (1 rank)' if world == 1 else 'torch.distributed'})",
                       "rccl_ranks": cnt.get("comm_ranks", 0), "rccl_rank": cnt.get("comm_rank", -1), "shards": shard,
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "solver": os.environ.get("MMHN_SOLVER", "substitution, per-problem dispatch (k_wsolve: window layout, chains of patients; k_psolve2: one workgroup "
                                                               "per patient; k_csolve: all remaining tiles in one cooperative launch)"),
                       "objective_value": float(val), "grad_norm": float(np.linalg.norm(grad))},
            # dominant kernel of the timed step
            "roofline": dominant,
            "roofline_adjoint": rf_adj, "roofline_marginals": rf_marg, "roofline_other_solves": rf_other,
            "measured_stream": stream or None,
        }
        if live_bytes:
            # compulsory traffic of the substitution formulation per evaluation: write pi and q_J once, read each once
            floor = 4.0 * live_bytes
            out["eval_floor"] = {"bytes_per_step": floor, "floor_ms_at_peak": floor / HBM_PEAK_GBPS / 1e6,
                                 "frac_of_floor": floor / HBM_PEAK_GBPS / 1e6 / ms_per_step,
                                 "note": "4 x live bytes (write pi, q_J; read each once) / 8 TB/s over the measured step"}
            if stream:
                out["eval_floor"]["frac_of_floor_measured_copy"] = floor / stream["copy_GBps"] / 1e6 / ms_per_step
            # "report achieved fraction against B_pat anyway and say which solver ran" - B_pat prices one paired
            # log-lik + gradient with the reference's k+1 Jacobi sweeps per solve (likelihood.py:231-262):
            # [(k+1)*3 + (k+1)*4 + 4] * 2^k * s  (forward solve, transposed solve, 1/diag build, one gradient pass)
            if a.workload == "full-k":
                kk = n
                b_pat = ((kk + 1) * 3 + (kk + 1) * 4 + 4) * (2.0 ** kk) * esz
                b_step = b_pat * unit_rows
                out["eval_floor"]["B_pat_equivalent"] = {
                    "bytes_per_patient": b_pat, "bytes_per_step": b_step, "equivalent_GBps": b_step / ms_per_step / 1e6,
                    "frac_of_peak": b_step / ms_per_step / 1e6 / HBM_PEAK_GBPS,
                    "note": "Jacobi pricing of SURVEY 8(d) over the measured step; solver that ran = substitution "
                            "(each state computed once), so this is an equivalence figure, not moved bytes"}
        if ek:
            tot = sum(v.get("bytes_per_launch", 0) * v.get("launches_per_step", 1) for v in ek.values())
            out["eval_traffic"] = {"bytes_per_step": tot, "over_floor": tot / (4.0 * live_bytes) if live_bytes else None,
                                   "source": traffic_source}

        if not a.no_extras and a.workload == "full-k":
            # ---- metric 2: batched kronvec Q_off p on `kb` resident 2^k vectors (working set > Infinity
            # Cache), HIP events on the engine's stream around 20 launches; B_kv = 2 * 2^k * s per vector
            st = dat[0, :2 * n + 1]
            kb = a.kronvec_batch
            V = (2 ** int(st.sum())) * esz
            tkv = traffic.get("kronvec", {}) if (n == 20 and kb == 64 and a.dtype == "f64") else {}
            kv = {}
            for name, tr, jac, mult in (("kronvec", 0, 0, 2), ("kronvec_T", 1, 0, 2), ("jacobi_step", 0, 1, 4)):
                # the timed launch = mmhn_kronvec_batched's launch: y = Q_off p into a NaN-filled y, every tile of every
                # vector (tiles where Q_off has no entries are zeroed inside the launch)
                ms, live, tot = eng.bench_kronvec(lt, st, kb, 20, transpose=tr, jacobi=jac, tiles=True)
                alg = mult * V * kb                              # read p once, write y once
                tile_b = V * kb / tot                            # bytes of one tile of one vector
                # what the launch has to move: read the tiles of p that carry values, write all of y (the Jacobi step also
                # reads 1/diag and the right-hand side of every tile: a tile without entries still gets lidg * rhs)
                live_b = (live * tile_b + tot * tile_b) if not jac else (live * tile_b + 3 * tot * tile_b)
                kv[name] = {"ms_per_launch": ms, "batch": kb, "tiles_per_launch": tot, "tiles_with_entries": live,
                            "alg_bytes_per_launch": alg, "alg_GBps": alg / ms / 1e6,
                            "frac_of_peak": alg / ms / 1e6 / HBM_PEAK_GBPS,
                            "live_bytes_per_launch": live_b, "live_GBps": live_b / ms / 1e6,
                            "frac_of_peak_live": live_b / ms / 1e6 / HBM_PEAK_GBPS}
                if name in tkv:
                    kv[name]["traffic"] = tkv[name]["bytes_per_launch"]
                    kv[name]["moved_GBps"] = tkv[name]["bytes_per_launch"] / ms / 1e6
                    kv[name]["frac_of_peak_moved"] = tkv[name]["bytes_per_launch"] / ms / 1e6 / HBM_PEAK_GBPS
                    kv[name]["traffic_source"] = traffic_source
                if stream:
                    kv[name]["frac_of_measured_copy_live"] = live_b / ms / 1e6 / stream["copy_GBps"]
            out["kronvec"] = kv
            note("kronvec leg done")
            # ---- small-k regime (BASELINE configs[0]): one LUAD-sized evaluation, launch-bound
            try:
                sdat, slt, sdp, sdm, spm, sname = luad_cohort()
                sp = np.concatenate((np.asarray(slt).flatten(), sdp, sdm))
                ro.score_and_grad_reg(sp, sdat, spm, ro.symmetric_penal, 1e-3)
                ts = []
                for _ in range(10):
                    t1 = time.perf_counter()
                    ro.score_and_grad_reg(sp, sdat, spm, ro.symmetric_penal, 1e-3)
                    ts.append(time.perf_counter() - t1)
                tsc = []
                for _ in range(10):
                    t1 = time.perf_counter()
                    ro.score_reg(sp, sdat, spm, ro.symmetric_penal, 1e-3)
                    tsc.append(time.perf_counter() - t1)
                out["small_cohort"] = {"workload": sname, "rows": int(sdat.shape[0]), "ms_per_eval_with_grad": float(np.median(ts) * 1e3),
                                       "ms_per_eval_score_only": float(np.median(tsc) * 1e3)}
            except Exception as exc:
                out["small_cohort"] = {"error": repr(exc)}
            note("small-cohort leg done")
            # ---- the reference's real workload: the 28-event LUAD cohort examples/analysis.py fits (heterogeneous: paired rows
            # with k = 1 .. 21, single-tumour spaces of up to 17 bits) - per-problem dispatch + the cooperative tile launch
            try:
                out["real_cohort"] = real_cohort_leg(ro, note)
            except Exception as exc:
                out["real_cohort"] = {"error": repr(exc)}
            note("real-cohort leg done")
        if world == 1 and not a.no_cpu and a.workload == "full-k":
            out["cpu_baseline"] = cpu_baseline(n, a.patients, a.cpu_sample)
            note("cpu baseline (reference structure) done")
            opt, ref = cpu_baseline_optimised(n, a.patients)
            out["cpu_baseline_optimised"] = opt
            if ref is not None:
                # parity of the timed workload itself: the engine's per-patient results for the rows the CPU just did
                m = ref["lp"].shape[0]
                lp, g, gp, gm = eng.patient_grads(lt, dp, dm)
                def rel(x, y):
                    return float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300))
                errs = {"lp": rel(lp[:m], ref["lp"]), "d_theta": rel(g[:m], ref["g"]), "d_dp": rel(gp[:m], ref["gp"]),
                        "d_dm": rel(gm[:m], ref["gm"])}
                out["checked_patients"] = int(m)
                out["max_rel_err"] = max(errs.values())
                out["check"] = {"against": "oracle/metmhn_fast.c on the host (fp64), first rows of the bench cohort", "errors": errs,
                                "tolerance": 1e-6 if a.dtype == "f64" else 1e-2}
            note("optimised CPU baseline + cross-check done")
        if world == 1 and not a.no_extras and a.workload == "full-k" and a.config == 2 and a.n == 20:
            # ---- BASELINE configs[4] shape (n = k = 25, fp32, 128 MiB vectors) on a bounded sample: one resident batch of
            # 768 patients (three per CU), in a child process after this one's engines are gone - a reported figure beside the
            # headline, not part of `value`
            try:
                ro.invalidate()
                import gc
                gc.collect()
                cmd = [sys.executable, os.path.abspath(__file__), "--n", "25", "--dtype", "f32", "--patients", "768", "--steps", "2",
                       "--warmup", "1", "--min-seconds", "0", "--no-cpu", "--no-extras"]
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                line = json.loads(res.stdout.strip().splitlines()[-1])
                out["config4_batch"] = {
                    "workload": "synthetic full-k cohort, n=25 events, 768 paired patients (one workspace batch of BASELINE configs[4]), "
                                "2^25-state vectors, f32", "patients": 768, "ms_per_step": line["ms_per_step"],
                    "ms_per_patient": line["ms_per_step"] / 768.0, "dtype": "f32",
                    "roofline": {k: line["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "alg_bytes_per_launch")},
                    "roofline_marginals": {k: line.get("roofline_marginals", {}).get(k) for k in ("kernel", "frac", "avg_launch_ms")}}
            except Exception as exc:
                out["config4_batch"] = {"error": repr(exc)}
            note("configs[4]-shaped batch done")
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
