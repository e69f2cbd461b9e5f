#!/usr/bin/env python3
"""profiles/<tag>_pmc/*_counter_collection.csv -> profiles/<tag>_traffic.json (read by bench.py, marked "recorded offline").

    python scripts/make_traffic_json.py [tag=r3]


The PMC passes are separate runs (scripts/profile_round.sh): FETCH_SIZE and WRITE_SIZE over scripts/eval_only.py 5000
(two evaluations of the bench cohort) and over scripts/kv_only.py (batched kronvec, 64 vectors of 2^20 states).
HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE reports half of the bytes of these
coalesced streams (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on 32 768 single-tile problems: 744 258 KB
reported against 1 478 656 KB read), WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os

import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r4"
PMC = os.path.join(ROOT, "profiles", f"{TAG}_pmc")
sys.path.insert(0, ROOT)


def per_launch(tag, counter):
    """{kernel short name: mean counter value per launch (KB)} over the second half of the launches (warm-up dropped)"""
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(PMC, f"{tag}*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"].split("(")[0].replace("void mmhn::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in vals.items()}


def main():
    P = 5000
    from bench import csrc_sha16
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = "?"
    out = {"_comment": __doc__.strip().split("\n\n")[2].replace("\n", " "),
           # the tree the PMC passes were taken on: bench.py quotes these figures only while csrc/ is unchanged
           "source": {"git_head": head, "csrc_sha16": csrc_sha16()}}
    ef, ew = per_launch("eval_f", "FETCH_SIZE"), per_launch("eval_w", "WRITE_SIZE")
    ek = {}
    for name, prefixes in (("psolve_fwd", ("k_wsolve<double, false", "k_psolve2<double, false")),
                           ("psolve_adj", ("k_wsolve<double, true", "k_psolve2<double, true")), ("pclass", ("k_wclass<double", "k_pclass<double"))):
        kern = next((kname for pre in prefixes for kname in ef if kname.startswith(pre)), None)
        if kern is None:
            continue
        b = (2 * ef[kern] + ew[kern]) * 1024
        ek[name] = {"kernel": kern, "patients": P, "fetch_kb_reported": ef[kern], "write_kb": ew[kern], "bytes_per_launch": b,
                    "bytes_per_patient": b / P, "launches_per_step": 1}
    out["eval_kernels"] = ek
    kf, kw = per_launch("kv_f", "FETCH_SIZE"), per_launch("kv_w", "WRITE_SIZE")
    alg = {"kronvec": 2, "kronvec_T": 2, "jacobi_step": 4}
    kv = {}
    for name, kern in (("kronvec", "k_kv<double, false, 1, false>"), ("kronvec_T", "k_kv<double, true, 1, false>"),
                       ("jacobi_step", "k_kv<double, false, 1, true>")):
        if kern not in kf:
            continue
        kv[name] = {"kernel": kern, "fetch_kb_reported": kf[kern], "write_kb": kw[kern], "bytes_per_launch": (2 * kf[kern] + kw[kern]) * 1024,
                    "alg_bytes_per_launch": alg[name] * 64 * (2 ** 20) * 8}
        kv[name]["moved_over_alg"] = kv[name]["bytes_per_launch"] / kv[name]["alg_bytes_per_launch"]
    out["kronvec"] = kv
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_traffic.json"), "w"), indent=1)
    # instruction counters of the same kernels (pass eval_i: SQ_INSTS_*), per launch
    ck = {}
    names = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY")
    per = {c: per_launch("eval_i", c) for c in names}
    for name in ek:
        kern = ek[name]["kernel"]
        if kern in per["SQ_INSTS_VALU"]:
            ck[name] = {c: per[c][kern] for c in names if kern in per[c]}
            ck[name]["kernel"] = kern
            ck[name]["waves_per_simd"] = 4 if kern.startswith(("k_wsolve", "k_wclass")) else 8
    json.dump({"_comment": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY over "
                           "scripts/eval_only.py 5000, mean per launch of the second evaluation; waves_per_simd from the launch geometry "
                           "(k_wsolve / k_wclass: one 1024-thread workgroup per CU)",
               "source": out["source"], "kernels": ck}, open(os.path.join(ROOT, "profiles", f"{TAG}_counters.json"), "w"), indent=1)
    print(json.dumps({k: round(v["bytes_per_patient"] / 1e6, 2) for k, v in ek.items()}))
    print(json.dumps({k: round(v["moved_over_alg"], 3) for k, v in kv.items()}))


if __name__ == "__main__":
    main()
