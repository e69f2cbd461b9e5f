#!/usr/bin/env python3
"""profiles/<tag>_n25_traffic.json from the two PMC passes of scripts/profile_n25.sh (BASELINE configs[4] on one GPU: n = k = 25,
10 000 paired patients, fp32, ONE evaluation per pass): HBM-side bytes per evaluation of the joint solves and the class marginals
= 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 correction, scripts/make_traffic_json.py) summed over the launches of the evaluation,
against the algorithmic bytes (solves: the seeded half written once; marginals: pi and q read once).
    python scripts/make_n25_traffic.py <tag> [patients=10000] [k=25]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 25
pmc = os.path.join(ROOT, "profiles", f"{tag}_n25_pmc")


def sums(path, counter):
    out, launches = {}, {}
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("void ", "").replace("mmhn::", "")
        n = n[:n.index("(")] if "(" in n else n
        out[n] = out.get(n, 0.0) + float(r["Counter_Value"])
        key = (n, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            launches[n] = launches.get(n, 0) + 1
    return out, launches


F, LF = sums(os.path.join(pmc, "f_counter_collection.csv"), "FETCH_SIZE")
W, _ = sums(os.path.join(pmc, "w_counter_collection.csv"), "WRITE_SIZE")
half = (1 << (k - 1)) * 4
groups = {
    "joint solve, forward": (["k_wsolve<float, false", "k_psolve2<float, false", "k_psolve<float, false", "k_tsolve<float, false, false"], P * half),
    "joint solve, adjoint": (["k_wsolve<float, true", "k_psolve2<float, true", "k_psolve<float, true", "k_tsolve<float, true, false"], P * half),
    "class marginals": (["k_wclass<float", "k_pclass<float", "k_class_marg<float"], 2 * P * half),
}
res = {}
for name, (prefixes, alg) in groups.items():
    ks = sorted(n for n in F if any(n.startswith(p) for p in prefixes))
    b = sum((2 * F[n] + W.get(n, 0.0)) * 1024.0 for n in ks)
    res[name] = dict(kernels=ks, launches={n: LF[n] for n in ks}, bytes_per_evaluation=b, alg_bytes_per_evaluation=alg,
                     traffic_over_alg=b / alg, bytes_per_patient=b / P)
line = {}
try:
    line = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_n25_bench_line.json")).read().strip().splitlines()[-1])
except Exception:
    pass
doc = {"_comment": __doc__.split("\n    python")[0], "kernels": res, "ms_per_evaluation": line.get("ms_per_step")}
json.dump(doc, open(os.path.join(ROOT, "profiles", f"{tag}_n25_traffic.json"), "w"), indent=1)
for n, v in res.items():
    print(f"{n:24s} {v['bytes_per_evaluation'] / 1e12:6.2f} TB  {v['traffic_over_alg']:5.2f}x  {v['kernels']}")
print("ms per evaluation", doc["ms_per_evaluation"])
