#!/usr/bin/env python3
"""Mean counter value per launch and kernel from a rocprofv3 --pmc output directory (last launch of each kernel only:
the first evaluation is the warm-up).  usage: scripts/pmc_table.py <dir> [kernel substring ...]"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    rows += list(csv.DictReader(open(f)))
want = sys.argv[2:] or ["k_wsolve", "k_wclass", "k_wconvert", "k_psolve", "k_pclass", "k_sweep", "k_tsolve", "k_grad_rows", "k_spatient"]
vals = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void mmhn::", "")
    if any(w in k for w in want):
        vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    half = v[len(v) // 2:]                       # second evaluation
    print(f"{k:40s} {c:28s} launches {len(v):3d}  mean(last half) {sum(half) / len(half):16.1f}")
