#!/usr/bin/env python3
"""Timeline of the last evaluation in a rocprofv3 --kernel-trace csv (scripts/eval_only.py): kernel, start offset, duration, gap to
the previous kernel's end (us).    python scripts/eval_timeline.py <kernel_trace.csv> [grad]
grad: the last evaluation that ran an adjoint solve (a trace that ends with score-only evaluations)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_begin_eval" in r["Kernel_Name"]]
first = starts[-1] if starts else 0
last = len(rows)
if len(sys.argv) > 2 and sys.argv[2] == "grad":
    for si in reversed(range(len(starts))):
        a, b = starts[si], (starts[si + 1] if si + 1 < len(starts) else len(rows))
        if any(re.search(r"solve2?<\w+, true", r["Kernel_Name"]) for r in rows[a:b]):
            first, last = a, b
            break
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[first:last]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("mmhn::", "").replace("void ", "")
    name = name[:name.index("(")] if "(" in name else name
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    shape = f"{grid // wg if wg else 0:6d} x {wg:4d}"
    print(f"{(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:8.1f}  {shape}  q{r.get('Queue_Id', '?')}  {name[:70]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
