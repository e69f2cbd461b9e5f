#!/usr/bin/env python3
"""Timeline of the last evaluation in a rocprofv3 --kernel-trace directory: start offset, duration, kernel.
    python scripts/trace_eval.py gpurun_out/prof_<tag> [nth-from-last=1]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pack_sums" in r["Kernel_Name"] or "k_reduce_parts_pack" in r["Kernel_Name"]]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = idx[-nth - 1] + 1, idx[-nth] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %7.1f us  wgs %6d x %4s  %s" % (s / 1000, (e - s) / 1000, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                                                 r["Workgroup_Size_X"], r["Kernel_Name"][:64]))
