import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-52s calls %4s total %9.2f ms  avg %8.3f ms  %s%%" % (r["Name"][:52], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
