import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        vals[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in vals.items():
    v = v[2:] if len(v) > 4 else v      # skip the two warm-up launches
    print(f"{d} {pat} {c}: n={len(v)} avg={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
