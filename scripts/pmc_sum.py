import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
tot = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void mmhn::', '')
    tot[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for (k, c), v in sorted(tot.items(), key=lambda x: -x[1])[:8]:
    print(f"{d} {k:34s} {c}: total {v/1e6:10.3f} GB(KB-units)  over {n[(k,c)]} launches")
