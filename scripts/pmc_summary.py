import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/*/*counter_collection.csv')
if not f: print("no counter file in", d); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    print(k)
    for c, val in sorted(v.items()):
        print(f"   {c:28s} {val:16.0f}  (dispatches {cnt[(k,c)]})")
