"""Sum raw PMC counters per kernel: python scripts/pmc_raw.py <dir> [...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        tot = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void mmhn::', '')
            tot[(k, r['Counter_Name'])] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
        for (k, c), v in sorted(tot.items()):
            if v > 0 and ('solve' in k or 'class' in k):
                print(f"{d.split('/')[-1]:10s} {k:28s} {c:24s} {v:16.0f}  /launch {v/n[(k,c)]:14.0f} ({n[(k,c)]})")
