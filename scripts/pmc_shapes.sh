#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one evaluation of single-shape cohorts (n = 25, fp32, 512 patients): reads per state by shape.
#   gpurun -- 'bash scripts/pmc_shapes.sh'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for sh in "12,12" "14,10" "16,8"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_s
    AB_SHAPES="$sh" timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_s -- python3 $R/scripts/eval_only.py 512 25 f32 > /dev/null 2>&1
    python3 - "$sh" $c <<'PY'
import csv, glob, sys, collections
v=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_s/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==sys.argv[2]:
            v[r['Kernel_Name'].split('(')[0].replace('void mmhn::','')].append(float(r['Counter_Value']))
for k,x in v.items():
    if k.startswith(('k_wsolve','k_wclass')):
        print(sys.argv[1], sys.argv[2], k, 'KB per launch', x[-1], ' per patient MB', x[-1]/512/1024)
PY
  done
done
