#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one evaluation with two workgroups per patient on the window route (MMHN_WSPLIT=1):
#   gpurun -- 'bash scripts/pmc_split.sh'  ->  gpurun_out/split_{FETCH,WRITE}_SIZE.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
export MMHN_WSPLIT=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_split_$ctr
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_split_$ctr -- python3 $R/scripts/eval_only.py 5000 > /dev/null 2>&1
  cp /tmp/pmc_split_$ctr/*/*counter_collection.csv $R/gpurun_out/split_${ctr}.csv
done
ls -la $R/gpurun_out/split_*
