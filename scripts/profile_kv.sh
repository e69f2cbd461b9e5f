#!/bin/bash
# usage: scripts/profile_kv.sh <tag>   kernel stats of the pure batched-kronvec leg (k_sweep<double,false>, 21 launches)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rm -rf $R/gpurun_out/prof_$tag
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/scripts/kv_only.py 20 20 64 20 0 0 > $R/gpurun_out/${tag}.log 2>&1
cp $R/gpurun_out/prof_$tag/*/*kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
