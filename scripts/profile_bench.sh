#!/bin/bash
# usage: scripts/profile_bench.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench_line.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rm -rf $R/gpurun_out/prof_$tag
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/${tag}_profiled_line.json 2> $R/gpurun_out/${tag}_profiled.err
cp $R/gpurun_out/prof_$tag/*/*kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
cd $R && timeout 900 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
