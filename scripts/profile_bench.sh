#!/bin/bash
# Profiling recipe of profiles/README.md, run on the GPU box (gpurun -- 'bash scripts/profile_bench.sh <tag> [bench args]'):
#   gpurun_out/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu --no-extras`
#   gpurun_out/<tag>_bench_line.json    the un-profiled bench line (never compare a profiled arm with an un-profiled one)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/prof_$tag
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu --no-extras "$@" > $R/gpurun_out/${tag}_profiled_line.json 2> $R/gpurun_out/${tag}_profiled.err
cp $R/gpurun_out/prof_$tag/*/*kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
cd $R && timeout 1500 python3 bench.py "$@" > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
python3 scripts/kstats.py gpurun_out/prof_$tag 14
