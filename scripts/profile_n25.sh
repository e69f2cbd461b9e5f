#!/bin/bash
# BASELINE configs[4] on ONE GPU (n = 25 events, 10 000 paired patients, fp32, 2^25-state vectors): kernel statistics and the
# two HBM-traffic passes over one evaluation.   gpurun -- 'bash scripts/profile_n25.sh r4'
#   gpurun_out/<tag>_n25_kernel_stats.csv, <tag>_n25_bench_line.json, <tag>_n25_pmc/{f,w}_counter_collection.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
ARGS="--config 4 --gpus 1 --steps 1 --warmup 0 --no-cpu --no-extras --min-seconds 0"
cd /tmp
rm -rf $R/gpurun_out/prof_${tag}_n25 /tmp/pmc_n25_f /tmp/pmc_n25_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_n25 -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_n25_profiled_line.json 2> /dev/null
cp $R/gpurun_out/prof_${tag}_n25/*/*kernel_stats.csv $R/gpurun_out/${tag}_n25_kernel_stats.csv
mkdir -p $R/gpurun_out/${tag}_n25_pmc
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_n25_f -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cp /tmp/pmc_n25_f/*/*counter_collection.csv $R/gpurun_out/${tag}_n25_pmc/f_counter_collection.csv
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_n25_w -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cp /tmp/pmc_n25_w/*/*counter_collection.csv $R/gpurun_out/${tag}_n25_pmc/w_counter_collection.csv
cd $R && timeout 900 python3 bench.py --config 4 --gpus 1 --steps 2 --warmup 1 --no-cpu --no-extras > gpurun_out/${tag}_n25_bench_line.json 2> /dev/null
python3 scripts/kstats.py gpurun_out/prof_${tag}_n25 10
