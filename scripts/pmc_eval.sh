#!/bin/bash
# PMC passes over scripts/eval_only.py (one hardware-compatible counter set per pass, own run each: no tracing domains
# beside --kernel-trace).  usage: scripts/pmc_eval.sh <tag> "<counters>" [eval_only.py args]
#   -> gpurun_out/pmc_<tag>/*counter_collection.csv ; scripts/pmc_table.py prints per-kernel means
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
rm -rf $R/gpurun_out/pmc_$tag
cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/eval_only.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 $R/scripts/pmc_table.py $R/gpurun_out/pmc_$tag
