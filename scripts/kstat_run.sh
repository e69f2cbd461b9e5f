#!/bin/bash
# usage: scripts/kstat_run.sh <tag> <n> <patients>   -> gpurun_out/<tag>_kstats.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/$tag
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/scripts/probe2.py "$@" > $R/gpurun_out/$tag.log 2>&1
cd $R && python3 scripts/kstats.py gpurun_out/$tag 12 > gpurun_out/${tag}_kstats.txt
