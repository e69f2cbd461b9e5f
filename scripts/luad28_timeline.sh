#!/bin/bash
# Timeline of one evaluation of the 28-event LUAD cohort (kernel trace of scripts/luad28_eval.py) + its timing:
#   gpurun -- 'bash scripts/luad28_timeline.sh <tag>'   -> gpurun_out/luad28_timeline_<tag>.txt, gpurun_out/luad28_<tag>_kernel_stats.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}
cd /tmp
rm -rf /tmp/prof_luad28_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_luad28_$tag -- python3 $R/scripts/luad28_eval.py 5 fit > /dev/null 2>&1
cp /tmp/prof_luad28_$tag/*/*kernel_stats.csv $R/gpurun_out/luad28_${tag}_kernel_stats.csv
python3 $R/scripts/eval_timeline.py /tmp/prof_luad28_$tag/*/*kernel_trace.csv grad > $R/gpurun_out/luad28_timeline_$tag.txt 2>&1
cd $R
python3 scripts/luad28_eval.py 30 fit >> gpurun_out/luad28_timeline_$tag.txt 2>&1
