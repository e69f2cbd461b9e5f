#!/bin/bash
# Timeline of one evaluation of the LUAD-reduced cohort (kernel trace of scripts/eval_only.py luad) + its timing:
#   gpurun -- 'bash scripts/luad_timeline.sh <tag>'   -> gpurun_out/luad_timeline_<tag>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}
cd /tmp
rm -rf /tmp/prof_luad_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_luad_$tag -- python3 $R/scripts/eval_only.py luad > /dev/null 2>&1
python3 $R/scripts/trace_eval.py /tmp/prof_luad_$tag > $R/gpurun_out/luad_timeline_$tag.txt 2>&1
cd $R
python3 bench.py --workload luad --steps 200 --warmup 10 --no-cpu --no-extras >> gpurun_out/luad_timeline_$tag.txt 2>&1
