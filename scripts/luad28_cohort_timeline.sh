#!/bin/bash
# Timeline of one evaluation of the 28-event LUAD cohort through score_and_grad_reg (the cohort entry the fit uses):
#   gpurun -- 'bash scripts/luad28_cohort_timeline.sh <tag>'   -> gpurun_out/luad28_cohort_timeline_<tag>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-x}
cd /tmp
rm -rf /tmp/prof_l28c_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l28c_$tag -- python3 $R/bench.py --workload ${WL:-luad28} --steps 5 --warmup 2 --min-seconds 0 --no-cpu --no-extras > /dev/null 2>&1
python3 $R/scripts/eval_timeline.py /tmp/prof_l28c_$tag/*/*kernel_trace.csv grad > $R/gpurun_out/luad28_cohort_timeline_$tag.txt 2>&1
cd $R
python3 bench.py --workload ${WL:-luad28} --steps 200 --warmup 20 --no-cpu --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('un-profiled: %.4f ms per evaluation' % d['ms_per_step'])" >> gpurun_out/luad28_cohort_timeline_$tag.txt
