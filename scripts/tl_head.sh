#!/bin/bash
# The stretch between the two joint solves of the bench cohort's evaluation (kernel trace of scripts/eval_only.py 5000) for a list
# of environment settings:   gpurun -- 'bash scripts/tl_head.sh "MMHN_STAGE_ASIDE=0" "MMHN_STAGE_ASIDE=1"'
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for v in "$@"; do
  i=$((i+1))
  export $v
  rm -rf /tmp/prof_head_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_head_$i -- python3 $R/scripts/eval_only.py 5000 > /dev/null 2>&1
  echo "== $v"
  python3 $R/scripts/eval_timeline.py /tmp/prof_head_$i/*/*kernel_trace.csv grad | awk '/k_wsolve<double, false/{f=1} f{print} /k_wsolve<double, true/{f=0}'
  python3 $R/scripts/eval_timeline.py /tmp/prof_head_$i/*/*kernel_trace.csv grad | tail -1
done
