#!/bin/bash
# One evaluation of the bench cohort in launch order (kernel trace of scripts/eval_only.py 5000): start, duration, gap, grid, queue.
#   gpurun -- 'bash scripts/tl_full.sh'  ->  gpurun_out/tl_full.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/prof_full
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_full -- python3 $R/scripts/eval_only.py 5000 > /dev/null 2>&1
python3 $R/scripts/eval_timeline.py /tmp/prof_full/*/*kernel_trace.csv grad > $R/gpurun_out/tl_full.txt 2>&1
