#!/bin/bash
# Interleaved A/B of GPU_MAX_HW_QUEUES (how many hardware queues the HIP runtime spreads the streams of a process over; runtime
# default 4, metmhn_amd's default 8) on the LUAD cohorts through score_and_grad_reg, each in a FRESH process:
#   gpurun -- 'bash scripts/ab_hwq.sh 4 8'
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "$@"; do
    export GPU_MAX_HW_QUEUES=$v
    python3 - <<PY
import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from metmhn_amd import regularized_optimization as ro
out = bench.real_cohort_leg(ro, '', reps=40)
print('round $r hwq=$v  luad28 %.4f / %.4f ms' % (out['ms_per_eval_with_grad'], out['ms_per_eval_score_only']), end='  ')
PY
    python3 bench.py --workload luad --steps 300 --warmup 20 --no-cpu --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('luad-reduced %.4f ms' % d['ms_per_step'])"
  done
done
