#!/bin/bash
# Interleaved A/B of library builds on the bench cohort by the step minus its three big kernels (their box-to-box noise is larger than
# what the small kernels can win):   gpurun -- 'bash scripts/ab_rest.sh old new'      (build_ab/lib<name>.so)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "$@"; do
    export MMHN_LIB=$GRAFT_REPO_ROOT/build_ab/lib$v.so
    python3 bench.py --no-cpu --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['ms_per_step']-d['roofline']['avg_launch_ms']-d['roofline_adjoint']['avg_launch_ms']-d['roofline_marginals']['avg_launch_ms']; print('round $r $v ms_per_step %.3f  rest (step - fwd - adj - marg) %.3f   fwd %.2f adj %.2f marg %.2f' % (d['ms_per_step'], o, d['roofline']['avg_launch_ms'], d['roofline_adjoint']['avg_launch_ms'], d['roofline_marginals']['avg_launch_ms']))"
  done
done
