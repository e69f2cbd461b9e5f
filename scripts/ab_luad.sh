#!/bin/bash
# Interleaved A/B of library builds on the LUAD-reduced cohort: gpurun -- 'bash scripts/ab_luad.sh build_ab/liba.so default ...'
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MMHN_LIB; else export MMHN_LIB=$GRAFT_REPO_ROOT/$lib; fi
    v=$(python3 bench.py --workload luad --steps 300 --warmup 20 --no-cpu --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  obj %.12f' % (d['ms_per_step'], d.get('objective_value', 0)))")
    echo "round $r $lib $v"
  done
done
