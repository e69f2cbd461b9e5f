#!/bin/bash
# Interleaved A/B of library builds / environment switches on the 28-event LUAD cohort:
#   gpurun -- 'bash scripts/ab_luad28.sh "MMHN_LIB=build_ab/libx.so" "MMHN_COOP=0" default ...'   (3 rounds, fresh process per run)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = default ]; then out=$(python3 scripts/luad28_eval.py 30 fit 2>/dev/null | tail -1)
    else out=$(env $v python3 scripts/luad28_eval.py 30 fit 2>/dev/null | tail -1); fi
    echo "round $r  $v  ::  $out"
  done
done
