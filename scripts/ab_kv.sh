#!/bin/bash
# Interleaved A/B of library builds on the batched kronvec: gpurun -- 'bash scripts/ab_kv.sh build_ab/liba.so build_ab/libb.so ...'
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MMHN_LIB; else export MMHN_LIB=$GRAFT_REPO_ROOT/$lib; fi
    echo "round $r $lib: $(python3 scripts/kv_only.py 2>/dev/null | awk '{printf "%s %s ms (%s)  ", $1, $2, $NF}')"
  done
done
