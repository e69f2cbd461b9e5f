#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && rocprofv3 --kernel-trace --pmc "$1" --output-format csv -d $R/gpurun_out/$tag -- python3 $R/scripts/kv_only.py 20 20 64 6 > $R/gpurun_out/$tag.log 2>&1
