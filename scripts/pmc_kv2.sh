#!/bin/bash
# usage: scripts/pmc_kv2.sh <tag> <counter> [jacobi] [transpose]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
rm -rf $R/gpurun_out/$tag
cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$ctr" --output-format csv -d $R/gpurun_out/$tag -- python3 $R/scripts/kv_only.py 20 20 64 6 "$@" > $R/gpurun_out/$tag.log 2>&1
