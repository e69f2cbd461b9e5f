#!/bin/bash
# usage: scripts/pmc_probe.sh <tag> <counters (one hardware-compatible set)> <n> <patients>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/$tag -- python3 $R/scripts/probe2.py "$@" > $R/gpurun_out/$tag.log 2>&1
