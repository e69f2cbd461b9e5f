#!/bin/bash
# usage: scripts/pmc.sh <outdir-name> <counters...>   (runs probe.py 20 256 under rocprofv3 --pmc)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$name -- python3 $R/scripts/probe.py 20 256 > $R/gpurun_out/$name.log 2>&1
