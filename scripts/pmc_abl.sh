#!/bin/bash
# usage: scripts/pmc_abl.sh <n> <patients> <lib tag>...   (tags: base or build_ab/<tag>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=$1; P=$2; shift; shift
for tag in "$@"; do
  if [ "$tag" = base ]; then unset MMHN_LIB; else export MMHN_LIB=$R/build_ab/$tag.so; fi
  cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/abl_$tag -- python3 $R/scripts/probe2.py $n $P > $R/gpurun_out/abl_$tag.log 2>&1
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ablt_$tag -- python3 $R/scripts/probe2.py $n $P > $R/gpurun_out/ablt_$tag.log 2>&1
done
cd $R
for tag in "$@"; do
  python3 scripts/pmc_raw.py gpurun_out/abl_$tag
  python3 scripts/kstats.py gpurun_out/ablt_$tag 2>/dev/null | grep -E "solve|class_marg" 
done > gpurun_out/abl_summary.txt
