#!/bin/bash
# Everything under profiles/ for one round, on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r2'
#   gpurun_out/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of bench.py (3 steps, no CPU legs)
#   gpurun_out/<tag>_bench_line.json           the un-profiled default bench line
#   gpurun_out/<tag>_kronvec_kernel_stats.csv  the same for scripts/kv_only.py (batched kronvec)
#   gpurun_out/<tag>_luad_kernel_stats.csv     the same for bench.py --workload luad
#   gpurun_out/<tag>_pmc/{eval,kv}_{f,w}_counter_collection.csv   FETCH_SIZE / WRITE_SIZE, one pass each
# then on the build side: cp into profiles/ and run scripts/make_traffic_json.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
bash $R/scripts/profile_bench.sh $tag > $R/gpurun_out/${tag}_kstats.txt 2>&1
cd /tmp
rm -rf $R/gpurun_out/prof_${tag}_kv $R/gpurun_out/prof_${tag}_luad $R/gpurun_out/${tag}_pmc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_kv -- python3 $R/scripts/kv_only.py > $R/gpurun_out/${tag}_kv.log 2>&1
cp $R/gpurun_out/prof_${tag}_kv/*/*kernel_stats.csv $R/gpurun_out/${tag}_kronvec_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_luad -- python3 $R/bench.py --workload luad --steps 20 --warmup 3 --min-seconds 0 --no-cpu --no-extras > $R/gpurun_out/${tag}_luad_line.json 2> /dev/null
cp $R/gpurun_out/prof_${tag}_luad/*/*kernel_stats.csv $R/gpurun_out/${tag}_luad_kernel_stats.csv
mkdir -p $R/gpurun_out/${tag}_pmc
for pass in "eval_f FETCH_SIZE eval_only.py 5000" "eval_w WRITE_SIZE eval_only.py 5000" "kv_f FETCH_SIZE kv_only.py" "kv_w WRITE_SIZE kv_only.py" \
            "eval_i SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY eval_only.py 5000"; do
  set -- $pass
  name=$1; ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc ${ctr//,/ } --output-format csv -d /tmp/pmc_$name -- python3 $R/scripts/"$@" > /dev/null 2>&1
  cp /tmp/pmc_$name/*/*counter_collection.csv $R/gpurun_out/${tag}_pmc/${name}_counter_collection.csv
done
ls -la $R/gpurun_out/${tag}_pmc
