export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in baseg whb3g; do
  export MMHN_LIB=$R/build_ab/lib$lib.so
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$lib_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${lib}_$ctr -- python3 $R/scripts/eval_only.py 5000 > /dev/null 2>&1
    cp /tmp/pmc_${lib}_$ctr/*/*counter_collection.csv $R/gpurun_out/h8_${lib}_${ctr}.csv
  done
done
ls -la $R/gpurun_out/h8_*
