export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in coop0 wgs256; do
rm -rf /tmp/prof_$v
if [ $v = coop0 ]; then export MMHN_COOP=0; else unset MMHN_COOP; export MMHN_COOP_WGS=256; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python3 $R/scripts/luad28_eval.py 5 fit > /dev/null 2>&1
python3 $R/scripts/eval_timeline.py /tmp/prof_$v/*/*kernel_trace.csv grad > $R/gpurun_out/luad28_timeline_$v.txt 2>&1
done
