set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dd -o dd -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_dd.log 2>&1
cd $R
python3 tools/step_trace.py $(ls $O/prof_dd/*kernel_trace.csv | head -1) > $O/step_trace_dd.txt
rm -f $O/prof_dd/*kernel_trace.csv
cat $O/step_trace_dd.txt
