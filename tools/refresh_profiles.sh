set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout -k 10 200 python3 bench.py > $O/bench_dd.json 2> $O/bench_dd.err
timeout -k 10 120 python3 bench.py --no-cpu-baseline --linkpred > $O/bench_dd_linkpred.json 2>/dev/null
timeout -k 10 120 python3 bench.py --no-cpu-baseline --workload enzymes > $O/bench_enzymes.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload er > $O/bench_er.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes_s2s > $O/bench_enzymes_s2s.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes_p3 > $O/bench_enzymes_p3.json 2>/dev/null
echo benches done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dd -o dd -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_dd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o probe -- python3 $R/bench.py --probe-only > $O/prof_probe.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --probe-only > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --probe-only > $O/pmc_write.log 2>&1
echo pmc done
cd $R
python3 tools/pmc_summary.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) k_aggregate > $O/pmc_fetch_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_write/*counter_collection.csv | head -1) k_aggregate > $O/pmc_write_summary.csv
python3 tools/step_trace.py $(ls $O/prof_dd/*kernel_trace.csv | head -1) > $O/step_trace_dd.txt
rm -rf $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv $O/prof_dd/*kernel_trace.csv $O/prof_probe/*kernel_trace.csv
cat $O/pmc_fetch_summary.csv $O/pmc_write_summary.csv; tail -1 $O/bench_dd.json | cut -c1-400
