R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/t_all3.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/t_all3.log | tail -8
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_bar.json 2>/dev/null; cut -c1-230 $O/bench_bar.json
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes_s2s > $O/bench_s2s.json 2>/dev/null; cut -c1-230 $O/bench_s2s.json
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes > $O/bench_enz.json 2>/dev/null; cut -c1-230 $O/bench_enz.json
