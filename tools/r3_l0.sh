# persistent level-0 kernel: parity tests (model + edge cases + parallel), then the DD bench and a step trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_edge_cases.py -m gpu -x -q > $O/t_l0.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/t_l0.log | tail -8
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_l0.json 2> $O/bench_l0.err; cut -c1-260 $O/bench_l0.json
DP_NO_L0_PERSIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_nol0.json 2> $O/bench_nol0.err; cut -c1-260 $O/bench_nol0.json
