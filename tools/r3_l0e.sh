R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_edge_cases.py -m gpu -x -q > $O/t_l0b.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/t_l0b.log | tail -8
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/bench_l0b.json 2> $O/bench_l0b.err; cut -c1-260 $O/bench_l0b.json
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 200 python3 tools/l0_stamps.py > $O/l0_stamps.txt 2>&1
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 200 python3 tools/l0b_stamps.py > $O/l0b_stamps.txt 2>&1
head -42 $O/l0_stamps.txt; head -34 $O/l0b_stamps.txt
