R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 200 python3 tools/l0b_stamps.py > $O/l0b_stamps.txt 2>&1
head -48 $O/l0b_stamps.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/t_all2.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/t_all2.log | tail -8
