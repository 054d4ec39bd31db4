R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 200 python3 tools/l0_stamps.py > $O/l0_stamps.txt 2>&1
cat $O/l0_stamps.txt
