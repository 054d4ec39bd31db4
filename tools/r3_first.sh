# round 3, first GPU session: the whole GPU suite on the new build, the DD bench, the A/B of the dominant kernel
# against the round-1-final build (one process), and the rocprof summary of the probe
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/t_all.log 2>&1 || true
grep -E "^(FAILED|ERROR)|passed|failed" $O/t_all.log | tail -15
timeout -k 10 200 python3 bench.py > $O/bench_dd.json 2> $O/bench_dd.err
cut -c1-300 $O/bench_dd.json
timeout -k 10 200 python3 tools/agg_ab_probe.py graph_pooling_amd/libdiffpool_hip.so graph_pooling_amd/libdiffpool_hip_r01f.so > $O/agg_ab.txt 2>&1
cat $O/agg_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o probe -- python3 $R/bench.py --probe-only > $O/prof_probe.log 2>&1
grep -h "k_aggregate" $O/prof_probe/*kernel_stats.csv | head -3
