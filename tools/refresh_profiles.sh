# Re-measure everything profiles/ and DESIGN quote, on the CURRENT build, in one gpurun call:
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/refresh/* into profiles/ under the round's prefix.  bench lines for every workload (DD with its
# CPU baseline, DD eval), rocprofv3 kernel stats + one-step timeline for DD, kernel stats for ER, the dominant-kernel
# probes (DD: HBM-bound aggregation, ER: wide aggregation + MFMA-bound A^T S) with their PMC passes — FETCH_SIZE,
# WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES each in its OWN run (--pmc is never combined with another trace domain).
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; mkdir -p $O
cd $R
timeout -k 10 200 python3 bench.py > $O/bench_dd.json 2> $O/bench_dd.err
timeout -k 10 120 python3 bench.py --no-cpu-baseline --linkpred > $O/bench_dd_linkpred.json 2>/dev/null
timeout -k 10 120 python3 bench.py --no-cpu-baseline --eval > $O/bench_dd_eval.json 2>/dev/null
timeout -k 10 120 python3 bench.py --no-cpu-baseline --workload enzymes > $O/bench_enzymes.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload er > $O/bench_er.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes_s2s > $O/bench_enzymes_s2s.json 2>/dev/null
timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload enzymes_p3 > $O/bench_enzymes_p3.json 2>/dev/null
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 120 python3 tools/l0_stamps.py > $O/l0_stamps.txt 2>&1 || true
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 120 python3 tools/l0b_stamps.py > $O/l0b_stamps.txt 2>&1 || true
DP_LIB=graph_pooling_amd/libdiffpool_hip_stamp.so PYTHONPATH=. timeout -k 10 120 python3 tools/s2s_stamps.py > $O/s2s_stamps.txt 2>&1 || true
PYTHONPATH=. timeout -k 10 300 python3 tools/e2e_train_bench.py > $O/e2e_train_bench.json 2>/dev/null || true
echo benches done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dd -o dd -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_dd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o probe -- python3 $R/bench.py --probe-only > $O/prof_probe.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_er -o er -- python3 $R/bench.py --workload er --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_er.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_er_probe -o erp -- python3 $R/bench.py --workload er --probe-only > $O/prof_er_probe.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --probe-only > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --probe-only > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_er_fetch -o f -- python3 $R/bench.py --workload er --probe-only > $O/pmc_er_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_er_write -o w -- python3 $R/bench.py --workload er --probe-only > $O/pmc_er_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_er_mfma -o m -- python3 $R/bench.py --workload er --probe-only > $O/pmc_er_mfma.log 2>&1
# the persistent level-0 kernels, in the step itself (eager launches: counters per dispatch)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_step_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $O/pmc_step_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_step_write -o w -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $O/pmc_step_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_step_mfma -o m -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-graph > $O/pmc_step_mfma.log 2>&1
echo pmc done
cd $R
python3 tools/pmc_summary.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) k_aggregate > $O/pmc_dd_probe_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_write/*counter_collection.csv | head -1) k_aggregate | tail -n +2 >> $O/pmc_dd_probe_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_er_fetch/*counter_collection.csv | head -1) k_aggregate > $O/pmc_er_probe_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_er_write/*counter_collection.csv | head -1) k_aggregate | tail -n +2 >> $O/pmc_er_probe_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_er_mfma/*counter_collection.csv | head -1) k_aggregate | tail -n +2 >> $O/pmc_er_probe_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_step_fetch/*counter_collection.csv | head -1) k_level0 > $O/pmc_dd_step_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_step_write/*counter_collection.csv | head -1) k_level0 | tail -n +2 >> $O/pmc_dd_step_summary.csv
python3 tools/pmc_summary.py $(ls $O/pmc_step_mfma/*counter_collection.csv | head -1) k_level0 | tail -n +2 >> $O/pmc_dd_step_summary.csv
python3 tools/step_trace.py $(ls $O/prof_dd/*kernel_trace.csv | head -1) > $O/step_trace_dd.txt
rm -rf $O/pmc_*/*kernel_trace.csv $O/pmc_*/*counter_collection.csv $O/prof_*/*kernel_trace.csv
cat $O/pmc_dd_probe_summary.csv $O/pmc_er_probe_summary.csv; tail -1 $O/bench_dd.json | cut -c1-400
