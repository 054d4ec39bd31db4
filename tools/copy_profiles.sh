# Copy what tools/refresh_profiles.sh (+ gemm_split_probe, er_step_breakdown) left under gpurun_out/ into profiles/
# under a round prefix:  bash tools/copy_profiles.sh r02
set -e
P=${1:?round prefix}; O=gpurun_out/refresh; D=profiles
for w in dd dd_eval dd_linkpred enzymes enzymes_p3 enzymes_s2s er; do tail -1 $O/bench_$w.json > $D/${P}_bench_$w.json; done
cp $O/prof_dd/dd_kernel_stats.csv $D/${P}_dd_kernel_stats.csv
cp $O/prof_probe/probe_kernel_stats.csv $D/${P}_dd_probe_only_kernel_stats.csv
cp $O/prof_er/er_kernel_stats.csv $D/${P}_er_kernel_stats.csv
cp $O/prof_er_probe/erp_kernel_stats.csv $D/${P}_er_probe_kernel_stats.csv
cp $O/pmc_dd_probe_summary.csv $D/${P}_dd_probe_pmc_summary.csv
cp $O/pmc_er_probe_summary.csv $D/${P}_er_probe_pmc_summary.csv
cp $O/step_trace_dd.txt $D/${P}_dd_step_trace.txt
cp $O/pmc_dd_step_summary.csv $D/${P}_dd_step_pmc_summary.csv
for s in l0 l0b s2s; do [ -f $O/${s}_stamps.txt ] && grep -v amdgpu.ids $O/${s}_stamps.txt > $D/${P}_${s}_stamps.txt; done
[ -f $O/e2e_train_bench.json ] && cp $O/e2e_train_bench.json $D/${P}_e2e_train_bench.json
[ -f $O/gemm_split_probe.txt ] && cp $O/gemm_split_probe.txt $D/${P}_gemm_split_probe.txt
[ -f gpurun_out/er_breakdown/kernels.txt ] && cp gpurun_out/er_breakdown/kernels.txt $D/${P}_er_step_kernels.txt
[ -f gpurun_out/er_breakdown/gemm_shapes.txt ] && cp gpurun_out/er_breakdown/gemm_shapes.txt $D/${P}_er_step_gemm_shapes.txt
python3 tools/update_pmc_traffic.py $P > /dev/null
echo copied
