# Per-kernel time of ONE eager ER step (the third of three), and its fp32 GEMM launches by shape:
#   gpurun -- 'bash tools/er_step_breakdown.sh'   -> gpurun_out/er_breakdown/{kernels.txt,gemm_shapes.txt}
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/er_breakdown; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DP_GEMM_TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o gs -- python3 $R/tools/er_gemm_shapes.py run 2> $O/shapes.err
cd $R
python3 tools/er_gemm_shapes.py join $O/shapes.err $O/gs_kernel_trace.csv > $O/gemm_shapes.txt
python3 tools/er_gemm_shapes.py kernels $O/gs_kernel_trace.csv > $O/kernels.txt
rm -f $O/gs_kernel_trace.csv
