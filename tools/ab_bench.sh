# A/B of two builds of the library in ONE gpurun call (box-to-box variance is larger than most single changes):
#   cp graph_pooling_amd/libdiffpool_hip.so graph_pooling_amd/libdiffpool_hip_base.so   (the baseline build)
#   ... rebuild with the change ...
#   gpurun -- 'bash tools/ab_bench.sh [bench args]'
# alternates base / new three times and prints ms_per_step of each run.
R=${GRAFT_REPO_ROOT:-.}
cd $R
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then L=$R/graph_pooling_amd/libdiffpool_hip_base.so; else L=$R/graph_pooling_amd/libdiffpool_hip.so; fi
    ms=$(DP_LIB=$L python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $ms"
  done
done
