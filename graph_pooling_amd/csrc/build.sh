#!/bin/bash
# Build libdiffpool_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdiffpool_hip.so
BUILD=build
# DP_STAMP=1: diagnostic library with in-kernel phase stamps (never the product; load it with DP_LIB=<path>)
if [ "${DP_STAMP:-0}" = "1" ]; then OUT=../libdiffpool_hip_stamp.so; BUILD=build_stamp; EXTRA="-DDP_STAMP"; else EXTRA=""; fi
# DP_VARIANT=<name> DP_EXTRA_FLAGS="...": an experimental build next to the product (libdiffpool_hip_<name>.so)
if [ -n "${DP_VARIANT:-}" ]; then OUT=../libdiffpool_hip_${DP_VARIANT}.so; BUILD=build_${DP_VARIANT}; EXTRA="$EXTRA ${DP_EXTRA_FLAGS:-}"; fi
SRCS="dp_api.hip dp_gemm.hip dp_gemm_split.hip dp_rowops.hip dp_linkpred.hip dp_model.hip dp_set2set.hip dp_meanagg.hip dp_agg.hip dp_small.hip dp_level0.hip dp_head.hip dp_optim.hip dp_batch.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden $EXTRA"
mkdir -p $BUILD
pids=()
for s in $SRCS; do
  o=$BUILD/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ dp_common.h -nt "$o" ] || [ ../../include/diffpool_hip.h -nt "$o" ]; then
    hipcc $FLAGS -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $BUILD/*.o
echo "built $(realpath $OUT)"
