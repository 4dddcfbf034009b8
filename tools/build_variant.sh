#!/bin/bash
# A/B builds of the product library with different compiler flags (diagnostic; output
# under tools/build/, selected at run time with BOOM_AMD_LIB=...; never the default).
# usage: tools/build_variant.sh NAME "flags for ssvs_kernel.hip" ["flags for the other kernels"]
set -e
NAME=$1; SSVS_FLAGS=${2:-}; OTHER_FLAGS=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/boom_amd/csrc
OUT=$ROOT/tools/build/$NAME
mkdir -p $OUT
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in engine group ssvs_kernel ssvs_big_kernel ssvs_adaptive_kernel ssm_kernel probit_kernel xtwx_cols_kernel predict_kernel suf_kernel kalman_kernel; do
  fl=$OTHER_FLAGS
  [ $f = ssvs_kernel ] && fl=$SSVS_FLAGS
  [ $f = group ] && fl=""; [ $f = engine ] && fl=$ENGINE_FLAGS
  ( /opt/rocm/bin/hipcc $BASE $fl -c $SRC/$f.hip -o $OUT/$f.o 2>/dev/null ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -ldl -o $OUT/libboomamd.so
ls -la $OUT/libboomamd.so
