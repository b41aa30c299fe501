#!/bin/bash
# Same-box A/B of small-batch latency between builds of the library: tools/small_ab_libs.sh "<dtypes>" "<Bs>" <lib.so|default> ...  (two interleaved rounds)
dts=$1; Bs=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset RGBM_HIP_LIB; else export RGBM_HIP_LIB=$PWD/$lib; fi
    echo "== round $round $(basename $lib)"
    python tools/small_lat.py $dts $Bs 2>&1 | grep -v amdgpu.ids
  done
done
