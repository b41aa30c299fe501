#!/bin/bash
# Same-box A/B of whole-forward time (batch 256, dense cost regularisation) between builds of the library:
#   tools/ab_libs.sh "<dtype...>" <lib.so|default> ...     two interleaved rounds; prints the median ms per forward of each run
dts=$1; shift
for round in 1 2; do
  for dt in $dts; do
    for lib in "$@"; do
      if [ "$lib" = default ]; then unset RGBM_HIP_LIB; else export RGBM_HIP_LIB=$PWD/$lib; fi
      echo "round $round $dt $(basename $lib): $(python tools/ab_option.py $dt sparse_dec 0 2>&1 | tail -n 1 | sed 's/.*median//')"
    done
  done
done
