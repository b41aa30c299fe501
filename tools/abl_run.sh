#!/bin/bash
# same-box run of experiment builds (tools/abl_build2.sh) on one conv shape: abl_run.sh "<bench_conv args>" name...   ("base" = the tree's library)
args=$1; shift
for n in "$@"; do
  if [ "$n" = base ]; then unset RGBM_HIP_LIB; else export RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_$n.so; fi
  echo "== $n"; timeout 120 python tools/bench_conv.py $args 2>&1 | grep -v amdgpu.ids | grep "variant"
done
