#!/bin/bash
dt=${1:-bf16x3}; shift
echo "== default"; AB_SHAPES=${AB_SHAPES:-3} python tools/bench_gemm_ab.py 2 2 $dt 2>&1 | grep -v amdgpu.ids
for n in "$@"; do
  echo "== $n: kernel 2"; AB_SHAPES=${AB_SHAPES:-3} RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_$n.so python tools/bench_gemm_ab.py 2 2 $dt 2>&1 | grep -v amdgpu.ids
done
