#!/bin/bash
export AB_SHAPES=${AB_SHAPES:-2}
for n in t768; do
  echo "== $n: kernel 2 (timers)"; AB_DEBUG=1 RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_$n.so python tools/bench_gemm_ab.py 2 2 2>&1 | grep -v amdgpu.ids
done
