#!/bin/bash
export AB_SHAPES=${AB_SHAPES:-3}
for n in 257 260 261; do
  echo "== M32_ABL=$n: kernel 1 (timers)"; AB_DEBUG=1 RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_M32_ABL_$n.so python tools/bench_gemm_ab.py 2 1 2>&1 | grep -v amdgpu.ids
done
