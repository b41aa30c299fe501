#!/bin/bash
export AB_SHAPES=${AB_SHAPES:-0,3}
echo "== default"; python tools/bench_gemm_ab.py 3 0,1 2>&1 | grep -v amdgpu.ids
for n in 512 1024 1536; do
  echo "== M32_ABL=$n: kernel 1"; RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_M32_ABL_$n.so python tools/bench_gemm_ab.py 3 1 2>&1 | grep -v amdgpu.ids
done
for n in 256 768 1792; do
  echo "== M32_ABL=$n: kernel 1 (timers)"; AB_DEBUG=1 RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_M32_ABL_$n.so python tools/bench_gemm_ab.py 2 1 2>&1 | grep -v amdgpu.ids
done
