#!/bin/bash
# timing builds of conv_igemm_m32_kernel (M32_ABL, /tmp/abl_many.sh) against the default library, same box: layer3 / layer4 shapes
export AB_SHAPES=${AB_SHAPES:-0,3}
echo "== default library: kernels 0 1 2"; python tools/bench_gemm_ab.py 3 0,1,2 2>&1 | grep -v amdgpu.ids
for n in "$@"; do
  k=1,2; if [ $(( n & 8 )) -ne 0 ]; then k=1; fi
  echo "== M32_ABL=$n: kernels $k"; RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_M32_ABL_$n.so python tools/bench_gemm_ab.py 3 $k 2>&1 | grep -v amdgpu.ids
done
