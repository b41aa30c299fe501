#!/bin/bash
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_kernel_variants and not bf16x3" 2>&1 | tail -3
AB_SHAPES=0,3,4 python tools/bench_gemm_ab.py 3 0,1,2 2>&1 | grep -v amdgpu.ids
echo "== M32_ABL=256: kernel 1 (timers)"; AB_SHAPES=3 AB_DEBUG=1 RGBM_HIP_LIB=$PWD/rgbmanip_amd/abl/librgbm_hip_M32_ABL_256.so python tools/bench_gemm_ab.py 2 1 2>&1 | grep -v amdgpu.ids
