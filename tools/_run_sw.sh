mkdir -p gpurun_out/r5k
SPARSE_DEC=2 timeout 300 python tools/kernel_ms.py bf16 2>&1 | grep -v amdgpu | awk '{printf "%-64s %s %s %s\n", substr($0, 47, 62), $(NF-2), $(NF-1), $NF}' | sort -k4 -n -r -t' ' | tee gpurun_out/r5k/rows.txt
