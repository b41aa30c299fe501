mkdir -p gpurun_out/r5k
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee gpurun_out/r5k/tests.txt
timeout 900 python bench.py > gpurun_out/r5k/bench_default.json 2> gpurun_out/r5k/bench_default.err
timeout 600 python tools/check_determinism.py 256 6 bf16 2>&1 | tail -2 | tee gpurun_out/r5k/determinism.txt
