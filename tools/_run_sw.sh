mkdir -p gpurun_out/r5i
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -6 | tee gpurun_out/r5i/tests.txt
timeout 900 python bench.py > gpurun_out/r5i/bench_default.json 2> gpurun_out/r5i/bench_default.err
timeout 600 python tools/check_determinism.py 256 6 bf16 2>&1 | tail -3 | tee gpurun_out/r5i/determinism.txt
