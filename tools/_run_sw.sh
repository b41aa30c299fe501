mkdir -p gpurun_out/r5h
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -25 | tee gpurun_out/r5h/tests.txt
timeout 900 python bench.py > gpurun_out/r5h/bench_default.json 2> gpurun_out/r5h/bench_default.err; tail -c 1500 gpurun_out/r5h/bench_default.json | head -c 600
