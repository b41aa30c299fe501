mkdir -p gpurun_out/r5f3
timeout 900 python bench.py > gpurun_out/r5f3/bench_default.json 2> gpurun_out/r5f3/bench_default.err
timeout 900 python -m pytest tests -m gpu -q -k "persistent or hip_options or skips_view2 or sweep_f16_off" 2>&1 | tail -3 > gpurun_out/r5f3/new_tests.txt
timeout 300 python tools/kernel_ms.py bf16 conv0_sweep 2>&1 | tail -1 >> gpurun_out/r5f3/new_tests.txt
