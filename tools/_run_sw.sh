mkdir -p gpurun_out/r5f4
bash tools/profile_round.sh r5_bf16 bf16 > gpurun_out/r5f4/profile_bf16.log 2>&1
bash tools/profile_round.sh r5_bx3 bf16x3 > gpurun_out/r5f4/profile_bx3.log 2>&1
cp gpurun_out/prof_r5_bf16/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/prof_r5_bx3/hbm_traffic.json profiles/hbm_traffic_bf16x3.json
timeout 900 python bench.py > gpurun_out/r5f4/bench_default.json 2> gpurun_out/r5f4/bench_default.err
timeout 600 python bench.py --dtype bf16x3 --no-modes > gpurun_out/r5f4/bench_bf16x3.json 2> gpurun_out/r5f4/bench_bf16x3.err
timeout 600 python bench.py --inputs survey --no-modes > gpurun_out/r5f4/bench_survey.json 2> gpurun_out/r5f4/bench_survey.err
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -4 > gpurun_out/r5f4/gpu_tests.txt
timeout 600 python tools/check_determinism.py 256 6 bf16 2>&1 | tail -1 >> gpurun_out/r5f4/gpu_tests.txt
