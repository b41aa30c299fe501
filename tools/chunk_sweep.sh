for c in 64 128 256 512; do
  python bench.py --no-cpu-baseline --no-boundary --no-small-batch --no-peaks --no-prepare --ppo-envs 0 --steps 6 --warmup 2 --chunk $c > gpurun_out/chunk_$c.json 2>/dev/null
  python -c "import json; r=json.load(open('gpurun_out/chunk_$c.json')); print('chunk', $c, r['value'], r['ms_per_step'])"
done
