#!/bin/bash
# same-box A/B of the bf16 headline under kernel debug flags: ab_bench.sh flagsA flagsB ... (each run twice, interleaved)
for rep in 1 2; do for f in "$@"; do
  python bench.py --no-modes --no-cpu-baseline --no-boundary --no-small-batch --no-peaks --ppo-envs 0 --no-mixed --no-prepare --no-accuracy --debug-flags $f 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('flags $f: %.1f poses/s %.2f ms/step' % (d['value'], d['ms_per_step']))
if $rep == 1:
    for k in d['conv_kernels'][:7]: print('     %-64s %5.1f x %.3f ms = %6.2f ms  %.0f TFLOP/s' % (k['kernel'][:64], k['launches_per_step'], k['avg_launch_ms'], k['total_ms_per_step'], k['tflops']))"
done; done
