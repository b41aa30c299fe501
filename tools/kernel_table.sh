#!/bin/bash
# in-library profiler table of one bench configuration: kernel_table.sh <dtype> [debug flags]
python bench.py --dtype $1 --no-modes --no-cpu-baseline --no-boundary --no-small-batch --no-peaks --ppo-envs 0 --no-mixed --no-prepare --no-accuracy --debug-flags ${2:-0} 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 flags ${2:-0}: %.1f poses/s %.2f ms/step' % (d['value'], d['ms_per_step']))
tot=0
for k in d['conv_kernels']:
    tot+=k['total_ms_per_step']; print('     %-86s %5.1f x %7.3f ms = %6.2f ms  %5.0f TFLOP/s' % (k['kernel'][:86], k['launches_per_step'], k['avg_launch_ms'], k['total_ms_per_step'], k['tflops']))
print('     profiled conv kernels: %.2f ms' % tot)"
