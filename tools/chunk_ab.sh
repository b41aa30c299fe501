#!/bin/bash
# Step time against the cost-volume chunk size (views per chunk; 512 = the whole batch of 256 poses in one chunk): does keeping a
# chunk's 3-D intermediates (c0: 19.3 MB per view in 16-bit storage) inside the 256 MB infinity cache pay for the shorter launches?
# usage: tools/chunk_ab.sh <dtype> [chunks ...]
dt=$1; shift
F="--steps 10 --warmup 3 --no-mixed --ppo-envs 0 --no-prepare --no-modes --no-dense-leg --no-accuracy --no-cpu-baseline --no-boundary --no-small-batch --no-peaks"
for c in ${@:-512 64 32 16 8 4 512}; do
  python3 bench.py --dtype $dt --chunk $c $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$dt chunk $c ms_per_step', d['ms_per_step'], 'value', d['value'])"
done
