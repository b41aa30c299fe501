#!/bin/bash
# Builds librgbm_hip.so for gfx950 in-tree (no cmake; hipcc only).  Usage: build.sh [outdir]
set -e
cd "$(dirname "$0")"
OUT=${1:-..}
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in conv_igemm.hip conv_igemm_glds.hip conv3d_tile.hip conv0_sweep.hip conv0_sweep_x3.hip prob_sparse.hip misc_kernels.hip upconv.hip head_kernels.hip postproc.hip prepare.hip ppo_kernels.hip policy_kernels.hip control.hip synth_env.hip align.hip; do
  [ -f "$f" ] || continue
  EXTRA=""
  # files that must round like numpy / torch elementwise ops: no mul+add -> fma contraction
  case "$f" in postproc.hip|ppo_kernels.hip|prepare.hip|control.hip|synth_env.hip|align.hip|misc_kernels.hip) EXTRA="-ffp-contract=off";; esac
  if [ ! -f build/${f%.hip}.o ] || [ "$f" -nt build/${f%.hip}.o ] || [ common.h -nt build/${f%.hip}.o ] || [ kernels.h -nt build/${f%.hip}.o ] || [ control.h -nt build/${f%.hip}.o ] || [ bbox_emit.h -nt build/${f%.hip}.o ] || [ build.sh -nt build/${f%.hip}.o ]; then
    hipcc $FLAGS $EXTRA -c "$f" -o build/${f%.hip}.o &
    pids+=($!)
  fi
done
for f in layers.cpp adapose.cpp capi.cpp prof.cpp; do
  hipcc $FLAGS -x hip -c "$f" -o build/${f%.cpp}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/librgbm_hip.so" build/*.o
echo "built $OUT/librgbm_hip.so"
