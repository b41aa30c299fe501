#!/bin/bash
# Builds librgbm_hip.so for gfx950 in-tree (no cmake; hipcc only).  Usage: build.sh [outdir]
set -e
cd "$(dirname "$0")"
OUT=${1:-..}
mkdir -p build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
# No packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in any kernel.  Round 4: forwards that overlap on the device
# (two streams) differed intermittently from the one-stream forward, and the difference CORRELATES with this code-generation switch: same
# source with packed ops 16 of 20 overlapped runs differ, without them 0 of 60 (tools/check_two_stream_forwards.py).
# The mechanism is NOT isolated: the stand-alone reproducers (tools/micro/pk_after_load.hip, valu_under_mfma.hip) are negative, and a
# library-wide flag also moves register allocation and scheduling in every kernel, so a latent ordering bug elsewhere is not excluded.
# What holds is the gate: 12 overlapped forwards x 3 storage types bit for bit (tests/test_gpu_at_batch.py) and the check that the shipped
# code objects hold no v_pk_*_f32 (tests/test_cabi_symbols.py).  The packed forms buy nothing here (same-box A/B of every kernel file with
# and without them: within 0.3 %), so they are off everywhere.
FLAGS="$FLAGS -Xclang -target-feature -Xclang -packed-fp32-ops"
echo "$FLAGS" > build/.flags.new 2>/dev/null || true
if ! cmp -s build/.flags.new build/.flags 2>/dev/null; then rm -f build/*.o build/*.asm_ok; cp build/.flags.new build/.flags; fi      # new flags: everything is rebuilt AND the sweep kernels' ISA is re-checked
pids=()
for f in conv_igemm.hip conv_igemm_glds.hip conv3d_tile.hip conv0_sweep.hip conv0_sweep_x3.hip prob_sparse.hip misc_kernels.hip bn_kernels.hip upconv.hip upconv_final.hip stem.hip head_kernels.hip postproc.hip prepare.hip ppo_kernels.hip policy_kernels.hip control.hip synth_env.hip align.hip pnp.hip microbench.hip; do
  [ -f "$f" ] || continue
  EXTRA=""
  # files that must round like numpy / torch elementwise ops: no mul+add -> fma contraction
  case "$f" in postproc.hip|ppo_kernels.hip|prepare.hip|control.hip|synth_env.hip|align.hip|pnp.hip|misc_kernels.hip|bn_kernels.hip) EXTRA="-ffp-contract=off";; esac
  if [ ! -f build/${f%.hip}.o ] || [ "$f" -nt build/${f%.hip}.o ] || [ common.h -nt build/${f%.hip}.o ] || [ kernels.h -nt build/${f%.hip}.o ] || [ conv_igemm_m32.inc -nt build/${f%.hip}.o ] || [ conv3d_tile_table.h -nt build/${f%.hip}.o ] || [ control.h -nt build/${f%.hip}.o ] || [ bbox_emit.h -nt build/${f%.hip}.o ] || [ build.sh -nt build/${f%.hip}.o ]; then
    hipcc $FLAGS $EXTRA -c "$f" -o build/${f%.hip}.o &
    pids+=($!)
  fi
done
for f in layers.cpp adapose.cpp capi.cpp prof.cpp; do
  hipcc $FLAGS -x hip -c "$f" -o build/${f%.cpp}.o &
  pids+=($!)
done
# The plane-sweep kernels keep inline-asm gathers in flight across loop iterations and count them by hand: what hipcc made of
# THESE sources is checked (no copy / re-homing of an in-flight gather register, tools/check_asm_gathers.py) whenever one of them
# was recompiled; a finding fails the build.
chk=()
for f in conv0_sweep.hip conv0_sweep_x3.hip; do
  if [ ! -f build/${f%.hip}.asm_ok ] || [ "$f" -nt build/${f%.hip}.asm_ok ] || [ common.h -nt build/${f%.hip}.asm_ok ] || [ ../../tools/check_asm_gathers.py -nt build/${f%.hip}.asm_ok ] || [ build.sh -nt build/${f%.hip}.asm_ok ]; then
    ( python3 ../../tools/check_asm_gathers.py "$f" > build/${f%.hip}.asm_log 2>&1 && touch build/${f%.hip}.asm_ok ) &
    chk+=("$!:$f")
  fi
done
for p in "${pids[@]}"; do wait $p; done
for c in "${chk[@]}"; do
  if ! wait ${c%%:*}; then echo "check_asm_gathers FAILED for ${c#*:}"; cat build/$(basename ${c#*:} .hip).asm_log; exit 1; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/librgbm_hip.so" build/*.o
echo "built $OUT/librgbm_hip.so"
