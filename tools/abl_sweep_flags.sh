#!/bin/bash
# usage: abl_sweep_flags.sh name "flags" [name "flags"]...  -> rgbmanip_amd/abl/librgbm_hip_<name>.so (conv0_sweep.hip rebuilt with the flags)
cd "$(dirname "$0")/../rgbmanip_amd/csrc"
mkdir -p ../abl
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops $f -c conv0_sweep.hip -o /tmp/abl_$n.o 2>&1 | grep -v "recognized feature" | grep -A5 "error:"
    objs=$(ls build/*.o | grep -v "build/conv0_sweep.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/librgbm_hip_$n.so $objs /tmp/abl_$n.o ) &
done
wait
ls ../abl
