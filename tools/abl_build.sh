#!/bin/bash
# Ablation builds (timing experiments only).  usage: abl_build.sh <file.hip> <MACRO> <n>...  ->  rgbmanip_amd/abl/librgbm_hip_<MACRO>_<n>.so
set -e
cd "$(dirname "$0")/../rgbmanip_amd/csrc"
f=$1; m=$2; shift 2
bash build.sh >/dev/null
mkdir -p ../abl
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -D$m=$n -c $f -o /tmp/abl_$m$n.o
  objs=$(ls build/*.o | grep -v "build/${f%.hip}.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/librgbm_hip_${m}_$n.so $objs /tmp/abl_$m$n.o
done
ls ../abl
