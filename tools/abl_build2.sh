#!/bin/bash
# Experiment builds with arbitrary compiler flags: abl_build2.sh <file.hip> <name> <flags...> -> rgbmanip_amd/abl/librgbm_hip_<name>.so
set -e
cd "$(dirname "$0")/../rgbmanip_amd/csrc"
f=$1; name=$2; shift 2
mkdir -p ../abl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $f -o /tmp/abl2_$name.o
objs=$(ls build/*.o | grep -v "build/${f%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/librgbm_hip_$name.so $objs /tmp/abl2_$name.o
echo ../abl/librgbm_hip_$name.so
