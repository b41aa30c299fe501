#!/bin/bash
# AddressSanitizer build of the library's HOST side (GPU ASan is not available on this pool): every .cpp / .hip of
# rgbmanip_amd/csrc compiled with --cuda-host-only -fsanitize=address and linked against tests/asan/hip_stub.cpp instead of the
# HIP runtime -> rgbmanip_amd/librgbm_hip_asan_host.so.  Driven by tests/test_asan_host.py.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R/rgbmanip_amd/csrc"
B=build_asan
mkdir -p $B
FLAGS="--cuda-host-only -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -O1 -g -std=c++17 -fPIC"
pids=()
for f in *.hip *.cpp "$R/tests/asan/hip_stub.cpp"; do
  o=$B/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ common.h -nt $o ] || [ kernels.h -nt $o ] || [ layers.h -nt $o ] || [ adapose.h -nt $o ] || [ prof.h -nt $o ] || [ conv_igemm_m32.inc -nt $o ]; then
    hipcc $FLAGS -c $f -o $o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
# the host objects reference the embedded device code objects that a host-only compile does not produce
DEFS=$(nm -u $B/*.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u | sed 's/^/-Wl,--defsym=/; s/$/=0/' | tr '\n' ' ')
# plain clang++ for the link: no HIP runtime behind the stub (ASan runtime shared, so that a python driver can LD_PRELOAD it)
/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -fsanitize=address -shared-libasan $DEFS -o ../librgbm_hip_asan_host.so $B/*.o
ls -la ../librgbm_hip_asan_host.so
