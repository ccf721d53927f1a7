#!/bin/bash
# AddressSanitizer + UBSan build of the library's HOST code (device code is compiled as usual, not instrumented) and a sweep of
# the host-only entry points: workspace / split planning over random shapes, argument validation.  CPU only -- no GPU needed.
set -e
cd "$(dirname "$0")/.."
out=${1:-/tmp/rime_asan}
mkdir -p $out
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Wno-unused-function"
objs=""
for f in fringe.hip fringe_mfma.hip interp.hip alm.hip chisq.hip cal.hip eq2top.hip jones.hip capi.cpp comm.cpp; do
  o=$out/$(basename ${f%.*}).o
  case $f in *.cpp) x="-x hip";; *) x="";; esac
  /opt/rocm/bin/hipcc $FLAGS $x -c bayeslim_amd/csrc/$f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined $objs -ldl -o $out/librime_hip_asan.so
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address,undefined -fno-gpu-sanitize -x c++ tools/host_sanitize.cpp -o $out/host_sanitize -L$out -lrime_hip_asan -Wl,-rpath,$out
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $out/host_sanitize
