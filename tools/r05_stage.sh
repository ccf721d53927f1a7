#!/bin/bash
# round 5: the LDS-staged pointing vectors of the small-array forward kernels: parity first (every test that runs the
# 1 / 2-row-tile and packed forward kernels), then timing against the library of the commit before (tools/bin/librime_prestage.so)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05c}; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_objects_gpu.py tests/test_rime_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "not c5_size and not c4_size" > $out/parity.txt 2>&1; echo "parity rc=$?"; tail -4 $out/parity.txt
bash tools/r05_ab.sh ${1:-r05c} "c3 c2" prestage nofetch
