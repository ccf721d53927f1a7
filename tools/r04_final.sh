#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04q}; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/gputest_final.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/gputest_final.txt
timeout -k 10 600 python tools/fuzz_ops.py 300 > $out/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 $out/fuzz.txt; grep -c FAIL $out/fuzz.txt
timeout -k 10 300 python tools/fuzz_alm.py 40 > $out/fuzz_alm.txt 2>&1; echo "fuzz_alm rc=$?"; tail -1 $out/fuzz_alm.txt
timeout -k 10 600 python tools/soak_fullsize.py 4 > $out/soak_fullsize.txt 2>&1; echo "soak rc=$?"; tail -4 $out/soak_fullsize.txt
timeout -k 10 600 python tools/emulate_rank.py c4 > $out/emulate_rank.txt 2>&1; echo "emulate rc=$?"; cat $out/emulate_rank.txt | grep world
