#!/bin/bash
# store order of the pair forward's epilogue: parity, HBM counters of the headline launch, step time
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/pair_traffic
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conjugate_pair" > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
timeout -k 10 600 python -m pytest tests/test_rime_gpu.py -x -q -k "matrix_core_arrays_against_reference or bench_model" > $out/rime.txt 2>&1 || { tail -30 $out/rime.txt; exit 1; }
tail -2 $out/rime.txt
bash tools/r05_pair_ab.sh product
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_c4_$name -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-workloads --steps 2 --warmup 1 > $out/pmc_c4_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc_c4_FETCH_SIZE $out/pmc_c4_WRITE_SIZE $out/pmc_c4_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1; echo "summary rc=$?"
find $out -name "*counter_collection.csv" -size +2M -delete
grep -h "fringe_pair_fwd" $out/pmc_summary.txt | cut -c1-420
