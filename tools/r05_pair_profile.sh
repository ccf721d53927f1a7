#!/bin/bash
# profile set of the conjugate-pair kernels on the headline workload: bench line, rocprofv3 kernel stats, PMC passes
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05p
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "[1] bench c4"; timeout -k 10 400 python bench.py --no-other-workloads > $out/bench_c4.json 2> $out/bench_c4.err; echo rc=$?
cd /tmp && export TMPDIR=/tmp
echo "[2] rocprofv3 kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-workloads --steps 5 --warmup 2 > $out/prof_c4.log 2>&1; echo rc=$?
echo "[3] PMC passes"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_c4_$name -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-workloads --steps 2 --warmup 1 > $out/pmc_c4_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc_c4_FETCH_SIZE $out/pmc_c4_WRITE_SIZE $out/pmc_c4_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1; echo "summary rc=$?"
find $out -name "*kernel_trace.csv" -size +2M -delete; find $out -name "*counter_collection.csv" -size +2M -delete
grep -h "fringe_pair" $out/pmc_summary.txt | cut -c1-900
