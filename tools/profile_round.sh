#!/bin/bash
# Round profile set on ONE MI355X (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats and PMC
# passes of the same command.  Outputs under gpurun_out/$1/ (copy what is to be judged into profiles/).
set -u
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "[1] bench c4 (default command)"; timeout -k 10 400 python bench.py > $out/bench_c4.json 2> $out/bench_c4.err; echo rc=$?
echo "[2] bench c3 / c2 / c5 rank share"
timeout -k 10 200 python bench.py --workload c3 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err; echo rc=$?
timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err; echo rc=$?
timeout -k 10 300 python bench.py --workload c5 --nf 64 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c5_rankshare.json 2> $out/bench_c5_rankshare.err; echo rc=$?
cd /tmp && export TMPDIR=/tmp
echo "[3] rocprofv3 kernel stats, c4"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/prof_c4.log 2>&1; echo rc=$?
echo "[4] rocprofv3 kernel stats, c5 rank share"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c5 -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --nf 64 --steps 2 --warmup 1 --no-cpu-baseline > $out/prof_c5.log 2>&1; echo rc=$?
echo "[5] PMC passes, c4 (separate runs, counters only)"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$name -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1; echo "summary rc=$?"
# keep the merge-back small: the per-dispatch traces are not needed
find $out -name "*kernel_trace.csv" -size +2M -delete; find $out -name "*counter_collection.csv" -size +2M -delete
ls -la $out | head -40
