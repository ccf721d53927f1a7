#!/bin/bash
# Round profile set on ONE MI355X (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats and PMC
# passes of the same command.  Outputs under gpurun_out/$1/ (copy what is to be judged into profiles/).
set -u
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "[1] bench c4 (default command)"; timeout -k 10 400 python bench.py > $out/bench_c4.json 2> $out/bench_c4.err; echo rc=$?
echo "[2] bench c3 / c2 / c5 rank share"
timeout -k 10 200 python bench.py --workload c3 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err; echo rc=$?
timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err; echo rc=$?
timeout -k 10 300 python bench.py --workload c5 --nf 64 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c5_rankshare.json 2> $out/bench_c5_rankshare.err; echo rc=$?
echo "[2b] the N > 1 code path on one rank over RCCL (supervisor, both partitions, self-check)"
BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --no-cpu-baseline > $out/bench_c4_rccl_one_rank.json 2> $out/bench_c4_rccl_one_rank.err; echo rc=$?
cd /tmp && export TMPDIR=/tmp
echo "[3] rocprofv3 kernel stats, c4 and c3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c4 -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/prof_c4.log 2>&1; echo rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline --steps 5 --warmup 2 > $out/prof_c3.log 2>&1; echo rc=$?
echo "[5] PMC passes, c4 and c3 (separate runs, counters only)"
for wl in c4 c3; do
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_${wl}_$name -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_${wl}_$name.log 2>&1; echo "pmc $wl $name rc=$?"
done
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc_c4_FETCH_SIZE $out/pmc_c4_WRITE_SIZE $out/pmc_c4_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1; echo "summary rc=$?"
mkdir -p $out/c3 && python tools/pmc_summary.py $out/c3/pmc_summary.json $out/pmc_c3_FETCH_SIZE $out/pmc_c3_WRITE_SIZE $out/pmc_c3_SQ_INSTS_VALU > $out/c3/pmc_summary.txt 2>&1; echo "summary c3 rc=$?"
# keep the merge-back small: the per-dispatch traces are not needed
find $out -name "*kernel_trace.csv" -size +2M -delete; find $out -name "*counter_collection.csv" -size +2M -delete
ls -la $out | head -40
