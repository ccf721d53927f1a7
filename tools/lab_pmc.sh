#!/bin/bash
# PMC pass over forward-kernel lab variants (tools/bin/lab_*, built from tools/fringe_mfma_lab.hip with -DRIME_LAB_* switches)
# at the C4 lab shape: matrix-pipe busy cycles, VALU / MFMA instruction counts, clock.  Run through gpurun from the repo root.
set -u
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag/lab_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in base chain1 m3; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d $out/$v -o pmc -- $GRAFT_REPO_ROOT/tools/bin/lab_$v 128 256 98304 > $out/$v.log 2>&1
  echo "$v rc=$?"
done
cd $GRAFT_REPO_ROOT
for v in base chain1 m3; do
  python tools/pmc_summary.py $out/$v.json $out/$v > $out/$v.txt 2>&1
  grep fringe_ant_fwd $out/$v.txt | head -2
done
find $out -name "*counter_collection.csv" -size +1M -delete
