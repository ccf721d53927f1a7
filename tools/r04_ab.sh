#!/bin/bash
# A/B of library variants on ONE box: tools/r04_ab.sh <outdir> "<variants>" "<workloads>" [rounds]
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${4:-2}); do for v in $2; do for wl in $3; do
  lib=""; [ $v != base ] && lib=$GRAFT_REPO_ROOT/tools/bin/librime_$v.so
  RIME_LIB_PATH=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $out/bench_${wl}_${v}_$r.json 2> $out/bench_${wl}_${v}_$r.err
  python - <<E
import json
d=json.load(open("$out/bench_${wl}_${v}_$r.json")); k=d["roofline"]["kernels"]
print("$wl $v $r", round(d["ms_per_step"],3), {n.replace("fringe_ant_","").replace("_kernel",""):round(x["total_ms"]/10,3) for n,x in k.items()})
E
done; done; done 2>&1 | tee $out/summary.txt
