#!/bin/bash
# mirror pairs on / off (RIME_MIRROR), alternating on one box: tools/r05_mirror_ab.sh <tag> "<workloads>"
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05j}; mkdir -p $out
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for wl in ${2:-c4 c3 c2}; do for m in 1 0; do
  steps=10; [ $wl = c2 ] && steps=40
  RIME_MIRROR=$m timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$wl mirror=$m $r', round(d['ms_per_step'],3), {n[11:14]:round(x['total_ms']/d['steps'],3) for n,x in k.items() if 'fringe' in n})"
done; done; done 2>&1 | tee $out/ab.txt
