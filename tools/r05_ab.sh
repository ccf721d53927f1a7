#!/bin/bash
# A/B of lab variants against the library on ONE box, alternating: tools/r05_ab.sh <tag> "<workloads>" <variant> [<variant> ...]
# (variants: tools/bin/librime_<name>.so from tools/build_variant.sh; "base" = the in-tree library)
set -u
tag=$1; wls=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for wl in $wls; do for v in base "$@"; do
  if [ $v = base ]; then unset RIME_LIB_PATH; else export RIME_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/librime_$v.so; fi
  steps=10; [ $wl = c2 ] && steps=40
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$wl $v $r', round(d['ms_per_step'],3), {n[11:14]:round(x['total_ms']/d['steps'],3) for n,x in k.items() if 'fringe' in n})"
done; done; done 2>&1 | tee $out/ab.txt
unset RIME_LIB_PATH
