#!/bin/bash
# round 5: the mirror-pair kernels on the benchmark models, one box: default line (with the value self-check against the float64
# unsharded model), the same without the pairing (RIME_MIRROR=0), alternating; element-wise error on the reference fixtures
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05i}; mkdir -p $out
cd $GRAFT_REPO_ROOT
BENCH_SELFCHECK=1 timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python - <<EOF
import json
d=json.load(open("$out/bench_default.json"))
print('c4', round(d['ms_per_step'],2), d['roofline']['frac'], d.get('selfcheck'), d['config'].get('mirror'))
for o in d.get('other_workloads', []): print(o.get('workload'), o.get('ms_per_step'), o.get('kernels'))
EOF
for r in 1 2; do for wl in c4 c3 c2; do for m in 1 0; do
  steps=10; [ $wl = c2 ] && steps=40
  RIME_MIRROR=$m timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$wl mirror=$m $r', round(d['ms_per_step'],3), {n[11:14]:round(x['total_ms']/d['steps'],3) for n,x in k.items() if 'fringe' in n})"
done; done; done 2>&1 | tee $out/ab.txt
python tools/elementwise_error.py > $out/elementwise_error.txt 2>&1; tail -8 $out/elementwise_error.txt
