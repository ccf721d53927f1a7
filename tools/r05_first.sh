#!/bin/bash
# round 5, first GPU call: the new object-contract tests, the whole -m gpu suite, smoke, the default bench line (with other_workloads)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05a}; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_objects_gpu.py -x -q -m gpu > $out/objects.txt 2>&1; echo "objects rc=$?"; tail -5 $out/objects.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $out/gputest.txt 2>&1; echo "gpu suite rc=$?"; tail -5 $out/gputest.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.txt
( time timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; echo "bench rc=$?"; cat $out/bench_default.time
python - <<EOF
import json
d=json.load(open("$out/bench_default.json"))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
for o in d.get('other_workloads', []): print(o)
EOF
