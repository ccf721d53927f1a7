#!/bin/bash
# closing run of the round's last tree: -m gpu suite, smoke(), the driver's default bench command
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/closing
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gputest.txt 2>&1; echo "suite rc=$?"; tail -2 $out/gputest.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench_c4.json 2> $out/bench_c4.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/closing/bench_c4.json').read().strip().splitlines()[-1])
print('ms/step %.2f value %.3e frac %s useful %s traffic_ratio %s cpu %.0f speedup %.3g' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['useful_frac_of_pipe_peak'], d['roofline']['traffic_ratio'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline']))
for o in d['other_workloads']:
    print('  ', o['workload'], o.get('ms_per_step'), o.get('pair_blocks'), o.get('mirror_groups'))
PY
