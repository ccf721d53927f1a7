#!/bin/bash
# Lab: C3 step and sky-gradient kernel time against the forced number of time splits of the sky gradient (RIME_SKYGRAD_SPLITS)
set -e
out=gpurun_out/r04_c3_skygrad.txt; mkdir -p gpurun_out; : > $out
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3 4 6; do
  e=""; [ $s != 0 ] && e="RIME_SKYGRAD_SPLITS=$s"
  rm -rf /tmp/sg_$s
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg_$s -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline --steps 5 --warmup 2 > /tmp/sg_$s.json 2>/dev/null
  python3 - <<E >> $GRAFT_REPO_ROOT/$out
import csv, glob, json
d = json.loads(open('/tmp/sg_$s.json').readline())
st = glob.glob('/tmp/sg_$s/**/*kernel_stats.csv', recursive=True)[0]
rows = {r['Name'].split('(')[0].replace('void rime::', ''): r for r in csv.DictReader(open(st))}
pick = {k: round(float(v['AverageNs']) / 1e3, 1) for k, v in rows.items() if any(s in k for s in ('sky_grad', 'plane_sum', 'beam_sky', 'interp_scatter'))}
print('splits %s: ms/step %.3f' % ('$s' if '$s' != '0' else 'planned', d['ms_per_step']), pick)
E
done
cat $GRAFT_REPO_ROOT/$out
