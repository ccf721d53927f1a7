#!/bin/bash
# Kernel stats + FETCH/WRITE counters of the fused psky builder on the C4 arguments (tools/bench_beam_sky.py); run through
# gpurun from the repo root.  Outputs under gpurun_out/$1/beam_sky_*.
set -u
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/beam_sky_prof -o bs -- python3 $GRAFT_REPO_ROOT/tools/bench_beam_sky.py c4 > $out/beam_sky_prof.log 2>&1; echo "stats rc=$?"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/beam_sky_pmc_$set -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_beam_sky.py c4 > $out/beam_sky_pmc_$set.log 2>&1; echo "pmc $set rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/beam_sky_prof/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:10]:
        print('%-60s %5s calls  avg %8.1f us  min %8.1f  max %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
for cname in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(out + '/beam_sky_pmc_%s/**/*counter_collection.csv' % cname, recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == cname and 'rime::' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts 64-B fetch requests as 32 B (guide, HBM section): fetch bytes x 2
        m = (2 if cname == 'FETCH_SIZE' else 1) * 1024 / 1e9
        print('%-12s %-50s max %.3f GB  median %.3f GB  (%d dispatches)' % (cname, k, max(v) * m, sorted(v)[len(v) // 2] * m, len(v)))
PY
find $out -name "*kernel_trace.csv" -size +2M -delete; find $out -name "*counter_collection.csv" -size +2M -delete
