cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rankshare
timeout -k 10 300 python bench.py --nf 32 --no-cpu-baseline --no-other-workloads --steps 20 --warmup 5 > gpurun_out/rankshare/bench_nf32.json 2> gpurun_out/rankshare/bench_nf32.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/rankshare/bench_nf32.json').read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print('nf32 ms/step %.3f' % d['ms_per_step'], {n: round(v['total_ms'] / d['steps'], 3) for n, v in k.items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rankshare/prof -o nf32 -- python3 $GRAFT_REPO_ROOT/bench.py --nf 32 --no-cpu-baseline --no-other-workloads --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/rankshare/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/rankshare -name "*kernel_trace.csv" -size +2M -delete
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/rankshare/prof/nf32_kernel_stats.csv')))
tot = 0
for r in rows[:22]:
    ms = float(r['TotalDurationNs']) / 1e6 / 12
    tot += ms
    print('%7.3f ms/step x%-3d %s' % (ms, int(r['Calls']) // 12, r['Name'][:90]))
print('sum of all kernels per step: %.3f' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / 12))
PY
