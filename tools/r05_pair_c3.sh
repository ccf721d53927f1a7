#!/bin/bash
# one-tile pair form: parity tests, then C3 / C2 / C4 with the form on and off
set -u
mkdir -p gpurun_out/pair_c3
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conjugate_pair or mirror_pairs" > gpurun_out/pair_c3/tests.txt 2>&1 || { tail -30 gpurun_out/pair_c3/tests.txt; exit 1; }
tail -2 gpurun_out/pair_c3/tests.txt
timeout -k 10 600 python -m pytest tests/test_rime_gpu.py -x -q > gpurun_out/pair_c3/rime.txt 2>&1 || { tail -30 gpurun_out/pair_c3/rime.txt; exit 1; }
tail -2 gpurun_out/pair_c3/rime.txt
for wl in c3 c2; do for pair in 1 0; do
  RIME_PAIR=$pair timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/pair_c3/${wl}_pair$pair.json 2> gpurun_out/pair_c3/${wl}_pair$pair.err || { tail -5 gpurun_out/pair_c3/${wl}_pair$pair.err; exit 1; }
  python - $wl $pair <<'PY'
import json, sys
d = json.loads(open('gpurun_out/pair_c3/%s_pair%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print(sys.argv[1], 'RIME_PAIR=' + sys.argv[2], 'ms/step %.3f' % d['ms_per_step'], d['config'].get('antenna_pair_blocks'), d['config'].get('antenna_mirror_groups'),
      {n[11:14]: round(v['total_ms'] / d['steps'], 3) for n, v in k.items() if n.startswith('fringe')})
PY
done; done
