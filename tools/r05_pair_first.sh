#!/bin/bash
# first GPU run of the conjugate-pair kernels: parity tests, then the headline bench with the form on and off
set -o pipefail
mkdir -p gpurun_out
echo "(ops tests: run before)"

timeout -k 10 600 python -m pytest tests/test_rime_gpu.py -x -q -k "matrix_core_arrays_against_reference" > gpurun_out/pair_golden.txt 2>&1 || { tail -30 gpurun_out/pair_golden.txt; exit 1; }
tail -3 gpurun_out/pair_golden.txt
timeout -k 10 400 python bench.py --no-other-workloads > gpurun_out/pair_bench_on.json 2> gpurun_out/pair_bench_on.err || { tail -20 gpurun_out/pair_bench_on.err; exit 1; }
RIME_PAIR=0 timeout -k 10 400 python bench.py --no-other-workloads > gpurun_out/pair_bench_off.json 2> gpurun_out/pair_bench_off.err || { tail -20 gpurun_out/pair_bench_off.err; exit 1; }
python - <<'PY'
import json
for tag in ('on', 'off'):
    d = json.loads(open('gpurun_out/pair_bench_%s.json' % tag).read().strip().splitlines()[-1])
    k = d['roofline']['kernels']
    print(tag, 'ms/step %.2f' % d['ms_per_step'], 'value %.3e' % d['value'],
          {n: (v['total_ms'], v.get('frac')) for n, v in k.items() if n.startswith('fringe')})
PY
