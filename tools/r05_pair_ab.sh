#!/bin/bash
# A/B of pair-kernel variants on the headline workload: usage tools/r05_pair_ab.sh <name> ...   (name = product | tools/bin/librime_<name>.so)
set -u
mkdir -p gpurun_out/pair_ab
for name in "$@"; do
  for rep in 1 2; do
    if [ "$name" = product ]; then unset RIME_LIB_PATH; else export RIME_LIB_PATH=$PWD/tools/bin/librime_$name.so; fi
    timeout -k 10 300 python bench.py --no-other-workloads --no-cpu-baseline > gpurun_out/pair_ab/${name}_$rep.json 2> gpurun_out/pair_ab/${name}_$rep.err || { tail -5 gpurun_out/pair_ab/${name}_$rep.err; exit 1; }
    python - "$name" "$rep" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/pair_ab/%s_%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print(sys.argv[1], sys.argv[2], 'ms/step %.2f' % d['ms_per_step'], {n[11:14]: round(v['total_ms'] / d['steps'], 2) for n, v in k.items() if n.startswith('fringe')})
PY
  done
done
