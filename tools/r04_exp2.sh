#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04f}; mkdir -p $out; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_rime_gpu.py -x -q -k "${2:-matrix_core or packed_forward or degenerate or hex37 or c3 or c2 or self_blocks or small_groups or full_size or golden or stitch}" > $out/tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/tests.txt
for wl in c2 c3 c4; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "$wl rc=$?"
  python - <<E
import json
d=json.load(open("$out/bench_$wl.json")); k=d["roofline"]["kernels"]
print("$wl", round(d["ms_per_step"],3), {n:(round(x["total_ms"]/10,3), x["frac"]) for n,x in k.items()})
E
done
