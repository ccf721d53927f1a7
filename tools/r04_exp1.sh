#!/bin/bash
# round-4 experiments, one box: SIMD placement of a block's waves; forward pixel-split size (slab traffic vs accuracy);
# C2 step: eager vs hipGraph, kernel stats
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r04e; mkdir -p $out; cd $GRAFT_REPO_ROOT
./tools/bin/simd_map > $out/simd_map.txt 2>&1; head -12 $out/simd_map.txt
for v in base split16k split24k base; do
  lib=""; [ $v != base ] && lib=$GRAFT_REPO_ROOT/tools/bin/librime_$v.so
  RIME_LIB_PATH=$lib BENCH_SELFCHECK=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_c4_$v.json 2> $out/bench_c4_$v.err; echo "$v rc=$?"
  python - <<E
import json
d=json.load(open("$out/bench_c4_$v.json")); k=d["roofline"]["kernels"]
print("$v", round(d["ms_per_step"],2), {n:round(x["total_ms"]/10,2) for n,x in k.items()}, d.get("selfcheck",{}).get("vis_relmax"), d.get("selfcheck",{}).get("grad_relmax"))
E
done
timeout -k 10 300 python tools/graphed_step.py c2 > $out/graphed_c2.txt 2>&1; tail -2 $out/graphed_c2.txt
timeout -k 10 300 python tools/step_profile.py c2 > $out/step_profile_c2.txt 2>&1; head -3 $out/step_profile_c2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --no-cpu-baseline --steps 20 --warmup 3 > $out/prof_c2.log 2>&1; echo "prof rc=$?"
find $out -name "*kernel_trace.csv" -size +2M -delete
