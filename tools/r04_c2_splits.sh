#!/bin/bash
# Lab (needs the two getenv hooks named in profiles/r04/c2_splits.txt, not in the library): C2 step time against the number of pixel splits of the forward blocks (RIME_LAB_FWD_S) and the pixel tiles per
# backward block (RIME_LAB_BWD_PER): 1920 (t, f) rows against the 1280 / 256 blocks the chip holds at once.
set -e
out=gpurun_out/r04_c2_splits.txt
mkdir -p gpurun_out; : > $out
run() {
    echo "== $*" >> $out
    env "$@" python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels']
print('ms/step %.4f' % d['ms_per_step'], {n.replace('fringe_ant_',''): round(v['total_ms']/v['launches'],4) for n,v in k.items()})" >> $out
}
run A=0
for s in 2 3 4 6; do run RIME_LAB_FWD_S=$s; done
for p in 128 96 64 48 32 16; do run RIME_LAB_BWD_PER=$p; done
run RIME_LAB_FWD_S=2 RIME_LAB_BWD_PER=64
run A=0
cat $out
