#!/bin/bash
# Round 5, VERDICT r04 item 1 (phasor shared along the frequency axis in the <= 64-antenna kernels): the counters of the
# TIMING-ONLY stand-ins (tools/lab: -DRIME_LAB_FWD_CHEAP / -DRIME_LAB_BWD_CHEAP: 3 of 4 channels pay 4 plain FMAs per
# phasor instead of 3 f64 FMA + fract + cvt + sin + cos) beside the library, on C3 and C2, one box: kernel time, VALU and
# MFMA instruction counts, matrix-pipe busy cycles, clock.  Build the variants first (CPU):
#   tools/build_variant.sh fwdcheap -DRIME_LAB_FWD_CHEAP; tools/build_variant.sh bwdcheap -DRIME_LAB_BWD_CHEAP
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05cs}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in c3 c2; do
for v in base fwdcheap bwdcheap; do
  if [ $v = base ]; then unset RIME_LIB_PATH; else export RIME_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/librime_$v.so; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
      --output-format csv -d $out/pmc_${wl}_$v -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_${wl}_$v.log 2>&1
  echo "pmc $wl $v rc=$?"
done
done
unset RIME_LIB_PATH
cd $GRAFT_REPO_ROOT
# plain timing, alternating, same box
for r in 1 2; do for wl in c3 c2; do for v in base fwdcheap bwdcheap; do
  if [ $v = base ]; then unset RIME_LIB_PATH; else export RIME_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/librime_$v.so; fi
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$wl $v $r', round(d['ms_per_step'],3), {n[11:14]:round(x['total_ms']/d['steps'],3) for n,x in k.items() if 'fringe' in n})"
done; done; done > $out/timing.txt 2>&1
unset RIME_LIB_PATH
python - <<EOF > $out/counters.txt
import csv, glob, collections
for wl in ('c3', 'c2'):
  for v in ('base', 'fwdcheap', 'bwdcheap'):
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
    for path in glob.glob("$out/pmc_%s_%s/**/*counter_collection.csv" % (wl, v), recursive=True):
        seen=set()
        for r in csv.DictReader(open(path)):
            k=r['Kernel_Name'].replace('void ','').replace('rime::','').split('(')[0]
            if 'fringe_ant' not in k: continue
            per[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if (k, r['Dispatch_Id']) not in seen:
                seen.add((k, r['Dispatch_Id'])); cnt[k]+=1
                dur[k]+=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    for k in sorted(per):
        c={n: x/cnt[k] for n,x in per[k].items()}
        if c.get('SQ_INSTS_MFMA', 0) == 0: continue
        ms=dur[k]/cnt[k]/1e6
        simd=c['GRBM_GUI_ACTIVE']/8*1024
        print(wl, v, k, 'avg_ms %.3f' % ms, 'clock_GHz %.2f' % (c['GRBM_GUI_ACTIVE']/8/(ms*1e6)),
              'VALU_per_MFMA %.2f' % ((c['SQ_INSTS_VALU']-c['SQ_INSTS_MFMA'])/c['SQ_INSTS_MFMA']),
              'INSTS_VALU %.4g INSTS_MFMA %.4g' % (c['SQ_INSTS_VALU'], c['SQ_INSTS_MFMA']),
              'mfma_busy %.3f' % (c['SQ_VALU_MFMA_BUSY_CYCLES']/simd), 'valu_issue %.3f' % ((c['SQ_ACTIVE_INST_VALU']-c['SQ_INSTS_MFMA'])*4/simd))
EOF
cat $out/counters.txt $out/timing.txt
find $out -name "*counter_collection.csv" -size +1M -delete
