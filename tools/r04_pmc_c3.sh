#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04j}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/avail.txt 2>&1; grep -c "" $out/avail.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${2:-c3} --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_$i.log 2>&1; echo "pmc $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<E
import csv, glob, collections
for i in (1,2,3):
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.defaultdict(float)
    for path in glob.glob("$out/pmc_%d/**/*counter_collection.csv" % i, recursive=True):
        seen=set()
        for r in csv.DictReader(open(path)):
            k=r['Kernel_Name'].replace('void ','').replace('rime::','').split('(')[0]
            if 'fringe' not in k: continue
            per[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if (k, r['Dispatch_Id']) not in seen:
                seen.add((k, r['Dispatch_Id'])); cnt[k]+=1
                dur[k]+=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    for k in per:
        print(i, k, 'launches', cnt[k], 'avg_ms', round(dur[k]/cnt[k]/1e6,3), {c: '%.4g' % (v/cnt[k]) for c,v in per[k].items()})
E
find $out -name "*counter_collection.csv" -size +1M -delete
