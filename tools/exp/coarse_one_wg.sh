#!/bin/bash
# would the two or three coarsest levels run faster as ONE workgroup each (all 22 iterations in one launch, workgroup
# barriers only)?  Measured inputs of the estimate: (a) a level's 22 iterations as they run today, eager and as one HIP-graph replay
# (kbench mode 13: kernel time + the gap between dependent launches); (b) the VALU instructions one iteration of that level issues
# (PMC over kbench mode 7), i.e. what ONE CU's four SIMDs would have to issue by themselves.
O=$PWD/gpurun_out/coarse_one_wg; mkdir -p $O
K=$PWD/tools/kbench
for sz in "54 36" "77 51" "109 72"; do echo "== level of $sz"; timeout -k 10 60 $K $sz 20 13; done > $O/iter22.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for sz in "54 36" "77 51" "109 72"; do
  tag=$(echo $sz | tr ' ' 'x')
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_$tag -o p --output-format csv -- $K $sz 5 7 > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('$O/pmc_*/')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
    for p in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            k=r['Kernel_Name'].split('(')[0][:60]
            if 'small' not in k: continue
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    dur=collections.defaultdict(list)
    for p in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            k=r['Kernel_Name'].split('(')[0][:60]
            if 'small' in k: dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for k in acc:
        print(d.split('/')[-2], k, {c: round(v/n[k][c]) for c,v in acc[k].items()}, 'median us under PMC', sorted(dur[k])[len(dur[k])//2] if dur[k] else None, 'wg', None)
PY
cat $O/iter22.txt
