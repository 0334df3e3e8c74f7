#!/bin/bash
# round 3, experiment 7: k_cost_march4 on the large levels (standalone strip-height sweep, then the pipeline with wider ranges)
O=gpurun_out/exp7; mkdir -p $O
export UGSM_DEV=1
for sz in "4928 3264 10" "3484 2308 10"; do timeout -k 10 100 ./tools/kbench $sz 14; done > $O/kb14_large.txt 2>&1
cat $O/kb14_large.txt
for r in "150001,9000000" "150001,20000000"; do
  echo "== one slot UGSM_MARCH4=$r"; UGSM_MARCH4=$r timeout -k 10 200 python bench.py --slots 1 --no-events --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/one_slot.txt 2>&1
grep -v "amdgpu.ids\|^\[bench" $O/one_slot.txt
for r in "0,0" "150001,3000000" "150001,5000000" "150001,9000000" "150001,20000000" "1,3000000"; do
  echo "== four slots UGSM_MARCH4=$r"; UGSM_MARCH4=$r timeout -k 10 200 python bench.py --no-events --no-cpu-baseline --steps 48 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/four_slots.txt 2>&1
grep -v "amdgpu.ids\|^\[bench" $O/four_slots.txt
